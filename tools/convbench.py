#!/usr/bin/env python
"""tools/convbench.py -- local_mix (3^3 submanifold conv) and whole-ELKBlock timings, C=64:
HIP kernel vs the per-offset torch gather+GEMM loop it replaced, on a LiDAR-like frame and on cfg2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import link_amd as la
from link_amd.elk import subm_conv
from bench import s_uniform
from helpers import lidar_like
dev = torch.device("cuda", 0)
C = 64

def timeit(fn, k=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

def torch_loop(feats, kernel, nbr):
    n = feats.shape[0]
    idx = torch.where(nbr < 0, torch.full_like(nbr, n), nbr).long()
    padded = torch.cat([feats, feats.new_zeros(1, feats.shape[1])], 0)
    out = feats.new_zeros(n, kernel.shape[2])
    for k in range(kernel.shape[0]):
        out = out + padded[idx[:, k]].matmul(kernel[k])
    return out

for name, coords, s in (("lidar-like", torch.from_numpy(lidar_like(120000, seed=0)), 7), ("cfg2 S-uniform", s_uniform(100000), 7)):
    coords = coords.to(dev); n = coords.shape[0]
    torch.manual_seed(0)
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
    feats = torch.randn(n, C, device=dev)
    st = la.SparseTensor(feats, coords, 1)
    conv = blk.local_mix[0]
    nbr, order = conv._neighbor_table(st)
    present = float((nbr >= 0).float().mean()) * 27
    t_hip0 = timeit(lambda: subm_conv(feats, conv.kernel, nbr))
    t_hip = timeit(lambda: subm_conv(feats, conv.kernel, nbr, order))
    t_loop = timeit(lambda: torch_loop(feats, conv.kernel.detach(), nbr), k=5, warm=1)
    flops = 2.0 * float((nbr >= 0).sum()) * C * C
    def block():
        x = la.SparseTensor(feats, coords, 1); x.kmaps = st.kmaps; x.cmaps = st.cmaps
        with torch.no_grad(): return blk(x, s, 3)
    t_blk = timeit(block)
    print(f"{name}: N={n} neighbours/voxel={present:.1f}  conv HIP {t_hip:.1f} us (unordered tiles {t_hip0:.1f}) ({flops/t_hip/1e6:.1f} TFLOP/s useful) | "
          f"torch loop {t_loop:.1f} us | whole ELKBlock.forward (warm maps) {t_blk:.1f} us")

# training: conv forward + backward (input + weight gradients)
for name, coords in (("lidar-like", torch.from_numpy(lidar_like(120000, seed=0))), ("cfg2 S-uniform", s_uniform(100000))):
    coords = coords.to(dev); n = coords.shape[0]
    conv = la.Conv3d(C, C, 3).to(dev)
    feats = torch.randn(n, C, device=dev)
    st = la.SparseTensor(feats, coords, 1); conv._neighbor_table(st)
    gout = torch.randn(n, C, device=dev)
    def step():
        f = feats.detach().requires_grad_(True)
        x = la.SparseTensor(f, coords, 1); x.kmaps = st.kmaps; x.cmaps = st.cmaps
        conv(x).F.backward(gout)
    print(f"{name}: conv fwd+bwd (d feats, d kernel) {timeit(step, k=20):.1f} us")

# whole ELKBlock training step (fwd+bwd), warm maps
for name, coords, s in (("lidar-like", torch.from_numpy(lidar_like(120000, seed=0)), 7), ("cfg2 S-uniform", s_uniform(100000), 7)):
    coords = coords.to(dev); n = coords.shape[0]
    torch.manual_seed(0)
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).train()
    feats = torch.randn(n, C, device=dev); gout = torch.randn(n, C, device=dev)
    st = la.SparseTensor(feats, coords, 1)
    with torch.no_grad(): blk(la.SparseTensor(feats.clone(), coords, 1), s, 3)
    st0 = la.SparseTensor(feats.clone(), coords, 1); blk.local_mix[0]._neighbor_table(st0); la.voxel_to_aux(st0, s)
    def step():
        f = feats.detach().clone().requires_grad_(True)
        x = la.SparseTensor(f, coords, 1); x.kmaps = st0.kmaps; x.cmaps = st0.cmaps
        blk(x, s, 3).F.backward(gout)
    print(f"{name}: whole ELKBlock training step (fwd+bwd, warm maps) {timeit(step, k=20):.1f} us")
