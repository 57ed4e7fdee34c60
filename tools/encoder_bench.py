#!/usr/bin/env python
"""tools/encoder_bench.py -- cfg3-shaped timing: LinK encoder stages (stem + 4 x [down k2s2, 2 residual
blocks + tail || ELKBlock + tail, add/ReLU], C=64, cos_x (2x3)^3, groups=1; linkencoder.py:186-368) on a
LiDAR-like frame, forward (eval) and forward+backward (train), with the share spent inside the ELK blocks."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import link_amd as la
import link_encoder as LE
from helpers import lidar_like
dev = torch.device("cuda", 0)
coords = torch.from_numpy(lidar_like(120000, seed=0)).to(dev)
n = coords.shape[0]
feats = torch.rand(n, 4, device=dev)
torch.manual_seed(0)
net = LE.build_stages(la, 4, 64, "cos_x", 1, 4).to(dev)
elk_t = [0.0]
def timed_elk(mod):
    orig = mod.forward
    def fwd(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = orig(*a, **k)
        torch.cuda.synchronize(); elk_t[0] += time.perf_counter() - t0
        return out
    return fwd
def run(train, k, instrument=False):
    net.train(train)
    st0 = la.SparseTensor(feats, coords, 1)
    with torch.no_grad(): net(st0, 3, 2)                      # warm the kernel maps / indices (cached on kmaps)
    saved = [m.forward for m in net.elk]
    if instrument:
        for m in net.elk: m.forward = timed_elk(m)
    def step():
        f = feats.detach().requires_grad_(train)
        x = la.SparseTensor(f, coords, 1); x.kmaps = st0.kmaps; x.cmaps = st0.cmaps
        if train:
            outs = net(x, 3, 2); outs[-1].F.square().sum().backward()
        else:
            with torch.no_grad(): net(x, 3, 2)
    for _ in range(3): step()
    elk_t[0] = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
    for m, f0 in zip(net.elk, saved): m.forward = f0
    return dt, elk_t[0] / k
sizes = []
x = la.SparseTensor(feats, coords, 1)
with torch.no_grad():
    for o in net(x, 3, 2): sizes.append(o.C.shape[0])
print(f"frame: {n} voxels; stage voxels {sizes}")
t_inf, _ = run(False, 20)
_, e_inf = run(False, 10, instrument=True)
t_tr, _ = run(True, 10)
print(f"encoder forward (eval, warm maps): {t_inf*1e3:.2f} ms  (ELK blocks forward: {e_inf*1e3:.2f} ms, synchronised timing)")
print(f"encoder forward+backward (train, warm maps, sum-of-squares loss on stage 4): {t_tr*1e3:.2f} ms")
