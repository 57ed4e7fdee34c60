#!/usr/bin/env python
"""tools/pairbench.py -- the two forms of the sparse convolution (table kernel conv.hip vs pair-list kernels
conv_pairs.hip) across neighbourhood densities, C = 64: where the cross-over (elk.PAIR_DENSITY_MAX) sits, and
the whole ELKBlock.forward (R_block) on cfg2 and a LiDAR-like frame."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import link_amd as la
from link_amd.elk import subm_conv, subm_conv_ln_add_relu
from link_amd.synth import s_kitti
from bench import s_uniform
from helpers import lidar_like
dev = torch.device("cuda", 0)

def timeit(fn, k=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

def ev(fn, k=40, warm=5):
    for _ in range(warm): fn()
    ts = []
    for _ in range(k):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort(); return ts[len(ts) // 2]

frames = [("cfg2 S-uniform 256^3", s_uniform(100000)),
          ("S-uniform 96^3", s_uniform(100000, grid=96)),
          ("S-uniform 64^3", s_uniform(100000, grid=64)),
          ("lidar-like", torch.from_numpy(lidar_like(120000, seed=0))),
          ("S-kitti stride 1", torch.from_numpy(s_kitti(0)[0])),
          ("S-uniform 56^3", s_uniform(100000, grid=56)),
          ("S-uniform 50^3", s_uniform(100000, grid=50)),
          ("dense cube 47^3", s_uniform(100000, grid=47))]
C = int(os.environ.get("C", "64"))
if os.environ.get("FRAMES"):
    frames = [frames[int(i)] for i in os.environ["FRAMES"].split(",")]
for name, coords in frames:
    coords = coords.to(dev); n = coords.shape[0]
    torch.manual_seed(0)
    conv = la.Conv3d(C, C, 3).to(dev)
    feats = torch.randn(n, C, device=dev)
    st = la.SparseTensor(feats, coords, 1)
    nbr, order = conv._neighbor_table(st)
    dens = float((nbr >= 0).sum()) / n
    w = conv.kernel.detach()
    t_tab = ev(lambda: subm_conv(feats, w, nbr, order, form="table"))
    t_pair = ev(lambda: subm_conv(feats, w, nbr, order, form="pairs"))
    lw, lb, add = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.randn(n, C, device=dev)
    t_tab_t = ev(lambda: subm_conv_ln_add_relu(feats, w, nbr, order, lw, lb, 1e-6, add, form="table"))
    t_pair_t = ev(lambda: subm_conv_ln_add_relu(feats, w, nbr, order, lw, lb, 1e-6, add, form="pairs"))
    print(f"{name:24s} N={n:7d} nbrs/voxel={dens:5.2f} rows_pad={nbr._link_pairs.rows_pad:8d} | table {t_tab:7.1f} us  pairs {t_pair:7.1f} us | "
          f"with tail: table {t_tab_t:7.1f}  pairs {t_pair_t:7.1f}", flush=True)

if os.environ.get("FRAMES"):
    sys.exit(0)
for name, coords, s in (("cfg2", s_uniform(100000), 7), ("lidar-like", torch.from_numpy(lidar_like(120000, seed=0)), 7)):
    coords = coords.to(dev); n = coords.shape[0]
    torch.manual_seed(0)
    blk = la.ELKBlock(64, 64, groups=2, baseop="cos").to(dev).eval()
    feats = torch.randn(n, 64, device=dev)
    st = la.SparseTensor(feats, coords, 1)
    def block():
        x = la.SparseTensor(feats, coords, 1); x.kmaps = st.kmaps; x.cmaps = st.cmaps
        with torch.no_grad(): return blk(x, s, 3)
    import link_amd.elk as E
    for thr in (0.0, 10.0):
        E.PAIR_DENSITY_MAX = thr
        print(f"R_block {name}: PAIR_DENSITY_MAX={thr}: ELKBlock.forward warm {ev(block):.1f} us (events) {timeit(block):.1f} us (wall)")
