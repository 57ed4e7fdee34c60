#!/usr/bin/env python
"""tools/convsweep.py -- tiles-per-wave / workgroup-cap sweep of the submanifold conv kernel (C=64)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import link_amd as la
from link_amd import _lib as L
from link_amd.elk import subm_conv
from helpers import lidar_like, s_uniform
dev = torch.device("cuda", 0)
C = int(os.environ.get("C", 64))
def timeit(fn, k=30, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
frames = {"lidar51k": torch.from_numpy(lidar_like(120000, seed=0)),
          "dense100k": s_uniform(100000, grid=48, seed=1),
          "slab100k": torch.cat([s_uniform(100000, grid=320, seed=2)[:, :2] , torch.zeros(100000, 1, dtype=torch.int32), torch.zeros(100000, 1, dtype=torch.int32)], 1)}
for name, coords in frames.items():
    coords = torch.unique(coords, dim=0)
    coords = coords[torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(0))].to(dev).contiguous()
    n = coords.shape[0]
    conv = la.Conv3d(C, C, 3).to(dev)
    feats = torch.randn(n, C, device=dev)
    st = la.SparseTensor(feats, coords, 1)
    nbr, order = conv._neighbor_table(st)
    pres = float((nbr >= 0).float().mean()) * 27
    row = [f"{name}: N={n} nbrs={pres:.1f}"]
    for nt in (1, 2, 4):
        for wgs in (512, 1024):
            L.lib().link_conv_set_tuning(1, nt); L.lib().link_conv_set_tuning(0, wgs)
            row.append(f"nt{nt}/w{wgs}: {timeit(lambda: subm_conv(feats, conv.kernel, nbr, order)):.0f}")
    L.lib().link_conv_set_tuning(1, 0); L.lib().link_conv_set_tuning(0, 512)
    row.append(f"auto: {timeit(lambda: subm_conv(feats, conv.kernel, nbr, order)):.0f}  unordered: {timeit(lambda: subm_conv(feats, conv.kernel, nbr, None)):.0f}")
    print("  ".join(row))
