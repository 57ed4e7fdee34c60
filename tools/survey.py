#!/usr/bin/env python
"""tools/survey.py -- R_core step time (cold / warm, one frame in flight) on BASELINE-shaped workloads."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import link_amd as la
from bench import s_uniform
from helpers import lidar_like

dev = torch.device("cuda", 0)

def run(name, coords, C, baseop, groups, r, s, coord_div=1.0):
    n = coords.shape[0]
    torch.manual_seed(2)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).to(dev).eval()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(1)).to(dev)
    c = coords.to(dev)
    lo, hi = la.coords_bounds(c)
    plan = la.ElkCorePlan(n, C, baseop, C // groups, r, s, (lo, hi), dev, coord_div=coord_div)
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
              blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
    out = plan.run(feats, c); m = plan.blocks()
    assert torch.isfinite(out).all()
    res = []
    for cold in (True, False):
        for _ in range(5): plan.run(feats, c, cold)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(100): plan.run(feats, c, cold)
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 100 * 1e6)
    print(f"{name:44s} N={n:7d} M={m:6d} V={plan.grid.cells:8d} C={C:3d} {baseop:5s} r={r} s={s:2d}: cold {res[0]:7.1f} us  warm {res[1]:7.1f} us  "
          f"-> {n / res[0] * 1e-3:6.2f} Gvox/s cold", flush=True)

run("cfg1 S-uniform 10k C=16", s_uniform(10_000), 16, "cos", 2, 3, 7)
run("cfg2 S-uniform 100k C=64", s_uniform(100_000), 64, "cos", 2, 3, 7)
for C in (16, 32, 128):
    run(f"S-uniform 100k C={C}", s_uniform(100_000), C, "cos", 2, 3, 7)
run("S-uniform 100k C=64 cos_x (2x3)^3", s_uniform(100_000), 64, "cos_x", 1, 2, 3)
b4 = torch.cat([torch.cat([s_uniform(100_000, seed=k)[:, :3], torch.full((100_000, 1), k, dtype=torch.int32)], 1) for k in range(4)])
run("batch of 4 S-uniform frames (400k)", b4, 64, "cos", 2, 3, 7)
lid = torch.from_numpy(lidar_like(120_000, seed=0))
run("LiDAR-like stride 1, cos (3x7)^3", lid, 64, "cos", 2, 3, 7)
lid2 = torch.from_numpy(lidar_like(120_000, seed=1, stride=2))
run("LiDAR-like stride 2, cos_x (2x3)^3 s_eff=6", lid2, 64, "cos_x", 1, 2, 6)
lid8 = torch.from_numpy(lidar_like(120_000, seed=2, stride=8))
run("LiDAR-like stride 8, cos_x s_eff=24", lid8, 64, "cos_x", 1, 2, 24)

# per-stage timing for the LiDAR-like stride-1 case
import ctypes
from link_amd import _lib as L
def stages(coords, C=64, baseop="cos", groups=2, r=3, s=7):
    n = coords.shape[0]
    torch.manual_seed(2)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).to(dev).eval()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(1)).to(dev)
    c = coords.to(dev)
    lo, hi = la.coords_bounds(c)
    plan = la.ElkCorePlan(n, C, baseop, C // groups, r, s, (lo, hi), dev)
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
              blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
    plan.run(feats, c)
    lib, st = L.lib(), torch.cuda.current_stream().cuda_stream
    b, grid, desc = plan.buf, plan.grid, plan.desc
    N = n
    st_fns = {
        "index": lambda: lib.link_index_build(c.data_ptr(), N, ctypes.byref(grid), b.cell_counts, b.scratch, b.scratch_bytes, b.cell_blk, b.vox_blk, b.idx_query, b.perm, b.vox_sorted, b.pos_blk, b.blk_start, b.blk_coords, b.counts, b.hdr, st),
        "premix": lambda: lib.link_premix_ln(b.feats, b.w_pre, b.pre_ln_w, b.pre_ln_b, N, C, 1e-6, b.fin, st),
        "modsum": lambda: lib.link_modulate_block_sum(b.fin, b.vox_sorted, b.w_pos, b.alpha, b.blk_start, b.hdr, ctypes.byref(desc), N, N, b.S, st),
        "bgather": lambda: lib.link_block_gather(b.S, b.blk_coords, b.cell_blk, ctypes.byref(grid), b.hdr, ctypes.byref(desc), N, b.A, st),
        "vdemod": lambda: lib.link_voxel_demod_ln(b.A, b.fin, b.vox_sorted, b.pos_blk, b.w_pos, b.alpha, b.ln_w, b.ln_b, b.hdr, ctypes.byref(desc), N, b.out, st),
    }
    out = {}
    for k, fn in st_fns.items():
        for _ in range(3): fn()
        evs = []
        for _ in range(30):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); evs.append((e0, e1))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in evs); out[k] = round(ts[len(ts) // 2], 1)
    cnt = plan.counts[: plan.blocks()].float()
    print("stages:", out, "| voxels/block: mean %.1f max %d" % (cnt.mean().item(), int(cnt.max().item())))
stages(lid)
stages(s_uniform(10_000), C=16)
