#!/usr/bin/env python
"""tools/kbench.py -- per-stage timings of the R_core step under different launch geometries
(link_set_tuning), HIP-event timed, cfg2 workload.  Usage: python tools/kbench.py [key=v1,v2,...] ..."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform
from link_amd import _lib as L

N, C = int(os.environ.get("KB_N", 100000)), int(os.environ.get("KB_C", 64))
dev = torch.device("cuda", 0)
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
coords = s_uniform(N).to(dev)
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
plan = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev)
plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
          blk.norm.weight, blk.norm.bias)
plan.run(feats, coords)
lib, st = L.lib(), torch.cuda.current_stream().cuda_stream
b, grid, desc = plan.buf, plan.grid, plan.desc
stages = {
    "index": lambda: lib.link_index_build(coords.data_ptr(), N, ctypes.byref(grid), b.cell_counts, b.scratch,
                                          b.scratch_bytes, b.cell_blk, b.vox_blk, b.idx_query, b.perm,
                                          b.vox_sorted, b.pos_blk, b.blk_start, b.blk_coords, b.counts, b.hdr, st),
    "premix": lambda: lib.link_premix_ln(b.feats, b.w_pre, b.pre_ln_w, b.pre_ln_b, N, C, 1e-6, b.fin, st),
    "modsum": lambda: lib.link_modulate_block_sum(b.fin, b.vox_sorted, b.w_pos, b.alpha, b.blk_start, b.hdr,
                                                  ctypes.byref(desc), N, N, b.S, st),
    "gather": lambda: lib.link_gather_demod_ln(b.S, b.fin, b.vox_sorted, b.w_pos, b.alpha, b.ln_w, b.ln_b,
                                               b.blk_start, b.blk_coords, b.cell_blk, ctypes.byref(grid), b.hdr,
                                               ctypes.byref(desc), N, N, b.out, st),
    "bgather": lambda: lib.link_block_gather(b.S, b.blk_coords, b.cell_blk, ctypes.byref(grid), b.hdr,
                                             ctypes.byref(desc), N, b.A, st),
    "vdemod": lambda: lib.link_voxel_demod_ln(b.A, b.fin, b.vox_sorted, b.pos_blk, b.w_pos, b.alpha, b.ln_w,
                                              b.ln_b, b.hdr, ctypes.byref(desc), N, b.out, st),
}
KEYS = {"modsum": 0, "gather": 1, "premix": 2, "group": 3, "pair": 4, "bgather": 5, "split": 6, "wt": 8}


def time_stage(fn, k=50):
    for _ in range(5):
        fn()
    evs = []
    for _ in range(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in evs)
    return ts[len(ts) // 2]


print("default:", {k: round(time_stage(f), 2) for k, f in stages.items()})
for arg in sys.argv[1:]:
    key, vals = arg.split("=")
    for v in vals.split(","):
        lib.link_set_tuning(KEYS[key], int(v))
        if key == "wt":
            print(f"wt={v}:", {k: round(time_stage(f), 2) for k, f in stages.items() if k in ("premix", "modsum", "bgather", "vdemod")})
        elif key in ("group", "pair"):
            print(f"{key}={v}:", {k: round(time_stage(f), 2) for k, f in stages.items() if k in ("modsum", "gather")})
        else:
            print(f"{key} wgs={v}: {time_stage(stages[key]):.2f} us")
# whole step (one FFI call), cold and warm
import time
for ov in (0,):
  for cold in (True, False):
    for _ in range(10):
        plan.run(feats, coords, cold)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        plan.run(feats, coords, cold)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("overlap", ov, "step", "cold" if cold else "warm", f"{(t1 - t0) / 200 * 1e6:.1f} us")
# memory-system yardsticks: plain device copy / fill of one [N,C] fp32 tensor
src, dst = feats, torch.empty_like(feats)
print("torch copy [N,C] f32:", round(time_stage(lambda: dst.copy_(src)), 2), "us  (", round(2 * src.numel() * 4 / 1e6, 1), "MB moved )")
print("torch fill [N,C] f32:", round(time_stage(lambda: dst.fill_(1.0)), 2), "us")
big = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev); big2 = torch.empty_like(big)
print("torch copy 256MB:", round(time_stage(lambda: big2.copy_(big), 10), 2), "us (512 MB moved)")
# frame-level concurrency: S independent frames in flight, one plan + one stream each (no events)
for nstreams in (1, 2, 3, 4):
    plans, streams, fr = [], [], []
    for k in range(nstreams):
        pl = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev)
        pl.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
                blk.norm.weight, blk.norm.bias)
        plans.append(pl); streams.append(torch.cuda.Stream())
        fr.append((torch.randn(N, C, generator=torch.Generator().manual_seed(10 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
    torch.cuda.synchronize()
    def run(K):
        for it in range(K):
            k = it % nstreams
            with torch.cuda.stream(streams[k]):
                plans[k].run(fr[k][0], fr[k][1], True)
    run(4 * nstreams); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(240); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{nstreams} stream(s): {(t1 - t0) / 240 * 1e6:.1f} us per frame -> {N * 240 / (t1 - t0) / 1e9:.3f} Gvox/s")
