#!/usr/bin/env python
"""tools/densebench.py -- the two forms of the block gather on cfg2 (HIP-event timed stage) and the bench-style
frame rate for each (link_set_tuning key 9: 1 = dense-grid form allowed, 2 = column-walking form only)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
from link_amd import _lib as L
N, C = 100000, 64
dev = torch.device("cuda", 0)
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
lib = L.lib()
def make(k):
    pl = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev)
    pl.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    return pl, torch.cuda.Stream(), torch.randn(N, C, generator=torch.Generator().manual_seed(10 + k)).to(dev), s_uniform(N, seed=k).to(dev)
sets = [make(k) for k in range(3)]
def stage_us(k=60):
    pl, _, f, c = sets[0]; pl.run(f, c)
    b, st = pl.buf, torch.cuda.current_stream().cuda_stream
    fn = lambda: lib.link_block_gather(b.S, b.blk_coords, b.cell_blk, ctypes.byref(pl.grid), b.hdr, ctypes.byref(pl.desc), pl.m_cap if hasattr(pl, "m_cap") else N, b.A, st)
    for _ in range(5): fn()
    evs = []
    for _ in range(k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b_) * 1e3 for a, b_ in evs); return ts[len(ts) // 2]
def frame_us(nstreams, K=300):
    torch.cuda.synchronize()
    def run(K):
        for it in range(K):
            pl, sm, f, c = sets[it % nstreams]
            with torch.cuda.stream(sm): pl.run(f, c, True)
    run(30); torch.cuda.synchronize(); t0 = time.perf_counter(); run(K); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6
for name, keys in (("column-walking form only", [(9, 2)]), ("dense-grid form allowed", [(9, 1)])):
    for k, v in keys: lib.link_set_tuning(k, v)
    print(f"{name}: block_gather stage {stage_us():.2f} us | frame {frame_us(1):.1f} us (1 stream) {frame_us(3):.1f} us (3 in flight)")
