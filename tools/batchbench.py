#!/usr/bin/env python
"""tools/batchbench.py -- R_core on B cfg2 frames as ONE batched tensor (batch index in coords[:,3]) vs
B frames in flight on separate streams: per-frame time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
dev = torch.device("cuda", 0)
N, C = 100000, 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
for B in (1, 2, 3, 4, 6):
    cs = []
    for b in range(B):
        c = s_uniform(N, seed=b); c[:, 3] = b; cs.append(c)
    coords = torch.cat(cs).to(dev).contiguous()
    feats = torch.randn(N * B, C, device=dev)
    plan = la.ElkCorePlan(N * B, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, B - 1)), dev)
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
              blk.norm.weight, blk.norm.bias)
    for _ in range(10): plan.run(feats, coords)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 100
    for _ in range(K): plan.run(feats, coords)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
    print(f"batch of {B} frames in one call: {dt*1e6:.1f} us/call = {dt*1e6/B:.1f} us/frame = {N*B/dt/1e9:.3f} Gvox/s  (M={plan.blocks()})")
