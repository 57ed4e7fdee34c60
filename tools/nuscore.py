#!/usr/bin/env python
"""tools/nuscore.py -- R_core (TSELKBlock shape: cos, cg = C/2, r = 3, s = 7) on the S-nusc frame and its
down-sampled stages, dense-cell vs general layout, cold and warm index: how the two layouts take clumpy frames."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import link_amd as la
from link_amd.synth import s_nusc, block_stats
dev = torch.device("cuda", 0)

def wall(fn, k=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

co, _ = s_nusc(0)
for C, ds in ((16, 1), (32, 2), (64, 4), (128, 8), (64, 1)):
    c = co.copy(); c[:, :3] //= ds; c = np.unique(c, axis=0); n = c.shape[0]
    st = block_stats(c, 7)
    coords = torch.from_numpy(c).int().to(dev)
    feats = torch.randn(n, C, device=dev)
    torch.manual_seed(0)
    blk = la.TSELKBlock(C, C).to(dev).eval()
    hi = tuple(int(v) for v in c.max(0))
    line = f"C={C:3d} N={n:6d} M={st[1]:6d} N/M={st[2]:5.1f} max/block={st[3]:4d} |"
    for layout in ("dense", "general"):
        try:
            p = la.ElkCorePlan(n, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), hi), dev, layout=layout)
        except Exception as e:
            line += f" {layout}: n/a"; continue
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight[: C // 2], None,
               blk.norm.weight, blk.norm.bias)
        line += f" {layout}: cold {wall(lambda: p.run(feats, coords)):6.1f} warm {wall(lambda: p.run(feats, coords, build_index=False)):6.1f} us |"
    print(line, flush=True)
