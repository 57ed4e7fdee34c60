"""tools/k2prof.py -- per-wave s_memtime breakdown of the producer / consumer gather + de-modulate kernel on cfg2
(link_dc_tuning_t::k2_dbg): ticks waiting for the plane DMA, in the workgroup barrier, in the box sums, in the pair loop."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import link_amd as la
from bench import s_uniform

dev = torch.device("cuda")
N, C = 100000, 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
coords = s_uniform(N, seed=0).to(dev)
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
FORM = int(os.environ.get("K2_FORM", 0))
for zs in (0, 2):
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k2_zsplit=zs, k2_form=FORM)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
           blk.norm.weight, blk.norm.bias)
    dbg = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
    p.buf.tune.k2_dbg = dbg.data_ptr()
    for _ in range(3):
        p.run(feats, coords)
    torch.cuda.synchronize()
    d = dbg.view(-1, 8).cpu().numpy()
    for role, nm in ((1, "producer / compute"), (2, "consumer"), (3, "DMA wave")):
        e = d[d[:, 7] == role]
        if len(e) == 0:
            continue
        print(f"zsplit {zs} {nm}: {len(e)} waves, planes {e[:, 6].mean():.1f}; ticks per wave mean (per plane-step)")
        for i, k in enumerate(["total", "dma wait", "barrier", "box sums", "pairs"]):
            if (role == 2 and i == 1) or (role == 1 and i == 4):   # those words hold the wave's start time
                continue
            print(f"   {k:10s} {e[:, i].mean():9.0f}  ({e[:, i].sum() / e[:, 6].sum():7.0f})   max {e[:, i].max():9.0f}")
        print(f"   pair-loop iterations per plane-step {e[:, 5].sum() / e[:, 6].sum():.2f}")
