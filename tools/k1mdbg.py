"""tools/k1mdbg.py -- where the matrix-core sums form (k1_form 2) of the fused pre_mix kernel differs from the cell-range form
(k1_form 0): per-cell S tables and outputs on cfg2, by cell population."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import link_amd as la
from bench import s_uniform

dev = torch.device("cuda")
N, C = int(os.environ.get("N", 100000)), 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
coords = s_uniform(N, seed=0).to(dev)
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))


def plan(**kw):
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", **kw)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
           blk.norm.weight, blk.norm.bias)
    return p


p0, p2 = plan(k1_form=0), plan(k1_form=int(os.environ.get("FORM", 2)), k1_wgs=int(os.environ.get("WGS", 512)))
o0 = p0.run(feats, coords).clone()
o2 = p2.run(feats, coords).clone()
torch.cuda.synchronize()
S0, S2 = p0.S.clone(), p2.S.clone()
n0, n2 = p0.cell_n.clone(), p2.cell_n.clone()
print("cell_n equal:", bool(torch.equal(n0, n2)), " out max abs diff", float((o0 - o2).abs().max()), "rel", float((o0 - o2).abs().max() / o0.abs().max()))
d = (S0 - S2).abs()
print("S max abs diff", float(d.max()), " S max", float(S0.abs().max()))
rowd = d.max(dim=1).values[: n0.numel()]
for k in range(0, 12):
    m = n0 == k
    if int(m.sum()):
        print(f"  cells with {k:2d} voxels: {int(m.sum()):6d}  max |dS| {float(rowd[m].max()):.3e}  mean {float(rowd[m].mean()):.3e}")
worst = int(rowd.argmax())
print("worst cell", worst, "count", int(n0[worst]), "dS row (first 8 of each part):", d[worst][:8].tolist(), d[worst][64:72].tolist())
print("S0 row", S0[worst][:8].tolist()); print("S2 row", S2[worst][:8].tolist())
