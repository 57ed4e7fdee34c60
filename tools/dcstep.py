"""tools/dcstep.py -- N cold steps of the dense-cell R_core plan on cfg2, one stream (profiling target:
TAG=x SCRIPT=tools/dcstep.py bash tools/profile_cmd.sh, or tools/pmc_dc.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

N, C = int(os.environ.get("DC_N", 100000)), int(os.environ.get("DC_C", 64))
steps = int(os.environ.get("DC_STEPS", 200))
layout = os.environ.get("DC_LAYOUT", "dense")
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
coords = s_uniform(N, seed=0).to(dev)
tune = {k: int(os.environ[e]) for k, e in (("k1_form", "DC_K1_FORM"), ("k1_wgs", "DC_K1_WGS"), ("k2_zsplit", "DC_K2_ZSPLIT"),
                                           ("k2_form", "DC_K2_FORM")) if os.environ.get(e) not in (None, "")}
p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, layout=layout,
                   **(tune if layout == "dense" else {}))
p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
       blk.norm.weight, blk.norm.bias)
for _ in range(steps):
    p.run(feats, coords)
torch.cuda.synchronize()
print("done", p.blocks())
