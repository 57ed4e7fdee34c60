"""tools/dcstep3.py -- cold steps of the dense-cell R_core plan on cfg2 in the configuration bench.py TIMES: DC_STREAMS (3) plans
with frames_in_flight = DC_STREAMS (256 pre_mix workgroups, 2 z-segments, cell-range pre_mix form), three different frames, one HIP
stream each.  Profiling target for the PMC passes of round 5 (the counters of round 4 were collected with one frame in flight, i.e. on
the OTHER launch geometry).  Under a counter pass rocprofv3 serialises the dispatches, so what this gives is the instruction / byte /
cycle content of every launch of the timed geometry, not the concurrency."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

N, C = int(os.environ.get("DC_N", 100000)), int(os.environ.get("DC_C", 64))
steps = int(os.environ.get("DC_STEPS", 60))
NS = int(os.environ.get("DC_STREAMS", 3))
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
tune = {k: int(os.environ[e]) for k, e in (("k1_form", "DC_K1_FORM"), ("k1_wgs", "DC_K1_WGS"), ("k2_zsplit", "DC_K2_ZSPLIT"),
                                           ("k2_form", "DC_K2_FORM")) if os.environ.get(e) not in (None, "")}
frames, plans, streams = [], [], []
for k in range(NS):
    frames.append((torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, frames_in_flight=NS, **tune)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
           blk.norm.weight, blk.norm.bias)
    plans.append(p)
    streams.append(torch.cuda.Stream(device=dev))
torch.cuda.synchronize()
for _ in range(steps):
    for j in range(NS):
        with torch.cuda.stream(streams[j]):
            plans[j].run(*frames[j])
torch.cuda.synchronize()
print("done", [p.blocks() for p in plans])
