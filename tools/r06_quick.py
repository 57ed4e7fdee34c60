"""tools/r06_quick.py -- us/frame of the timed geometry (NS plans on NS streams, cold index) over per-plan tuning variants, on ONE box in
ONE process, alternating (box-to-box spread is larger than most effects).  VARIANTS: ';'-separated 'key=value,key=value' lists of
link_dc_tuning_t fields ('' = default), e.g.  VARIANTS=";k2_zsplit=1;k1_lds_pad=0" python tools/r06_quick.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

N, C, NS = 100000, 64, int(os.environ.get("DC_STREAMS", 3))
STEPS, PASSES = int(os.environ.get("STEPS", 300)), int(os.environ.get("PASSES", 3))
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
frames = [(torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)) for k in range(NS)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
variants = os.environ.get("VARIANTS", ";k2_zsplit=1").split(";")


def mk(spec):
    kw = {k: int(v) for k, v in (kv.split("=") for kv in spec.split(",") if kv)}
    ps = []
    for _ in range(NS):
        p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, frames_in_flight=NS, **kw)
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
               blk.norm.weight, blk.norm.bias)
        ps.append(p)
    return ps


def timed(ps, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        for j in range(NS):
            with torch.cuda.stream(streams[j]):
                ps[j].run(*frames[j])
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / (k * NS)


allp = {v: mk(v) for v in variants}
ref = None
for v, ps in allp.items():                               # outputs agree bit for bit across launch geometries
    outs = [ps[j].run(*frames[j]).clone() for j in range(NS)]
    torch.cuda.synchronize()
    if ref is None:
        ref = outs
    else:
        print(f"variant {v!r}: bitwise equal to the default: {all(torch.equal(a, b) for a, b in zip(ref, outs))}")
for ps in allp.values():
    timed(ps, 100)
for p_ in range(PASSES):
    for v, ps in allp.items():
        print(f"pass {p_} variant {v or 'default'!r}: {timed(ps, STEPS):.2f} us/frame", flush=True)
