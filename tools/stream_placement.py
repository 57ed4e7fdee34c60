"""tools/stream_placement.py -- does the three-stream figure depend on WHICH three HIP streams carry the frames?  (round 6: the batch entry
point's rate depends on the hardware queues / pipes its streams landed on; this asks the same of the headline geometry.)  TRIPLES
candidate triples of torch streams, created one after the other, each timed PASSES times alternately on the same three plans."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

N, C, NS = 100000, 64, 3
STEPS, PASSES, TRIPLES = int(os.environ.get("STEPS", 300)), int(os.environ.get("PASSES", 3)), int(os.environ.get("TRIPLES", 6))
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
frames = [(torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)) for k in range(NS)]
plans = []
for _ in range(NS):
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, frames_in_flight=NS)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    plans.append(p)
triples = [[torch.cuda.Stream(device=dev) for _ in range(NS)] for _ in range(TRIPLES)]


def timed(streams, k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        for j in range(NS):
            plans[j].run(*frames[j], stream=streams[j].cuda_stream)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / (k * NS)


import ctypes
from link_amd import _lib as L


def share(a, b):
    d = ctypes.c_double(0.0)
    L.check(L.lib().link_streams_share_queue(a.cuda_stream, b.cuda_stream, ctypes.byref(d)), "link_streams_share_queue")
    return round(d.value, 1)


for t in triples:
    timed(t, 100)
print("queue test per triple (0>1, 0>2, 1>2; us: ~5 = own queues, >= 150 = one queue):", [[share(t[0], t[1]), share(t[0], t[2]), share(t[1], t[2])] for t in triples])
for p_ in range(PASSES):
    print(f"pass {p_}: " + "  ".join(f"{timed(t, STEPS):.2f}" for t in triples), flush=True)
