"""tools/batch_bench.py -- the batch entry point (ElkCoreBatch / link_elk_core_dense_forward_batch: one insert grid + two persistent
kernels) against three plans on three streams (what bench.py timed through round 5), same frames, same process, alternating.
    B=24 SETS=2 STEPS=40 PASSES=3 python tools/batch_bench.py
B frames per call, SETS arena sets alternated on as many streams (2: the pre_mix role of call s+1 starts under the gather role of s)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

N, C = 100000, 64
B, SETS = int(os.environ.get("B", 24)), int(os.environ.get("SETS", 2))
STEPS, PASSES = int(os.environ.get("STEPS", 40)), int(os.environ.get("PASSES", 3))
NF = int(os.environ.get("NF", 6))                      # distinct frames (rotated through the arenas)
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
frames = [(torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)) for k in range(NF)]


def bind(o):
    return o.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)


plans = [bind(la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, frames_in_flight=3)) for _ in range(3)]
pstreams = [torch.cuda.Stream(device=dev) for _ in range(3)]
batches = [bind(la.ElkCoreBatch(B, N, C, "cos", C // 2, 3, 7, bounds, dev))]
for _ in range(SETS - 1):
    batches.append(bind(la.ElkCoreBatch(B, N, C, "cos", C // 2, 3, 7, bounds, dev, share=batches[0])))
bstreams = [torch.cuda.Stream(device=dev) for _ in range(SETS)]
bf = [[frames[(i + 7 * j) % NF][0] for i in range(B)] for j in range(SETS)]
bc = [[frames[(i + 7 * j) % NF][1] for i in range(B)] for j in range(SETS)]

# correctness first: every batch row against the per-frame plan, bit for bit
ref = [plans[0].run(*frames[k]).clone() for k in range(NF)]
for j in range(SETS):
    outs = batches[j].run(bf[j], bc[j])
    torch.cuda.synchronize()
    batches[j].check()
    bad = [i for i in range(B) if not torch.equal(outs[i], ref[(i + 7 * j) % NF])]
    print(f"set {j}: {B - len(bad)}/{B} frames bit-equal to the per-frame plan" + (f"; first differing frame {bad[0]}: max rel "
          f"{float((outs[bad[0]] - ref[(bad[0] + 7 * j) % NF]).abs().max() / ref[(bad[0] + 7 * j) % NF].abs().max()):.3g}" if bad else ""), flush=True)


def t_streams(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(k):
        for r in range(B // 3):
            for j in range(3):
                with torch.cuda.stream(pstreams[j]):
                    plans[j].run(*frames[(3 * r + j) % NF])
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / (k * (B // 3) * 3)


def t_batch(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(k):
        j = s % SETS
        batches[j].run(bf[j], bc[j], stream=bstreams[j].cuda_stream)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / (k * B)


def t_submit(k):
    """the same calls from ONE stream: submit call s + 1, then join call s (link_dc_batch_submit / link_dc_batch_join)"""
    one = pstreams[0].cuda_stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    prev = None
    for s in range(k):
        j = s % SETS
        _, tk = batches[j].submit(bf[j], bc[j], stream=one)
        if prev is not None and SETS > 1:
            batches[0].join(prev, stream=one)
        prev = tk
        if SETS == 1:
            batches[0].join(tk, stream=one)
    if SETS > 1:
        batches[0].join(prev, stream=one)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / (k * B)


t_streams(10); t_batch(10); t_submit(10)
for p in range(PASSES):
    print(f"pass {p}: three plans on three streams {t_streams(STEPS):.2f} us/frame | batch of {B} x {SETS} set(s), two caller streams {t_batch(STEPS):.2f} | "
          f"submit / join from one stream {t_submit(STEPS):.2f} us/frame", flush=True)
batches[0].check()
print("queue delays (pre_mix->gather, pre_mix->insert, gather->insert, caller->pre_mix, caller->gather, caller->insert):", [batches[0].probe_streams(b.cuda_stream) for b in bstreams], "frame streams",
      [batches[0].probe_streams(s.cuda_stream) for s in pstreams])
print("status ok")
