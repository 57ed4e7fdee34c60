"""tools/step3.py -- the three-frame step kernel (ElkCorePipeline / link_elk_core_dense_step3) on cfg2-sized frames:
results against ElkCorePlan(k1_form=2) bit for bit, then us / frame of a stream of pushes against three plans on three HIP
streams (what bench.py timed before).  Env: K1_WGS, ZS, IX_WGS, FRAMES."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

dev = torch.device("cuda")
N, C = int(os.environ.get("N", 100000)), 64
K = int(os.environ.get("FRAMES", 900))
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
par = (blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
NF = 5
frames = [(torch.randn(N - 1000 * k, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N - 1000 * k, seed=k).to(dev))
          for k in range(NF)]
tune = {k_: int(v) for k_, v in (("k1_wgs", os.environ.get("K1_WGS", "")), ("k2_zsplit", os.environ.get("ZS", ""))) if v != ""}
pipe = la.ElkCorePipeline(N, C, "cos", C // 2, 3, 7, bounds, dev, insert_wgs=int(os.environ.get("IX_WGS", 0)), **tune).bind(*par)
plan = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k1_form=2).bind(*par)
ref = [plan.run(f, c).clone() for f, c in frames]
got = []
for f, c in frames:
    r = pipe.push(f, c)
    if r is not None:
        got.append(r.clone())
got += [r.clone() for r in pipe.flush()]
torch.cuda.synchronize()
pipe.check()
assert len(got) == NF, len(got)
for k, (a, b) in enumerate(zip(got, ref)):
    print(f"frame {k}: n = {a.shape[0]}, equal = {bool(torch.equal(a, b))}, max |d| = {float((a - b).abs().max()):.3e}")


def timed(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(k):
        pipe.push(*frames[0])
    t_issue = time.perf_counter() - t0
    pipe.flush()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6, t_issue / k * 1e6


timed(300)
for _ in range(3):
    us, host = timed(K)
    print(f"step kernel: {us:.2f} us/frame (host issue {host:.2f} us/frame)")

plans = [la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", frames_in_flight=3).bind(*par) for _ in range(3)]
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]


def base(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(k):
        j = f % 3
        with torch.cuda.stream(streams[j]):
            plans[j].run(*frames[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


base(300)
for _ in range(3):
    print(f"three plans on three streams: {base(K):.2f} us/frame")
