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


# three different full-size frames in rotation, as bench.py does (one frame over and over would sit in the Infinity Cache:
# 33 against 37 us / frame)
same = [(torch.randn(N, C, generator=torch.Generator().manual_seed(11 + k)).to(dev), s_uniform(N, seed=20 + k).to(dev)) for k in range(3)]


def timed(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(k):
        pipe.push(*same[f % 3])
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
            plans[j].run(*same[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


base(300)
for _ in range(3):
    print(f"three plans on three streams: {base(K):.2f} us/frame")

if os.environ.get("SWEEP"):
    for k1 in (256, 384, 512, 768, 1024):
        for zs in (1, 2, 3, 4):
            for ix in (0, 16, 200):
                pipe = la.ElkCorePipeline(N, C, "cos", C // 2, 3, 7, bounds, dev, insert_wgs=ix, k1_wgs=k1, k2_zsplit=zs).bind(*par)
                timed(300)
                print(f"k1_wgs {k1:4d} zsplit {zs} insert_wgs {ix:3d}: {min(timed(K)[0] for _ in range(2)):.2f} us/frame", flush=True)
                del pipe

if os.environ.get("PROF"):
    # where each range of the step launch starts and ends (s_memtime, 10 ns ticks) in the steady state
    pipe = la.ElkCorePipeline(N, C, "cos", C // 2, 3, 7, bounds, dev, insert_wgs=int(os.environ.get("IX_WGS", 0)), **tune).bind(*par)
    d1 = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    d2 = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
    for p in pipe.plans:
        p.buf.tune.k1_dbg, p.buf.tune.k2_dbg = d1.data_ptr(), d2.data_ptr()
    for _ in range(8):
        pipe.push(*frames[0])
    torch.cuda.synchronize()
    pipe.push(*frames[0])
    torch.cuda.synchronize()
    a = d1.view(-1, 8).cpu().numpy()
    a = a[a[:, 5] > 0]
    b = d2.view(-1, 8).cpu().numpy()
    prod, cons = b[b[:, 7] == 1], b[b[:, 7] == 2]
    import numpy as np
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(50):
        pipe.push(*frames[0])
    ev1.record()
    torch.cuda.synchronize()
    print(f"launch period by events: {ev0.elapsed_time(ev1) * 20:.2f} us")
    # the counters of different XCDs are unrelated: group the waves by counter epoch, report each group on its own clock
    rows = np.concatenate([np.stack([a[:, 7], a[:, 5], np.zeros(len(a))], 1), np.stack([prod[:, 4], prod[:, 0], np.ones(len(prod))], 1),
                           np.stack([cons[:, 1], cons[:, 0], np.full(len(cons), 2.0)], 1)]).astype(np.float64)
    rows = rows[np.argsort(rows[:, 0])]
    cuts = np.where(np.diff(rows[:, 0]) > 1e7)[0] + 1
    for gi, grp in enumerate(np.split(rows, cuts)):
        t0 = grp[:, 0].min()
        line = f"clock group {gi}: {len(grp):4d} waves, span {(grp[:, 0] + grp[:, 1]).max() - t0:8.0f} ticks;"
        for r, nm in ((0, "K1"), (1, "K2 prod"), (2, "K2 cons")):
            e = grp[grp[:, 2] == r]
            if len(e):
                line += f"  {nm}: {len(e)} waves start <= {e[:, 0].max() - t0:6.0f}, end {(e[:, 0] + e[:, 1]).mean() - t0:6.0f} / {(e[:, 0] + e[:, 1]).max() - t0:6.0f}, run {e[:, 1].mean():6.0f}"
        print(line)
