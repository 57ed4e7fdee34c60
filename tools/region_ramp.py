"""tools/region_ramp.py -- where does a SHORT timed region (the driver's --steps 20: 480 frames, 17 ms) lose its 5 % against a long one?
The timed geometry (three plans, three streams), regions of REGION frames between synchronisations; an event after every frame; the
frame completion period over consecutive windows of the region.  STAGGER=1: streams 1 / 2 start the region behind a 32 / 64 us spin."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

N, C, NS = 100000, 64, 3
REGION, REGIONS = int(os.environ.get("REGION", 480)), int(os.environ.get("REGIONS", 14))
STAGGER = float(os.environ.get("STAGGER", 0))
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
frames = [(torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)) for k in range(NS)]
plans = []
for _ in range(NS):
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, frames_in_flight=NS)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    plans.append(p)
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
clock_mhz = 100.0  # torch.cuda._sleep counts shader cycles; ~2.4 GHz


def region(record):
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    if STAGGER:
        for j in range(1, NS):
            with torch.cuda.stream(streams[j]):
                torch.cuda._sleep(int(STAGGER * j * 2400))
    for i in range(REGION // NS):
        for j in range(NS):
            plans[j].run(*frames[j], stream=streams[j].cuda_stream)
            if record:
                e = torch.cuda.Event(enable_timing=True)
                e.record(streams[j])
                evs.append(e)
    torch.cuda.synchronize()
    wall = 1e6 * (time.perf_counter() - t0) / REGION
    if not record:
        return wall, None
    ts = sorted(1e3 * e0.elapsed_time(e) for e in evs)
    w = 60
    return wall, [round((ts[min(k + w, len(ts)) - 1] - (ts[k - 1] if k else 0.0)) / (min(k + w, len(ts)) - k), 1) for k in range(0, len(ts), w)]


for r in range(REGIONS):
    wall, _ = region(False)
    print(f"region {r}: {wall:.2f} us/frame", flush=True)
wall, win = region(True)
print(f"recorded region: {wall:.2f} us/frame (events add host time); period per window of 60 frames: {win}")
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(4000 // NS):
    for j in range(NS):
        plans[j].run(*frames[j], stream=streams[j].cuda_stream)
torch.cuda.synchronize()
print(f"long run (4000 frames): {1e6 * (time.perf_counter() - t0) / (4000 // NS * NS):.2f} us/frame")
