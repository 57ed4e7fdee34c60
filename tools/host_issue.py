"""How long the HOST takes to issue the bench's steps (cfg2, three frames in flight) against how long the GPU takes to run them:
per-frame host time with torch's stream context per frame (what bench.py did until round 5) and with the raw stream handle passed
to ElkCorePlan.run; timed regions of 20 steps (the driver's) and 200, frames per step 3 / 12 / 24.   python tools/host_issue.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import link_amd as la
from helpers import s_uniform
dev = torch.device("cuda:0")
N, C = 100000, 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
NS = 3
frames, plans, streams = [], [], []
for k in range(NS):
    frames.append((torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
    pl = la.ElkCorePlan(N, C, "cos", 32, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, frames_in_flight=NS)
    pl.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    plans.append(pl); streams.append(torch.cuda.Stream(device=dev))
raw = [s.cuda_stream for s in streams]
def region(k, rounds, ctx):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k):
        for _r in range(rounds):
            for j in range(NS):
                if ctx:
                    with torch.cuda.stream(streams[j]):
                        plans[j].run(frames[j][0], frames[j][1])
                else:
                    plans[j].run(frames[j][0], frames[j][1], stream=raw[j])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    nf = k * rounds * NS
    return 1e6 * (t1 - t0) / nf, 1e6 * (t2 - t0) / nf
for _ in range(30): region(20, 1, True)
for ctx in (True, False):
    for k, rounds in ((20, 1), (20, 4), (20, 8), (200, 1)):
        for _ in range(8): region(k, rounds, ctx)
        v = sorted(region(k, rounds, ctx) for _ in range(9))
        print(f"stream context per frame={ctx}: {k} steps x {rounds * NS} frames: host issue {v[4][0]:.1f} us/frame, region {v[4][1]:.2f} us/frame (min {v[0][1]:.2f})")
