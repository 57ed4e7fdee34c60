"""tools/graphstep.py -- the bench's three-frames-in-flight step as ONE hipGraph: NS independent chains (one per captured stream) of
REPS cold R_core steps each, replayed GRAPHS times, against the same work issued from the host on NS streams.  Does taking the host
(and its launch-time jitter) out of the loop change the frame rate?   NS=3 REPS=20 python tools/graphstep.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from bench import s_uniform

N, C = 100000, 64
NS, REPS, GRAPHS = int(os.environ.get("NS", 3)), int(os.environ.get("REPS", 20)), int(os.environ.get("GRAPHS", 30))
dev = torch.device("cuda")
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
frames, plans, streams = [], [], []
for k in range(NS):
    frames.append((torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, frames_in_flight=NS)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    plans.append(p)
    streams.append(torch.cuda.Stream(device=dev))


def host_batches(k):
    for _ in range(k):
        for j in range(NS):
            with torch.cuda.stream(streams[j]):
                plans[j].run(*frames[j])


host_batches(300)
torch.cuda.synchronize()
t0 = time.perf_counter()
host_batches(REPS * GRAPHS)
torch.cuda.synchronize()
t_host = (time.perf_counter() - t0) / (REPS * GRAPHS * NS)
ref = [p.out[:N].clone() for p in plans]

g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream(device=dev)
with torch.cuda.stream(cap):
    g.capture_begin()
    ev0 = torch.cuda.Event()
    ev0.record(cap)
    done = []
    for j in range(NS):
        streams[j].wait_event(ev0)                       # fork
        with torch.cuda.stream(streams[j]):
            for _ in range(REPS):
                plans[j].run(*frames[j])
            e = torch.cuda.Event()
            e.record(streams[j])
            done.append(e)
    for e in done:
        cap.wait_event(e)                                # join
    g.capture_end()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(GRAPHS):
    g.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / (REPS * GRAPHS * NS)
same = all(torch.equal(p.out[:N], r) for p, r in zip(plans, ref))
print(f"NS={NS} REPS={REPS}: host-issued {1e6 * t_host:.2f} us/frame, one graph of {NS * REPS * 3} kernels {1e6 * t_graph:.2f} us/frame, same bits {same}")

# short timed regions, as the driver's `--steps 20` gives: 12 samples of 20 host-issued batches between synchronisations (with and
# without the garbage collector), then the same through the graph (REPS must be 20 for that)
import gc
for label, fn in (("host-issued", lambda: host_batches(20)), ("host-issued, gc off", lambda: host_batches(20)),
                  ("graph", (lambda: g.replay()) if REPS == 20 else None)):
    if fn is None:
        continue
    if "gc off" in label:
        gc.disable()
    samples = []
    for _ in range(12):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        samples.append(1e6 * (time.perf_counter() - t0) / (20 * NS))
    gc.enable()
    print(f"20-batch regions, {label}: us/frame", " ".join(f"{v:.1f}" for v in samples))
