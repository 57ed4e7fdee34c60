#!/usr/bin/env python
"""tools/hostcost.py -- host enqueue cost per frame (plan.run) vs GPU time, eager and hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
N, C = 100000, 64
dev = torch.device("cuda", 0)
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
NS = 3
plans, streams, fr = [], [], []
for k in range(NS):
    pl = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev)
    pl.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
            blk.norm.weight, blk.norm.bias)
    plans.append(pl); streams.append(torch.cuda.Stream())
    fr.append((torch.randn(N, C, generator=torch.Generator().manual_seed(10 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
torch.cuda.synchronize()
# host-only cost: enqueue 300 frames on one stream, measure time until the last call RETURNS (GPU far behind)
for _ in range(20): plans[0].run(*fr[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): plans[0].run(*fr[0])
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"eager: host enqueue {1e6*(t1-t0)/300:.1f} us/frame ; GPU-complete {1e6*(t2-t0)/300:.1f} us/frame")
# graphs: one captured frame per stream
graphs = []
for k in range(NS):
    with torch.cuda.stream(streams[k]):
        for _ in range(3): plans[k].run(*fr[k])
        streams[k].synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=streams[k]):
            plans[k].run(*fr[k])
        graphs.append(g)
torch.cuda.synchronize()
for ns in (1, 2, 3):
    for _ in range(10):
        for k in range(ns):
            with torch.cuda.stream(streams[k]): graphs[k].replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(300):
        k = it % ns
        with torch.cuda.stream(streams[k]): graphs[k].replay()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"graph replay, {ns} stream(s): host {1e6*(t1-t0)/300:.1f} us/frame ; total {1e6*(t2-t0)/300:.1f} us/frame")
for ns in (1, 2, 3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for it in range(300):
        k = it % ns
        with torch.cuda.stream(streams[k]): plans[k].run(*fr[k])
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"eager, {ns} stream(s): host {1e6*(t1-t0)/300:.1f} us/frame ; total {1e6*(t2-t0)/300:.1f} us/frame")
