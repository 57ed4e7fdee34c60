#!/usr/bin/env python
"""tools/mstream.py -- frames-in-flight throughput vs launch geometry (link_set_tuning)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
from link_amd import _lib as L
N, C = 100000, 64
dev = torch.device("cuda", 0)
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
lib = L.lib()
NS = int(os.environ.get("NS", 3))
plans, streams, fr = [], [], []
for k in range(4):
    pl = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev)
    pl.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
            blk.norm.weight, blk.norm.bias)
    plans.append(pl); streams.append(torch.cuda.Stream())
    fr.append((torch.randn(N, C, generator=torch.Generator().manual_seed(10 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
torch.cuda.synchronize()
def run(K, ns):
    for it in range(K):
        k = it % ns
        with torch.cuda.stream(streams[k]):
            plans[k].run(fr[k][0], fr[k][1], True)
def measure(ns):
    run(4 * ns, ns); torch.cuda.synchronize()
    t0 = time.perf_counter(); run(240, ns); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 240 * 1e6
for modsum, premix, bg in itertools.product((512, 1024, 2048), (256, 512, 1024), (256, 512, 1024)):
    lib.link_set_tuning(0, modsum); lib.link_set_tuning(2, premix); lib.link_set_tuning(5, bg)
    print(f"modsum={modsum} premix={premix} bgather={bg}: 1s {measure(1):.1f}  2s {measure(2):.1f}  3s {measure(3):.1f}  4s {measure(4):.1f} us/frame", flush=True)
