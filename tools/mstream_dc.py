"""tools/mstream_dc.py -- frames-in-flight throughput of the dense-cell path under different launch geometries
(distinct frames per stream, as bench.py runs them)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from link_amd import _lib as L
from bench import s_uniform
dev = torch.device("cuda")
N, C = 100000, 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
lib = L.lib()
NSMAX = 4
frames, plans, streams = [], [], []
for k in range(NSMAX):
    frames.append((torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    plans.append(p); streams.append(torch.cuda.Stream())

def run(ns, k=300):
    for i in range(30):
        with torch.cuda.stream(streams[i % ns]): plans[i % ns].run(*frames[i % ns])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(k):
        with torch.cuda.stream(streams[i % ns]): plans[i % ns].run(*frames[i % ns])
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / k

for k1w, zs in ((512, 0), (256, 0), (512, 3), (256, 3), (256, 2), (384, 4), (320, 3)):
    lib.link_dc_set_tuning2(0, k1w); lib.link_dc_set_tuning2(3, zs)
    print(f"k1 wgs {k1w:4d}  k2 zsplit {zs}:  " + "  ".join(f"{ns} in flight {run(ns):6.2f} us" for ns in (1, 2, 3, 4)))
