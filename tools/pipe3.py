"""tools/pipe3.py -- frames in flight as a STAGE pipeline: the three launches of a step (index -> pre_mix -> gather) go to three
streams, one per stage, chained by events per frame, so that at any time at most one kernel of each kind runs and the kinds
that fit a CU together (one pre_mix workgroup + one gather workgroup) overlap by construction -- against three independent
streams each running whole steps (what bench.py does).  Prototype of the host side in Python (9 calls per frame)."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from link_amd import _lib as L
from bench import s_uniform

dev = torch.device("cuda")
N, C = 100000, 64
NP = int(os.environ.get("NP", 3))
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
lib = L.lib()
frames, plans = [], []
for k in range(NP):
    frames.append((torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(N, seed=k).to(dev)))
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, layout="dense", frames_in_flight=3,
                       **{k_: int(v) for k_, v in (("k1_wgs", os.environ.get("K1_WGS", "")), ("k2_zsplit", os.environ.get("ZS", "")),
                                                  ("k1_form", os.environ.get("K1_FORM", "")), ("k1_lds_pad", os.environ.get("PAD", ""))) if v != ""})
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    p.run(*frames[k])
    plans.append(p)
torch.cuda.synchronize()
ref = [p.out[:N].clone() for p in plans]
s_idx, s_k1, s_k2 = (torch.cuda.Stream(device=dev) for _ in range(3))
ev_idx = [torch.cuda.Event() for _ in range(NP)]
ev_k1 = [torch.cuda.Event() for _ in range(NP)]
ev_k2 = [torch.cuda.Event() for _ in range(NP)]


def issue(f):
    j = f % NP
    p = plans[j]
    b, g, d = p.buf, p.dcg, p.desc
    b.feats, b.coords = frames[j][0].data_ptr(), frames[j][1].data_ptr()
    b.out = p.out.data_ptr()
    s_idx.wait_event(ev_k2[j])                       # the slot lists of this plan's previous frame have been read
    lib.link_dc_index(b.coords, N, ctypes.byref(g), b.cnt, b.slots, b.vcell, b.hdr, s_idx.cuda_stream)
    ev_idx[j].record(s_idx)
    s_k1.wait_event(ev_idx[j])
    lib.link_dc_premix_modsum(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, 0, s_k1.cuda_stream)
    ev_k1[j].record(s_k1)
    s_k2.wait_event(ev_k1[j])
    lib.link_dc_gather_demod(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, s_k2.cuda_stream)
    ev_k2[j].record(s_k2)


def timed(k):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(k):
        issue(f)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6, t_issue / k * 1e6


for e in ev_k2:
    e.record(s_k2)
timed(600)
for _ in range(3):
    us, host = timed(600)
    print(f"stage pipeline, {NP} plans: {us:.2f} us/frame (host issue {host:.2f} us/frame)")
print("outputs equal:", all(bool(torch.equal(p.out[:N], r)) for p, r in zip(plans, ref)))
# baseline: three independent streams, one FFI call per frame
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
def base(k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for f in range(k):
        j = f % 3
        with torch.cuda.stream(streams[j]):
            plans[j].run(*frames[j])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6
base(600)
for _ in range(3):
    print(f"independent streams: {base(600):.2f} us/frame")
