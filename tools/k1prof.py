"""tools/k1prof.py -- the fused pre_mix kernel on cfg2: HIP-event time inside the full step over k1_wgs / k1_form, and the
per-wave s_memtime phases of the tile form (link_dc_tuning_t::k1_dbg).  Run on the GPU box."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import link_amd as la
from link_amd import _lib as L
from bench import s_uniform

dev = torch.device("cuda")
N, C = int(os.environ.get("N", 100000)), int(os.environ.get("C", 64))
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
coords = s_uniform(N, seed=0).to(dev)
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
lib = L.lib()


def plan(**kw):
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", **kw)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
           blk.norm.weight, blk.norm.bias)
    return p


def k1_time(p, iters=60):
    b, g, d = p.buf, p.dcg, p.desc
    st = torch.cuda.current_stream().cuda_stream
    idx = (lambda: lib.link_dc_index_ids(b.coords, N, ctypes.byref(g), b.cnt, b.sid, b.vcell, b.hdr, st)) if b.tune.k1_form == 1 else \
          (lambda: lib.link_dc_index(b.coords, N, ctypes.byref(g), b.cnt, b.slots, b.vcell, b.hdr, st))
    p.run(feats, coords)
    ts, ts2 = [], []
    for _ in range(iters):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        idx()
        e0.record()
        lib.link_dc_premix_modsum(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, 0, st)
        e1.record()
        lib.link_dc_gather_demod(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, st)
        e2.record()
        ts.append((e0, e1)); ts2.append((e1, e2))
    torch.cuda.synchronize()
    v = sorted(1e3 * a.elapsed_time(b_) for a, b_ in ts[5:])
    w = sorted(1e3 * a.elapsed_time(b_) for a, b_ in ts2[5:])
    return v[len(v) // 2], w[len(w) // 2]


for wgs, pad in ((512, 0), (512, 20000), (512, 45056), (768, 0), (768, 18000), (1024, 0), (1024, 5000)):
    p = plan(k1_form=1, k1_wgs=wgs, k1_lds_pad=pad)
    k1, k2 = k1_time(p)
    print(f"tile form k1_wgs {wgs:5d} lds pad {pad:6d}: K1 {k1:6.2f} us   K2 {k2:6.2f} us")
p = plan(k1_form=0, k1_wgs=512)
print("cell-range form 512: K1 %.2f K2 %.2f" % k1_time(p))

for wgs, pad in ((512, 0), (512, 45056), (768, 18000)):
    p = plan(k1_form=1, k1_wgs=wgs, k1_lds_pad=pad)
    dbg = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    p.buf.tune.k1_dbg = dbg.data_ptr()
    for _ in range(3):
        p.run(feats, coords)
    torch.cuda.synchronize()
    d = dbg.view(-1, 8).cpu().numpy()
    d = d[d[:, 5] > 0]
    names = ["W staging", "chunk section", "mfma (+row wait)", "theta/sincos/LN", "scan+stores", "total"]
    print(f"tile form, k1_wgs {wgs} pad {pad}: {len(d)} waves; s_memtime ticks per wave (mean / p50 / max), tiles per wave {d[:, 6].mean():.2f}")
    for i, nm in enumerate(names):
        print(f"  {nm:18s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {d[:, i].max():9.0f}")
    tl = d[:, 6].sum()
    print(f"  per tile: mfma {d[:, 2].sum() / tl:.0f}  valu {d[:, 3].sum() / tl:.0f}  scan {d[:, 4].sum() / tl:.0f}")
    span = (d[:, 7] + d[:, 5]).max() - d[:, 7].min()
    print(f"  first start -> last end: {span} ticks; start skew p50 {np.median(d[:, 7] - d[:, 7].min()):.0f} max {(d[:, 7] - d[:, 7].min()).max()}")
