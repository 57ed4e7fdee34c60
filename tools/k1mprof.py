"""tools/k1mprof.py -- cell-range form (k1_form 0) against the matrix-core sums form (k1_form 2) of the fused pre_mix kernel on
cfg2: HIP-event time of K1 / K2 inside the step over the number of waves, and the per-wave s_memtime phases (k1_dbg)."""
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


def k_times(p, iters=80):
    b, g, d = p.buf, p.dcg, p.desc
    st = torch.cuda.current_stream().cuda_stream
    p.run(feats, coords)
    ts, ts2 = [], []
    for _ in range(iters):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        lib.link_dc_index(b.coords, N, ctypes.byref(g), b.cnt, b.slots, b.vcell, b.hdr, st)
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


for form in (0, 2):
    for wgs in (128, 256, 384, 512, 768, 1024):
        k1, k2 = k_times(plan(k1_form=form, k1_wgs=wgs))
        print(f"k1_form {form} k1_wgs {wgs:5d} ({4 * wgs} waves): K1 {k1:6.2f} us   K2 {k2:6.2f} us")

names = {0: ["W staging", "cell section", "fill", "tile bodies", "per-cell sums", "total"],
         2: ["W staging", "cell section", "rows+contraction", "theta/sincos/LN", "second product", "total"]}
for form, wgs in ((0, 512), (2, 256), (2, 512), (2, 1024)):
    p = plan(k1_form=form, k1_wgs=wgs)
    dbg = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
    p.buf.tune.k1_dbg = dbg.data_ptr()
    for _ in range(3):
        p.run(feats, coords)
    torch.cuda.synchronize()
    d = dbg.view(-1, 8).cpu().numpy()
    d = d[d[:, 5] > 0]
    print(f"form {form}, k1_wgs {wgs}: {len(d)} waves; s_memtime ticks per wave (mean / p50 / max), tiles per wave {d[:, 6].mean():.2f}, tiles {d[:, 6].sum()}")
    for i, nm in enumerate(names[form]):
        print(f"  {nm:18s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {d[:, i].max():9.0f}")
    tl = max(d[:, 6].sum(), 1)
    print(f"  per tile: {names[form][2]} {d[:, 2].sum() / tl:.0f}  {names[form][3]} {d[:, 3].sum() / tl:.0f}  {names[form][4]} {d[:, 4].sum() / tl:.0f}")
    span = (d[:, 7] + d[:, 5]).max() - d[:, 7].min()
    print(f"  first start -> last end: {span} ticks; start skew p50 {np.median(d[:, 7] - d[:, 7].min()):.0f} max {(d[:, 7] - d[:, 7].min()).max()}")
