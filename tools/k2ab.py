"""tools/k2ab.py -- forms of the fused gather + de-modulate kernel on cfg2 (k2_form 0 default = quad consumers, 8 pair consumers, 4 own-cell, 1 single-role) x z-segments: HIP-event time of the kernel inside the full step + bitwise comparison of the outputs."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from link_amd import _lib as L
from bench import s_uniform

dev = torch.device("cuda")
N, C = 100000, 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
coords = s_uniform(N, seed=0).to(dev)
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
lib = L.lib()
ref = None
for form in (8, 0, 4, 1):
    for zs in (0, 2, 3, 8):
        p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k2_form=form, k2_zsplit=zs)
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
               blk.norm.weight, blk.norm.bias)
        out = p.run(feats, coords).clone()
        if ref is None:
            ref = out
        b, g, d = p.buf, p.dcg, p.desc
        st = torch.cuda.current_stream().cuda_stream
        ts = []
        for _ in range(60):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.link_dc_index(b.coords, N, ctypes.byref(g), b.cnt, b.slots, b.vcell, b.hdr, st)
            lib.link_dc_premix_modsum(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, 0, st)
            e0.record()
            lib.link_dc_gather_demod(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, st)
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(1e3 * a.elapsed_time(b_) for a, b_ in ts[5:])
        print(f"k2_form {form} zsplit {zs}: K2 {v[len(v) // 2]:6.2f} us   bitwise == first {bool(torch.equal(out, ref))}  max diff {float((out - ref).abs().max()):.2e}")

# cos_x (3-part rows: the producer / consumer form does not fit two workgroups per CU, so form 0 runs the single-role kernel)
for r, s in ((2, 6), (3, 7)):
    torch.manual_seed(3)
    blkx = la.ELKBlock(C, C, groups=1, baseop="cos_x").to(dev).eval()
    ref = None
    for form in (0, 4):
        p = la.ElkCorePlan(N, C, "cos_x", C, r, s, bounds, dev, layout="dense", k2_form=form)
        p.bind(blkx.pre_mix[0].weight, blkx.pre_mix[1].weight, blkx.pre_mix[1].bias, blkx.pos_weight[0].weight, blkx.alpha,
               blkx.norm.weight, blkx.norm.bias)
        out = p.run(feats, coords).clone()
        ref = out if ref is None else ref
        b, g, d = p.buf, p.dcg, p.desc
        st = torch.cuda.current_stream().cuda_stream
        ts = []
        for _ in range(60):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            lib.link_dc_index(b.coords, N, ctypes.byref(g), b.cnt, b.slots, b.vcell, b.hdr, st)
            lib.link_dc_premix_modsum(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, 0, st)
            e0.record()
            lib.link_dc_gather_demod(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, st)
            e1.record()
            ts.append((e0, e1))
        torch.cuda.synchronize()
        v = sorted(1e3 * a.elapsed_time(b_) for a, b_ in ts[5:])
        print(f"cos_x r {r} s {s} k2_form {form}: K2 {v[len(v) // 2]:6.2f} us   max diff vs form 0 {float((out - ref).abs().max()):.2e}")
