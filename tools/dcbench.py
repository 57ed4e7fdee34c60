"""tools/dcbench.py -- per-kernel HIP-event timings of the dense-cell R_core path (cfg2 by default), next to
the general path, with optional tuning sweeps.  Run on the GPU box: python tools/dcbench.py [--sweep]."""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import link_amd as la
from link_amd import _lib as L
from bench import s_uniform


def ev_time(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    v = sorted(1e3 * a.elapsed_time(b) for a, b in ts)
    return v[len(v) // 2], v[0]


def wall(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--voxels", type=int, default=100_000)
    ap.add_argument("--channels", type=int, default=64)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--unfused", action="store_true")
    ap.add_argument("--split", action="store_true", help="separate gather + demod kernels instead of the fused one")
    ap.add_argument("--pipecmp", action="store_true", help="fused kernel: software-pipelined vs plain tiles")
    ap.add_argument("--phases", action="store_true", help="per-wave phase timing of the fused kernel (s_memtime)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    N, C = a.voxels, a.channels
    torch.manual_seed(2)
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
    feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
    coords = s_uniform(N, seed=0).to(dev)
    bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
    plans = {}
    for layout in ("dense", "general"):
        p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout=layout)
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
               blk.norm.weight, blk.norm.bias)
        plans[layout] = p
    od = plans["dense"].run(feats, coords).clone()
    og = plans["general"].run(feats, coords).clone()
    torch.cuda.synchronize()
    print("max |dense - general| / max|general| =", float((od - og).abs().max() / og.abs().max()),
          "blocks", plans["dense"].blocks(), plans["general"].blocks())
    for layout, p in plans.items():
        print(f"{layout:8s} cold {wall(lambda: p.run(feats, coords)):7.2f} us/step   "
              f"warm {wall(lambda: p.run(feats, coords, build_index=False)):7.2f} us/step")
    lib = L.lib()
    p = plans["dense"]
    b, g, d = p.buf, p.dcg, p.desc
    st = torch.cuda.current_stream().cuda_stream
    unfused = {
        "premix_insert": lambda: lib.link_dc_premix_insert(b.feats, b.coords, b.w_pre, b.pre_ln_w, b.pre_ln_b, N, C, 1e-6,
                                                           ctypes.byref(g), 1, b.fin, b.cnt, b.slots, b.vrec, b.vcell, b.hdr, st),
        "modsum": lambda: lib.link_dc_modsum(b.fin, b.slots, b.cnt, b.cell_n, b.w_pos, b.alpha, ctypes.byref(d), ctypes.byref(g),
                                             0, b.S, b.hdr, st),
        "gather": lambda: lib.link_dc_gather(b.S, b.cell_n, ctypes.byref(d), ctypes.byref(g), b.A, st),
        "demod_c": lambda: lib.link_voxel_demod_ln(b.A, b.fin, b.vrec, b.vcell, b.w_pos, b.alpha, b.ln_w, b.ln_b, b.hdr,
                                                   ctypes.byref(d), N, b.out, st),
    }
    stages = {
        "index": lambda: lib.link_dc_index(b.coords, N, ctypes.byref(g), b.cnt, b.slots, b.vcell, b.hdr, st),
        "premix_modsum": lambda: lib.link_dc_premix_modsum(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, 0, st),
        "gather": unfused["gather"],
        "demod": lambda: lib.link_dc_demod(b.A, b.fin, b.coords, b.vcell, b.w_pos, b.alpha, b.ln_w, b.ln_b,
                                           ctypes.byref(d), ctypes.byref(g), N, b.out, 0, st),
    }
    if a.unfused:
        stages = unfused
    elif not a.split and C == 64:
        stages = {"index": stages["index"], "premix_modsum": stages["premix_modsum"],
                  "gather_demod": lambda: lib.link_dc_gather_demod(ctypes.byref(b), ctypes.byref(g), ctypes.byref(d), N, st)}

    def chain():
        for f in stages.values():
            f()

    def report(tag=""):
        # each stage timed inside the full chain (its inputs are always the step's real data)
        res = {}
        for name in stages:
            def one(name=name):
                for k, f in stages.items():
                    if k == name:
                        e0.record(); f(); e1.record()
                    else:
                        f()
            ts = []
            for it in range(40):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                one()
                ts.append((e0, e1))
            torch.cuda.synchronize()
            v = sorted(1e3 * x.elapsed_time(y) for x, y in ts[5:])
            res[name] = v[len(v) // 2]
        print(tag, " ".join(f"{k} {v:6.2f}" for k, v in res.items()), f"| chain {wall(chain):6.2f} us")
        return res

    report("default   ")
    if a.phases and not a.unfused:
        dbg = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
        lib.link_dc_set_debug_buffer(dbg.data_ptr())
        for _ in range(3):
            chain()
        torch.cuda.synchronize()
        lib.link_dc_set_debug_buffer(None)
        d = dbg.view(-1, 8).cpu().numpy()
        d = d[d[:, 5] > 0]
        names = ["W staging", "cell section", "pipeline fill", "tile bodies", "per-cell sums", "total"]
        print(f"fused kernel, {len(d)} waves; cycles per wave (mean / p50 / max), tiles per wave {d[:, 6].mean():.2f}")
        for i, nm in enumerate(names):
            print(f"  {nm:14s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {d[:, i].max():9.0f}")
        print(f"  per tile: body {d[:, 3].sum() / d[:, 6].sum():.0f}  sums {d[:, 4].sum() / d[:, 6].sum():.0f}")
        span = (d[:, 7] + d[:, 5]).max() - d[:, 7].min()
        print(f"  first start -> last end: {span} cycles; start skew p50 {np.median(d[:, 7] - d[:, 7].min()):.0f} max {(d[:, 7] - d[:, 7].min()).max()}")
    if a.sweep and not a.unfused:
        for wgs in (256, 384, 512, 768, 1024):
            lib.link_dc_set_tuning2(0, wgs); report(f"k1 wgs={wgs:5d}")
        lib.link_dc_set_tuning2(0, 512)
        for wgs in (256, 512, 1024, 2048, 4096):
            lib.link_dc_set_tuning2(1, wgs); report(f"demod wgs={wgs:5d}")
        lib.link_dc_set_tuning2(1, 1024)
        for wgs in (128, 256, 512):
            lib.link_dc_set_tuning2(2, wgs); report(f"index wgs={wgs:5d}")
        lib.link_dc_set_tuning2(2, 0)
        for zs in (2, 3, 4, 5):
            lib.link_dc_set_tuning(2, zs); lib.link_dc_set_tuning2(3, zs); report(f"gather zsplit={zs:3d}")
        lib.link_dc_set_tuning(2, 0); lib.link_dc_set_tuning2(3, 0)
    if a.pipecmp:
        for pipe in (1, 0):
            lib.link_dc_set_tuning2(5, pipe)
            report(f"k1 pipe={pipe}")
            for ns in (1, 3):
                ps, ss = [], []
                for k in range(ns):
                    q = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense")
                    q.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
                           blk.norm.weight, blk.norm.bias)
                    ps.append(q); ss.append(torch.cuda.Stream())
                cnt = [0]

                def step():
                    j = cnt[0] % ns
                    cnt[0] += 1
                    with torch.cuda.stream(ss[j]):
                        ps[j].run(feats, coords)
                print(f"   pipe={pipe}, {ns} frames in flight: {wall(step, iters=300):6.2f} us/frame")
        lib.link_dc_set_tuning2(5, 1)
        return
    # frames in flight
    for ns in (1, 2, 3, 4):
        ps, ss = [], []
        for k in range(ns):
            q = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense")
            q.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
                   blk.norm.weight, blk.norm.bias)
            ps.append(q); ss.append(torch.cuda.Stream())
        cnt = [0]

        def step():
            j = cnt[0] % ns
            cnt[0] += 1
            with torch.cuda.stream(ss[j]):
                ps[j].run(feats, coords)
        print(f"dense, {ns} frames in flight: {wall(step, iters=300):6.2f} us/frame")


if __name__ == "__main__":
    main()
