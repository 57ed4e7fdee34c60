"""R_core on LiDAR-shaped stage frames (cfg3: S-kitti encoder stages, cfg5: S-nusc detection stages), outside the
modules: captures the (coords, C, op, s_eff, r, cg, coord_div) every LinK block of the two networks is called with,
then times ElkCorePlan steps on each (one FFI call per step, preallocated arena) with HIP events over back-to-back
steps, cold (index rebuilt) and warm, next to the module path's host-inclusive time.  STAGE=<k> restricts the run to
one stage (for a rocprofv3 pass: kernel names are shared by the stages); LAYOUT=general|sparse picks the plan layout;
FORM=four|tiles|sparse|lean restricts the run to one form (lean: link_elk_core_lean_forward, three launches rebuilt).

    python tools/lidar_core.py                  # table over the 8 stages
    STAGE=4 ITERS=200 python tools/lidar_core.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import link_amd as la
from harness import networks as LE
from link_amd.elk import ElkCorePlan
from link_amd.index import coords_bounds
from link_amd.synth import s_kitti, s_nusc


def capture(blocks, step):
    rec, saved = [], [b._core for b in blocks]

    def wrap(b, f0):
        def core(st, s_eff, r, w_pos, alpha, cg, coord_div):
            rec.append(dict(blk=b, coords=st.C.contiguous(), feats=st.F.contiguous().float(), s_eff=int(s_eff), r=int(r), w_pos=w_pos,
                            alpha=alpha, cg=int(cg), coord_div=float(coord_div), stride=int(st.s[0]) if hasattr(st, "s") else 1))
            return f0(st, s_eff, r, w_pos, alpha, cg, coord_div)
        return core
    for b, f0 in zip(blocks, saved):
        b._core = wrap(b, f0)
    try:
        step()
    finally:
        for b, f0 in zip(blocks, saved):
            b._core = f0
    return rec


def stages(dev):
    out = []
    co, fe = s_kitti(seed=0)
    coords, feats = torch.from_numpy(co).to(dev), torch.from_numpy(fe).to(dev)
    torch.manual_seed(0)
    net = la.fuse_for_inference(LE.build_reference_shaped_encoder(la, 64, "cos_x", 1)).to(dev).eval()
    with torch.no_grad():
        out += [("cfg3", r) for r in capture([net.elk1, net.elk2, net.elk3, net.elk4],
                                              lambda: net(la.SparseTensor(feats, coords, 1), 3, 2))]
    co, fe = s_nusc(seed=0)
    indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
    f5 = torch.from_numpy(fe).to(dev)
    torch.manual_seed(0)
    det = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
    with torch.no_grad():
        out += [("cfg5", r) for r in capture([det.elk1, det.elk2, det.elk3, det.elk4],
                                              lambda: det(f5, indices, 1, [1440, 1440, 40]))]
    return out


def ev_time(fn, iters):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    iters = int(os.environ.get("ITERS", 100))
    only = os.environ.get("STAGE")
    layout = os.environ.get("LAYOUT", "general")
    rows = []
    for k, (cfg, r) in enumerate(stages(dev)):
        if only is not None and int(only) != k:
            continue
        b, coords, feats = r["blk"], r["coords"], r["feats"]
        n, c = feats.shape
        bounds = coords_bounds(coords)
        st = la.SparseTensor(feats, coords, 1)
        parts = 3 if b.baseop == "cos_x" else 2
        row = dict(stage=k, cfg=cfg, n=n, c=c, op=b.baseop, s=r["s_eff"], r=r["r"])
        with torch.no_grad():
            ref = b._core(st, r["s_eff"], r["r"], r["w_pos"], r["alpha"], r["cg"], r["coord_div"]).float()
            row["module_warm_us"] = round(ev_time(lambda: b._core(st, r["s_eff"], r["r"], r["w_pos"], r["alpha"], r["cg"], r["coord_div"]), iters), 1)
            cap = max(1, r["s_eff"] // max(r["stride"], 1)) ** 3      # voxel sites of a block at this tensor stride
            for name, kw in (("four", dict(tiles=False)), ("tiles", dict(tiles=True)), 
                             ("lean", dict(layout="lean", slot_cap=min(cap, 343), **({"lean_cs": bool(int(os.environ["CS"]))} if os.environ.get("CS") else {}),
                                           **({"lean_pm": bool(int(os.environ["PM"]))} if os.environ.get("PM") else {})))):
                if os.environ.get("FORM", name) != name:
                    continue
                if name == "sparse" and (cap > 64 or c not in (16, 32, 64)):
                    continue                                 # big blocks stay on the general layout's tile form
                kw = dict(kw)
                if os.environ.get("ORDER") and name not in ("sparse", "lean"):
                    kw["block_order"] = os.environ["ORDER"]
                try:
                    plan = ElkCorePlan(n, c, b.baseop, r["cg"], r["r"], r["s_eff"], bounds, dev, coord_div=r["coord_div"],
                                       layout=kw.pop("layout", layout), **kw)
                except la._lib.LinkAmdError as e:
                    row[name + "_skipped"] = str(e)[:80]
                    continue
                plan.bind(b.pre_mix[0].weight, b.pre_mix[1].weight, b.pre_mix[1].bias, r["w_pos"], r["alpha"], b.norm.weight, b.norm.bias)
                got = plan.run(feats, coords, build_index=True).clone()
                m = plan.blocks()
                alg = n * 16 + 2 * n * 4 * c + 2 * m * 4 * (parts * c + 1)
                t_cold = ev_time(lambda: plan.run(feats, coords, build_index=True), iters)
                t_warm = ev_time(lambda: plan.run(feats, coords, build_index=False), iters)
                again = plan.run(feats, coords, build_index=False)
                row.update({"m": m, "cells": int(plan.grid.cells), "slot_cap": cap, "roof_us": round(alg / 8e12 * 1e6, 2),
                            name + "_cold_us": round(t_cold, 1), name + "_warm_us": round(t_warm, 1),
                            name + "_frac_warm": round(alg / 8e12 * 1e6 / t_warm, 4), name + "_err": float((got - ref).abs().max()),
                            name + "_repeat_bitwise": bool(torch.equal(got, again))})
        rows.append(row)
        print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    main()
