"""tools/lidar_core_parity.py -- R_core ALONE on the full-size S-kitti stage frames against the oracle and against a float64
evaluation of the same formula ("What's weak" 1: nothing arbitrated between the forms).

For both segmentation variants (linkunet.py:165 theta on stride-multiplied coordinates; linkencoder.py:165 theta on coords / stride)
it captures what each of the four ELKBlock._core calls of a forward on S-kitti seed 0 receives, then evaluates the core through the
forms of ElkCorePlan (tile form, four-kernel form, lean form) and prints per stage and form

    rel32 = max|HIP - oracle_fp32| / max|oracle_fp32|       (north_star's measure; the oracle is torch fp32 + the C aggregation)
    rel64 = max|HIP - fp64| / max|fp64|                      (same formula in float64: theta, sin / cos, sums, LayerNorm)
    o64   = max|oracle_fp32 - fp64| / max|fp64|              (what fp32 conditioning alone costs: theta of a few thousand radians)

A form is RIGHT when rel64 <= o64 (+ a little): it is then as close to the exact result as the fp32 reference itself.
    python tools/lidar_core_parity.py            # one JSON line per (variant, stage)
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
from link_amd.synth import s_kitti
from oracle import link_oracle as O
from tools.lidar_core import capture


def seg_stage_calls(dev, variant, seed=0):
    co, fe = s_kitti(seed=seed)
    coords, feats = torch.from_numpy(co).to(dev), torch.from_numpy(fe).to(dev)
    torch.manual_seed(0)
    if variant == "encoder":
        net = la.fuse_for_inference(LE.build_reference_shaped_encoder(la, 64, "cos_x", 1)).to(dev).eval()
        run = lambda: net(la.SparseTensor(feats, coords, 1), 3, 2)
    else:
        net = la.fuse_for_inference(LE.build_reference_shaped_unet(la, 1.0, "cos_x", 1, 3, 2)).to(dev).eval()
        run = lambda: net(la.SparseTensor(feats, coords, 1))
    with torch.no_grad():
        return capture([net.elk1, net.elk2, net.elk3, net.elk4], run)


def core_refs(r, variant):
    """(oracle fp32 through the C aggregation, float64 evaluation of the same formula) for one captured call."""
    b = r["blk"]
    params = {k: v.detach().cpu() for k, v in b.state_dict().items()}
    feats, coords = r["feats"].cpu(), r["coords"].cpu()
    groups = b.groups if hasattr(b, "groups") else 1
    ref32 = O.elk_core_torch(feats, coords, params, r["s_eff"], r["r"], b.baseop, groups, variant=variant, tensor_stride=r["stride"],
                             agg=O.aggregate_c)
    p64 = {k: v.double() for k, v in params.items()}
    ref64 = O.elk_core_torch(feats.double(), coords, p64, r["s_eff"], r["r"], b.baseop, groups, variant=variant,
                             tensor_stride=r["stride"])
    return ref32, ref64


def forms_of(r, dev):
    b, coords, feats = r["blk"], r["coords"], r["feats"]
    n, c = feats.shape
    cap = max(1, r["s_eff"] // max(r["stride"], 1)) ** 3
    out = {}
    for name, kw in (("tiles", dict(layout="general", tiles=True)), ("four", dict(layout="general", tiles=False)),
                     ("lean", dict(layout="lean", slot_cap=min(cap, 343)))):
        try:
            plan = ElkCorePlan(n, c, b.baseop, r["cg"], r["r"], r["s_eff"], coords_bounds(coords), dev, coord_div=r["coord_div"], **kw)
        except la._lib.LinkAmdError:
            continue
        plan.bind(b.pre_mix[0].weight, b.pre_mix[1].weight, b.pre_mix[1].bias, r["w_pos"], r["alpha"], b.norm.weight, b.norm.bias)
        out[name] = plan.run(feats, coords, build_index=True).clone().cpu()
        plan.check()
    return out


def main():
    dev = torch.device("cuda:0")
    for variant in ("encoder", "unet"):
        for k, r in enumerate(seg_stage_calls(dev, variant)):
            ref32, ref64 = core_refs(r, variant)
            s64 = float(ref64.abs().max())
            row = dict(variant=variant, stage=k + 1, n=int(r["feats"].shape[0]), s_eff=r["s_eff"], stride=r["stride"],
                       theta_max=float(O.theta_torch(r["coords"].cpu(), r["blk"].pos_weight[0].weight.detach().cpu(), r["blk"].baseop, 1,
                                                     r["blk"].alpha.detach().cpu() if r["blk"].baseop == "cos_x" else None, variant,
                                                     r["stride"]).abs().max()),
                       o64=float((ref32.double() - ref64).abs().max() / s64))
            for name, got in forms_of(r, dev).items():
                row[name + "_rel32"] = float((got - ref32).abs().max() / ref32.abs().max())
                row[name + "_rel64"] = float((got.double() - ref64).abs().max() / s64)
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
