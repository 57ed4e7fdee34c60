"""tools/core_dbg.py -- where the general layout's four-kernel form of R_core loses accuracy against a float64 evaluation on a
cos_x LiDAR stage with large theta: compares its intermediates (fin, block-sum table S, normalised neighbour sums A, output) with
float64 ones built on the plan's own block numbering.   VARIANT=unet STAGE=3 python tools/core_dbg.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import torch.nn.functional as TF

import link_amd as la
from link_amd.elk import ElkCorePlan
from link_amd.index import coords_bounds
from oracle import link_oracle as O
from tools.lidar_core_parity import seg_stage_calls

dev = torch.device("cuda:0")
variant, stage = os.environ.get("VARIANT", "unet"), int(os.environ.get("STAGE", 3))
r = seg_stage_calls(dev, variant)[stage - 1]
b, coords, feats = r["blk"], r["coords"], r["feats"]
n, c = feats.shape
plan = ElkCorePlan(n, c, b.baseop, r["cg"], r["r"], r["s_eff"], coords_bounds(coords), dev, coord_div=r["coord_div"], layout="general", tiles=False)
plan.bind(b.pre_mix[0].weight, b.pre_mix[1].weight, b.pre_mix[1].bias, r["w_pos"], r["alpha"], b.norm.weight, b.norm.bias)
out = plan.run(feats, coords, build_index=True).clone().cpu()
m = plan.blocks()
P = 3
fin = plan.fin[:n].cpu()
S = plan.S[: plan.m_cap * P * c].view(plan.m_cap, P * c)[:m].cpu()
A = plan.A[:m].cpu()
blk = plan.vox_blk[:n].cpu().long()
p = {k: v.detach().cpu().double() for k, v in b.state_dict().items()}
f64 = TF.layer_norm(TF.linear(feats.cpu().double(), p["pre_mix.0.weight"]), (c,), p["pre_mix.1.weight"], p["pre_mix.1.bias"], 1e-6)
th64 = O.theta_torch(coords.cpu(), p["pos_weight.0.weight"], b.baseop, 1, p.get("alpha"), variant, r["stride"])
th32 = O.theta_torch(coords.cpu(), p["pos_weight.0.weight"].float(), b.baseop, 1, p["alpha"].float(), variant, r["stride"])
X64 = torch.cat([f64 * th64.cos(), f64 * th64.sin(), f64 * th64], 1)
S64 = torch.zeros(m, P * c, dtype=torch.float64).index_add_(0, blk, X64)
cnt = torch.bincount(blk, minlength=m).double()
# neighbour structure from the oracle's index on the plan's numbering: use coordinates of blocks
bc = plan.blk_coords[:m].cpu().numpy()
nbr = torch.from_numpy(O.neighbor_index(bc, r["r"]).astype("int64"))
w = (nbr >= 0).double()
num = (S64[nbr.clamp(min=0)] * w[..., None]).sum(1)
den = (cnt[nbr.clamp(min=0)] * w).sum(1)
A64 = num / den[:, None]
new64 = A64[blk][:, :c] * th64.cos() + A64[blk][:, c:2 * c] * th64.sin() + (A64[blk][:, 2 * c:] - f64 * th64)
out64 = TF.layer_norm(new64, (c,), p["norm.weight"], p["norm.bias"], 1e-6)


def rel(a, b_):
    return float((a.double() - b_).abs().max() / b_.abs().max())


print("theta max", float(th64.abs().max()), " theta fp32 vs fp64 max abs", float((th32.double() - th64).abs().max()))
print("fin rel", rel(fin, f64))
for k, nm in enumerate(("S cos", "S sin", "S lin")):
    print(nm, "rel", rel(S[:, k * c:(k + 1) * c], S64[:, k * c:(k + 1) * c]), " max|.|", float(S64[:, k * c:(k + 1) * c].abs().max()))
for k, nm in enumerate(("A cos", "A sin", "A lin")):
    print(nm, "rel", rel(A[:, k * c:(k + 1) * c], A64[:, k * c:(k + 1) * c]), " max|.|", float(A64[:, k * c:(k + 1) * c].abs().max()))
print("out rel64", rel(out, out64), " max|new64|", float(new64.abs().max()))
# which part of the output error comes from the linear term: recompute the output in float64 from the kernel's A and fin
A_ = A.double()[blk]
for nm, lin in (("kernel A, fp64 fin*theta", f64 * th64), ("kernel A, kernel fin * fp32 theta", fin.double() * th32.double()),
                ("kernel A, fp32 product", (fin * th32).double())):
    nw = A_[:, :c] * th64.cos() + A_[:, c:2 * c] * th64.sin() + (A_[:, 2 * c:] - lin)
    print("out from", nm, rel(TF.layer_norm(nw, (c,), p["norm.weight"], p["norm.bias"], 1e-6), out64))
