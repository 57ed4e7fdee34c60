"""debug: quad-consumer K2 (k2_form 0) vs pair form (8): where do the outputs differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
dev = torch.device("cuda")
N, C = 100000, 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
coords = s_uniform(N, seed=0).to(dev)
bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
outs = {}
for form in (8, 0):
    p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k2_form=form, k2_zsplit=int(os.environ.get("ZS", 0)))
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    p.out.zero_()
    outs[form] = p.run(feats, coords).clone()
    cell_n = p.cell_n.clone(); vcell = p.vcell.clone(); g = p.dcg
d = (outs[0] - outs[8]).abs()
bad = d.max(1).values > 1e-4
print("bad voxels", int(bad.sum()), "of", N, "nan", int(torch.isnan(outs[0]).any(1).sum()), "zero rows", int((outs[0].abs().sum(1) == 0).sum()))
cb = torch.div(coords[:, :3], 7, rounding_mode="floor")
print("bad by z-plane of cell:", torch.bincount(cb[bad][:, 2], minlength=37).tolist())
print("bad by x of cell:", torch.bincount(cb[bad][:, 0], minlength=37).tolist())
print("bad by y of cell:", torch.bincount(cb[bad][:, 1], minlength=37).tolist())
nvox = cell_n[vcell.long()]
print("bad by voxels-in-cell:", torch.bincount(nvox[bad], minlength=10).tolist(), "all:", torch.bincount(nvox, minlength=10).tolist())
ch = d[bad].max(0).values
print("max diff per channel (bad voxels):", [round(float(x), 2) for x in ch.tolist()])
i = int(torch.nonzero(bad)[0]) if bad.any() else 0
print("voxel", i, "got", outs[0][i, :8].tolist(), "want", outs[8][i, :8].tolist())
# rank of each bad voxel inside its cell (ids ascending) and the cell's count; position of the cell in its 4x4 tile
order = torch.argsort(vcell.long() * N + torch.arange(N, device=dev))
vs = vcell[order]
first = torch.ones(N, dtype=torch.bool, device=dev); first[1:] = vs[1:] != vs[:-1]
idx = torch.arange(N, device=dev)
start = torch.cummax(torch.where(first, idx, torch.zeros_like(idx)), 0).values
rank = torch.empty(N, dtype=torch.long, device=dev); rank[order] = idx - start
import collections
cnt = collections.Counter()
for r_, n_ in zip(rank[bad].tolist(), nvox[bad].tolist()):
    cnt[(r_, n_)] += 1
print("bad (rank, cell count):", sorted(cnt.items())[:40])
# plane-level: total voxels in the (tile, plane) of each bad voxel and the voxel's global index v inside the plane's list
tile = (cb[:, 0] // 4) * 10 + (cb[:, 1] // 4)
cpos = (cb[:, 0] % 4) * 4 + (cb[:, 1] % 4)
key = (tile * 37 + cb[:, 2]).long()
tot = torch.bincount(key, minlength=100 * 37)
print("bad by plane total:", torch.bincount(tot[key][bad], minlength=70).tolist())
