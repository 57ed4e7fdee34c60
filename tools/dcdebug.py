import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import link_amd as la
from helpers import rel_err, s_uniform
from oracle import link_oracle as O
C, groups, baseop, s, r, grid, n = 32, 2, "sin", 3, 2, 40, 6000
torch.manual_seed(5)
blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
coords = s_uniform(n, grid=grid, seed=C + r).cuda()
feats = torch.randn(n, C, generator=torch.Generator().manual_seed(3)).cuda()
bounds = ((0, 0, 0, 0), (grid - 1,) * 3 + (0,))
outs = {}
for layout in ("dense", "general"):
    p = la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, torch.device("cuda"), layout=layout)
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    outs[layout] = p.run(feats, coords).clone().cpu().numpy()
    if layout == "dense":
        cn = p.cell_n.cpu().numpy(); vcell = p.vcell.cpu().numpy()
params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
ref = O.elk_core_torch(feats.cpu(), coords.cpu(), params, s, r, baseop, groups, agg=O.aggregate_c).numpy()
for k, o in outs.items():
    print(k, "vs oracle", rel_err(o, ref))
d = np.abs(outs["dense"] - outs["general"]).max(1)
idx = np.argsort(-d)[:10]
print("worst rows", idx, d[idx], "cell counts", cn[vcell[idx]])
e = np.abs(outs["dense"] - ref).max(1); print("dense-oracle worst", np.sort(e)[-5:], "general-oracle worst", np.sort(np.abs(outs["general"] - ref).max(1))[-5:])
print("hist of counts", np.bincount(cn[cn > 0]))
