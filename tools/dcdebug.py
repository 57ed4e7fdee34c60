import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import link_amd as la
from link_amd import _lib as L
from helpers import rel_err, s_uniform
C, groups, baseop, s, r, grid, n = 64, 1, "cos_x", 3, 3, 30, 4000
torch.manual_seed(5)
blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
coords = s_uniform(n, grid=grid, seed=C + r).cuda()
feats = torch.randn(n, C, generator=torch.Generator().manual_seed(3)).cuda()
bounds = ((0, 0, 0, 0), (grid - 1,) * 3 + (0,))
lib = L.lib()
for mode, single in ((7, 0), (7, 1)):
    lib.link_dc_set_tuning(3, mode); lib.link_dc_set_tuning2(4, single)
    p = la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, torch.device("cuda"), layout="dense")
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, blk.alpha, blk.norm.weight, blk.norm.bias)
    o0 = p.run(feats, coords).clone()
    fin0 = p.fin.clone()
    bad = 0
    for it in range(6):
        o = p.run(feats, coords).clone()
        d = (o != o0).any(1)
        fd = (p.fin != fin0).any(1)
        if d.any() or fd.any():
            bad += 1
            idx = torch.nonzero(d).flatten()[:8].tolist()
            cn = p.cell_n[p.vcell[idx].long()].tolist() if idx else []
            print("mode", mode, "iter", it, "rows differing", int(d.sum()), "fin rows differing", int(fd.sum()), idx, "cell counts", cn,
                  "max abs diff", float((o - o0).abs().max()))
    print("mode", mode, "single", single, "bad iterations", bad)
