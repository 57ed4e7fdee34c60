import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import link_amd as la
from link_amd import _lib as L
from helpers import rel_err, s_uniform
C, groups, baseop, s, r, grid, n = 64, 2, "cos", 7, 3, 80, 9000
torch.manual_seed(5)
blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
coords = s_uniform(n, grid=grid, seed=C + r).cuda()
feats = torch.randn(n, C, generator=torch.Generator().manual_seed(3)).cuda()
bounds = ((0, 0, 0, 0), (grid - 1,) * 3 + (0,))
lib = L.lib()
res = {}
for mode in (0, 2, 3):
    lib.link_dc_set_tuning(3, mode)
    p = la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, torch.device("cuda"), layout="dense")
    p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    out = p.run(feats, coords).clone()
    torch.cuda.synchronize()
    res[mode] = dict(out=out.cpu().numpy(), S=p.S.clone().cpu().numpy(), A=p.A.clone().cpu().numpy(), cell_n=p.cell_n.clone().cpu().numpy(), vcell=p.vcell.clone().cpu().numpy())
for mode in (2, 3):
    for k in ("S", "A", "out"):
        a, b = res[mode][k], res[0][k]
        print("mode", mode, k, "rel err vs mode 0:", rel_err(a, b))
    print("cell_n equal", np.array_equal(res[mode]["cell_n"], res[0]["cell_n"]), "vcell equal", np.array_equal(res[mode]["vcell"], res[0]["vcell"]))
S3, S0 = res[3]["S"], res[0]["S"]
bad = np.where(np.abs(S3 - S0).max(1) > 1e-4)[0]
print("bad S rows", len(bad), bad[:20], "counts there", res[0]["cell_n"][bad[:20]])
if len(bad):
    r0 = bad[0]; print(S3[r0][:8], S0[r0][:8]); print(S3[r0][64:72], S0[r0][64:72])
o2, o0 = res[2]["out"], res[0]["out"]
d = np.abs(o2 - o0).max(1)
print("rows wrong:", (d > 1e-3).sum(), "of", len(d), "first wrong rows", np.where(d > 1e-3)[0][:16])
i = int(np.where(d > 1e-3)[0][0]) if (d > 1e-3).any() else 0
print("row", i, o2[i][:8], o0[i][:8])
print("chan err profile", np.abs(o2 - o0).max(0)[:64].round(3))
# is o2 row i equal to o0 row j for some neighbour?
for j in (i - 1, i + 1, i ^ 1):
    if 0 <= j < len(d): print("vs row", j, np.abs(o2[i] - o0[j]).max())
