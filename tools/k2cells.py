"""Dense-cell layout at the widths other than 64: the two-kernel gather (box sums -> A, per-voxel de-modulate) against the fused
cells form (mode bit 3, dense_gather_cells_impl.h).   python tools/k2cells.py"""
import sys, torch
sys.path.insert(0, "/root/repo")
import link_amd as la
sys.path.insert(0, "/root/repo/tests")
from helpers import s_uniform
def ev(fn, it=200):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / it
for C, n, grid, s in ((16, 10000, 256, 7), (32, 60000, 200, 7), (128, 30000, 120, 7), (16, 100000, 256, 7)):
    torch.manual_seed(0)
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    coords = s_uniform(n, grid=grid, seed=1).cuda()
    feats = torch.randn(n, C, device="cuda")
    bounds = ((0, 0, 0, 0), (grid - 1, grid - 1, grid - 1, 0))
    outs = {}
    for name, mode in (("two-kernel gather", 7), ("cells form", 15)):
        p = la.ElkCorePlan(n, C, "cos", C // 2, 3, s, bounds, feats.device, layout="dense", mode=mode)
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
        outs[name] = p.run(feats, coords).clone()
        print(C, n, name, "cold %.1f us, warm %.1f us" % (ev(lambda: p.run(feats, coords)), ev(lambda: p.run(feats, coords, build_index=False))))
    print("   max diff", float((outs["two-kernel gather"] - outs["cells form"]).abs().max()))
