"""cfg2 frames as ONE batched step (batch index in coords[:, 3], as the reference's collate does) against the same frames as
separate steps in flight: ElkCorePlan on B x 100k voxels, dense-cell layout, index rebuilt every step; HIP events over
back-to-back steps.   python tools/batchstep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform

dev = torch.device("cuda:0")
N, C = 100000, 64
torch.manual_seed(0)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()


def run(B, tuning, iters=300):
    cs = []
    for b in range(B):
        c = s_uniform(N, seed=b).clone()
        c[:, 3] = b
        cs.append(c)
    coords = torch.cat(cs).to(dev).contiguous()
    feats = torch.randn(B * N, C, device=dev)
    plan = la.ElkCorePlan(B * N, C, "cos", 32, 3, 7, ((0, 0, 0, 0), (255, 255, 255, B - 1)), dev, layout="dense", **tuning)
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
    for _ in range(500):
        plan.run(feats, coords, build_index=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        plan.run(feats, coords, build_index=True)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters / B


for B in (1, 2, 3, 4, 6):
    for tuning in ({}, {"k1_wgs": 512}, {"k2_zsplit": 1}, {"k1_wgs": 512, "k2_zsplit": 1}):
        try:
            print(B, tuning, "%.2f us/frame" % run(B, tuning), flush=True)
        except Exception as e:
            print(B, tuning, "ERR", str(e)[:100], flush=True)
