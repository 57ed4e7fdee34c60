"""Phase ticks of the channel-split form of the lean form's first launch (library built with LINK_AMD_CXXFLAGS=-DLEAN_DBG):
mean ticks per workgroup for loads | contraction | LayerNorm | theta + stores, on the LiDAR stage frames.  STAGE=<k>."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import link_amd as la
from link_amd.elk import ElkCorePlan
from link_amd.index import coords_bounds
from tools.lidar_core import stages

dev = torch.device("cuda:0")
for k, (cfg, r) in enumerate(stages(dev)):
    if os.environ.get("STAGE") and int(os.environ["STAGE"]) != k:
        continue
    b, coords, feats = r["blk"], r["coords"], r["feats"]
    n, c = feats.shape
    cap = min(max(1, r["s_eff"] // max(r["stride"], 1)) ** 3, 343)
    plan = ElkCorePlan(n, c, b.baseop, r["cg"], r["r"], r["s_eff"], coords_bounds(coords), dev, coord_div=r["coord_div"], layout="lean", slot_cap=cap)
    plan.bind(b.pre_mix[0].weight, b.pre_mix[1].weight, b.pre_mix[1].bias, r["w_pos"], r["alpha"], b.norm.weight, b.norm.bias)
    for _ in range(20):
        plan.run(feats, coords)
    torch.cuda.synchronize()
    plan.hdr[16:].zero_()
    for _ in range(10):
        plan.run(feats, coords)
    torch.cuda.synchronize()
    h = plan.hdr.tolist()
    cnt = max(h[20], 1)
    print(k, n, c, "wgs/launch", cnt / 10, "ticks per wg: loads %.0f mfma %.0f ln %.0f trig+stores %.0f" % tuple(16 * h[16 + j] / cnt for j in range(4)))
    ci = max(h[29], 1)
    print("   gather: items/launch", ci / 10, "ticks per item: item fetch %.0f | records + counts + first rows %.0f | further rows %.0f | A %.0f | xl wait %.0f | voxel loop %.0f | store wait %.0f | steps %.2f"
          % (tuple(256 * h[j] / ci for j in (24, 25, 26, 27, 28, 30, 31)) + (h[32] / ci,)))
