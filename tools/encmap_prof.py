import os, sys, cProfile, pstats, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch, link_amd as la, link_encoder as LE
from link_amd.synth import s_kitti
dev = torch.device("cuda", 0)
co, fe = s_kitti(0)
coords, feats = torch.from_numpy(co).to(dev), torch.from_numpy(fe).to(dev)
torch.manual_seed(0)
net = la.fuse_for_inference(LE.build_reference_shaped_encoder(la, 64, "cos_x", 1)).to(dev).eval()
def cold():
    with torch.no_grad(): net(la.SparseTensor(feats, coords, 1), 3, 2)
for _ in range(3): cold()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): cold()
torch.cuda.synchronize(); print("cold encoder forward ms:", (time.perf_counter() - t0) / 10 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): cold()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(40)
