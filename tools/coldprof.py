import os, sys, cProfile, pstats, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, link_amd as la
from bench import s_uniform
dev = torch.device("cuda", 0)
torch.manual_seed(0)
blk = la.ELKBlock(64, 64, groups=2, baseop="cos").to(dev).eval()
coords = s_uniform(100000).to(dev); feats = torch.randn(100000, 64, device=dev)
def cold():
    with torch.no_grad(): blk(la.SparseTensor(feats, coords, 1), 7, 3)
for _ in range(3): cold()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): cold()
torch.cuda.synchronize(); print("R_block cold wall us:", (time.perf_counter() - t0) / 20 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(20): cold()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
