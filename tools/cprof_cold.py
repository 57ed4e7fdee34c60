"""Host side of ELKBlock.forward on cfg2 with a new coordinate set every call (R_block cold): cProfile over 200 calls, by
internal time -- which Python / FFI / torch calls the ~230 us of host work per call are.   python tools/cprof_cold.py"""
import sys, time, torch, cProfile, pstats
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import link_amd as la
from helpers import s_uniform
dev = torch.device("cuda:0")
N, C = 100000, 64
coords = s_uniform(N, grid=256, seed=0).to(dev)
feats = torch.randn(N, C, device=dev)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
def cold():
    st = la.SparseTensor(feats, coords.clone(), 1)
    with torch.no_grad():
        blk(st, 7, 3)
for _ in range(10): cold()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): cold()
torch.cuda.synchronize(); print("cold wall %.1f us" % (1e6 * (time.perf_counter() - t0) / 200))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): cold()
torch.cuda.synchronize(); pr.disable()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(28)
