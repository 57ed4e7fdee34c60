"""Where ELKBlock.forward spends its time on cfg2 when every map is rebuilt (R_block cold): wall time incl. the syncs each
builder contains (nested: the convolution tail includes its pair plan).   python tools/rblock_cold.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo")
import link_amd as la
from link_amd import elk, index, aggregate
sys.path.insert(0, "/root/repo/tests")
from helpers import s_uniform
acc = {}
def timed(mod, name, label):
    f0 = getattr(mod, name)
    def f(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = f0(*a, **k)
        torch.cuda.synchronize()
        d = acc.setdefault(label, [0.0, 0]); d[0] += time.perf_counter() - t0; d[1] += 1
        return r
    setattr(mod, name, f)
dev = torch.device("cuda:0")
N, C = 100000, 64
coords = s_uniform(N, grid=256, seed=0).to(dev)
feats = torch.randn(N, C, device=dev)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
def cold():
    st = la.SparseTensor(feats, coords.clone(), 1)
    with torch.no_grad():
        blk(st, 7, 3)
for _ in range(5): cold()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): cold()
torch.cuda.synchronize(); print("R_block cold wall %.1f us" % (1e6 * (time.perf_counter() - t0) / 20))
timed(index, "coords_bounds", "coords_bounds")
timed(elk, "neighbor_table_of", "neighbor table")
timed(elk._PairPlan, "__init__", "pair plan")
timed(elk._ELKBase, "_core_dense", "_core_dense (probe + plan.run)")
timed(elk, "subm_conv_ln_add_relu", "conv + tail")
for _ in range(10): cold()
for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    print("  %-34s %7.1f us per call, %d calls per block" % (k, 1e6 * t / c, c // 10))
