import os, sys, ctypes, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, link_amd as la
from link_amd import _lib as L
from bench import s_uniform
dev = torch.device("cuda")
N, C = 100000, 64
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).to(dev)
coords = s_uniform(N, seed=0).to(dev)
p = la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, ((0, 0, 0, 0), (255, 255, 255, 0)), dev, layout="dense")
p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
lib = L.lib()
lib.link_dc_set_tuning2(6, 0)
ref = p.run(feats, coords).clone()
lib.link_dc_set_tuning2(6, 1)
got = p.run(feats, coords).clone()
got2 = p.run(feats, coords).clone()
torch.cuda.synchronize()
print("split vs plain max abs diff", float((got - ref).abs().max()), "equal:", bool(torch.equal(got, ref)), "repeat equal:", bool(torch.equal(got, got2)))
def wall(fn, k=200, w=20):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
for zs in (0, 4, 5, 6, 8, 10):
    lib.link_dc_set_tuning2(3, zs)
    for sp in (0, 1):
        lib.link_dc_set_tuning2(6, sp)
        print(f"zsplit={zs} split={sp}: step {wall(lambda: p.run(feats, coords)):.2f} us", flush=True)
