"""Would a finer z-split of the gather kernel pay with frames in flight if it did not overflow the CUs?  cfg2-like frames on a
252^3 grid (36 blocks per axis = 9 x 9 tiles of 4 x 4 columns, no rim tiles): 2 z-segments = 162 workgroups, 3 = 243 (<= 256 CUs),
4 = 324; on the 256^3 grid of cfg2 (10 x 10 tiles, 19 of them rim tiles) 3 segments are 300 workgroups.   python tools/zsplit_probe.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import link_amd as la
from helpers import s_uniform
dev = torch.device("cuda:0")
torch.manual_seed(2)
blk = la.ELKBlock(64, 64, groups=2, baseop="cos").to(dev).eval()
def run(grid, n, zs, k1w=256):
    NS = 3
    plans, frames, streams = [], [], []
    for k in range(NS):
        frames.append((torch.randn(n, 64, generator=torch.Generator().manual_seed(1 + k)).to(dev), s_uniform(n, grid=grid, seed=k).to(dev)))
        pl = la.ElkCorePlan(n, 64, "cos", 32, 3, 7, ((0, 0, 0, 0), (grid - 1, grid - 1, grid - 1, 0)), dev, frames_in_flight=NS, k2_zsplit=zs, k1_wgs=k1w)
        pl.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight, blk.norm.bias)
        plans.append(pl); streams.append(torch.cuda.Stream(device=dev))
    def region(k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k):
            for j in range(NS):
                with torch.cuda.stream(streams[j]):
                    plans[j].run(frames[j][0], frames[j][1])
        torch.cuda.synchronize()
        return 1e6 * (time.perf_counter() - t0) / (k * NS)
    for _ in range(5): region(200)
    v = sorted(region(400) for _ in range(7))
    return v[3], v[0]
for grid, n in ((252, 95400), (256, 100000)):
    for zs in (2, 3, 4):
        m, lo = run(grid, n, zs)
        print(f"grid {grid} n {n} z-segments {zs}: {m:.2f} us/frame (best {lo:.2f})")
