"""R_block on cfg2 with every map rebuilt (a new coordinate set per call), through the one-call block driver and through the
per-stage module path, next to the warm-map figure: host wall time per call and device-event medians.
   python tools/block_driver_check.py"""
import sys, time, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import link_amd as la
from link_amd import elk
from helpers import s_uniform
dev = torch.device("cuda:0")
N, C = 100000, 64
coords = s_uniform(N, grid=256, seed=0).to(dev)
feats = torch.randn(N, C, device=dev)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
stb = la.SparseTensor(feats, coords, 1)
with torch.no_grad():
    blk(stb, 7, 3)
def call(warm):
    st = la.SparseTensor(feats, coords, 1)
    if warm:
        st.kmaps, st.cmaps = stb.kmaps, stb.cmaps
    with torch.no_grad():
        return blk(st, 7, 3).F
def med(fn, iters=40):
    for _ in range(5): fn()
    ev = []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); ev.append((a, b))
    torch.cuda.synchronize(); wall = 1e6 * (time.perf_counter() - t0) / iters
    v = sorted(1e3 * a.elapsed_time(b) for a, b in ev)
    return round(v[len(v) // 2], 1), round(wall, 1)
for drv in (True, False, True):
    elk.BLOCK_DRIVER = drv
    c0 = dict(elk.BLOCK_DRIVER_CALLS)
    cold, warm = med(lambda: call(False)), med(lambda: call(True))
    d = {k: elk.BLOCK_DRIVER_CALLS[k] - c0[k] for k in c0}
    print(f"driver={drv}: cold event-median {cold[0]} us, wall {cold[1]} us | warm {warm[0]} us, wall {warm[1]} us | ratio {cold[0] / warm[0]:.2f} | {d}")
elk.BLOCK_DRIVER = True
a = call(False); elk.BLOCK_DRIVER = False; b = call(False)
print("bitwise equal:", bool(torch.equal(a, b)), float((a - b).abs().max()))
