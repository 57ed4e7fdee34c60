"""The convolution's finish kernel (k_conv_centre_sum) on the cfg2 block: device-event time of subm_conv_ln_add_relu on warm maps
(pair GEMM of a few granules + the finish kernel).   python tools/centre_ab.py"""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import link_amd as la
from link_amd import elk
from helpers import s_uniform
dev = torch.device("cuda:0")
for n, grid, C in ((100000, 256, 64), (30000, 96, 64)):
    coords = s_uniform(n, grid=grid, seed=0).to(dev); feats = torch.randn(n, C, device=dev); add = torch.randn(n, C, device=dev)
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).eval()
    st = la.SparseTensor(feats, coords, 1); conv = blk.local_mix[0]
    nbr, order = conv._neighbor_table(st)
    def f(): return elk.subm_conv_ln_add_relu(feats, conv.kernel, nbr, order, blk.norm_local.weight, blk.norm_local.bias, 1e-6, add)
    for _ in range(20): f()
    torch.cuda.synchronize()
    ev = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); ev.append((a, b))
    torch.cuda.synchronize()
    v = sorted(1e3 * a.elapsed_time(b) for a, b in ev)
    print(f"n={n} C={C}: conv finish (GEMM + centre_sum) median {v[100]:.1f} us, min {v[0]:.1f}")
