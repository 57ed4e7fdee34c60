#!/usr/bin/env python
"""tools/regions.py -- timings of the other regions of SURVEY.md section 8d on cfg2:
R_agg (voxel_to_aux + aux_to_voxel through the drop-in Python surface, X[N,128]) cold/warm, the
module-level fused R_core (ELKBlock._core, allocating path), and the differentiable R_core fwd+bwd."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
dev = torch.device("cuda", 0)
N, C = 100000, 64
coords = s_uniform(N).to(dev)
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev)
x = torch.randn(N, 2 * C, generator=torch.Generator().manual_seed(1)).to(dev)
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(3)).to(dev)
gout = torch.randn(N, C, generator=torch.Generator().manual_seed(4)).to(dev)

def timeit(fn, k=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

def r_agg(cached):
    st = la.SparseTensor(x, coords, 1)
    if cached: st.kmaps = kcache; st.cmaps = ccache
    small, idx, counts = la.voxel_to_aux(st, 7)
    return la.aux_to_voxel(small, st, idx, counts, 3).F
st0 = la.SparseTensor(x, coords, 1); la.voxel_to_aux(st0, 7); kcache, ccache = st0.kmaps, st0.cmaps
print(f"R_agg surface, cold (index + bbox sync per call): {timeit(lambda: r_agg(False), warm=40):.1f} us")   # first calls grow the allocator pools
print(f"R_agg surface, warm (index cached on kmaps):      {timeit(lambda: r_agg(True)):.1f} us")
blk.eval()
def core_infer():
    st = la.SparseTensor(feats, coords, 1); st.kmaps = kcache; st.cmaps = ccache
    with torch.no_grad():
        return blk._core(st, 7, 3, blk.pos_weight[0].weight, None, 32, 1.0)
print(f"R_core module path (allocating, warm index):      {timeit(core_infer):.1f} us")
blk.train()
def core_train():
    f = feats.detach().requires_grad_(True)
    st = la.SparseTensor(f, coords, 1); st.kmaps = kcache; st.cmaps = ccache
    out = blk._core(st, 7, 3, blk.pos_weight[0].weight, None, 32, 1.0)
    out.backward(gout)
print(f"R_core differentiable fwd+bwd (warm index):       {timeit(core_train, 20):.1f} us")
