#!/usr/bin/env python
"""tools/trainprof.py -- loop the differentiable R_core (fwd+bwd, cfg2, warm index) for a kernel trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
dev = torch.device("cuda", 0)
N, C = 100000, 64
coords = s_uniform(N).to(dev)
torch.manual_seed(2)
blk = la.ELKBlock(C, C, groups=2, baseop="cos").to(dev).train()
feats = torch.randn(N, C, generator=torch.Generator().manual_seed(3)).to(dev)
st0 = la.SparseTensor(feats, coords, 1); la.voxel_to_aux(st0, 7); kc, cc = st0.kmaps, st0.cmaps
gout = torch.randn(N, C, device=dev)
def step():
    f = feats.detach().requires_grad_(True)
    st = la.SparseTensor(f, coords, 1); st.kmaps = kc; st.cmaps = cc
    out = blk._core(st, 7, 3, blk.pos_weight[0].weight, None, 32, 1.0)
    out.backward(gout)
K = int(os.environ.get("K", 30))
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(K): step()
torch.cuda.synchronize(); print(f"fwd+bwd {(time.perf_counter()-t0)/K*1e6:.1f} us/step")
