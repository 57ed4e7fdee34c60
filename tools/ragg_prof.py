"""R_agg on cfg2 through the drop-in surface (voxel_to_aux + aux_to_voxel on X[N,128], warm index) for a rocprofv3 pass:
which kernels the 65 us are.   rocprofv3 --kernel-trace --stats -d out -- python tools/ragg_prof.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import link_amd as la
from bench import s_uniform
dev = torch.device("cuda", 0)
N, C = 100000, 64
coords = s_uniform(N).to(dev)
x = torch.randn(N, 2 * C, generator=torch.Generator().manual_seed(1)).to(dev)
st0 = la.SparseTensor(x, coords, 1); la.voxel_to_aux(st0, 7); kcache, ccache = st0.kmaps, st0.cmaps
for _ in range(int(os.environ.get("ITERS", 200))):
    st = la.SparseTensor(x, coords, 1); st.kmaps = kcache; st.cmaps = ccache
    small, idx, counts = la.voxel_to_aux(st, 7)
    out = la.aux_to_voxel(small, st, idx, counts, 3).F
torch.cuda.synchronize()
