"""ELKBlock.forward on cfg2 on warm maps, in a loop (target of rocprofv3 passes: SCRIPT=tools/block_warm_loop.py bash tools/pmc_block.sh)."""
import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import link_amd as la
from helpers import s_uniform
dev = torch.device("cuda:0")
coords = s_uniform(100000, grid=256, seed=0).to(dev); feats = torch.randn(100000, 64, device=dev)
blk = la.ELKBlock(64, 64, groups=2, baseop="cos").to(dev).eval()
stb = la.SparseTensor(feats, coords, 1)
with torch.no_grad():
    blk(stb, 7, 3)
    for _ in range(60):
        st = la.SparseTensor(feats, coords, 1); st.kmaps, st.cmaps = stb.kmaps, stb.cmaps
        blk(st, 7, 3)
torch.cuda.synchronize()
