#!/usr/bin/env python
"""tools/pairwarm.py -- pair-list convolution with the LayerNorm tail, C=64, on cfg2 / LiDAR-like / S-kitti frames (warm maps),
30 calls each, for rocprofv3 kernel stats (tools/ab_variants.sh)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import link_amd as la
from link_amd.elk import subm_conv_ln_add_relu
from link_amd.synth import s_kitti
from bench import s_uniform
from helpers import lidar_like
dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
coords = {"cfg2": lambda: s_uniform(100000), "lidar": lambda: torch.from_numpy(lidar_like(120000, seed=0)),
          "kitti": lambda: torch.from_numpy(s_kitti(0)[0])}[which]().to(dev)
C = 64
n = coords.shape[0]
conv = la.Conv3d(C, C, 3).to(dev)
feats = torch.randn(n, C, device=dev)
st = la.SparseTensor(feats, coords, 1)
nbr, order = conv._neighbor_table(st)
w = conv.kernel.detach()
lw, lb, add = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.randn(n, C, device=dev)
for _ in range(30):
    subm_conv_ln_add_relu(feats, w, nbr, order, lw, lb, 1e-6, add, form="pairs")
torch.cuda.synchronize()
