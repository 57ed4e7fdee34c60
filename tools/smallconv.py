#!/usr/bin/env python
"""tools/smallconv.py -- the 3^3 convolution on the small stages of an encoder (2-5 k voxels, C = 64): per-call GPU
time of the two forms (run under rocprofv3 --kernel-trace --stats for kernel durations; wall = host-bound here)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
import link_amd as la
from link_amd.elk import subm_conv
from link_amd.synth import s_kitti
dev = torch.device("cuda", 0)
co, _ = s_kitti(0)
for ds in (8, 16):
    c = co.copy(); c[:, :3] = c[:, :3] // ds; c = np.unique(c, axis=0)
    n = c.shape[0]
    coords = torch.from_numpy(c).int().to(dev)
    conv = la.Conv3d(64, 64, 3).to(dev)
    feats = torch.randn(n, 64, device=dev)
    st = la.SparseTensor(feats, coords, 1)
    nbr, order = conv._neighbor_table(st)
    w = conv.kernel.detach()
    for form in ("table", "pairs"):
        for _ in range(20): subm_conv(feats, w, nbr, order, form=form)
        torch.cuda.synchronize()
    print(f"stage /{ds}: N={n} neighbours/voxel={float((nbr >= 0).sum()) / n:.2f} pairs rows_pad={nbr._link_pairs.rows_pad}")
