#!/usr/bin/env python
"""tools/detstage_bench.py -- cfg5's backbone half, one stage at a time: link_amd.ELKv3Stage (scn.py:477-494,586-590)
on the S-nusc frame, stage 1 (C=16, full resolution) and the C=32/64/128 widths on 2x/4x/8x
down-sampled coordinates; fused inference path vs module-by-module execution, warm maps."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
import link_amd as la
from link_amd.synth import s_nusc
dev = torch.device("cuda", 0)

def ev(fn, k=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6

co, fe = s_nusc(0)
print(f"S-nusc frame: {co.shape[0]} voxels")
for planes, ds in ((16, 1), (32, 2), (64, 4), (128, 8)):
    c = co.copy(); c[:, :3] //= ds
    c = np.unique(c, axis=0)
    n = c.shape[0]
    indices = torch.from_numpy(c[:, [3, 2, 1, 0]].copy()).int().to(dev)
    shape = [41 // ds + 1, 1440 // ds, 1440 // ds]
    torch.manual_seed(0)
    stage = la.ELKv3Stage(planes).to(dev).eval()
    feats = torch.randn(n, planes, device=dev)
    sct = la.SparseConvTensor(feats, indices, shape, 1)
    with torch.no_grad():
        stage(sct)
        nbr = list(v for k, v in sct.indice_dict.items() if k[0] == "link_ts")[0][3]
        tab = [v for k, v in nbr.items() if k[0] == "link_conv_nbr"][0][0]
        dens = float((tab >= 0).sum()) / n
        t_f = ev(lambda: stage(sct))
        t_m = ev(lambda: stage._modules_path(sct))
    print(f"stage C={planes:3d}: N={n:6d} neighbours/voxel={dens:5.2f}  fused {t_f:7.1f} us   module-by-module {t_m:7.1f} us")
