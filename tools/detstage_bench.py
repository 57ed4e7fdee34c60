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

# whole sparse half of the backbone (scn.py:570-626) on the full S-nusc frame: 5 input features, grid 1440 x 1440 x 40
net = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
feats = torch.from_numpy(fe).to(dev)
with torch.no_grad():
    bev, scales = net(feats, indices, 1, [1440, 1440, 40])
    sizes = [scales[f"conv{k}"].features.shape[0] for k in (1, 2, 3, 4)]
    t_f = ev(lambda: net(feats, indices, 1, [1440, 1440, 40]), k=10)
    maps = {}
    t_w = ev(lambda: net(feats, indices, 1, [1440, 1440, 40], indice_dict=maps), k=10)
net.train(False)
def mods():
    with torch.no_grad():
        x = la.SparseConvTensor(feats, indices, [41, 1440, 1440], 1)
        from link_amd import detstage as D
        x = D._seq_conv_bn(net.conv_input, x, True)
        for k in (1, 2, 3, 4):
            if k > 1: x = D._seq_conv_bn(getattr(net, f"down{k}"), x, True)
            x = D._stage_modules(getattr(net, f"conv{k}"), getattr(net, f"conv{k}_tail"), getattr(net, f"elk{k}"),
                                 getattr(net, f"elk{k}_tail"), getattr(net, f"act{k}"), x, 7)
        return D.to_dense(D._seq_conv_bn(net.extra_conv, x, True))
t_m = ev(mods, k=10)
print(f"backbone sparse half: stage voxels {sizes}, BEV {tuple(bev.shape)}: fused {t_f / 1e3:.2f} ms with the kernel maps built per call (as the reference does), {t_w / 1e3:.2f} ms on warm maps; module-by-module {t_m / 1e3:.2f} ms (maps per call)")
