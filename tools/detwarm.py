#!/usr/bin/env python
"""tools/detwarm.py [f16] -- cfg5's sparse backbone half on warm kernel maps, 20 frames, for `rocprofv3 --kernel-trace --stats`."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import link_amd as la
from link_amd.synth import s_nusc
dev = torch.device("cuda", 0)
co, fe = s_nusc(0)
torch.manual_seed(0)
net = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
f = torch.from_numpy(fe).to(dev)
if "f16" in sys.argv: f = f.half()
maps = {}
with torch.no_grad():
    for _ in range(23):
        net(f, indices, 1, [1440, 1440, 40], indice_dict=maps)
torch.cuda.synchronize()
