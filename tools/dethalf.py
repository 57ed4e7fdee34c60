#!/usr/bin/env python
"""tools/dethalf.py -- cfg5's sparse backbone half with fp16 / bf16 feature rows (BASELINE.json configs[4] is quoted
in fp16): agreement with the fp32 run and time per frame, kernel maps per call and warm."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import link_amd as la
from link_amd.synth import s_nusc
dev = torch.device("cuda", 0)

def ev(fn, k=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3

co, fe = s_nusc(0)
torch.manual_seed(0)
net = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
f32 = torch.from_numpy(fe).to(dev)
with torch.no_grad():
    ref, _ = net(f32, indices, 1, [1440, 1440, 40])
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        f = f32.to(dt)
        bev, scales = net(f, indices, 1, [1440, 1440, 40])
        err = float((bev.float() - ref).norm() / ref.norm())
        t_c = ev(lambda: net(f, indices, 1, [1440, 1440, 40]))
        maps = {}
        t_w = ev(lambda: net(f, indices, 1, [1440, 1440, 40], indice_dict=maps))
        print(f"{str(dt):16s} BEV dtype {bev.dtype}, rel. difference to fp32 {err:.2e}; {t_c:.2f} ms maps per call, {t_w:.2f} ms warm maps")
