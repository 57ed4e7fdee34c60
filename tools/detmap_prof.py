import os, sys, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import link_amd as la
from link_amd.synth import s_nusc
dev = torch.device("cuda", 0)
co, fe = s_nusc(0)
net = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
feats = torch.from_numpy(fe).to(dev)
with torch.no_grad():
    maps = {} if os.environ.get("WARM") else None
    for _ in range(3): net(feats, indices, 1, [1440, 1440, 40], indice_dict=maps)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10): net(feats, indices, 1, [1440, 1440, 40], indice_dict=maps)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
