import os, sys, cProfile, pstats
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
import link_amd as la
from link_amd.synth import s_nusc
dev = torch.device("cuda", 0)
co, fe = s_nusc(0)
c = co.copy(); c[:, :3] //= 4; c = np.unique(c, axis=0); n = c.shape[0]
indices = torch.from_numpy(c[:, [3, 2, 1, 0]].copy()).int().to(dev)
stage = la.ELKv3Stage(64).to(dev).eval()
feats = torch.randn(n, 64, device=dev)
sct = la.SparseConvTensor(feats, indices, [11, 360, 360], 1)
with torch.no_grad():
    for _ in range(5): stage(sct)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): stage(sct)
    torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
