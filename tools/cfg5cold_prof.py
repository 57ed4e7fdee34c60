"""tools/cfg5cold_prof.py -- the sparse half of the detection backbone on the S-nusc frame with EVERY kernel map rebuilt per frame
(what bench.py --workload cfg5 reports as ms_per_step): wall time per frame, then either a cProfile of the host side (CPROF=1) or a
plain loop for a kernel trace (TAG=cfg5cold SCRIPT=tools/cfg5cold_prof.py bash tools/profile_cmd.sh: kernel time per frame against
the wall time = how much of the frame the GPU is busy)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import link_amd as la
from link_amd.synth import s_nusc

dev = torch.device("cuda:0")
co, fe = s_nusc(seed=0)
indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
feats = torch.from_numpy(fe).to(dev)
if os.environ.get("IO") == "f16":
    feats = feats.half()
torch.manual_seed(0)
net = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
if os.environ.get("IO") == "f16":
    net = net.half()
shape = [1440, 1440, 40]
K = int(os.environ.get("K", 20))
with torch.no_grad():
    for _ in range(5):
        net(feats, indices, 1, shape)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        net(feats, indices, 1, shape)
    torch.cuda.synchronize()
    print(f"cfg5 backbone, every map rebuilt: {1e3 * (time.perf_counter() - t0) / K:.3f} ms per frame, {indices.shape[0]} voxels")
    if os.environ.get("CPROF"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(K):
            net(feats, indices, 1, shape)
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(32)
