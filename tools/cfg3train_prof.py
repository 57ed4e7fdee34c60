"""tools/cfg3train_prof.py -- the cfg3 training step (encoder common to both segmentation models on one full-size S-kitti frame,
forward + backward of the sum-of-squares loss on stage 4, warm kernel maps: what bench.py --workload cfg3 reports as fwd_bwd_ms)
in a loop for a kernel trace:   TAG=cfg3train SCRIPT=tools/cfg3train_prof.py LINES=60 bash tools/profile_cmd.sh
Prints wall time per step and, with HIP events around the loop, the GPU-busy share a trace can be checked against."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import link_amd as la
from harness import networks as LE
from link_amd.synth import s_kitti

from link_amd import elk as E

if os.environ.get("BN_FUSE") == "0":                   # A/B: the BatchNorm + ReLU passes of round 5 off (torch elementwise kernels)
    from link_amd import modules as _M
    _M.BatchNorm._hip_train_ok_orig = _M.BatchNorm._hip_train_ok
    _M._FusedSequential.forward = lambda self, input: _plain(self, input)

    def _plain(self, input):
        for m in self:
            input = m(input)
        return input
if os.environ.get("WGRAD_TABLE") is not None:          # A/B: table weight-gradient kernel (1, default) against the pair-list form (0)
    E.WGRAD_TABLE_SQUARE = bool(int(os.environ["WGRAD_TABLE"]))
dev = torch.device("cuda", 0)
co, fe = s_kitti(seed=0)
coords, feats = torch.from_numpy(co).to(dev), torch.from_numpy(fe).to(dev)
torch.manual_seed(0)
net = la.fuse_for_inference(LE.build_reference_shaped_encoder(la, 64, "cos_x", 1)).to(dev).train()
st0 = la.SparseTensor(feats, coords, 1)
with torch.no_grad():
    net.eval()(st0, 3, 2)
net.train()


def step():
    f = feats.detach().requires_grad_(True)
    x = la.SparseTensor(f, coords, 1)
    x.kmaps, x.cmaps = st0.kmaps, st0.cmaps
    if os.environ.get("ZERO_GRAD", "1") == "1":
        net.zero_grad(set_to_none=True)
    net(x, 3, 2)[1][-1].F.square().sum().backward()


K = int(os.environ.get("K", 12))
for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(int(os.environ.get("REPS", 1))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(K):
        step()
    e1.record()
    torch.cuda.synchronize()
    print(f"cfg3 train step: wall {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step, device span {e0.elapsed_time(e1) / K:.3f} ms/step, N = {coords.shape[0]}")
if os.environ.get("CPROF"):
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)
