"""Where the per-frame ("cold": every kernel map rebuilt) milliseconds of the cfg5 backbone go on the HOST side: wall time
(including the device syncs they contain) inside the map builders, per forward.   python tools/coldprof.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import link_amd as la
from link_amd import detstage, elk
from link_amd.synth import s_nusc

acc = {}


def timed(mod, name, label):
    f0 = getattr(mod, name)

    def f(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = f0(*a, **k)
        torch.cuda.synchronize()
        d = acc.setdefault(label, [0.0, 0])
        d[0] += time.perf_counter() - t0
        d[1] += 1
        return r
    setattr(mod, name, f)


def main():
    dev = torch.device("cuda:0")
    co, fe = s_nusc(seed=0)
    indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
    feats = torch.from_numpy(fe).to(dev)
    torch.manual_seed(0)
    net = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
    shape = [1440, 1440, 40]
    with torch.no_grad():
        for _ in range(3):
            net(feats, indices, 1, shape)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net(feats, indices, 1, shape)
        torch.cuda.synchronize()
        base = (time.perf_counter() - t0) / 10
        maps = {}
        net(feats, indices, 1, shape, indice_dict=maps)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net(feats, indices, 1, shape, indice_dict=maps)
        torch.cuda.synchronize()
        warm = (time.perf_counter() - t0) / 10
    print(f"per frame: maps rebuilt {1e3 * base:.3f} ms, warm maps {1e3 * warm:.3f} ms")
    timed(elk._PairPlan, "__init__", "pair plan (count, sync, layout, fill)")
    timed(elk, "neighbor_table_of", "neighbour table of a submanifold convolution")
    timed(detstage.SparseConv3d, "_map", "strided convolution: sites + table")
    timed(elk, "link_index_of", "LinK block index")
    with torch.no_grad():
        for _ in range(5):
            net(feats, indices, 1, shape)
    for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:50s} {1e3 * t / 5:7.3f} ms per frame in {c / 5:.0f} calls")


if __name__ == "__main__":
    main()
