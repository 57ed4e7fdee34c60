"""Where the per-frame milliseconds of the segmentation encoder (cfg3 / cfg4 shape) go when every kernel map is rebuilt:
wall time (with the syncs they contain) inside the map builders, per forward.   python tools/coldprof_seg.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import link_amd as la
from link_amd import aggregate, elk, index
from harness import networks as LE
from link_amd.synth import s_kitti

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
    co, fe = s_kitti(seed=0)
    coords, feats = torch.from_numpy(co).to(dev), torch.from_numpy(fe).to(dev)
    torch.manual_seed(0)
    net = la.fuse_for_inference(LE.build_reference_shaped_encoder(la, 64, "cos_x", 1)).to(dev).eval()

    def cold():
        with torch.no_grad():
            net(la.SparseTensor(feats, coords.clone(), 1), 3, 2)
    st0 = la.SparseTensor(feats, coords, 1)

    def warm():
        x = la.SparseTensor(feats, coords, 1)
        x.kmaps, x.cmaps = st0.kmaps, st0.cmaps
        with torch.no_grad():
            net(x, 3, 2)
    for fn, name in ((cold, "maps rebuilt"), (warm, "warm maps")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"{name}: {1e2 * (time.perf_counter() - t0):.3f} ms per frame")
    timed(elk, "neighbor_table_of", "neighbour table (incl. bounds)")
    timed(elk._PairPlan, "__init__", "pair plan")
    timed(elk.Conv3d, "_strided_map", "strided map (k2 s2)")
    timed(elk, "link_index_of", "LinK block index")
    timed(index, "coords_bounds", "coords_bounds")
    timed(elk._ELKBase, "_core_dense", "_core_dense")
    for _ in range(5):
        cold()
    for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
        print(f"  {k:36s} {1e3 * t / 5:7.3f} ms per frame in {c / 5:.0f} calls")


if __name__ == "__main__":
    main()
