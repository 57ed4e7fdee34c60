"""Convolution form A/B on the S-nusc backbone (cfg5 shape), warm kernel maps: pair-list everywhere / resident-weights
kernel on the narrow layers (with and without the spatial tile order).   python tools/convform.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

import link_amd as la
from link_amd import elk
from link_amd.synth import s_nusc


def main():
    dev = torch.device("cuda:0")
    co, fe = s_nusc(seed=0)
    indices = torch.from_numpy(co[:, [3, 2, 1, 0]].copy()).int().to(dev)
    feats = torch.from_numpy(fe).to(dev)
    torch.manual_seed(0)
    net = la.SpMiddleResNetFHDELKv3(num_input_features=5).to(dev).eval()
    shape = [1440, 1440, 40]
    ref = None
    cases = [("pair-list", False, False), ("resident", True, False), ("resident + tile order", True, True)]
    if os.environ.get("ONLY"):
        cases = [c for c in cases if c[0] == os.environ["ONLY"]]
    for name, res, order in cases:
        elk.RESIDENT_FORM, elk.RESIDENT_TILE_ORDER = res, order
        maps = {}
        with torch.no_grad():
            for _ in range(3):
                bev, _ = net(feats, indices, 1, shape, indice_dict=maps)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                net(feats, indices, 1, shape, indice_dict=maps)
            torch.cuda.synchronize()
            warm = 1e3 * (time.perf_counter() - t0) / 20
            t0 = time.perf_counter()
            for _ in range(10):
                net(feats, indices, 1, shape)
            torch.cuda.synchronize()
            cold = 1e3 * (time.perf_counter() - t0) / 10
        ref = bev if ref is None else ref
        print(f"{name:24s} warm maps {warm:.3f} ms, maps per frame {cold:.3f} ms, max rel diff vs first {float((bev - ref).abs().max() / ref.abs().max()):.2e}")


if __name__ == "__main__":
    main()
