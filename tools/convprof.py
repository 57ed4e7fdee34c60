"""Per-wave phases of the resident-weights convolution kernel (conv.hip: k_subm_conv_resident) on an S-nusc stage frame.
Needs a profiling build (LINK_AMD_CXXFLAGS=-DCONV_RES_DBG).   C=16 python tools/convprof.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import link_amd as la
from link_amd import _lib as L, elk
from link_amd.synth import s_nusc


def main():
    dev = torch.device("cuda:0")
    c = int(os.environ.get("C", 16))
    co, _ = s_nusc(seed=0)
    coords = torch.from_numpy(co).to(dev)
    n = coords.shape[0]
    st = la.SparseTensor(torch.randn(n, c, device=dev), coords, 1)
    nbr, _ = elk.neighbor_table_of(st, (3, 3, 3))
    w = torch.randn(27, c, c, device=dev) * 0.1
    sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    split = os.environ.get("EXACT", "0") == "0"       # the fused inference entry takes the fp16-split products
    for _ in range(5):
        if split:
            out = elk.subm_conv_ln_add_relu(st.F, w, nbr, None, sc, sh, 0.0, None, relu=True, affine=True)
        else:
            out = elk.subm_conv(st.F, w, nbr, None)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(L.lib()._name)
    buf = np.zeros(8 * 16384, dtype=np.uint64)
    assert lib.link_conv_resident_debug_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.nbytes)) == 0
    d = buf.reshape(-1, 8)
    d = d[d[:, 7] > 0]
    t0 = d[:, 6].astype(np.float64)
    print(f"n={n} C={c}: {len(d)} waves; start spread {t0.max() - t0.min():.0f} ticks; s_memtime ticks")
    for i, nm in enumerate(["stage W + barrier", "neighbour ids", "wait rows", "products", "store phase", "(tiles)", None, "wave total"]):
        if nm:
            v = d[:, i].astype(np.float64)
            print(f"  {nm:20s} mean {v.mean():9.1f}  p50 {np.median(v):9.1f}  max {v.max():9.1f}")
    last = (d[:, 6] + d[:, 7]).astype(np.float64)
    print(f"  first start -> last end {last.max() - t0.min():.0f} ticks")


if __name__ == "__main__":
    main()
