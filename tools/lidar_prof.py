"""Per-wave phases of k_elk_tiles (general-layout tile form) on one LiDAR stage frame.  Needs a profiling build of the
library (LINK_AMD_CXXFLAGS=-DELK_T_DBG python link_amd/build.py -> link_amd/lib/variants/lib_DBG.so, copied over
liblink_amd.so on the GPU box by tools/ab_variants.sh or by hand): the kernel then leaves 8 s_memtime deltas per wave
in a device array.   STAGE=0 python tools/lidar_prof.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import link_amd as la
from link_amd import _lib as L
from link_amd.elk import ElkCorePlan
from link_amd.index import coords_bounds
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from lidar_core import stages


def main():
    dev = torch.device("cuda:0")
    k = int(os.environ.get("STAGE", 0))
    cfg, r = stages(dev)[k]
    b, coords, feats = r["blk"], r["coords"], r["feats"]
    n, c = feats.shape
    plan = ElkCorePlan(n, c, b.baseop, r["cg"], r["r"], r["s_eff"], coords_bounds(coords), dev, coord_div=r["coord_div"], layout="general")
    plan.bind(b.pre_mix[0].weight, b.pre_mix[1].weight, b.pre_mix[1].bias, r["w_pos"], r["alpha"], b.norm.weight, b.norm.bias)
    for _ in range(5):
        plan.run(feats, coords, build_index=True)
    torch.cuda.synchronize()
    lib = ctypes.CDLL(L.lib()._name)
    for which, title, names in (
            (0, "k_elk_tiles", ["boundary", "stage W + barrier", "wait rows", "mfma", "theta prep + LayerNorm", "sincos + scans + stores", "combine", "total"]),
            (1, "k_elk_gather_tiles", ["positions, records, run heads", "neighbour ids", "A rows", "voxel steps", "(runs in the tile)", "-", "-", "total"])):
        buf = np.zeros(8 * 32768, dtype=np.uint64)
        assert lib.link_elk_tiles_debug_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(buf.nbytes), ctypes.c_int(which)) == 0
        d = buf.reshape(-1, 8)
        d = d[d[:, 7] > 0]
        print(f"stage {k} ({cfg}): n={n} C={c} op={b.baseop}; {title}: {len(d)} waves with work; s_memtime ticks")
        for i, nm in enumerate(names):
            if nm == "-":
                continue
            v = d[:, i].astype(np.float64)
            print(f"  {nm:30s} mean {v.mean():8.1f}  p50 {np.median(v):8.1f}  max {v.max():8.1f}")


if __name__ == "__main__":
    main()
