"""tests/golden/make_golden_stridedconv.py -- golden fixtures for the strided part of row N1 (SURVEY.md
section 8f): the reference's spnn.Conv3d (torchsparse/nn/modules/conv.py, CPU branch of
nn/functional/conv.py:47-61) in the four forms the LinK encoders chain (linkunet.py:40-92): k3 s1
(channel change), k2 s2 down-sampling, k3 s1 at tensor stride 2, k2 s2 transposed.  Imported reference,
build container only.  Run:  python tests/golden/make_golden_stridedconv.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402

torchsparse, backend = build_ref.import_reference_python()
from torchsparse import SparseTensor  # noqa: E402
import torchsparse.nn as spnn  # noqa: E402

META = {"generator": "tests/golden/make_golden_stridedconv.py",
        "reference": "MCG-NJU/LinK @ 2024_08_07, imported from /root/reference",
        "hash_query_cpu": "oracle restatement (sparsehash absent; oracle/ref_bind.cpp)", "torch": torch.__version__}


def main():
    for tag, (n, grid, batches, seed) in {"a": (1500, 14, 1, 0), "b": (1200, 12, 2, 1)}.items():
        g = torch.Generator().manual_seed(seed)
        per = []
        for b in range(batches):
            lin = torch.randperm(grid ** 3, generator=g)[: n // batches]
            per.append(torch.stack([lin % grid, (lin // grid) % grid, lin // (grid * grid), torch.full_like(lin, b)], 1))
        coords = torch.cat(per).int()
        feats = torch.randn(coords.shape[0], 8, generator=g)
        torch.manual_seed(seed)
        c1 = spnn.Conv3d(8, 16, kernel_size=3, stride=1)
        c2 = spnn.Conv3d(16, 16, kernel_size=2, stride=2)
        c3 = spnn.Conv3d(16, 24, kernel_size=3, stride=1)
        c4 = spnn.Conv3d(24, 8, kernel_size=2, stride=2, transposed=True)
        x0 = SparseTensor(feats, coords, 1)
        x0.cmaps.setdefault(x0.stride, x0.coords)
        x1 = c1(x0); x2 = c2(x1); x3 = c3(x2); x4 = c4(x3)
        arrays = dict(coords=coords.numpy(), feats=feats.numpy(), k1=c1.kernel.detach().numpy(), k2=c2.kernel.detach().numpy(),
                      k3=c3.kernel.detach().numpy(), k4=c4.kernel.detach().numpy(), x1_F=x1.F.detach().numpy(),
                      x2_F=x2.F.detach().numpy(), x2_C=x2.C.numpy(), x3_F=x3.F.detach().numpy(), x4_F=x4.F.detach().numpy(),
                      x4_C=x4.C.numpy())
        m = dict(META); m.update(what="x1=Conv3d(8,16,3)(x0); x2=Conv3d(16,16,2,stride=2)(x1); x3=Conv3d(16,24,3)(x2); "
                                       "x4=Conv3d(24,8,2,stride=2,transposed=True)(x3)", batches=batches,
                 x2_stride=list(x2.s), x4_stride=list(x4.s),
                 features_valid=(batches == 1),
                 note="batches > 1: the reference CPU kernel_hash uses row 0's batch index for every row "
                      "(hash_cpu.cpp:29), so neighbour maps of batch > 0 rows -- and the features -- are "
                      "defective on CPU; only the coordinate outputs (x2_C, x4_C: spdownsample ordering) are "
                      "usable from this file" if batches > 1 else "")
        np.savez_compressed(os.path.join(HERE, f"g_stridedconv_{tag}.npz"), meta=np.array(json.dumps(m)), **arrays)
        print("wrote", tag, {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    main()
