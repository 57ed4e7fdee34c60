"""tests/golden/make_golden_pointvoxel.py -- golden fixtures for row N4 of SURVEY.md section 8f: the
point<->voxel helpers of the reference (segmentation/core/models/utils.py:234-324: initial_voxelize,
point_to_voxel, voxel_to_point) run through the imported reference on its CPU ops (build container
only; same provenance caveats as make_golden.py: hash_query_cpu is the oracle restatement, and the
CPU kernel_hash batch defect (hash_cpu.cpp:29) is avoided by single-batch inputs for voxel_to_point).
Run:  python tests/golden/make_golden_pointvoxel.py
"""
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
from torchsparse import PointTensor, SparseTensor  # noqa: E402
import core.models.utils as ref_utils  # noqa: E402

META = {"generator": "tests/golden/make_golden_pointvoxel.py",
        "reference": "MCG-NJU/LinK @ 2024_08_07, imported from /root/reference",
        "hash_query_cpu": "oracle restatement (sparsehash absent; oracle/ref_bind.cpp)",
        "torch": torch.__version__}


def save(name, meta, **arrays):
    m = dict(META); m.update(meta)
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(json.dumps(m)), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def main():
    for tag, (npts, extent, init_res, after_res, batches, seed) in {
            "a": (3000, 30.0, 1, 1, 1, 0), "b": (4000, 12.0, 0.05, 0.1, 1, 1), "c": (2500, 20.0, 1, 1, 2, 2)}.items():
        g = torch.Generator().manual_seed(seed)
        xyz = torch.rand(npts, 3, generator=g) * extent
        if init_res != 1:
            xyz = torch.floor(xyz / init_res)             # integer point coords in init_res units, as the datasets give
        b = torch.randint(0, batches, (npts, 1), generator=g).float()
        pc = torch.cat([xyz, b], 1)
        feats = torch.randn(npts, 4, generator=g)
        z = PointTensor(feats.clone(), pc.clone())
        st = ref_utils.initial_voxelize(z, init_res, after_res)
        out = dict(points=pc.numpy(), feats=feats.numpy(), vox_F=st.F.numpy(), vox_C=st.C.numpy(),
                   idx_query=z.additional_features["idx_query"][1].numpy(),
                   counts=z.additional_features["counts"][1].numpy(), z_C=z.C.numpy())
        # point_to_voxel on the voxel set just built, with fresh point features
        z.F = torch.randn(npts, 6, generator=g)
        out["p2v_feats_in"] = z.F.numpy()
        out["p2v_F"] = ref_utils.point_to_voxel(st, z).F.numpy()
        if batches == 1:
            # voxel_to_point: trilinear devoxelisation from stride-1 voxels and from a stride-2 voxel set
            x1 = SparseTensor(torch.randn(st.C.shape[0], 5, generator=g), st.C, 1)
            p1 = ref_utils.voxel_to_point(x1, z)
            out.update(v2p1_F_in=x1.F.numpy(), v2p1_F=p1.F.numpy(), v2p1_idx=z.idx_query[x1.s].numpy(),
                       v2p1_w=z.weights[x1.s].numpy())
            c2 = torch.unique(torch.cat([(st.C[:, :3] // 2) * 2, st.C[:, 3:]], 1), dim=0).int()
            x2 = SparseTensor(torch.randn(c2.shape[0], 5, generator=g), c2, 2)
            p2 = ref_utils.voxel_to_point(x2, z)
            out.update(v2p2_C=c2.numpy(), v2p2_F_in=x2.F.numpy(), v2p2_F=p2.F.numpy(),
                       v2p2_idx=z.idx_query[x2.s].numpy(), v2p2_w=z.weights[x2.s].numpy())
            pn = ref_utils.voxel_to_point(SparseTensor(x1.F, st.C, 1), PointTensor(z.F, z.C), nearest=True)
            out["v2p1_nearest_F"] = pn.F.numpy()
        save(f"g_pointvoxel_{tag}.npz", {"what": "core/models/utils.py:234-324 initial_voxelize / point_to_voxel / "
                                                 "voxel_to_point", "init_res": init_res, "after_res": after_res,
                                         "batches": batches}, **out)


if __name__ == "__main__":
    main()
