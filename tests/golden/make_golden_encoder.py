"""tests/golden/make_golden_encoder.py -- network-level fixture (build container only).

Imports the REFERENCE's own ELKEncoder (segmentation/core/models/semantic_kitti/linkencoder.py:186-380) on
its CPU path (same import recipe and caveats as make_golden.py: reference C++ CPU ops compiled where they
lie, hash_query through the oracle restatement because sparsehash is absent), runs the encoder half of its
forward (linkencoder.py:339-368: stem, then four stages of down-conv, residual stage + tail || ELKBlock +
tail, add, ReLU) on a small seeded S-kitti-shaped frame, and stores inputs, the encoder's state_dict and the
four stage outputs.  r = 2, so the reference's CPU spdevoxelize (8 neighbours hard-wired) is exact: every
number in the fixture is reference output.  Fixtures are data only.  Run: python tests/golden/make_golden_encoder.py
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
from torchsparse import SparseTensor  # noqa: E402
from core.models.semantic_kitti.linkencoder import ELKEncoder  # noqa: E402

spec = __import__("importlib.util").util.spec_from_file_location("synth", os.path.join(ROOT, "link_amd", "synth.py"))
synth = __import__("importlib.util").util.module_from_spec(spec)
spec.loader.exec_module(synth)


def main():
    torch.manual_seed(7)
    coords_np, feats_np = synth.s_kitti(3, n_az=160, voxel=0.25)      # ~3-4k voxels, LiDAR-shaped
    coords, feats = torch.from_numpy(coords_np), torch.from_numpy(feats_np)
    net = ELKEncoder(cr=0.25, baseop="cos_x", groups=1, s=3, r=2, num_classes=19).eval()
    # non-trivial BatchNorm statistics / affine parameters (fresh modules would be identities)
    g = torch.Generator().manual_seed(11)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.num_features, generator=g))
    s, r = 3, 2
    outs = {}
    with torch.no_grad():
        x = SparseTensor(feats.clone(), coords.clone(), 1)
        x0 = net.stem(x)
        prev = x0
        for i in (1, 2, 3, 4):
            d = getattr(net, f"down{i}")(prev)
            xi = getattr(net, f"stage{i}_tail")(getattr(net, f"stage{i}")(d))
            lk = getattr(net, f"elk{i}_tail")(getattr(net, f"elk{i}")(d, d.s[0] * s, r))
            xi.F = getattr(net, f"activate{i}")(xi.F + lk.F)
            outs[f"x{i}_F"], outs[f"x{i}_C"] = xi.F.numpy().copy(), xi.C.numpy().copy()
            prev = xi
        outs["x0_F"] = x0.F.numpy().copy()
    keep = ("stem.", "down", "stage", "elk")
    sd = {"sd::" + k: v.numpy() for k, v in net.state_dict().items() if k.startswith(keep)}
    meta = {"generator": "tests/golden/make_golden_encoder.py", "reference": "MCG-NJU/LinK, imported from /root/reference",
            "what": "ELKEncoder(cr=0.25, cos_x, groups=1, s=3, r=2).eval(): stem + 4 encoder stages, reference CPU path",
            "hash_query_cpu": "oracle restatement (sparsehash absent; oracle/ref_bind.cpp)", "torch": torch.__version__,
            "n": int(coords.shape[0]), "stage_voxels": [int(outs[f"x{i}_C"].shape[0]) for i in (1, 2, 3, 4)]}
    np.savez_compressed(os.path.join(HERE, "g_encoder_cosx_s3_r2.npz"), meta=np.array(json.dumps(meta)),
                        coords=coords_np, feats=feats_np, **outs, **sd)
    print("wrote g_encoder_cosx_s3_r2.npz", meta)


if __name__ == "__main__":
    main()
