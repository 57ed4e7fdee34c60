"""tests/golden/make_golden_unet.py -- network-level fixture of the WHOLE segmentation network (build container only).

Imports the REFERENCE's own ELKUNet (segmentation/core/models/semantic_kitti/linkunet.py:186-385) on its CPU path (same
import recipe and caveats as make_golden.py / make_golden_encoder.py: reference C++ CPU ops compiled where they lie,
hash_query through the oracle restatement because sparsehash is absent) and runs its unmodified forward -- stem, four
encoder stages, four decoder stages (transposed convolution, torchsparse.cat with the skip, two residual blocks), classifier --
on a small seeded S-kitti-shaped frame.  r = 2, so the reference's CPU spdevoxelize (8 neighbours hard-wired) is exact:
every number in the fixture is reference output.  Stored: inputs, the state_dict, the logits and (through forward hooks on
the reference's modules) the decoder stage outputs.  Fixtures are data only.  Run: python tests/golden/make_golden_unet.py
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
from core.models.semantic_kitti.linkunet import ELKUNet  # noqa: E402

spec = __import__("importlib.util").util.spec_from_file_location("synth", os.path.join(ROOT, "link_amd", "synth.py"))
synth = __import__("importlib.util").util.module_from_spec(spec)
spec.loader.exec_module(synth)


def main():
    torch.manual_seed(9)
    coords_np, feats_np = synth.s_kitti(5, n_az=160, voxel=0.25)      # ~3-4k voxels, LiDAR-shaped
    coords, feats = torch.from_numpy(coords_np), torch.from_numpy(feats_np)
    net = ELKUNet(cr=0.25, baseop="cos_x", groups=1, s=3, r=2, num_classes=19).eval()
    g = torch.Generator().manual_seed(13)
    for m in net.modules():                                          # non-trivial BatchNorm statistics / affine parameters
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.copy_(0.1 * torch.randn(m.num_features, generator=g))
            m.running_var.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.weight.data.copy_(0.5 + torch.rand(m.num_features, generator=g))
            m.bias.data.copy_(0.1 * torch.randn(m.num_features, generator=g))
    outs = {}

    def keep(name):
        def hook(_m, _inp, out):
            outs[name + "_F"], outs[name + "_C"] = out.F.detach().numpy().copy(), out.C.detach().numpy().copy()
        return hook
    for i in (1, 2, 3, 4):
        getattr(net, f"up{i}")[1].register_forward_hook(keep(f"y{i}"))
    with torch.no_grad():
        logits = net(SparseTensor(feats.clone(), coords.clone(), 1))
    sd = {"sd::" + k: v.numpy() for k, v in net.state_dict().items()}
    meta = {"generator": "tests/golden/make_golden_unet.py", "reference": "MCG-NJU/LinK, imported from /root/reference",
            "what": "ELKUNet(cr=0.25, cos_x, groups=1, s=3, r=2, 19 classes).eval(): unmodified forward, reference CPU path",
            "hash_query_cpu": "oracle restatement (sparsehash absent; oracle/ref_bind.cpp)", "torch": torch.__version__,
            "n": int(coords.shape[0]), "decoder_voxels": [int(outs[f"y{i}_C"].shape[0]) for i in (1, 2, 3, 4)]}
    np.savez_compressed(os.path.join(HERE, "g_unet_cosx_s3_r2.npz"), meta=np.array(json.dumps(meta)), coords=coords_np,
                        feats=feats_np, logits=logits.numpy(), **outs, **sd)
    print("wrote g_unet_cosx_s3_r2.npz", meta)


if __name__ == "__main__":
    main()
