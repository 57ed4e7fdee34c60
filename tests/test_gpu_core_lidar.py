"""R_core ALONE on the full-size S-kitti stage frames, every form, against the oracle at north_star's 1e-4 AND against a float64
evaluation of the same formula ("pin R_core on the S-kitti stage frames at 1e-4 and arbitrate the
forms").  Captures what the four ELKBlock._core calls of a forward on S-kitti seed 0 receive, for both segmentation variants
(linkunet.py:165: theta on stride-multiplied coordinates, up to ~1500 rad; linkencoder.py:165: theta on coords / stride), and runs
the tile form, the four-kernel form and the lean form of ElkCorePlan on them.

What the float64 arbiter found in round 5 (tools/lidar_core_parity.py, tools/core_dbg.py; both fixed, this test keeps them fixed):
  * the fp16 hi | lo split of the pre_mix contraction (2^-22 operands) is multiplied by theta in cos_x's linear channel group:
    1.1e-4 / 1.6e-4 of max|out| on stage 1 where the fp32 reference itself is 2.3e-5 / 2.9e-5 from float64 -> cos_x takes the
    exact fp32 matrix instruction (LINK_COSX_EXACT, elk_common.h);
  * `v - fin * theta` compiled to ONE fma (unrounded product) while the block sums hold the rounded product: on the sparse stages
    3 / 4, where a voxel is often alone in its neighbourhood and the reference's two uses cancel exactly, that left 1.5e-5 against
    the reference's 4.7e-7 -> link_mul_rn (common.h).
A form is RIGHT when it is as close to float64 as the fp32 reference is: rel64 <= 2 * o64 + 2e-6."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
TOL = 1e-4              # north_star: within 1e-4 rel on fp32 features (max|a - b| / max|b|)


# Round 6: seeds 0, 1 and 7 -- on seed 0 the biggest stage sits at 0.93 of the gate for the four-kernel and
# lean forms (the float64 arbiter shows the ORACLE is 2-3e-5 from the truth there), so one frame proves little.  The per-form maxima
# of every (variant, seed, stage) go to gpurun_out/lidar_parity_maxima.jsonl (committed as profiles/r06_lidar_parity_maxima.jsonl).
@pytest.mark.parametrize("seed", [0, 1, 7])
@pytest.mark.parametrize("variant", ["encoder", "unet"])
def test_core_forms_on_full_size_s_kitti_stage_frames_vs_oracle_and_float64(variant, seed):
    import json
    from tools.lidar_core_parity import core_refs, forms_of, seg_stage_calls
    dev = torch.device("cuda:0")
    calls = seg_stage_calls(dev, variant, seed=seed)
    rows = []
    assert len(calls) == 4 and calls[0]["feats"].shape[0] > 40000          # full-size frame: ~59k voxels at stride 2
    for k, r in enumerate(calls):
        ref32, ref64 = core_refs(r, variant)
        s64 = float(ref64.abs().max())
        o64 = float((ref32.double() - ref64).abs().max() / s64)             # what evaluating the reference in fp32 costs
        got = forms_of(r, dev)
        assert {"tiles", "four"} <= set(got), (variant, k, sorted(got))
        if r["feats"].shape[0] * 3 * r["feats"].shape[1] <= 8_000_000 * 3:
            assert "lean" in got, (variant, k)
        for name, out in got.items():
            rel32 = float((out - ref32).abs().max() / ref32.abs().max())
            rel64 = float((out.double() - ref64).abs().max() / s64)
            rows.append({"variant": variant, "seed": seed, "stage": k + 1, "voxels": int(r["feats"].shape[0]), "form": name,
                         "rel32": rel32, "rel64": rel64, "oracle_rel64": o64, "share_of_gate": rel32 / TOL})
    try:                                                  # the table first, the verdicts after: a failing form still leaves its row
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "lidar_parity_maxima.jsonl"), "a") as f:
            for row in rows:
                f.write(json.dumps(row) + "\n")
    except OSError:
        pass
    for row in rows:
        assert row["rel32"] < TOL, row
        assert row["rel64"] <= 2.0 * row["oracle_rel64"] + 2e-6, row
