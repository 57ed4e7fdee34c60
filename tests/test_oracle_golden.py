"""Pin the oracle (oracle/link_oracle.{c,py}) against the golden fixtures generated from the
reference (tests/golden/make_golden.py) and -- when prebuilt -- the reference's own C++ CPU ops in
oracle/_ref.  CPU-only, seconds."""
import os

import numpy as np
import pytest

from helpers import golden_files, load_golden, rel_err
from oracle import link_oracle as O


def test_hash_known_answers():
    g = load_golden("g_hash.npz")
    # SURVEY.md section 8c known answers, independently reproduced by the reference CPU op
    kat = {(0, 0, 0, 0): 947293587111810033, (1, 2, 3, 0): 1043245732202901914,
           (-1, 5, 7, 1): 348679674271016180, (255, 255, 255, 0): 670648651708156917,
           (1, 1, 1, 0): 419080468822848237, (1, 1, 1, 1): 419081568334476434}
    for row, h in zip(g["kat_coords"], g["kat_hash"]):
        assert kat[tuple(int(v) for v in row)] == int(h)
    assert np.array_equal(O.sphash(g["kat_coords"]), g["kat_hash"])
    assert np.array_equal(O.sphash(g["rnd_coords"]), g["rnd_hash"])


def test_kernel_hash():
    g = load_golden("g_hash.npz")
    assert np.array_equal(O.sphash_offsets(g["single_coords"], O.get_kernel_offsets(3)), g["khash_r3"])
    assert np.array_equal(O.sphash_offsets(g["single_coords"], O.get_kernel_offsets(2)), g["khash_r2"])
    # the reference CPU twin's batch defect (hash_cpu.cpp:29) reproduced bit-exactly on demand
    assert np.array_equal(O.sphash_offsets(g["rnd_coords"], O.get_kernel_offsets(3), cpu_batch_bug=True),
                          g["khash_r3_cpu_defect"])
    # and the CUDA semantics differ from it exactly where batch != batch[0]
    good = O.sphash_offsets(g["rnd_coords"], O.get_kernel_offsets(3))
    same = (g["rnd_coords"][:, 3] == g["rnd_coords"][0, 3])
    assert np.array_equal(good[:, same], g["khash_r3_cpu_defect"][:, same])
    assert not np.array_equal(good[:, ~same], g["khash_r3_cpu_defect"][:, ~same])


def test_query_count_unique_offsets():
    g = load_golden("g_hash.npz")
    assert np.array_equal(O.sphashquery(g["query_q"], g["query_ref"]), g["query_out"])
    assert list(g["query_out"]) == [0, 3, -1, -1]
    assert np.array_equal(O.spcount(g["count_idx"], 7), g["count_out"])
    assert np.array_equal(O.unique_rows(g["unique_in"]), g["unique_out"])
    k = load_golden("g_koff.npz")
    for r in (2, 3, 4, 5):
        assert np.array_equal(O.get_kernel_offsets(r), k[f"r{r}"])
    assert O.get_kernel_offsets(2)[:4].tolist() == [[0, 0, 0], [0, 0, 1], [0, 1, 0], [0, 1, 1]]
    assert O.get_kernel_offsets(3)[:4].tolist() == [[-1, -1, -1], [0, -1, -1], [1, -1, -1], [-1, 0, -1]]


@pytest.mark.parametrize("name", golden_files("g_agg_*.npz"))
def test_aggregate_vs_reference(name):
    g = load_golden(name)
    s, r = g["meta"]["s"], g["meta"]["r"]
    small_c, idx, counts = O.voxel_to_aux_index(g["coords"], s)
    assert np.array_equal(small_c, g["small_c"])          # bit-exact indexing
    assert np.array_equal(idx, g["idx_query"])
    assert np.array_equal(counts, g["counts"])
    if "nbr" in g:
        assert np.array_equal(O.neighbor_index(small_c, r), g["nbr"])
    aux_f, _, _, _ = O.voxel_to_aux(g["feats"], g["coords"], s)
    assert np.array_equal(aux_f, g["aux_f"])               # same op order as voxelize_cpu.cpp -> exact
    out = O.aux_to_voxel(aux_f, small_c, idx, counts, r)
    assert rel_err(out, g["out"]) < 2e-6
    if r == 3:
        # round 6: the r = 3 forward pinned on reference OUTPUT -- the fixture also holds the same aux_to_voxel call with spdevoxelize
        # routed through the reference's compiled devoxelize_forward_cpu (devoxelize_cpu.cpp:9-31: K = 8 per call -> four 8-wide
        # slices of the [M, 27] map padded to 32 columns with (-1, 0), partial outputs added; make_golden.py::_CompiledDevox).
        # It agrees with the torch restatement of devoxelize_cuda.cu:21-33 to summation order (27 terms in one chain / four groups)
        assert "r=3 forward: reference compiled ops" in g["meta"]["devoxelize"]
        assert rel_err(g["out"], g["out_refcpu"]) < 5e-7
        assert rel_err(out, g["out_refcpu"]) < 2e-6


@pytest.mark.parametrize("name", golden_files("g_agg_*.npz"))
def test_aggregate_fixture_is_the_region_mean(name):
    """Independent pin of the aggregation fixtures -- above all the r = 3 ones, whose spdevoxelize call went
    through a restatement of the CUDA kernel because the reference's CPU op hard-wires 8 neighbours (SURVEY.md
    section 8a "aggregate identity"): out_i must be the plain mean of X over every voxel whose block
    lies in the r^3 neighbourhood of voxel i's block (offsets of get_kernel_offsets(r): {-1,0,1}^3 for r = 3,
    {0,1}^3 for r = 2), computed here by brute force in fp64 from the fixture's own inputs, nothing else."""
    g = load_golden(name)
    s, r = g["meta"]["s"], g["meta"]["r"]
    coords, x = g["coords"].astype(np.int64), g["feats"].astype(np.float64)
    blk = np.concatenate([np.floor_divide(coords[:, :3], s), coords[:, 3:]], 1)
    lo, hi = (-1, 1) if r == 3 else (0, 1)
    keys, inv = np.unique(blk, axis=0, return_inverse=True)
    inv = inv.reshape(-1)
    sums = np.zeros((keys.shape[0], x.shape[1]))
    np.add.at(sums, inv, x)
    cnt = np.bincount(inv, minlength=keys.shape[0]).astype(np.float64)
    lut = {tuple(k): j for j, k in enumerate(keys.tolist())}
    tot, den = np.zeros_like(sums), np.zeros(keys.shape[0])
    for j, k in enumerate(keys.tolist()):
        for dx in range(lo, hi + 1):
            for dy in range(lo, hi + 1):
                for dz in range(lo, hi + 1):
                    q = lut.get((k[0] + dx, k[1] + dy, k[2] + dz, k[3]))
                    if q is not None:
                        tot[j] += sums[q]
                        den[j] += cnt[q]
    brute = (tot / den[:, None])[inv]
    assert rel_err(g["out"], brute) < 2e-6


@pytest.mark.parametrize("name", golden_files("g_block_*.npz"))
def test_block_core_and_grads_vs_reference(name):
    import torch
    g = load_golden(name)
    m = g["meta"]
    params = {k[4:]: torch.from_numpy(v).clone().requires_grad_(True) for k, v in g.items()
              if k.startswith("sd__") and "local_mix" not in k and "norm_local" not in k}
    feats = torch.from_numpy(g["feats"]).clone().requires_grad_(True)
    coords = torch.from_numpy(g["coords"])
    core = O.elk_core_torch(feats, coords, params, m["s"], m["r"], m["baseop"], m["groups"],
                            m["variant"], m["tensor_stride"])
    assert rel_err(core.detach().numpy(), g["core"]) < 1e-5
    core_c = O.elk_core_torch(feats.detach(), coords, {k: v.detach() for k, v in params.items()},
                              m["s"], m["r"], m["baseop"], m["groups"], m["variant"],
                              m["tensor_stride"], agg=O.aggregate_c)
    assert rel_err(core_c.numpy(), g["core"]) < 1e-5
    if m["r"] == 3:              # r = 3: the block's forward through the reference's compiled CPU devoxelize (see test_aggregate_vs_reference)
        assert rel_err(g["core"], g["core_refcpu"]) < 1e-6 and rel_err(g["out"], g["out_refcpu"]) < 1e-6
        assert rel_err(core.detach().numpy(), g["core_refcpu"]) < 1e-5 and rel_err(core_c.numpy(), g["core_refcpu"]) < 1e-5
    names = sorted(params)
    grads = torch.autograd.grad(core, [feats] + [params[n] for n in names], torch.from_numpy(g["grad_out"]))
    assert rel_err(grads[0].numpy(), g["grad_feats"]) < 1e-4
    for n, gr in zip(names, grads[1:]):
        assert rel_err(gr.numpy(), g["grad__" + n]) < 1e-4, n


@pytest.mark.parametrize("name", golden_files("g_block_*.npz"))
def test_conv_restatement_vs_reference_local_mix(name):
    """oracle.subm_conv_torch against the reference's own local_mix output (spnn.Conv3d, CPU branch)."""
    import torch
    g = load_golden(name)
    out = O.subm_conv_torch(torch.from_numpy(g["feats"]), g["coords"], torch.from_numpy(g["sd__local_mix.0.kernel"]),
                             g["meta"]["tensor_stride"])
    assert rel_err(out.numpy(), g["local"]) < 1e-5


@pytest.mark.parametrize("name", golden_files("g_pointvoxel_*.npz"))
def test_pointvoxel_restatements_vs_reference(name):
    """Row N4: oracle restatements of initial_voxelize / point_to_voxel / voxel_to_point against the
    reference's own outputs (tests/golden/make_golden_pointvoxel.py)."""
    g = load_golden(name)
    m = g["meta"]
    vf, vc, idx, counts, zc = O.initial_voxelize_np(g["points"], g["feats"], m["init_res"], m["after_res"])
    assert np.array_equal(vc, g["vox_C"]) and np.array_equal(idx, g["idx_query"])
    assert np.array_equal(counts, g["counts"]) and np.array_equal(zc, g["z_C"])
    assert rel_err(vf, g["vox_F"]) < 1e-6
    p2v, _, _ = O.point_to_voxel_np(zc, g["p2v_feats_in"], vc, 1)
    assert rel_err(p2v, g["p2v_F"]) < 1e-6
    if "v2p1_F" in g:
        f1, i1, w1 = O.voxel_to_point_np(g["v2p1_F_in"], vc, 1, zc)
        assert np.array_equal(i1, g["v2p1_idx"]) and rel_err(w1, g["v2p1_w"]) < 1e-6 and rel_err(f1, g["v2p1_F"]) < 1e-6
        f2, i2, w2 = O.voxel_to_point_np(g["v2p2_F_in"], g["v2p2_C"], 2, zc)
        assert np.array_equal(i2, g["v2p2_idx"]) and rel_err(w2, g["v2p2_w"]) < 1e-6 and rel_err(f2, g["v2p2_F"]) < 1e-6
        fn, _, _ = O.voxel_to_point_np(g["v2p1_F_in"], vc, 1, zc, nearest=True)
        assert rel_err(fn, g["v2p1_nearest_F"]) < 1e-6


@pytest.mark.parametrize("name", golden_files("g_stridedconv_*.npz"))
def test_strided_conv_restatement_vs_reference(name):
    """Row N1 (strided part): oracle restatements of spdownsample + the k2-s2 / transposed convolution against
    the reference's spnn.Conv3d chain (tests/golden/make_golden_stridedconv.py)."""
    import torch
    g = load_golden(name)
    c0 = g["coords"]
    c2 = O.downsample_coords(c0, 2, 1)
    assert np.array_equal(c2, g["x2_C"]) and np.array_equal(c0, g["x4_C"])
    if not g["meta"]["features_valid"]:
        return
    t = lambda a: torch.from_numpy(a)
    x1 = O.subm_conv_torch(t(g["feats"]), c0, t(g["k1"]), 1)
    assert rel_err(x1.numpy(), g["x1_F"]) < 1e-5
    down = O.strided_conv_table(c0, c2, 2, 1)
    x2 = O.gather_conv_torch(x1, down, t(g["k2"]))
    assert rel_err(x2.numpy(), g["x2_F"]) < 1e-5
    x3 = O.subm_conv_torch(x2, c2, t(g["k3"]), 2)
    assert rel_err(x3.numpy(), g["x3_F"]) < 1e-5
    x4 = O.gather_conv_torch(x3, None, t(g["k4"]), n_out=c0.shape[0], transposed_of=down)
    assert rel_err(x4.numpy(), g["x4_F"]) < 1e-5


def test_size_checkpoints():
    """SURVEY.md section 8d generator checkpoints: M and sha256 of the index arrays at cfg1/cfg2."""
    import hashlib
    import json
    from helpers import GOLDEN, s_uniform
    with open(os.path.join(GOLDEN, "g_size.json")) as f:
        sizes = json.load(f)["sizes"]
    for key, n, s in (("N10000_s7", 10_000, 7), ("N100000_s7", 100_000, 7)):
        coords = s_uniform(n).numpy()
        small_c, idx, counts = O.voxel_to_aux_index(coords, s)
        assert small_c.shape[0] == sizes[key]["M"]
        assert hashlib.sha256(idx.astype(np.int64).tobytes()).hexdigest() == sizes[key]["sha256_idx"]
        assert hashlib.sha256(counts.astype(np.int32).tobytes()).hexdigest() == sizes[key]["sha256_counts"]
        assert hashlib.sha256(small_c.astype(np.int32).tobytes()).hexdigest() == sizes[key]["sha256_small_c"]
        nbr = O.neighbor_index(small_c, 3)
        assert hashlib.sha256(nbr.astype(np.int32).tobytes()).hexdigest() == sizes[key]["sha256_nbr_r3"]
    assert sizes["N10000_s7"]["M"] == 9047 and sizes["N100000_s7"]["M"] == 43334


def test_oracle_vs_compiled_reference_ops():
    """Bit-exact check of the C restatement against the reference's own C++ ops (oracle/_ref)."""
    from oracle import build_ref
    if not os.path.exists(build_ref.SO) and not build_ref.available():
        pytest.skip("oracle/_ref not built and /root/reference absent")
    import torch
    ref = build_ref.load_module()
    rng = np.random.default_rng(0)
    coords = rng.integers(-500, 500, (5000, 4)).astype(np.int32)
    coords[:, 3] = 0
    t = torch.from_numpy(coords)
    assert np.array_equal(ref.hash_cpu(t).numpy(), O.sphash(coords))
    off = torch.from_numpy(O.get_kernel_offsets(3))
    assert np.array_equal(ref.kernel_hash_cpu(t, off).numpy(), O.sphash_offsets(coords, off.numpy()))
    idx = rng.integers(-1, 300, 5000).astype(np.int32)
    assert np.array_equal(ref.count_cpu(torch.from_numpy(idx), 300).numpy(), O.spcount(idx, 300))
    idxp = np.abs(idx)
    counts = O.spcount(idxp, 300)
    feats = rng.standard_normal((5000, 24)).astype(np.float32)
    a = ref.voxelize_forward_cpu(torch.from_numpy(feats), torch.from_numpy(idxp), torch.from_numpy(counts)).numpy()
    assert np.array_equal(a, O.spvoxelize_fwd(feats, idxp, counts))
    top = rng.standard_normal((300, 24)).astype(np.float32)
    b = ref.voxelize_backward_cpu(torch.from_numpy(top), torch.from_numpy(idxp), torch.from_numpy(counts), 5000).numpy()
    assert np.array_equal(b, O.spvoxelize_bwd(top, idxp, counts, 5000))
    ind = rng.integers(-1, 300, (700, 8)).astype(np.int32)
    w = rng.random((700, 8)).astype(np.float32)
    c = ref.devoxelize_forward_cpu(torch.from_numpy(top), torch.from_numpy(ind), torch.from_numpy(w)).numpy()
    assert np.array_equal(c, O.spdevoxelize_fwd(top, ind, w))   # K=8: identical op order


@pytest.mark.parametrize("name", golden_files("g_agg_*.npz"))
def test_aggregate_fixture_indices_hold_by_definition(name):
    """Pins the fixtures' `idx_query` / `nbr` arrays -- which went through the oracle's restatement of hash_query_cpu when they
    were generated (query_cpu.cpp needs sparsehash: oracle/ref_bind.cpp) -- on the DEFINITIONS of utils.py:44-58 and :61-73,
    with nothing from oracle/ in the check: (a) small_c is the sorted unique set of floor(coords / s) rows (torch.unique order);
    (b) small_c[idx_query[i]] is voxel i's block; (c) counts is its histogram; (d) small_c[nbr[m, k]] == small_c[m] + offset_k
    wherever nbr >= 0, and where nbr == -1 that block is absent (checked against a Python set); offsets in the reference's
    get_kernel_offsets order (nn/utils/kernel.py:11-32), restated here by its rule: odd volume x fastest, even volume z fastest."""
    g = load_golden(name)
    s, r = g["meta"]["s"], g["meta"]["r"]
    coords = g["coords"].astype(np.int64)
    blk = np.concatenate([np.floor_divide(coords[:, :3], s), coords[:, 3:]], 1)
    small_c = g["small_c"].astype(np.int64)
    assert np.array_equal(small_c, np.unique(blk, axis=0))                                  # (a) sorted unique rows
    assert np.array_equal(small_c[g["idx_query"]], blk)                                     # (b)
    assert np.array_equal(g["counts"], np.bincount(g["idx_query"], minlength=small_c.shape[0]))   # (c)
    if "nbr" not in g:
        return
    ax = list(range(-r // 2 + 1, r // 2 + 1))
    offs = [(x, y, z) for z in ax for y in ax for x in ax] if (r ** 3) % 2 == 1 else [(x, y, z) for x in ax for y in ax for z in ax]
    present = {tuple(row) for row in small_c.tolist()}
    nbr = g["nbr"]
    assert nbr.shape == (small_c.shape[0], r ** 3)
    for k, (dx, dy, dz) in enumerate(offs):
        want = small_c + np.array([dx, dy, dz, 0])
        hit = nbr[:, k] >= 0
        assert np.array_equal(small_c[nbr[hit, k]], want[hit]), f"offset {k}"
        assert not any(tuple(row) in present for row in want[~hit].tolist()), f"offset {k}: a present block reported absent"
    assert (nbr >= 0).any(axis=1).all()                                                     # the own block is always there


@pytest.mark.parametrize("name", golden_files("g_pointvoxel_*.npz"))
def test_pointvoxel_fixture_indices_hold_by_definition(name):
    """The same for the point <-> voxel fixtures (core/models/utils.py:234-324): idx_query maps every point to the voxel that
    holds floor(point / voxel size) -- vox_C[idx_query[i]] equals the point's cell row -- and counts is its histogram; the
    v2p index arrays name, for every point and corner, a voxel whose coordinate IS that corner (or -1 with the corner absent)."""
    g = load_golden(name)
    vox_c = g["vox_C"].astype(np.int64)
    cell = np.concatenate([np.floor(g["z_C"][:, :3]).astype(np.int64), g["z_C"][:, 3:].astype(np.int64)], 1)
    assert np.array_equal(vox_c[g["idx_query"]], cell)
    assert np.array_equal(g["counts"], np.bincount(g["idx_query"], minlength=vox_c.shape[0]))
    assert len({tuple(r) for r in vox_c.tolist()}) == vox_c.shape[0]                        # voxels are unique
    present = {tuple(r): j for j, r in enumerate(vox_c.tolist())}
    if "v2p1_idx" not in g:
        return
    idx = g["v2p1_idx"]
    offs = [(x, y, z) for x in (0, 1) for y in (0, 1) for z in (0, 1)]                      # get_kernel_offsets(2): z fastest
    for k, (dx, dy, dz) in enumerate(offs):
        want = cell + np.array([dx, dy, dz, 0])
        got = idx[:, k]
        exp = np.array([present.get(tuple(r), -1) for r in want.tolist()])
        assert np.array_equal(got, exp), f"corner {k}"
