"""GPU parity of the block index (include/link_amd.h section B, index half): bit-exact against the
oracle / golden fixtures for small_x.C, idx_query, counts and the neighbour maps."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, golden_files, lidar_like, load_golden, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu


def check_index(coords_np, s, rs=(2, 3), bounds=None):
    import link_amd as la
    c = torch.from_numpy(np.ascontiguousarray(coords_np)).cuda()
    idx = la.BlockIndex(c, s, bounds=bounds)
    small_c, idxq, counts = O.voxel_to_aux_index(coords_np, s)
    m = idx.M
    assert m == small_c.shape[0]
    assert np.array_equal(idx.block_coords.cpu().numpy(), small_c)
    assert idx.idx_query.dtype == torch.int64 and np.array_equal(idx.idx_query.cpu().numpy(), idxq)
    assert idx.counts.dtype == torch.int32 and np.array_equal(idx.counts.cpu().numpy(), counts)
    assert np.array_equal(idx.vox_blk.cpu().numpy(), idxq.astype(np.int32))
    # perm groups voxels by block, ascending voxel id inside a block (stable sort by block id)
    perm = idx.perm.cpu().numpy()
    assert np.array_equal(perm, np.argsort(idxq, kind="stable").astype(np.int32))
    bs = idx.blk_start.cpu().numpy()[: m + 1]
    assert np.array_equal(bs, np.concatenate([[0], np.cumsum(counts)]).astype(np.int32))
    for r in rs:
        nbr = O.neighbor_index(small_c, r)
        assert np.array_equal(idx.neighbor_map(r).cpu().numpy(), nbr)
        # adjoint relation: j in nbr_t(m) <=> m in nbr(j)
        nt = idx.neighbor_map(r, transpose=True).cpu().numpy()
        fwd = {(i, int(j)) for i in range(m) for j in nbr[i] if j >= 0}
        bwd = {(int(j), i) for i in range(m) for j in nt[i] if j >= 0}
        if m <= 2000:
            assert fwd == bwd
    # scratch is self-cleaning: counters are all zero again
    from link_amd.index import _workspace
    assert int(_workspace(c.device).cell_counts.abs().sum()) == 0
    return idx


@pytest.mark.parametrize("name", golden_files("g_agg_*_w8.npz"))
def test_index_golden(name):
    g = load_golden(name)
    idx = check_index(g["coords"], g["meta"]["s"], rs=(g["meta"]["r"],))
    assert np.array_equal(idx.block_coords.cpu().numpy(), g["small_c"])
    assert np.array_equal(idx.idx_query.cpu().numpy(), g["idx_query"])
    assert np.array_equal(idx.counts.cpu().numpy(), g["counts"])
    if "nbr" in g:
        assert np.array_equal(idx.neighbor_map(g["meta"]["r"]).cpu().numpy(), g["nbr"])


def test_index_edge_cases():
    import link_amd as la
    rng = np.random.default_rng(0)
    # negatives, several batches, duplicates, s not dividing the extent
    c = rng.integers(-40, 40, (5000, 4)).astype(np.int32)
    c[:, 3] = rng.integers(0, 3, 5000)
    for s in (1, 2, 3, 7, 50):
        check_index(c, s)
    # single voxel, two identical voxels
    check_index(np.array([[5, -3, 2, 0]], np.int32), 3)
    check_index(np.array([[5, -3, 2, 0], [5, -3, 2, 0]], np.int32), 3)
    # one huge block (all voxels in one block): long segment sort
    c = rng.integers(0, 30, (6000, 4)).astype(np.int32); c[:, 3] = 0
    check_index(c, 64)
    # supplied bounds (no bbox sync) incl. generous ones
    c = s_uniform(3000, grid=64, seed=3).numpy()
    check_index(c, 7, bounds=((0, 0, 0, 0), (63, 63, 63, 0)))
    check_index(c, 7, bounds=((-10, -20, -30, 0), (100, 90, 80, 2)))
    # voxels outside supplied bounds are reported, not silently dropped
    idx = la.BlockIndex(torch.from_numpy(c).cuda(), 7, bounds=((0, 0, 0, 0), (31, 31, 31, 0)))
    with pytest.raises(la._lib.LinkAmdError):
        idx.M
    # empty frame
    e = la.BlockIndex(torch.empty((0, 4), dtype=torch.int32, device="cuda"), 3)
    assert e.M == 0


def test_index_lidar_like_large_grid():
    """Sparse surface-like frame: big dense grid (multi-workgroup look-back scan), few occupied cells."""
    c = lidar_like(60000, seed=1)
    check_index(c, 3, rs=(2,))
    check_index(c, 6, rs=(3,))


def test_index_config_checkpoints():
    """cfg1 / cfg2 of BASELINE.json: M and sha256 of the index arrays as produced by the reference."""
    import link_amd as la
    with open(os.path.join(GOLDEN, "g_size.json")) as f:
        sizes = json.load(f)["sizes"]
    for key, n, s in (("N10000_s7", 10_000, 7), ("N100000_s7", 100_000, 7), ("N100000_s3", 100_000, 3)):
        idx = la.BlockIndex(s_uniform(n).cuda(), s)
        assert idx.M == sizes[key]["M"]
        sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()
        assert sha(idx.idx_query) == sizes[key]["sha256_idx"]
        assert sha(idx.counts) == sizes[key]["sha256_counts"]
        assert sha(idx.block_coords) == sizes[key]["sha256_small_c"]
        assert sha(idx.neighbor_map(3)) == sizes[key]["sha256_nbr_r3"]


def test_foreign_neighbor_map():
    from link_amd.index import foreign_neighbor_map
    rng = np.random.default_rng(5)
    rows = np.unique(rng.integers(-9, 9, (800, 4)).astype(np.int32), axis=0)
    rows[:, 3] = np.abs(rows[:, 3]) % 2
    rows = np.unique(rows, axis=0)
    rng.shuffle(rows)          # NOT sorted: ids are row positions
    for r in (2, 3):
        out = foreign_neighbor_map(torch.from_numpy(rows).cuda(), r).cpu().numpy()
        assert np.array_equal(out, O.neighbor_index(rows, r))


def test_index_without_bounds_is_built_on_the_last_grid_in_one_round_trip():
    """BlockIndex(coords, s) with nobody telling the bounds (voxel_to_aux through the reference's surface): from the second frame
    of a block edge on, the index is built on the last frame's grid while the bounding box is taken in the same pass (index.py:
    SPECULATE_GRID).  Same extents / a frame inside the grid: accepted, everything bit for bit what the oracle says, `bounds` = the
    measured box, M already on the host.  A frame beyond the grid, and one that fills less than half of it: built again."""
    import link_amd as la
    from link_amd import index as I
    s = 5
    I._SPEC_GRID.clear()
    a = s_uniform(9000, grid=60, seed=3).numpy(); a[0, :3] = 0; a[1, :3] = 59
    b = s_uniform(8000, grid=60, seed=4).numpy(); b[0, :3] = 0; b[1, :3] = 59
    inside = s_uniform(7000, grid=50, seed=5).numpy() + np.array([3, 4, 5, 0], np.int32)
    beyond = s_uniform(7000, grid=60, seed=6).numpy() + np.array([30, 0, 0, 0], np.int32)
    small = s_uniform(800, grid=20, seed=7).numpy() + np.array([35, 2, 2, 0], np.int32)
    negative = s_uniform(5000, grid=40, seed=8).numpy() - np.array([17, 3, 9, 0], np.int32)
    spec_before = None
    for name, c in [("a", a), ("b", b), ("inside", inside), ("beyond", beyond), ("small", small), ("negative", negative), ("negative2", negative)]:
        ct = torch.from_numpy(np.ascontiguousarray(c)).cuda()
        idx = la.BlockIndex(ct, s)
        small_c, idxq, counts = O.voxel_to_aux_index(c, s)
        spec_taken = idx._m is not None                       # M came with the one round trip
        assert spec_taken == (name in ("b", "inside", "negative2")), (name, spec_taken)
        assert idx.M == small_c.shape[0], name
        assert np.array_equal(idx.block_coords.cpu().numpy(), small_c), name
        assert np.array_equal(idx.idx_query.cpu().numpy(), idxq) and np.array_equal(idx.counts.cpu().numpy(), counts), name
        assert np.array_equal(idx.perm.cpu().numpy(), np.argsort(idxq, kind="stable").astype(np.int32)), name
        assert idx.bounds == (tuple(int(v) for v in c.min(0)), tuple(int(v) for v in c.max(0))), name
        for r in (2, 3):
            assert np.array_equal(idx.neighbor_map(r).cpu().numpy(), O.neighbor_index(small_c, r)), (name, r)
        if spec_taken:
            assert I._SPEC_GRID[(ct.device, s)] == spec_before   # an accepted guess does not move the remembered grid
        spec_before = I._SPEC_GRID[(ct.device, s)]
    # voxel_to_aux / aux_to_voxel through the surface on a guessed grid: the same rows as with the guess switched off
    x = torch.randn(b.shape[0], 32, generator=torch.Generator().manual_seed(1)).cuda()
    ct = torch.from_numpy(np.ascontiguousarray(b)).cuda()
    outs = []
    for flag in (True, False):
        I.SPECULATE_GRID = flag
        try:
            st = la.SparseTensor(x, ct.clone(), 1)
            sm, iq, cn = la.voxel_to_aux(st, s)
            outs.append((la.aux_to_voxel(sm, st, iq, cn, 3).F, sm.C, iq, cn))
        finally:
            I.SPECULATE_GRID = True
    for u, v in zip(*outs):
        assert torch.equal(u, v)
