"""link_index_build_first (include/link_amd.h section B): the block index with blocks numbered in first-voxel order -- a scan over
the voxels instead of over every cell of the grid -- against link_index_build (the reference's numbering, utils.py:44-58) on the
same frames: the same blocks, counts and voxel -> block map up to the order of the blocks; and ElkCorePlan's general layout on
either numbering against the CPU oracle."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import lidar_like, rel_err, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


class _Tables:
    def __init__(self, n_cap, grid, first):
        from link_amd import _lib as L
        dev = "cuda"
        i32 = dict(dtype=torch.int32, device=dev)
        self.L, self.grid, self.first, v = L, grid, first, grid.cells
        self.cell_counts = torch.zeros(v, **i32)
        self.cell_pair = torch.zeros(v, dtype=torch.int64, device=dev)
        self.nbytes = L.lib().link_index_scratch_bytes(n_cap, v)
        self.scratch = torch.randint(0, 255, (self.nbytes,), dtype=torch.uint8, device=dev)
        self.cell_blk = torch.zeros(v, **i32)
        self.vox_blk = torch.full((n_cap,), -7, **i32)
        self.idx_query = torch.empty(n_cap, dtype=torch.int64, device=dev)
        self.perm = torch.full((n_cap,), -7, **i32)
        self.vox_sorted = torch.empty((n_cap, 4), **i32)
        self.pos_blk = torch.empty(n_cap, **i32)
        self.blk_start = torch.full((n_cap + 1,), -7, **i32)
        self.blk_coords = torch.full((n_cap, 4), -7, **i32)
        self.counts = torch.full((n_cap,), -7, **i32)
        self.hdr = torch.zeros(L.HDR_WORDS, **i32)

    def build(self, coords):
        L, n = self.L, coords.shape[0]
        st = L.current_stream_handle()
        tail = (self.cell_blk.data_ptr(), self.vox_blk.data_ptr(), self.idx_query.data_ptr(), self.perm.data_ptr(),
                self.vox_sorted.data_ptr(), self.pos_blk.data_ptr(), self.blk_start.data_ptr(), self.blk_coords.data_ptr(),
                self.counts.data_ptr(), self.hdr.data_ptr(), st)
        if self.first:
            L.check(L.lib().link_index_build_first(coords.data_ptr(), n, ctypes.byref(self.grid), self.cell_pair.data_ptr(),
                                                   self.scratch.data_ptr(), self.nbytes, *tail),
                    "link_index_build_first")
        else:
            L.check(L.lib().link_index_build(coords.data_ptr(), n, ctypes.byref(self.grid), self.cell_counts.data_ptr(),
                                             self.scratch.data_ptr(), self.nbytes, *tail), "link_index_build")
        torch.cuda.synchronize()
        h = self.hdr.tolist()
        return h[L.HDR_M], h[L.HDR_STATUS], h[L.HDR_NVALID]


def _check_frame(a, b, coords, s):
    """a: first-voxel numbering, b: cell-order numbering of the same frame."""
    n = coords.shape[0]
    (ma, sa, na), (mb, sb, nb) = a.build(coords), b.build(coords)
    assert (ma, sa, na) == (mb, sb, nb) and sa == 0 and na == n
    m = ma
    ca, cb = a.blk_coords[:m].cpu().numpy(), b.blk_coords[:m].cpu().numpy()
    # the same set of blocks; a's order = order of each block's smallest voxel id
    order = np.lexsort(ca.T[::-1])
    assert np.array_equal(ca[order], cb[np.lexsort(cb.T[::-1])])
    va, vb = a.vox_blk[:n].cpu().numpy(), b.vox_blk[:n].cpu().numpy()
    assert np.array_equal(ca[va], cb[vb])                                   # every voxel in the same block
    assert np.array_equal(a.idx_query[:n].cpu().numpy(), va.astype(np.int64))
    first_id = np.full(m, n, np.int64)
    np.minimum.at(first_id, va, np.arange(n))
    assert np.all(np.diff(first_id) > 0)                                    # blocks in order of their first voxel
    cnt = np.bincount(va, minlength=m)
    assert np.array_equal(a.counts[:m].cpu().numpy(), cnt)
    bs = a.blk_start[:m + 1].cpu().numpy()
    assert np.array_equal(bs, np.concatenate([[0], np.cumsum(cnt)]))
    perm = a.perm[:n].cpu().numpy()
    assert np.array_equal(np.sort(perm), np.arange(n))
    assert np.array_equal(va[perm], np.repeat(np.arange(m), cnt))           # grouped by block ...
    assert np.array_equal(a.pos_blk[:n].cpu().numpy(), va[perm])
    for k in np.flatnonzero(cnt > 1)[:2000]:
        assert np.all(np.diff(perm[bs[k]:bs[k + 1]]) > 0)                    # ... ascending voxel id inside a block
    vs = a.vox_sorted[:n].cpu().numpy()
    assert np.array_equal(vs[:, 3], perm) and np.array_equal(vs[:, :3], coords.cpu().numpy()[perm, :3])
    # the cell table: block + 1 at the cells of this frame, 0 everywhere else (nothing of an earlier frame left)
    cell_a, cell_b = a.cell_blk.cpu().numpy(), b.cell_blk.cpu().numpy()
    assert np.array_equal(cell_a > 0, cell_b > 0)
    occ = np.flatnonzero(cell_a > 0)
    assert np.array_equal(ca[cell_a[occ] - 1], cb[cell_b[occ] - 1])
    assert int(a.cell_pair.abs().sum()) == 0                                # the pair words cleaned themselves
    return m


def test_first_voxel_numbering_matches_cell_numbering_up_to_block_order():
    from link_amd import _lib as L
    from link_amd.index import coords_bounds
    frames = [torch.from_numpy(lidar_like(40000, seed=1)).cuda(), torch.from_numpy(lidar_like(9000, seed=2)).cuda(),
              torch.from_numpy(lidar_like(60000, seed=3)).cuda()]
    frames.append(frames[2][torch.randperm(frames[2].shape[0], device="cuda")].contiguous())     # random voxel order
    frames.append(frames[0][:1].contiguous())
    frames.append(frames[1])
    allc = torch.cat(frames)
    s = 12
    bounds = coords_bounds(allc)
    grid = L.grid_from_bounds(bounds[0], bounds[1], s)
    n_cap = max(f.shape[0] for f in frames)
    a, b = _Tables(n_cap, grid, True), _Tables(n_cap, grid, False)
    ms = [_check_frame(a, b, f, s) for f in frames]              # every frame on the tables the previous one left behind
    assert ms[2] == ms[3] and ms[4] == 1
    # a voxel outside the grid: status bit 0, as link_index_build reports it
    bad = frames[1].clone()
    bad[5, 0] = bounds[1][0] + 10 * s
    (_, sa, _), (_, sb, _) = a.build(bad), b.build(bad)
    assert sa == sb == 1


@pytest.mark.parametrize("C,groups,baseop,s,r", [(64, 1, "cos_x", 6, 2), (32, 2, "cos", 12, 3), (16, 2, "sin", 8, 2)])
def test_general_layout_plan_on_either_numbering(C, groups, baseop, s, r):
    """ElkCorePlan(layout='general') with the index from the voxel scan (default) and from the cell scan: both against the oracle,
    each bitwise repeatable, frames of different sizes alternating through the same arena."""
    import link_amd as la
    from link_amd.index import coords_bounds
    torch.manual_seed(C + r)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop, variant="encoder").cuda().eval()
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    frames = []
    for seed, npts in ((3, 30000), (4, 8000), (5, 30000)):
        coords = torch.from_numpy(lidar_like(npts, seed=seed, voxel=0.2 if C == 16 else 0.1))      # (C = 16: coarser voxels keep fp32 theta at large coordinates inside the gate)
        frames.append((coords, torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(seed))))
    bounds = coords_bounds(torch.cat([c for c, _ in frames]).cuda())
    n_cap = max(c.shape[0] for c, _ in frames)
    plans = {}
    for order in ("first", "cell"):
        p = la.ElkCorePlan(n_cap, C, baseop, C // groups, r, s, bounds, "cuda", layout="general", block_order=order)
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
               blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
        plans[order] = p
    for coords, feats in frames + frames[:2]:
        ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", agg=O.aggregate_c).numpy()
        f, c = feats.cuda(), coords.cuda()
        got = {k: p.run(f, c).clone() for k, p in plans.items()}
        assert plans["first"].blocks() == plans["cell"].blocks() > 0
        for k, p in plans.items():
            assert rel_err(got[k].cpu().numpy(), ref) < TOL, k
            assert torch.equal(p.run(f, c), got[k])                          # rebuilt: bitwise
            assert torch.equal(p.run(f, c, build_index=False), got[k])       # warm: bitwise
        assert rel_err(got["first"].cpu().numpy(), got["cell"].cpu().numpy()) < 1e-5
