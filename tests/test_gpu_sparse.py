"""Sparse-cell layout of R_core (include/link_amd.h: link_elk_core_sparse_forward; link_amd/csrc/dense_gather_sparse_impl.h) --
dense-cell addressing with a sparse iteration, three launches with the index rebuilt -- against the CPU oracle and the general
layout on the frames it is made for: LiDAR-shaped block grids with small blocks (reference call sites linkunet.py:345-363)."""
import numpy as np
import pytest
import torch

from helpers import lidar_like, rel_err, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _plans(la, blk, n, C, baseop, groups, r, s, coords, slot_cap, coord_div=1.0):
    from link_amd.index import coords_bounds
    bounds = coords_bounds(coords)
    out = []
    for layout, kw in (("sparse", {"slot_cap": slot_cap}), ("general", {})):
        p = la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, coords.device, coord_div=coord_div, layout=layout, **kw)
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
               blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
        out.append(p)
    assert out[0].sparse and out[0].dense and not out[1].dense
    return out


@pytest.mark.parametrize("C,groups,baseop,stride,s,r", [(64, 1, "cos_x", 2, 6, 2), (64, 2, "cos", 2, 6, 3), (32, 2, "sin", 2, 6, 2),
                                                        (16, 2, "cos", 2, 8, 3), (64, 1, "cos_x", 4, 12, 2), (32, 1, "cos_x", 2, 4, 3)])
def test_sparse_layout_on_lidar_like_frames(C, groups, baseop, stride, s, r):
    """LiDAR-like frame at tensor stride `stride` (coordinates are multiples of it), block edge s in coordinate units: a block
    holds (s / stride)^3 voxel sites = the slot capacity.  Oracle, general layout (block count too), warm == rebuilt, and a
    second frame through the same plan (stale counts of the first frame's cells must be gone)."""
    import link_amd as la
    torch.manual_seed(C + r)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop, variant="encoder").cuda().eval()
    with torch.no_grad():
        for nme, p in blk.named_parameters():
            if "norm" in nme or "pre_mix.1" in nme or nme == "alpha":
                p.add_(0.2 * torch.randn_like(p))
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    cap = (s // stride) ** 3
    frames = []
    for seed, npts in ((3, 30000), (4, 9000)):
        # (C = 16: a coarser voxel grid -- theta = w . p in fp32 carries ~|p| 2^-24 of rounding, and against the oracle's
        # differently associated fp32 Linear that alone is 2e-4 of a 16-channel row at |p| ~ 2000; every layout of this
        # library agrees there to 3e-7)
        coords = torch.from_numpy(lidar_like(npts, seed=seed, stride=stride, voxel=0.2 if C == 16 else 0.05))
        feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(seed))
        frames.append((coords, feats))
    n_cap = max(c.shape[0] for c, _ in frames)
    allc = torch.cat([c for c, _ in frames])
    sp, ge = _plans(la, blk, n_cap, C, baseop, groups, r, s, allc.cuda(), cap, coord_div=float(stride) if baseop == "cos_x" else 1.0)
    for coords, feats in frames + frames[:1]:           # frame A, frame B, frame A again: the marks alternate, counts are cleaned
        n = coords.shape[0]
        ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                               agg=O.aggregate_c).numpy()
        f, c = feats.cuda(), coords.cuda()
        got = sp.run(f, c).clone()
        gen = ge.run(f, c).clone()
        assert sp.blocks() == ge.blocks() > 0
        assert rel_err(got.cpu().numpy(), ref) < TOL
        assert rel_err(got.cpu().numpy(), gen.cpu().numpy()) < 2 * TOL     # two results, each within TOL of the oracle
        assert torch.equal(sp.run(f, c, build_index=False), got)          # warm index: bitwise
        for _ in range(3):
            # rebuilt: the wave that owns a cell is the one whose id range holds the voxel inserted FIRST (atomic order), so a
            # cell's rows can sit at other positions of a matrix-core tile and its fp32 sum associate differently: last-bit
            # agreement, not bitwise (the cell-range form of cos_x sums a cell's rows in id order: bitwise)
            again = sp.run(f, c)
            if baseop == "cos_x":
                assert torch.equal(again, got)
            else:
                assert rel_err(again.cpu().numpy(), got.cpu().numpy()) < 2e-6
        assert int(sp.cnt.abs().sum().item()) == 0                         # the counters cleaned themselves


def test_sparse_layout_uniform_frame_and_half_rows():
    """A frame in random voxel order (every cell's first voxel anywhere) with blocks of up to 27 voxels, and fp16 rows."""
    import link_amd as la
    torch.manual_seed(1)
    C, groups, baseop, s, r = 64, 2, "cos", 3, 3
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    coords = s_uniform(20000, grid=60, seed=9)
    feats = torch.randn(20000, C, generator=torch.Generator().manual_seed(2))
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, agg=O.aggregate_c).numpy()
    sp, ge = _plans(la, blk, 20000, C, baseop, groups, r, s, coords.cuda(), 27)
    got = sp.run(feats.cuda(), coords.cuda()).clone()
    ge.run(feats.cuda(), coords.cuda())
    assert sp.blocks() == ge.blocks() > 0
    assert rel_err(got.cpu().numpy(), ref) < TOL
    h = feats.half()
    ref_h = O.elk_core_torch(h.float(), coords, params, s, r, baseop, groups, agg=O.aggregate_c).numpy()
    got_h = sp.run(h.cuda(), coords.cuda()).float()
    assert rel_err(got_h.cpu().numpy(), ref_h) < 2e-3
    with pytest.raises(la._lib.LinkAmdError):            # blocks bigger than the layout takes: refused, not silently slow
        la.ElkCorePlan(1000, C, baseop, C // groups, r, 7, ((0, 0, 0, 0), (59, 59, 59, 0)), torch.device("cuda"), layout="sparse")
