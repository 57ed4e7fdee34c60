"""Tile form of R_core on the general layout (include/link_amd.h section C: link_elk_premix_modsum_tiles +
link_elk_gather_demod_tiles, two launches) against the CPU oracle and against the four-kernel form, on the frames it is
made for: LiDAR-like block grids with big blocks (reference call sites linkunet.py:345-363, scn.py:586-607)."""
import numpy as np
import pytest
import torch

from helpers import lidar_like, rel_err, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _plan_pair(la, blk, n, C, baseop, groups, r, s, coords, coord_div=1.0):
    from link_amd.index import coords_bounds
    bounds = coords_bounds(coords)
    plans = []
    for tiles in (True, False):
        # block_order="cell": the numbering of the module path (BlockIndex = the reference's), so that the bitwise comparisons
        # with it below hold; the voxel-scan numbering is covered by tests/test_gpu_index_first.py
        p = la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, coords.device, coord_div=coord_div, layout="general", tiles=tiles,
                           block_order="cell")
        p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
               blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
        plans.append(p)
    assert plans[0].tiles and not plans[1].tiles
    return plans


def _solid_and_sparse(n_box, n_noise, seed):
    """A fully occupied box (blocks of s^3 voxels: every block spans many 16-voxel tiles, waves and workgroups) next to
    scattered single voxels (one-voxel blocks: a tile holds 16 runs), shuffled."""
    g = np.random.default_rng(seed)
    e = int(round(n_box ** (1 / 3)))
    box = np.stack(np.meshgrid(np.arange(e), np.arange(e), np.arange(e), indexing="ij"), -1).reshape(-1, 3) + 40
    noise = np.unique(g.integers(0, 200, size=(n_noise, 3)), axis=0)
    noise = noise[~((noise >= 40) & (noise < 40 + e)).all(1)]
    xyz = np.concatenate([box, noise], 0)
    xyz = xyz[g.permutation(len(xyz))]
    return torch.from_numpy(np.concatenate([xyz, np.zeros((len(xyz), 1), np.int64)], 1).astype(np.int32))


@pytest.mark.parametrize("C,groups,baseop,s,r", [(16, 2, "cos", 7, 3), (32, 2, "cos", 7, 3), (64, 2, "cos", 7, 3),
                                                 (128, 2, "cos", 7, 3), (64, 1, "cos_x", 6, 2), (32, 1, "sin", 5, 3),
                                                 (128, 1, "cos_x", 4, 2), (16, 1, "cos_x", 3, 3), (64, 4, "cos", 7, 3)])
def test_tiles_vs_oracle_big_and_tiny_blocks(C, groups, baseop, s, r):
    import link_amd as la
    torch.manual_seed(C + r)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    with torch.no_grad():
        for nme, p in blk.named_parameters():
            if "norm" in nme or "pre_mix.1" in nme or nme == "alpha":
                p.add_(0.2 * torch.randn_like(p))
    coords = _solid_and_sparse(21 ** 3, 3000, seed=C)
    n = coords.shape[0]
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(1))
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, agg=O.aggregate_c).numpy()
    tp, fp = _plan_pair(la, blk, n, C, baseop, groups, r, s, coords.cuda())
    f, c = feats.cuda(), coords.cuda()
    got = tp.run(f, c).clone()
    four = fp.run(f, c).clone()
    assert tp.blocks() == fp.blocks() > 0
    assert rel_err(got.cpu().numpy(), ref) < TOL
    assert rel_err(got.cpu().numpy(), four.cpu().numpy()) < 1e-4      # both inside the gate; hardware vs polynomial sincos, summation order
    # warm index, repeated: bitwise the same (fixed summation order, no atomics)
    for _ in range(5):
        assert torch.equal(tp.run(f, c, build_index=False), got)


@pytest.mark.parametrize("stride,baseop,groups,s,r", [(1, "cos", 2, 14, 3), (2, "cos_x", 1, 6, 2), (1, "cos", 2, 7, 3)])
def test_tiles_on_lidar_like_frame(stride, baseop, groups, s, r):
    import link_amd as la
    torch.manual_seed(4)
    C = 64
    coords = torch.from_numpy(lidar_like(40000, seed=3, stride=stride))
    n = coords.shape[0]
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop, variant="encoder").cuda().eval()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(5))
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                           agg=O.aggregate_c).numpy()
    div = float(stride) if baseop == "cos_x" else 1.0
    tp, fp = _plan_pair(la, blk, n, C, baseop, groups, r, s, coords.cuda(), coord_div=div)
    got = tp.run(feats.cuda(), coords.cuda())
    assert rel_err(got.cpu().numpy(), ref) < TOL
    # the module path takes the same two kernels once the coordinate set has a block index (without one it runs the lean
    # form, three launches with the index rebuilt: tests/test_gpu_lean.py), and always with the lean form switched off
    from link_amd import elk as E
    from link_amd.aggregate import link_index_of
    st = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
    link_index_of(st, s)
    with torch.no_grad():
        core = blk._core(st, s, r, blk.pos_weight[0].weight, blk.alpha if baseop == "cos_x" else None, C // groups, div)
    assert torch.equal(core, got)
    E.LEAN_FORM = False
    try:
        st = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
        with torch.no_grad():
            core = blk._core(st, s, r, blk.pos_weight[0].weight, blk.alpha if baseop == "cos_x" else None, C // groups, div)
        assert torch.equal(core, got)
    finally:
        E.LEAN_FORM = True


@pytest.mark.parametrize("n", [1, 15, 16, 17, 63, 64, 65, 257, 1000])
def test_tiles_ragged_sizes(n):
    """Frame sizes around the tile (16), wave (64) and workgroup boundaries, one block and many blocks."""
    import link_amd as la
    torch.manual_seed(n)
    C, groups, s, r = 32, 2, 4, 3
    blk = la.ELKBlock(C, C, groups=groups, baseop="cos").cuda().eval()
    one_block = np.stack(np.unravel_index(np.random.default_rng(n).permutation(11 ** 3)[:n], (11, 11, 11)), 1)
    one_block = torch.from_numpy(np.concatenate([one_block, np.zeros((n, 1), np.int64)], 1).astype(np.int32))
    for coords, s_ in ((one_block, 16), (s_uniform(n, grid=40, seed=n), s)):     # everything in one block / scattered
        feats = torch.randn(n, C, generator=torch.Generator().manual_seed(2))
        params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
        ref = O.elk_core_torch(feats, coords, params, s_, r, "cos", groups, agg=O.aggregate_c).numpy()
        tp, _ = _plan_pair(la, blk, n, C, "cos", groups, r, s_, coords.cuda())
        got = tp.run(feats.cuda(), coords.cuda())
        assert rel_err(got.cpu().numpy(), ref) < TOL, (n, s_)


def test_tiles_plan_capacity_larger_than_frame_and_large_rows():
    """A plan sized for more voxels than the frame holds (the table's scratch rows are sized by capacity); feature rows and
    weights outside the fp16 split's range take the fp32 matrix-core instruction."""
    import link_amd as la
    torch.manual_seed(9)
    C, groups, s, r, n = 64, 2, 7, 3, 20000
    blk = la.ELKBlock(C, C, groups=groups, baseop="cos").cuda().eval()
    coords = torch.from_numpy(lidar_like(n, seed=8, stride=1))
    n = coords.shape[0]
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(3))
    feats[::97] *= 1e5                                       # rows beyond 2^15
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, "cos", groups, agg=O.aggregate_c).numpy()
    from link_amd.index import coords_bounds
    plan = la.ElkCorePlan(3 * n + 11, C, "cos", C // groups, r, s, coords_bounds(coords.cuda()), torch.device("cuda:0"), layout="general")
    assert plan.tiles
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
              blk.norm.weight, blk.norm.bias)
    got = plan.run(feats.cuda(), coords.cuda())
    assert rel_err(got.cpu().numpy(), ref) < TOL


def test_tiles_randomized_against_the_four_kernel_form():
    """60 random (frame shape, size, block edge, r, width, op) draws: the tile form against the four-kernel form of the same
    layout (both inside the oracle's gate on every case the oracle tests above cover; here the point is the rare paths --
    blocks ending exactly at tile / wave / workgroup boundaries, runs through several waves, tiles of one-voxel blocks)."""
    import link_amd as la
    g = np.random.default_rng(2024)
    worst = 0.0
    for case in range(60):
        C = int(g.choice([16, 32, 64, 128]))
        baseop = str(g.choice(["cos", "sin", "cos_x"]))
        groups = int(g.choice([1, 2])) if baseop != "cos_x" else 1
        r = int(g.choice([2, 3]))
        s = int(g.choice([2, 3, 4, 7, 12]))
        kind = int(g.integers(0, 3))
        if kind == 0:                                   # solid box (+ noise): blocks of s^3 voxels
            coords = _solid_and_sparse(int(g.integers(8, 20)) ** 3, int(g.integers(1, 2000)), seed=case)
        elif kind == 1:                                 # LiDAR-like surface
            coords = torch.from_numpy(lidar_like(int(g.integers(500, 30000)), seed=case, stride=1))
        else:                                           # scattered
            coords = s_uniform(int(g.integers(1, 20000)), grid=int(g.integers(30, 120)), seed=case)
        n = coords.shape[0]
        torch.manual_seed(case)
        blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
        feats = torch.randn(n, C, generator=torch.Generator().manual_seed(case)).cuda()
        tp, fp = _plan_pair(la, blk, n, C, baseop, groups, r, s, coords.cuda())
        a = tp.run(feats, coords.cuda()).clone()
        b = fp.run(feats, coords.cuda()).clone()
        assert tp.blocks() == fp.blocks()
        err = rel_err(a.cpu().numpy(), b.cpu().numpy())
        worst = max(worst, err)
        assert err < 1e-4, (case, C, baseop, groups, r, s, kind, n, err)
        assert torch.equal(tp.run(feats, coords.cuda(), build_index=False), a), case
    assert worst > 0.0                                  # the two forms do differ in the last bits (sanity of the comparison)


@pytest.mark.parametrize("dtype,tol_round,tol_oracle", [(torch.float16, 1.2e-3, 6e-3), (torch.bfloat16, 9e-3, 5e-2)])
@pytest.mark.parametrize("C,groups,baseop,s,r", [(16, 2, "cos", 7, 3), (32, 2, "cos", 7, 3), (64, 1, "cos_x", 6, 2), (128, 2, "cos", 7, 3)])
def test_tiles_half_rows_at_the_boundary(dtype, tol_round, tol_oracle, C, groups, baseop, s, r):
    """fp16 / bf16 feature rows in and out of the two tile-form kernels (link_elk_*_tiles_io; what TSELKBlock runs under
    autocast, scn.py:586-607 with fp16_enabled): against the fp32 tile form on the SAME (already rounded) rows -- the only
    difference left is the output rounding -- and against the oracle on the unrounded rows.  Tolerances as for the dense
    layout's half rows (tests/test_gpu_dense.py)."""
    import link_amd as la
    from link_amd.elk import elk_core_fused
    from link_amd.index import BlockIndex
    torch.manual_seed(C)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    coords = torch.from_numpy(lidar_like(20000, seed=C + r)).cuda()
    n = coords.shape[0]
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(2))
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords.cpu(), params, s, r, baseop, groups, agg=O.aggregate_c).numpy()
    index = BlockIndex(coords, s)
    args = (coords, index, blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
            blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias, baseop, C // groups, r)
    fh = feats.cuda().to(dtype)
    with torch.no_grad():
        got = elk_core_fused(fh, *args)
        full = elk_core_fused(fh.float(), *args)
    assert got.dtype == dtype and full.dtype == torch.float32
    assert rel_err(got.float().cpu().numpy(), full.cpu().numpy()) < tol_round
    # the plan path (one FFI call per step, link_elk_buffers_t::io_dtype): the same rows, bit for bit; the four-kernel form
    # refuses 16-bit rows
    tp, fp = _plan_pair(la, blk, n, C, baseop, groups, r, s, coords)
    assert torch.equal(tp.run(fh, coords), got)
    from link_amd._lib import LinkAmdError
    with pytest.raises(LinkAmdError):
        fp.run(fh, coords)
    assert rel_err(got.float().cpu().numpy(), ref) < tol_oracle
    # a module cast to half as a whole (parameters too) reads its parameters as fp32 copies: same rows out
    blk_h = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    blk_h.load_state_dict(blk.state_dict())
    blk_h = blk_h.to(dtype)
    # (the position weights stay fp32: theta = w . xyz with |xyz| ~ 10^3 does not survive a 16-bit w)
    args_h = (coords, index, blk_h.pre_mix[0].weight, blk_h.pre_mix[1].weight, blk_h.pre_mix[1].bias, blk.pos_weight[0].weight,
              blk.alpha if baseop == "cos_x" else None, blk_h.norm.weight, blk_h.norm.bias, baseop, C // groups, r)
    with torch.no_grad():
        got_h = elk_core_fused(fh, *args_h)
    assert got_h.dtype == dtype and rel_err(got_h.float().cpu().numpy(), ref) < 4 * tol_oracle
