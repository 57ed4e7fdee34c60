"""Lean form of R_core with the index rebuilt every call (include/link_amd.h: link_elk_core_lean_forward;
link_amd/csrc/elk_lean_impl.h) -- three launches, tables addressed by grid cell -- against the CPU oracle and the general layout
on the frames it is made for: LiDAR-shaped block grids (reference call sites linkunet.py:345-363, scn.py:586-607), every width of
the detection backbone, blocks of a few to a few hundred voxels."""
import numpy as np
import pytest
import torch

from helpers import lidar_like, rel_err, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _bind(p, blk, baseop):
    return p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
                  blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)


def _plans(la, blk, n, C, baseop, groups, r, s, coords, slot_cap, coord_div=1.0):
    from link_amd.index import coords_bounds
    bounds = coords_bounds(coords)
    lean = _bind(la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, coords.device, coord_div=coord_div, layout="lean",
                                slot_cap=slot_cap), blk, baseop)
    gen = _bind(la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, coords.device, coord_div=coord_div, layout="general"), blk, baseop)
    assert lean.lean and not lean.dense and not gen.dense
    return lean, gen


def _block(la, C, groups, baseop, seed):
    torch.manual_seed(seed)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop, variant="encoder").cuda().eval()
    with torch.no_grad():
        for nme, p in blk.named_parameters():
            if "norm" in nme or "pre_mix.1" in nme or nme == "alpha":
                p.add_(0.2 * torch.randn_like(p))
    return blk, {k: v.detach().cpu() for k, v in blk.state_dict().items()}


@pytest.mark.parametrize("C,groups,baseop,stride,s,r", [(64, 1, "cos_x", 2, 6, 2), (64, 2, "cos", 2, 6, 3), (32, 2, "sin", 2, 6, 2),
                                                        (16, 2, "cos", 2, 8, 3), (64, 1, "cos_x", 4, 12, 2), (32, 1, "cos_x", 2, 4, 3),
                                                        (128, 2, "cos", 2, 6, 3), (128, 1, "cos_x", 2, 6, 2)])
def test_lean_form_on_lidar_like_frames(C, groups, baseop, stride, s, r):
    """LiDAR-like frames at tensor stride `stride`, block edge s in coordinate units (slot capacity (s / stride)^3): oracle,
    general layout (block count too), warm == rebuilt == rebuilt again BITWISE (sums run in id order whatever order the
    atomics handed out), and frames of different sizes alternating through one plan (the counters of the frame before must be
    gone, also when it was the bigger one)."""
    import link_amd as la
    blk, params = _block(la, C, groups, baseop, C + r)
    cap = (s // stride) ** 3
    frames = []
    for seed, npts in ((3, 30000), (4, 9000)):
        coords = torch.from_numpy(lidar_like(npts, seed=seed, stride=stride, voxel=0.2 if C == 16 else 0.05))
        feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(seed))
        frames.append((coords, feats))
    n_cap = max(c.shape[0] for c, _ in frames)
    allc = torch.cat([c for c, _ in frames])
    le, ge = _plans(la, blk, n_cap, C, baseop, groups, r, s, allc.cuda(), cap, coord_div=float(stride) if baseop == "cos_x" else 1.0)
    for coords, feats in frames + frames[:1]:
        n = coords.shape[0]
        ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                               agg=O.aggregate_c).numpy()
        f, c = feats.cuda(), coords.cuda()
        got = le.run(f, c).clone()
        gen = ge.run(f, c).clone()
        assert le.blocks() == ge.blocks() > 0
        assert rel_err(got.cpu().numpy(), ref) < TOL
        assert rel_err(got.cpu().numpy(), gen.cpu().numpy()) < 2 * TOL
        assert torch.equal(le.run(f, c, build_index=False), got)
        for _ in range(3):
            assert torch.equal(le.run(f, c), got)
        le.check()
        assert int(le.cnt2[le._cur ^ 1].abs().sum().item()) == 0          # the other parity's counters are clean


@pytest.mark.parametrize("C,r", [(16, 3), (64, 3), (128, 3), (32, 2)])
def test_lean_form_big_blocks(C, r):
    """Blocks of up to 343 voxels (s = 7 at tensor stride 1: the detection blocks, ts_elk.py:87,168): cells span several
    chunks of 32, chunk rows of the 27 neighbours are summed in a fixed order."""
    import link_amd as la
    groups, baseop, s = 2, "cos", 7
    blk, params = _block(la, C, groups, baseop, 7 * C + r)
    n = 24000
    coords = s_uniform(n, grid=40, seed=5)              # 64000 sites, 37.5 % occupied: ~130 voxels per 7^3 block
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(6))
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", agg=O.aggregate_c).numpy()
    le, ge = _plans(la, blk, n, C, baseop, groups, r, s, coords.cuda(), 0)
    got = le.run(feats.cuda(), coords.cuda()).clone()
    ge.run(feats.cuda(), coords.cuda())
    assert le.blocks() == ge.blocks() > 0
    assert rel_err(got.cpu().numpy(), ref) < TOL
    assert torch.equal(le.run(feats.cuda(), coords.cuda()), got)
    le.check()


def test_lean_form_half_rows_status_and_limits():
    """fp16 / bf16 rows at the kernel boundary; a voxel outside the plan's bounds is dropped and reported; an empty frame;
    geometry the form does not take is refused."""
    import link_amd as la
    C, groups, baseop, s, r = 64, 2, "cos", 3, 3
    blk, params = _block(la, C, groups, baseop, 11)
    n = 20000
    coords = s_uniform(n, grid=60, seed=9)
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(2))
    le, ge = _plans(la, blk, n, C, baseop, groups, r, s, coords.cuda(), 27)
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", agg=O.aggregate_c).numpy()
    got = le.run(feats.cuda(), coords.cuda()).clone()
    assert rel_err(got.cpu().numpy(), ref) < TOL
    for dt, tol in ((torch.float16, 2e-3), (torch.bfloat16, 1.6e-2)):
        h = feats.to(dt)
        ref_h = O.elk_core_torch(h.float(), coords, params, s, r, baseop, groups, variant="encoder", agg=O.aggregate_c).numpy()
        got_h = le.run(h.cuda(), coords.cuda())
        assert got_h.dtype == dt
        assert rel_err(got_h.float().cpu().numpy(), ref_h) < tol
    # empty frame, then the full one again
    le.run(feats[:0].cuda(), coords[:0].cuda())
    assert torch.equal(le.run(feats.cuda(), coords.cuda()), got)
    # a voxel outside the bounds: status bit 0, every other row as before
    bad = coords.clone()
    bad[17, 0] = 10_000
    out_bad = le.run(feats.cuda(), bad.cuda()).clone()
    with pytest.raises(la._lib.LinkAmdError):
        le.check()
    keep = torch.ones(n, dtype=torch.bool)
    keep[17] = False
    ref_bad = O.elk_core_torch(feats[keep], coords[keep], params, s, r, baseop, groups, variant="encoder", agg=O.aggregate_c).numpy()
    assert rel_err(out_bad.cpu().numpy()[keep.numpy()], ref_bad) < TOL
    assert torch.equal(le.run(feats.cuda(), coords.cuda()), got)
    le.check()
    with pytest.raises(la._lib.LinkAmdError):            # blocks of 8^3 voxels: beyond the slot capacity the form takes
        la.ElkCorePlan(1000, C, baseop, C // groups, r, 8, ((0, 0, 0, 0), (59, 59, 59, 0)), torch.device("cuda"), layout="lean")
    assert not la.ElkCorePlan.lean_supported(1000, 48, baseop, r, 3, ((0, 0, 0, 0), (59, 59, 59, 0)))


def test_module_path_takes_the_lean_form_for_coordinate_sets_without_an_index():
    """ELKBlock.forward (inference) on a LiDAR-like frame runs the lean form (a plan in the module's cache, no block index
    built), the same coordinate set again reuses the plan's lists and gives the same bits, a fresh coordinate set too; a
    coordinate set that already has a block index (an aggregation op built it) takes the tile form; all within the oracle's
    tolerance."""
    import link_amd as la
    from link_amd import elk as E
    from link_amd.aggregate import link_index_of
    C, groups, baseop, stride, s, r = 64, 1, "cos_x", 2, 6, 2
    blk, params = _block(la, C, groups, baseop, 5)
    coords = torch.from_numpy(lidar_like(20000, seed=8, stride=stride))
    feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(8))
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                           agg=O.aggregate_c)

    def core(st):
        with torch.no_grad():
            return blk._core(st, s, r, blk.pos_weight[0].weight, blk.alpha, C // groups, float(stride)).float().cpu()

    st = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
    first = core(st)
    assert len(blk.__dict__.get("_lean_plans", {})) == 1 and next(iter(blk._lean_plans.values())).lean
    assert not any(k[0] == "link_block_index" for k in st.kmaps)
    next(iter(blk._lean_plans.values())).check()
    # the same maps again: a coordinate set that comes back gets a block index and moves to the tile form (faster on a built index);
    # within the forms' agreement of the first visit, and the third visit gives the second's bits
    second = core(st)
    assert any(k[0] == "link_block_index" for k in st.kmaps)
    assert rel_err(second.numpy(), first.numpy()) < 2e-6 and rel_err(second.numpy(), ref.numpy()) < TOL
    assert torch.equal(core(st), second)
    E.LEAN_SECOND_VISIT_INDEX = False
    try:                                                # switch off: the set stays on the lean form, lists reused, same bits
        stb = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
        fb = core(stb)
        assert torch.equal(fb, first) and torch.equal(core(stb), first)
        assert not any(k[0] == "link_block_index" for k in stb.kmaps)
    finally:
        E.LEAN_SECOND_VISIT_INDEX = True
    st2 = la.SparseTensor(feats.cuda(), coords.cuda(), stride)          # a fresh coordinate set: rebuilt, same bits
    assert torch.equal(core(st2), first)
    assert rel_err(first.numpy(), ref.numpy()) < TOL
    st3 = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
    link_index_of(st3, s)                               # an index exists: the two tile launches
    tiles = core(st3)
    assert rel_err(tiles.numpy(), ref.numpy()) < TOL
    E.LEAN_FORM = False
    try:
        st4 = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
        assert torch.equal(core(st4), tiles)
    finally:
        E.LEAN_FORM = True


def test_module_path_reports_a_frame_that_breaks_the_stride_promise():
    """the module path sizes a block's slot list as (s_eff / stride)^3.  A tensor whose stride is set but whose
    coordinates are NOT multiples of it can put more voxels into a block: the kernel drops the surplus and raises status bit 1.
    The module path must not hand back uninitialised rows silently: the rows of dropped voxels are zero, the NEXT call raises
    (the status word travels behind the kernels, no sync), and from then on the general layout -- which has no such limit --
    serves that block edge / stride and agrees with the oracle."""
    import link_amd as la
    from link_amd import _lib as L
    C, groups, baseop, stride, s, r = 32, 2, "cos", 2, 6, 3
    blk, params = _block(la, C, groups, baseop, 11)
    coords = s_uniform(6000, grid=20, seed=3)            # unit-spaced coordinates: up to 216 voxels per block of edge 6, slot list holds 27
    feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(3))

    def core(st):
        with torch.no_grad():
            return blk._core(st, s, r, blk.pos_weight[0].weight, None, C // groups, 1.0).float().cpu()

    st = la.SparseTensor(feats.cuda(), coords.cuda(), stride)       # claims stride 2: the promise is broken
    first = core(st)
    assert torch.isfinite(first).all()
    assert (first.abs().sum(1) == 0).any()               # dropped voxels: zero rows, not whatever the allocator returned
    torch.cuda.synchronize()
    st2 = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
    with pytest.raises(L.LinkAmdError, match="tensor stride"):
        core(st2)
    st3 = la.SparseTensor(feats.cuda(), coords.cuda(), stride)      # reported once; now the general layout
    got = core(st3)
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride, agg=O.aggregate_c)
    assert rel_err(got.numpy(), ref.numpy()) < TOL
    assert any(k[0] == "link_block_index" for k in st3.kmaps)


def test_lean_plan_state_survives_a_refused_call_and_rejects_unknown_tuning():
    """a call the library refuses (more voxels than the plan's capacity: LINK_ERR_ARG, nothing launched) must
    not advance the alternating counter state -- the next frame through the plan is still right; tuning keywords the lean form does
    not have are rejected instead of ignored."""
    import link_amd as la
    from link_amd import _lib as L
    C, groups, baseop, stride, s, r = 32, 2, "cos", 2, 6, 3
    blk, params = _block(la, C, groups, baseop, 12)
    coords = torch.from_numpy(lidar_like(9000, seed=5, stride=stride))
    feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(5))
    n = coords.shape[0]
    le, ge = _plans(la, blk, n, C, baseop, groups, r, s, coords.cuda(), (s // stride) ** 3)
    a = le.run(feats.cuda(), coords.cuda()).clone()
    keep = le.buf.seg_cap
    le.buf.seg_cap = 1                                              # the library checks the segment capacity against n: LINK_ERR_ARG
    with pytest.raises(L.LinkAmdError):
        le.run(feats.cuda(), coords.cuda())
    le.buf.seg_cap = keep
    b = le.run(feats.cuda(), coords.cuda()).clone()                 # rebuilt on the counters the refused call left alone
    c_ = le.run(feats.cuda(), coords.cuda()).clone()
    le.check()
    assert torch.equal(a, b) and torch.equal(a, c_)
    ge.run(feats.cuda(), coords.cuda())
    assert le.blocks() == ge.blocks()
    from link_amd.index import coords_bounds
    with pytest.raises(L.LinkAmdError):
        la.ElkCorePlan(n, C, baseop, C // groups, r, s, coords_bounds(coords.cuda()), coords.device, layout="lean", slot_cap=27, k1_wgs=256)


def test_lean_form_batched_frames():
    """Two frames in one tensor (batch index in the coordinates, hash_cuda.cu:46): blocks never mix across the batch axis."""
    import link_amd as la
    C, groups, baseop, stride, s, r = 32, 2, "cos", 2, 6, 3
    blk, params = _block(la, C, groups, baseop, 21)
    a = torch.from_numpy(lidar_like(6000, seed=1, stride=stride))
    b = torch.from_numpy(lidar_like(9000, seed=2, stride=stride))
    b[:, 3] = 1
    coords = torch.cat([a, b])[torch.randperm(a.shape[0] + b.shape[0], generator=torch.Generator().manual_seed(0))].contiguous()
    feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(3))
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride, agg=O.aggregate_c).numpy()
    le, ge = _plans(la, blk, coords.shape[0], C, baseop, groups, r, s, coords.cuda(), (s // stride) ** 3)
    got = le.run(feats.cuda(), coords.cuda()).clone()
    ge.run(feats.cuda(), coords.cuda())
    assert le.blocks() == ge.blocks() > 0
    assert rel_err(got.cpu().numpy(), ref) < TOL
    # each frame alone gives the same rows
    for bi, part in ((0, a), (1, b)):
        m = (coords[:, 3] == bi)
        ref_b = O.elk_core_torch(feats[m], coords[m], params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                                 agg=O.aggregate_c).numpy()
        assert rel_err(got.cpu().numpy()[m.numpy()], ref_b) < TOL


@pytest.mark.parametrize("C,groups,baseop,stride,s,r", [(64, 1, "cos_x", 2, 6, 2), (64, 2, "cos", 1, 7, 3), (32, 2, "sin", 2, 6, 2),
                                                        (16, 2, "cos", 1, 7, 3), (32, 1, "cos_x", 2, 4, 3)])
def test_lean_form_without_the_scratch_matrix(C, groups, baseop, stride, s, r):
    """The form whose first launch is the slot insert alone and whose second launch runs pre_mix on the gathered rows of each
    chunk (no n x P*C matrix in between; selected by link_elk_desc_t::flags / ElkCorePlan(lean_pm=True)), on a small and a bigger frame:
    oracle, the form with the matrix (same sums in another association), bitwise repeatable rebuilt / reused, fp16 rows."""
    import link_amd as la
    from link_amd.index import coords_bounds
    blk, params = _block(la, C, groups, baseop, 3 * C + r)
    cap = min((s // stride) ** 3, 343)
    for seed, npts in ((5, 40000), (6, 5000)):
        # (stride-1 coordinates with 7^3 blocks: a coarser voxel grid, so that the tables addressed by cell stay small)
        coords = torch.from_numpy(lidar_like(npts, seed=seed, stride=stride, voxel=0.2 if (C == 16 or stride == 1) else 0.05))
        n = coords.shape[0]
        feats = torch.randn(n, C, generator=torch.Generator().manual_seed(seed))
        div = float(stride) if baseop == "cos_x" else 1.0
        ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                               agg=O.aggregate_c).numpy()
        bounds = coords_bounds(coords.cuda())
        plans = [_bind(la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, torch.device("cuda"), coord_div=div, layout="lean",
                                      slot_cap=cap, lean_pm=pm), blk, baseop) for pm in (True, False)]
        f, c = feats.cuda(), coords.cuda()
        got = plans[0].run(f, c).clone()
        other = plans[1].run(f, c).clone()
        assert rel_err(got.cpu().numpy(), ref) < TOL
        assert rel_err(got.cpu().numpy(), other.cpu().numpy()) < 2e-5
        assert torch.equal(plans[0].run(f, c), got) and torch.equal(plans[0].run(f, c, build_index=False), got)
        plans[0].check()
        h = feats.half()
        ref_h = O.elk_core_torch(h.float(), coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                                 agg=O.aggregate_c).numpy()
        assert rel_err(plans[0].run(h.cuda(), c).float().cpu().numpy(), ref_h) < 2e-3


def test_module_lean_overflow_is_reported_by_flush_and_never_dropped():
    """a frame that breaks the stride promise (coordinates off the tensor stride's lattice: up to 8 x the voxels a
    block of edge s_eff may hold at stride 2) overflows a slot list on the module's sync-free lean path.  Its rows come back (dropped
    voxels as zeros), and `flush_lean_verdicts()` raises for it at once -- the verdict is not left to an unrelated later frame."""
    import link_amd as la
    from link_amd import _lib as L
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    C, n = 32, 6000
    blk = la.ELKBlock(C, C, groups=1, baseop="cos_x", variant="encoder").to(dev).eval()
    g = torch.Generator().manual_seed(3)
    lin = torch.randperm(24 ** 3, generator=g)[:n]                      # EVERY integer site of a 24^3 box ...
    coords = torch.stack([lin % 24, (lin // 24) % 24, lin // 576, torch.zeros_like(lin)], 1).int().to(dev)
    st = la.SparseTensor(torch.randn(n, C, generator=g).to(dev), coords, 2)      # ... declared as a stride-2 tensor
    with torch.no_grad():
        out = blk._core_lean(st, 6, 2, blk.pos_weight[0].weight, blk.alpha, C, 2.0)
    if out is None:
        pytest.skip("the lean form did not take this frame")
    assert out.shape == (n, C) and bool(torch.isfinite(out).all())
    with pytest.raises(L.LinkAmdError, match="earlier frame held more than"):
        blk.flush_lean_verdicts()
    blk.flush_lean_verdicts()                                            # reported once; nothing pending afterwards
    assert (6, 2) in blk._lean_distrust
