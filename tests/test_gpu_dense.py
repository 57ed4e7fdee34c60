"""Dense-cell layout of R_core (include/link_amd.h section E, link_amd/csrc/dense.hip) against the oracle
restatement of linkunet.py:124-185 and against the general layout (sections B + C) on the same inputs."""
import numpy as np
import pytest
import torch

from helpers import rel_err, s_uniform

pytestmark = pytest.mark.gpu


def _plan(la, blk, n, C, groups, baseop, r, s, bounds, layout, coord_div=1.0, **tuning):
    plan = la.ElkCorePlan(n, C, baseop, C // groups, r, s, bounds, torch.device("cuda"), coord_div=coord_div,
                          layout=layout, **(tuning if layout != "general" else {}))
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
              blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
    return plan


def _oracle(blk, feats, coords, s, r, baseop, groups):
    from oracle import link_oracle as O
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    return O.elk_core_torch(feats.cpu(), coords.cpu(), params, s, r, baseop, groups, agg=O.aggregate_c).numpy()


@pytest.mark.parametrize("C,groups,baseop,s,r,grid,n", [
    (64, 2, "cos", 7, 3, 80, 9000), (32, 2, "sin", 3, 2, 40, 6000), (16, 2, "cos", 7, 3, 256, 10000),
    (128, 2, "cos", 5, 3, 60, 7000), (64, 1, "cos_x", 3, 2, 36, 5000), (64, 1, "cos_x", 3, 3, 30, 4000),
    (32, 1, "cos", 4, 3, 50, 8000), (16, 1, "cos_x", 2, 2, 24, 3000), (128, 1, "cos_x", 4, 2, 40, 3000)])
@pytest.mark.parametrize("k1_form,k2_form", [(0, 0), (1, 0), (2, 0), (0, 4), (1, 4), (2, 4), (0, 8)])
def test_dense_vs_oracle_and_general(C, groups, baseop, s, r, grid, n, k1_form, k2_form):
    """k1_form: 0 = cell-range form of the fused pre_mix kernel, 1 = tile form (round 3), 2 = matrix-core sums form (round 4;
    C = 32 / 64, the other widths take the cell-range form); k2_form (C = 64): 0 =
    by measurement (quad consumers where the rows allow), 4 = own-cell form, 8 = pair consumers (the round-2 kernel)."""
    import link_amd as la
    torch.manual_seed(5)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    coords = s_uniform(n, grid=grid, seed=C + r).cuda()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(3)).cuda()
    bounds = ((0, 0, 0, 0), (grid - 1, grid - 1, grid - 1, 0))
    dense = _plan(la, blk, n, C, groups, baseop, r, s, bounds, "dense", k1_form=k1_form, k2_form=k2_form)
    general = _plan(la, blk, n, C, groups, baseop, r, s, bounds, "general")
    assert dense.dense and not general.dense
    od = dense.run(feats, coords).clone()
    og = general.run(feats, coords).clone()
    assert dense.blocks() == general.blocks() > 0
    ref = _oracle(blk, feats, coords, s, r, baseop, groups)
    assert rel_err(od.cpu().numpy(), ref) < 1e-4            # the parity gate (fp32, BASELINE.json)
    assert rel_err(od.cpu().numpy(), og.cpu().numpy()) < 1e-4     # both sit on the oracle within the gate
    # warm (index reused) == cold, bitwise; and a second cold run is bitwise identical (no fp atomics,
    # slot order removed by the id sort)
    warm = dense.run(feats, coords, build_index=False).clone()
    assert torch.equal(od, warm)
    for _ in range(3):
        assert torch.equal(od, dense.run(feats, coords))
    assert int(dense.cnt.abs().sum().item()) == 0           # the counters cleaned themselves


@pytest.mark.parametrize("k1_form,k2_form", [(0, 0), (1, 0), (2, 0), (0, 4), (0, 8)])
def test_dense_large_cells_negative_coords_batches(k1_form, k2_form):
    """Cells with many voxels (~80 per cell: the counting-rank pass of the tile form, the insertion path of the cell-range
    form), negative coordinates, two batch items."""
    import link_amd as la
    torch.manual_seed(9)
    C, groups, baseop, s, r = 64, 2, "cos", 7, 3
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    a = s_uniform(5000, grid=24, seed=1)                     # ~ 80 voxels per 7^3 block
    b = s_uniform(3000, grid=24, seed=2, batch=1)
    coords = torch.cat([a, b])
    coords[:, :3] -= 11                                      # blocks straddle zero: floor division
    perm = torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(0))
    coords = coords[perm].contiguous().cuda()
    n = coords.shape[0]
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(4)).cuda()
    bounds = ((-11, -11, -11, 0), (12, 12, 12, 1))
    dense = _plan(la, blk, n, C, groups, baseop, r, s, bounds, "dense", k1_form=k1_form, k2_form=k2_form)
    od = dense.run(feats, coords).clone()
    assert dense.blocks() > 0
    assert int(dense.cell_n.max().item()) > 8
    ref = _oracle(blk, feats, coords, s, r, baseop, groups)
    assert rel_err(od.cpu().numpy(), ref) < 1e-4
    for _ in range(3):
        assert torch.equal(od, dense.run(feats, coords))


def test_dense_status_word_and_capacity_reuse():
    import link_amd as la
    torch.manual_seed(7)
    C, n = 64, 20000
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    coords = s_uniform(n, grid=128, seed=11).cuda()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(4)).cuda()
    bounds = ((0, 0, 0, 0), (127, 127, 127, 0))
    plan = _plan(la, blk, n, C, 2, "cos", 3, 7, bounds, "auto")
    assert plan.dense                                        # 21^3 padded cells <= 4 n
    out = plan.run(feats, coords).clone()
    ref = _oracle(blk, feats, coords, 7, 3, "cos", 2)
    assert rel_err(out.cpu().numpy(), ref) < 1e-4
    n2 = 5000                                                # a smaller frame through the same plan
    out2 = plan.run(feats[:n2].contiguous(), coords[:n2].contiguous()).clone()
    ref2 = _oracle(blk, feats[:n2], coords[:n2], 7, 3, "cos", 2)
    assert rel_err(out2.cpu().numpy(), ref2) < 1e-4
    assert plan.blocks() > 0
    bad = coords.clone()
    bad[0, 0] = 500                                          # outside the plan's bounds -> status bit 0
    plan.run(feats, bad)
    with pytest.raises(la._lib.LinkAmdError):
        plan.blocks()
    plan.run(feats, coords)                                  # the status word is per step
    assert plan.blocks() > 0
    dup = coords.clone()                                     # 400 copies of one voxel: > 7^3 slots -> bit 1
    dup[:400] = dup[0]
    plan.run(feats, dup)
    with pytest.raises(la._lib.LinkAmdError):
        plan.blocks()
    assert torch.equal(out, plan.run(feats, coords))


@pytest.mark.parametrize("tuning", [{}, {"k1_form": 1}, {"k1_form": 2}, {"k1_form": 2, "k1_wgs": 256, "k2_zsplit": 2}, {"k1_form": 2, "k1_wgs": 1024},
                                    {"k1_wgs": 256, "k2_zsplit": 2}, {"k1_wgs": 256, "k2_zsplit": 2, "k1_form": 1, "k1_lds_pad": 2048},
                                    {"k1_wgs": 1024}, {"k2_zsplit": 1}, {"k2_form": 1}, {"k2_form": 2}, {"k2_form": 4}, {"k2_form": 8}, {"k2_form": 8, "k2_zsplit": 2},
                                    {"k2_form": 4, "k2_zsplit": 2, "k1_wgs": 256}])
def test_dense_cfg2_full_size(tuning):
    """BASELINE.json configs[1] at full size: dense-cell vs general layout vs oracle; M = 43 334 -- under every launch
    geometry bench.py times (one frame: defaults; frames in flight: 256 workgroups + 2 z-segments (+ LDS pad on the
    cell-range form)) and both forms of the fused pre_mix kernel."""
    import link_amd as la
    torch.manual_seed(2)
    N, C = 100_000, 64
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    coords = s_uniform(N, seed=0).cuda()
    feats = torch.randn(N, C, generator=torch.Generator().manual_seed(1)).cuda()
    bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
    dense = _plan(la, blk, N, C, 2, "cos", 3, 7, bounds, "auto", **tuning)
    general = _plan(la, blk, N, C, 2, "cos", 3, 7, bounds, "general")
    assert dense.dense
    od, og = dense.run(feats, coords).clone(), general.run(feats, coords).clone()
    assert dense.blocks() == general.blocks() == 43334
    assert rel_err(od.cpu().numpy(), og.cpu().numpy()) < 1e-4     # both sit on the oracle within the gate
    ref = _oracle(blk, feats, coords, 7, 3, "cos", 2)
    assert rel_err(od.cpu().numpy(), ref) < 1e-4
    for _ in range(5):
        assert torch.equal(od, dense.run(feats, coords))


@pytest.mark.parametrize("dtype,tol_round,tol_oracle", [(torch.float16, 1.2e-3, 6e-3), (torch.bfloat16, 9e-3, 5e-2)])
@pytest.mark.parametrize("C,groups,baseop,s,r,grid,n", [(64, 2, "cos", 7, 3, 80, 9000), (32, 1, "cos_x", 3, 2, 40, 6000),
                                                       (64, 1, "cos_x", 3, 2, 36, 5000)])
def test_dense_half_rows(dtype, tol_round, tol_oracle, C, groups, baseop, s, r, grid, n):
    """fp16 / bf16 feature rows at the kernel boundary (SURVEY.md section 8b "AMP": the reference wraps its
    voxelize / devoxelize ops in custom_fwd(cast_inputs=torch.half), fp32 accumulation).  Everything inside is
    fp32, so (a) against the fp32 path fed the SAME rounded inputs only the output rounding remains (half an ulp
    of the row maximum: 2^-11 fp16, 2^-8 bf16), and (b) against the fp32 oracle on the unrounded inputs the
    stated looser gate holds."""
    import link_amd as la
    torch.manual_seed(5)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    coords = s_uniform(n, grid=grid, seed=C + r).cuda()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(3)).cuda()
    bounds = ((0, 0, 0, 0), (grid - 1, grid - 1, grid - 1, 0))
    plan = _plan(la, blk, n, C, groups, baseop, r, s, bounds, "dense")
    fh = feats.to(dtype)
    oh = plan.run(fh, coords).clone()
    assert oh.dtype == dtype and oh.shape == (n, C)
    o32 = plan.run(fh.float(), coords).clone()
    assert rel_err(oh.float().cpu().numpy(), o32.cpu().numpy()) < tol_round
    ref = _oracle(blk, feats, coords, s, r, baseop, groups)
    assert rel_err(oh.float().cpu().numpy(), ref) < tol_oracle
    assert torch.equal(oh, plan.run(fh, coords))                 # reproducible
    # the module path keeps the row type (what a caller under torch.autocast hands over)
    st = la.SparseTensor(fh, coords, 1)
    with torch.no_grad():
        core = blk._core(st, s, r, blk.pos_weight[0].weight, blk.alpha if baseop == "cos_x" else None, C // groups, 1.0)
    assert core.dtype == dtype and rel_err(core.float().cpu().numpy(), o32.cpu().numpy()) < tol_round


@pytest.mark.parametrize("C,what", [(64, "rows"), (32, "rows"), (64, "weights"), (16, "weights")])
def test_dense_premix_values_outside_the_fp16_split_range(C, what):
    """The fused pre_mix kernel multiplies as an fp16 hi + lo split; feature rows or pre_mix weights with
    |value| >= 2^15 must take its fp32-instruction path and still agree with the general layout (fp32 MFMA) and the
    oracle.  Rows: a third of the voxels scaled by 1e5 (LayerNorm removes the scale from the result's size); weights: one
    entry of 5e4."""
    import link_amd as la
    torch.manual_seed(11)
    groups, baseop, s, r, grid, n = 2, "cos", 7, 3, 64, 6000
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    coords = s_uniform(n, grid=grid, seed=2).cuda()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(4)).cuda()
    if what == "rows":
        feats[::3] *= 1.0e5
    else:
        with torch.no_grad():
            blk.pre_mix[0].weight[3, 5] = 5.0e4
    bounds = ((0, 0, 0, 0), (grid - 1, grid - 1, grid - 1, 0))
    dense = _plan(la, blk, n, C, groups, baseop, r, s, bounds, "dense")
    general = _plan(la, blk, n, C, groups, baseop, r, s, bounds, "general")
    od, og = dense.run(feats, coords).clone(), general.run(feats, coords).clone()
    assert torch.isfinite(od).all()
    assert rel_err(od.cpu().numpy(), og.cpu().numpy()) < 2e-5
    assert rel_err(od.cpu().numpy(), _oracle(blk, feats, coords, s, r, baseop, groups)) < 1e-4
    assert torch.equal(od, dense.run(feats, coords))


def test_three_plans_in_flight_on_three_streams():
    """What bench.py times: three frames in flight on separate HIP streams, each with its own plan (arena + per-plan launch
    geometry, frames_in_flight = 3); every output equals the same plan run alone with the single-frame geometry within
    rounding of the summation tree (bitwise where the geometry does not change the order of any sum) and sits on the
    oracle."""
    import link_amd as la
    torch.manual_seed(2)
    N, C = 100_000, 64
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
    frames, plans, streams = [], [], []
    for k in range(3):
        frames.append((torch.randn(N, C, generator=torch.Generator().manual_seed(1 + k)).cuda(), s_uniform(N, seed=k).cuda()))
        pl = la.ElkCorePlan(N, C, "cos", 32, 3, 7, bounds, torch.device("cuda"), frames_in_flight=3)
        pl.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
                blk.norm.weight, blk.norm.bias)
        plans.append(pl)
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    for i in range(12):
        j = i % 3
        with torch.cuda.stream(streams[j]):
            plans[j].run(*frames[j])
    torch.cuda.synchronize()
    outs = [pl.out[:N].clone() for pl in plans]
    for pl in plans:
        pl.check()
    ref0 = _oracle(blk, frames[0][0], frames[0][1], 7, 3, "cos", 2)
    assert rel_err(outs[0].cpu().numpy(), ref0) < 1e-4
    for j in range(3):
        alone = la.ElkCorePlan(N, C, "cos", 32, 3, 7, bounds, torch.device("cuda"))
        alone.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
                   blk.norm.weight, blk.norm.bias)
        o1 = alone.run(*frames[j])
        assert rel_err(outs[j].cpu().numpy(), o1.cpu().numpy()) < 2e-6


@pytest.mark.parametrize("C,groups,baseop,s,r", [(64, 2, "cos", 7, 3), (32, 1, "cos_x", 3, 2), (128, 2, "cos", 5, 3)])
def test_module_first_visits_probe_once_and_guess_the_grid(C, groups, baseop, s, r):
    """A stream of new coordinate sets through one module (elk.py:_core_dense): the first visit's occupancy probe is the
    step's own slot insert (link_dc_index_probe, then build_index = 2), and from the second frame on bounding box and probe
    share one round trip on the guess that the frame lands on the last plan's grid -- right for frames 1-2 (same extents),
    wrong for frame 3 (shifted and smaller: that plan forgets the frame) and frame 4 (clumped: the general layout runs).
    Every frame against a fresh module on the same weights, and the kept plan's counters are clean afterwards."""
    import link_amd as la
    torch.manual_seed(9)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    grid, n = 72, 7000
    frames = []
    for k in range(3):
        c = s_uniform(n, grid=grid, seed=20 + k)
        c[0, :3], c[1, :3] = 0, grid - 1                      # same extents, hence the same block-aligned grid
        frames.append(c)
    shifted = s_uniform(6000, grid=grid // 2, seed=31)        # same row capacity as the frames before: the guess is made, and wrong
    shifted[:, :3] += 11
    clumped = s_uniform(n, grid=8, seed=32)                   # ~ 14 voxels per cell position: far above DENSE_MAX_MEAN
    clumped = torch.unique(clumped, dim=0)
    frames += [shifted, clumped, frames[0].clone()]
    for k, c in enumerate(frames):
        c = c.cuda()
        feats = torch.randn(c.shape[0], C, generator=torch.Generator().manual_seed(40 + k)).cuda()
        with torch.no_grad():
            got = blk._core(la.SparseTensor(feats, c, 1), s, r, blk.pos_weight[0].weight, blk.alpha if baseop == "cos_x" else None,
                            C // groups, 1.0).float()
            fresh = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
            fresh.load_state_dict(blk.state_dict())
            fresh.dense_layout = False                       # the general layout: no probe, no shared plan
            ref = fresh._core(la.SparseTensor(feats, c.clone(), 1), s, r, fresh.pos_weight[0].weight,
                              fresh.alpha if baseop == "cos_x" else None, C // groups, 1.0).float()
        assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 2e-5, k
    for plan in blk._dc_plans.values():
        if plan is not None and plan.__dict__.get("_indexed") is None:
            assert int(plan.cnt.sum()) == 0 and int(plan.hdr.abs().sum()) == 0


def test_a_frame_result_does_not_depend_on_the_frame_before_it():
    """The same frame through a fresh plan and through a plan that has just run a bigger, different frame: bitwise equal.
    (Round 4: quads of the gather kernel that hold no voxel read the record slot of an empty cell -- whatever an earlier frame
    left there -- and used to take part in the wave's choice between the two sincos paths.)"""
    import link_amd as la
    torch.manual_seed(7)
    blk = la.ELKBlock(64, 64, groups=2, baseop="sin").cuda().eval()
    par = (blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight,
           blk.norm.bias)
    bounds = ((0, 0, 0, 0), (63, 63, 63, 0))
    a = (torch.randn(20000, 64, generator=torch.Generator().manual_seed(40)).cuda(), s_uniform(20000, grid=64, seed=50).cuda())
    b = (torch.randn(7000, 64, generator=torch.Generator().manual_seed(41)).cuda(), s_uniform(7000, grid=64, seed=51).cuda())
    junk = [torch.randint(-2 ** 31, 2 ** 31 - 1, (32 << 20,), dtype=torch.int32, device="cuda") for _ in range(4)]
    del junk                                             # what torch.empty hands the plans next is not zero
    for form in (0, 2):
        mk = lambda: la.ElkCorePlan(20000, 64, "sin", 32, 2, 5, bounds, "cuda", layout="dense", k1_form=form).bind(*par)
        p1, p2 = mk(), mk()
        p1.run(*a)
        after, alone = p1.run(*b).clone(), p2.run(*b).clone()
        assert torch.equal(after, alone)
