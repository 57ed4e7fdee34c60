"""include/link_amd.h section G -- the one-call block driver (csrc/block.hip, elk.py:_block_native): ELKBlock.forward on a
coordinate set nothing is known about, as ONE host call.  Everything it launches is tested stage by stage elsewhere; here:
the call gives the per-stage path's rows bit for bit, registers what it built where later blocks on the same coordinates look
for it, leaves nothing behind when the frame misses the plan it was tried on, and agrees with the oracle."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from helpers import rel_err, s_uniform  # noqa: E402

pytestmark = pytest.mark.gpu


def _blk(C, groups, baseop, seed=3):
    import link_amd as la
    torch.manual_seed(seed)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    with torch.no_grad():                                    # LayerNorm parameters away from (1, 0): the finish phase must use them
        for ln in (blk.pre_mix[1], blk.norm, blk.norm_local):
            ln.weight.uniform_(0.5, 1.5)
            ln.bias.uniform_(-0.3, 0.3)
    return blk


def _frames(n, grid, k):
    out = []
    for i in range(k):
        c = s_uniform(n, grid=grid, seed=60 + i)
        c[0, :3], c[1, :3] = 0, grid - 1                      # same extents: the frames share a block-aligned grid
        out.append(c.cuda())
    return out


@pytest.mark.parametrize("C,groups,baseop,s,r", [(64, 2, "cos", 7, 3), (64, 1, "sin", 5, 2), (64, 1, "cos_x", 6, 2), (32, 2, "cos", 7, 3)])
def test_block_driver_rows_equal_the_per_stage_path(C, groups, baseop, s, r):
    import link_amd as la
    from link_amd import elk
    blk = _blk(C, groups, baseop)
    ref = _blk(C, groups, baseop)
    ref.load_state_dict(blk.state_dict())
    grid, n = 96, 12000                                      # 14^3 .. 20^3 blocks: 1.5 - 4.4 voxels per cell, a dense-cell frame
    frames = _frames(n, grid, 4)
    feats = [torch.randn(n, C, generator=torch.Generator().manual_seed(80 + i)).cuda() for i in range(4)]
    before = dict(elk.BLOCK_DRIVER_CALLS)
    with torch.no_grad():
        for i, (f, c) in enumerate(zip(feats, frames)):
            st = la.SparseTensor(f, c, 1)
            got = blk(st, s, r).F
            elk.BLOCK_DRIVER = False
            try:
                want = ref(la.SparseTensor(f, c.clone(), 1), s, r).F
            finally:
                elk.BLOCK_DRIVER = True
            assert torch.equal(got, want), (i, float((got - want).abs().max()))
            # what the call built is where the per-stage path looks: a second block on the same tensor runs on warm maps
            again = blk(la.SparseTensor(f, c, 1), s, r).F if i == 0 else None
            st2 = la.SparseTensor(f, c, 1)
            st2.cmaps, st2.kmaps = st.cmaps, st.kmaps
            warm = blk(st2, s, r).F
            assert rel_err(warm.cpu().numpy(), want.cpu().numpy()) < 5e-6, i        # (the GEMM of a plan whose counts have arrived covers fewer granules)
            if again is not None:
                assert torch.equal(again, want)
    done = elk.BLOCK_DRIVER_CALLS["done"] - before["done"]
    resident = elk._resident_form(C, C, 27, False, "auto")
    assert done == (0 if resident else 4), (done, elk.BLOCK_DRIVER_CALLS)     # the first pass of frame 0 goes stage by stage: it makes the plan
    torch.cuda.synchronize()
    for plan in blk._dc_plans.values():
        if plan is not None and plan.__dict__.get("_indexed") is None:
            assert int(plan.cnt.sum()) == 0 and int(plan.hdr.abs().sum()) == 0


def test_block_driver_miss_leaves_nothing_behind_and_the_per_stage_path_takes_over():
    import link_amd as la
    from link_amd import elk
    C, s, r = 64, 7, 3
    blk = _blk(C, 2, "cos")
    ref = _blk(C, 2, "cos")
    ref.load_state_dict(blk.state_dict())
    ref.dense_layout = False
    n, grid = 12000, 96
    a, b = _frames(n, grid, 2)
    shifted = s_uniform(n - 2000, grid=grid, seed=71).cuda()          # same row capacity, extents beyond the plan's grid
    shifted[:, 0] += 40
    clumped = torch.unique(s_uniform(1700, grid=12, seed=72), dim=0)          # 8 blocks of 7^3 hold them: ~ 200 voxels per cell
    clumped = torch.cat([clumped, s_uniform(n - clumped.shape[0] - 300, grid=grid, seed=73) + torch.tensor([20, 20, 20, 0], dtype=torch.int32)])
    clumped = torch.unique(clumped, dim=0).cuda()
    clumped[0, :3], clumped[1, :3] = 0, grid - 1
    f = lambda k, m: torch.randn(m, C, generator=torch.Generator().manual_seed(90 + k)).cuda()
    before = dict(elk.BLOCK_DRIVER_CALLS)
    with torch.no_grad():
        for k, c in enumerate([a, b, shifted, clumped, a.clone()]):
            x = f(k, c.shape[0])
            got = blk(la.SparseTensor(x, c, 1), s, r).F
            want = ref(la.SparseTensor(x, c.clone(), 1), s, r).F
            assert rel_err(got.cpu().numpy(), want.cpu().numpy()) < 2e-5, k
    d = {k: elk.BLOCK_DRIVER_CALLS[k] - before[k] for k in before}
    assert d["done"] >= 1 and d["miss"] >= 1, d
    torch.cuda.synchronize()
    for plan in blk._dc_plans.values():
        if plan is not None and plan.__dict__.get("_indexed") is None:
            assert int(plan.cnt.sum()) == 0 and int(plan.hdr.abs().sum()) == 0


def test_block_driver_vs_oracle():
    """relu(R_core + LayerNorm(local_mix)) of linkunet.py:124-185 from the oracle's restatements, against the one-call path."""
    import link_amd as la
    from link_amd import elk
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import link_oracle as O
    C, s, r, groups = 64, 7, 3, 2
    blk = _blk(C, groups, "cos")
    par = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    n, grid = 9000, 96
    frames = _frames(n, grid, 2)
    before = elk.BLOCK_DRIVER_CALLS["done"]
    with torch.no_grad():
        for i, c in enumerate(frames):
            x = torch.randn(n, C, generator=torch.Generator().manual_seed(100 + i))
            got = blk(la.SparseTensor(x.cuda(), c, 1), s, r).F.cpu()
            core = O.elk_core_torch(x, c.cpu(), par, s, r, baseop="cos", groups=groups)
            local = O.subm_conv_torch(x, c.cpu(), par["local_mix.0.kernel"])
            want = torch.relu(core + torch.nn.functional.layer_norm(local, (C,), par["norm_local.weight"], par["norm_local.bias"], 1e-6))
            assert rel_err(got.numpy(), want.numpy()) < 1e-4, i
    assert elk.BLOCK_DRIVER_CALLS["done"] - before == 1


@pytest.mark.parametrize("s,step,grid,n", [(7, 1, 64, 9000), (5, 1, 40, 6000), (6, 2, 48, 3000), (3, 1, 24, 5000), (3, 2, 30, 2500), (4, 4, 48, 1200)])
def test_dc_neighbor_map_equals_the_cell_table_map(s, step, grid, n):
    """link_dc_neighbor_map (the 27-neighbour table read off a frame's freshly inserted slot lists) against
    link_cell_table_build + link_neighbor_map, the table every convolution test pins on the reference's kernel maps: bit-exact,
    negative coordinates, a second batch item, a tensor stride."""
    import ctypes
    import link_amd as la
    from link_amd import _lib as L
    from link_amd.index import foreign_neighbor_map
    c = s_uniform(n, grid=grid, seed=5 + s)
    c[:, :3] = (c[:, :3] // step) * step                     # coordinates of a stride-`step` tensor
    c = torch.unique(c, dim=0)
    c[:, :3] -= 9                                            # negative coordinates: floor division
    c2 = c.clone(); c2[:, 3] = 1
    c = torch.cat([c, c2[: c.shape[0] // 2]])[torch.randperm(c.shape[0] + c.shape[0] // 2, generator=torch.Generator().manual_seed(1))]
    c = c.contiguous().cuda()
    m = c.shape[0]
    lo = [int(v) for v in c.min(0).values.tolist()]
    hi = [int(v) for v in c.max(0).values.tolist()]
    plan = la.ElkCorePlan(m, 64, "cos", 32, 3, s, (tuple(lo), tuple(hi)), c.device, layout="dense")
    stats = torch.zeros(256, dtype=torch.int32, device=c.device)
    plan._probe(c, m, stats)
    nbr = torch.empty((m, 27), dtype=torch.int32, device=c.device)
    L.check(L.lib().link_dc_neighbor_map(c.data_ptr(), m, ctypes.byref(plan.dcg), plan.cnt.data_ptr(), plan.slots.data_ptr(), step,
                                         nbr.data_ptr(), L.current_stream_handle()), "link_dc_neighbor_map")
    want = foreign_neighbor_map(c, 3, step=step)
    assert torch.equal(nbr, want)
    assert int((nbr[:, 13] != torch.arange(m, device=c.device, dtype=torch.int32)).sum()) == 0
    plan._unprobe()


def test_block_driver_across_streams_release_and_workspace_errors():
    """The driver's context keeps per-call scratch that the previous call lays out in stream order: calls that alternate between
    two torch streams (the context then re-initialises its scratch behind a synchronisation) give the same rows; a pair-plan arena
    or contribution buffer smaller than link_pair_plan_arena says is refused with LINK_ERR_WORKSPACE before anything is launched;
    release_block_driver() frees contexts and scratch and the next call builds them again."""
    import ctypes
    import link_amd as la
    from link_amd import elk
    from link_amd import _lib as L
    C, s, r = 64, 7, 3
    blk = _blk(C, 2, "cos")
    n, grid = 12000, 96
    frames = _frames(n, grid, 3)
    x = torch.randn(n, C, generator=torch.Generator().manual_seed(7)).cuda()
    with torch.no_grad():
        want = [blk(la.SparseTensor(x, c.clone(), 1), s, r).F.clone() for c in frames]       # frame 0 per stage, 1-2 through the driver
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        torch.cuda.synchronize()
        before = elk.BLOCK_DRIVER_CALLS["done"]
        for rep in range(3):
            for k, c in enumerate(frames):
                with torch.cuda.stream(streams[(rep + k) % 2]):
                    got = blk(la.SparseTensor(x, c.clone(), 1), s, r).F
                torch.cuda.synchronize()
                assert torch.equal(got, want[k]), (rep, k)
        assert elk.BLOCK_DRIVER_CALLS["done"] - before == 9
        elk.release_block_driver()
        assert not elk._BLOCK_CTX and not elk._BLOCK_CONTRIB
        got = blk(la.SparseTensor(x, frames[1].clone(), 1), s, r).F
        assert torch.equal(got, want[1]) and len(elk._BLOCK_CTX) == 1
    # a call whose arena is too small launches nothing
    plan = blk._dc_last[1]
    a = L.LinkBlockArgs()
    ctypes.memmove(ctypes.byref(a), ctypes.byref(plan._blk_args), ctypes.sizeof(a))
    a.pair_arena_words = 16
    ctx = elk._block_ctx(x.device)
    assert L.lib().link_elk_block_forward(ctx, ctypes.byref(a), L.current_stream_handle()) == L.LINK_ERR_WORKSPACE
    ctypes.memmove(ctypes.byref(a), ctypes.byref(plan._blk_args), ctypes.sizeof(a))
    a.contrib_rows = 128
    assert L.lib().link_elk_block_forward(ctx, ctypes.byref(a), L.current_stream_handle()) == L.LINK_ERR_WORKSPACE
    torch.cuda.synchronize()
    assert int(plan.cnt.sum()) == 0 or plan.__dict__.get("_indexed") is not None
