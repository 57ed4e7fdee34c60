"""The alternative kernel paths of the C ABI must agree: split gather (link_block_gather +
link_voxel_demod_ln) vs fused group gather vs lane=channel generic gather, and group vs generic
modulate kernels -- selected per call through link_elk_desc_t::flags (no process-global switches) -- on the same inputs."""
import ctypes

import numpy as np
import pytest
import torch

from helpers import rel_err, s_uniform

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("C,groups,baseop,s,r", [(64, 2, "cos", 7, 3), (32, 2, "sin", 3, 2), (16, 2, "cos", 7, 3),
                                                 (128, 2, "cos", 5, 3), (48, 1, "cos_x", 4, 2), (64, 1, "cos_x", 3, 3)])
def test_kernel_paths_agree(C, groups, baseop, s, r):
    import link_amd as la
    from link_amd import _lib as L
    lib = L.lib()
    torch.manual_seed(5)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    n = 9000
    coords = s_uniform(n, grid=80, seed=C + r).cuda()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(3)).cuda()
    lo, hi = (0, 0, 0, 0), (79, 79, 79, 0)
    # the flags select general-layout kernels; tiles=False: the four-kernel plan (with the A matrix of the split gather)
    plan = la.ElkCorePlan(n, C, baseop, C // groups, r, s, (lo, hi), feats.device, layout="general", tiles=False)
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
              blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
    outs = {}
    for name, flags in {"split+group+pair": 0, "fused-group": L.ELK_FUSED_GATHER, "no-pair": L.ELK_NO_PAIR,
                        "column-walking gather": L.ELK_NO_DENSE_GRID,
                        "generic": L.ELK_LANE_CHANNEL | L.ELK_NO_PAIR | L.ELK_FUSED_GATHER}.items():
        plan.desc.flags = flags
        outs[name] = plan.run(feats, coords).clone()
        assert plan.blocks() > 0
    plan.desc.flags = 0
    if C in (16, 32, 64, 128):          # the two-launch tile form of the same layout
        tplan = la.ElkCorePlan(n, C, baseop, C // groups, r, s, (lo, hi), feats.device, layout="general")
        assert tplan.tiles and tplan.desc.flags & L.ELK_TILES
        tplan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
                   blk.alpha if baseop == "cos_x" else None, blk.norm.weight, blk.norm.bias)
        outs["tiles"] = tplan.run(feats, coords).clone()
    # every path within the parity gate of the ORACLE (1e-4 rel), and within 5e-5 of each other
    from oracle import link_oracle as O
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    oracle = O.elk_core_torch(feats.cpu(), coords.cpu(), params, s, r, baseop, groups, agg=O.aggregate_c).numpy()
    ref = outs["generic"].cpu().numpy()
    for name, o in outs.items():
        assert rel_err(o.cpu().numpy(), oracle) < 1e-4, name
        assert rel_err(o.cpu().numpy(), ref) < 5e-5, name
    # (paths may differ in the last bit: different fma contraction / summation order per code path)


def test_plan_matches_module_and_warm_index():
    import link_amd as la
    torch.manual_seed(7)
    C, n = 64, 20000
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    coords = s_uniform(n, grid=128, seed=11).cuda()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(4)).cuda()
    plan = la.ElkCorePlan(n, C, "cos", 32, 3, 7, ((0, 0, 0, 0), (127, 127, 127, 0)), feats.device)
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
              blk.norm.weight, blk.norm.bias)
    cold = plan.run(feats, coords, build_index=True).clone()
    warm = plan.run(feats, coords, build_index=False).clone()
    assert torch.equal(cold, warm)
    st = la.SparseTensor(feats, coords, 1)
    with torch.no_grad():
        ref = blk._core(st, 7, 3, blk.pos_weight[0].weight, None, 32, 1.0)
    assert rel_err(cold.cpu().numpy(), ref.cpu().numpy()) < 2e-6
    # a smaller frame through the same plan (capacity reuse), then voxels outside the plan's bounds
    n2 = 5000
    out2 = plan.run(feats[:n2].contiguous(), coords[:n2].contiguous())
    st2 = la.SparseTensor(feats[:n2].contiguous(), coords[:n2].contiguous(), 1)
    with torch.no_grad():
        ref2 = blk._core(st2, 7, 3, blk.pos_weight[0].weight, None, 32, 1.0)
    assert rel_err(out2.cpu().numpy(), ref2.cpu().numpy()) < 2e-6
    bad = coords.clone()
    bad[0, 0] = 500
    plan.run(feats, bad)
    with pytest.raises(la._lib.LinkAmdError):
        plan.blocks()


def test_plan_is_hipgraph_capturable():
    """The one-call step allocates nothing and never syncs, so it can be captured in a hipGraph (how a
    serving loop would replay it) and the replay reproduces the eager result bit for bit."""
    import link_amd as la
    torch.manual_seed(8)
    C, n = 64, 15000
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    coords = s_uniform(n, grid=100, seed=12).cuda()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(6)).cuda()
    plan = la.ElkCorePlan(n, C, "cos", 32, 3, 7, ((0, 0, 0, 0), (99, 99, 99, 0)), feats.device)
    plan.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
              blk.norm.weight, blk.norm.bias)
    eager = plan.run(feats, coords).clone()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        plan.run(feats, coords)
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            plan.run(feats, coords)
    torch.cuda.synchronize()
    plan.out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.out[:n], eager)
    feats.mul_(2.0)                      # same buffers, new contents: the graph recomputes from them
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(plan.out[:n], plan.run(feats, coords))


def test_alternating_frames_through_one_plan():
    """Stale-cache check for the write-through stores: two DIFFERENT frames alternate through the same
    plan buffers (so every intermediate buffer changes content between launches); each result must equal
    the one computed through a separate, fresh plan."""
    import link_amd as la
    torch.manual_seed(9)
    C, n = 64, 30000
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    bounds = ((0, 0, 0, 0), (127, 127, 127, 0))

    def mkplan():
        p = la.ElkCorePlan(n, C, "cos", 32, 3, 7, bounds, torch.device("cuda"))
        return p.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight,
                      None, blk.norm.weight, blk.norm.bias)
    frames = []
    for k in range(2):
        coords = s_uniform(n - 5000 * k, grid=128, seed=30 + k).cuda()
        feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(40 + k)).cuda()
        want = mkplan().run(feats, coords).clone()
        frames.append((feats, coords, want))
    shared = mkplan()
    for it in range(40):
        feats, coords, want = frames[it % 2]
        got = shared.run(feats, coords)
        assert torch.equal(got, want), it
