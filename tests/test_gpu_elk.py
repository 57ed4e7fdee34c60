"""GPU parity of the LinK block (include/link_amd.h section C + link_amd/elk.py) vs the fixtures
generated from the reference's ELKBlock and vs the oracle.  fp32 tolerance: 1e-4 rel (BASELINE.json)."""
import numpy as np
import pytest
import torch

from helpers import golden_files, load_golden, rel_err, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def make_block(g, variant=None):
    import link_amd as la
    m = g["meta"]
    blk = la.ELKBlock(m["C"], m["C"], groups=m["groups"], baseop=m["baseop"],
                      variant=variant or m["variant"]).cuda().eval()
    sd = {k[4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd__")}
    missing, unexpected = blk.load_state_dict(sd, strict=True)      # reference checkpoint loads as is
    return blk


@pytest.mark.parametrize("name", golden_files("g_block_*.npz"))
def test_block_forward_vs_reference(name):
    import link_amd as la
    g = load_golden(name)
    m = g["meta"]
    blk = make_block(g)
    st = la.SparseTensor(torch.from_numpy(g["feats"]).cuda(), torch.from_numpy(g["coords"]).cuda(),
                         m["tensor_stride"])
    cap = {}
    h = blk.local_mix.register_forward_hook(lambda mod, i, o: cap.__setitem__("local", o.F))
    with torch.no_grad():
        core = blk._core(st, m["s"], m["r"], blk.pos_weight[0].weight,
                         blk.alpha if m["baseop"] == "cos_x" else None, m["C"] // m["groups"],
                         float(m["tensor_stride"]) if (m["variant"] == "encoder" and m["baseop"] == "cos_x") else 1.0)
        assert rel_err(core.cpu().numpy(), g["core"]) < TOL
        if m["r"] == 3:      # round 6: r = 3 forward pinned on reference OUTPUT (compiled devoxelize_forward_cpu, make_golden.py)
            assert rel_err(core.cpu().numpy(), g["core_refcpu"]) < TOL
        out = blk(st, m["s"], m["r"])
    h.remove()
    assert out is st                                             # in-place contract
    assert rel_err(cap["local"].cpu().numpy(), g["local"]) < TOL
    assert rel_err(st.F.cpu().numpy(), g["out"]) < TOL
    if m["r"] == 3:
        assert rel_err(st.F.cpu().numpy(), g["out_refcpu"]) < TOL


@pytest.mark.parametrize("name", golden_files("g_block_*.npz"))
def test_block_grads_vs_reference(name):
    import link_amd as la
    g = load_golden(name)
    m = g["meta"]
    blk = make_block(g).train()
    feats = torch.from_numpy(g["feats"]).cuda().requires_grad_(True)
    st = la.SparseTensor(feats, torch.from_numpy(g["coords"]).cuda(), m["tensor_stride"])
    core = blk._core(st, m["s"], m["r"], blk.pos_weight[0].weight,
                     blk.alpha if m["baseop"] == "cos_x" else None, m["C"] // m["groups"],
                     float(m["tensor_stride"]) if (m["variant"] == "encoder" and m["baseop"] == "cos_x") else 1.0)
    assert rel_err(core.detach().cpu().numpy(), g["core"]) < TOL
    core.backward(torch.from_numpy(g["grad_out"]).cuda())
    assert rel_err(feats.grad.cpu().numpy(), g["grad_feats"]) < 5e-4
    params = dict(blk.named_parameters())
    for k, v in g.items():
        if k.startswith("grad__"):
            assert rel_err(params[k[6:]].grad.cpu().numpy(), v) < 5e-4, k


@pytest.mark.parametrize("C,groups,baseop,s,r,n", [(64, 2, "cos", 7, 3, 20000), (16, 2, "cos", 7, 3, 10000),
                                                    (32, 1, "cos_x", 3, 2, 8000), (128, 2, "sin", 5, 3, 6000),
                                                    (48, 1, "cos_x", 4, 2, 5000), (20, 2, "cos", 3, 2, 3000)])
def test_core_fused_vs_oracle_wide(C, groups, baseop, s, r, n):
    """Widths of the BASELINE configs (16..128) incl. the non-MFMA width 20, on S-uniform inputs."""
    import link_amd as la
    torch.manual_seed(2)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    with torch.no_grad():
        for nme, p in blk.named_parameters():
            if "norm" in nme or "pre_mix.1" in nme or nme == "alpha":
                p.add_(0.2 * torch.randn_like(p))
    coords = s_uniform(n, grid=96, seed=C)
    gen = torch.Generator().manual_seed(1)
    feats = torch.randn(n, C, generator=gen)
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, agg=O.aggregate_c)
    st = la.SparseTensor(feats.cuda(), coords.cuda(), 1)
    with torch.no_grad():
        core = blk._core(st, s, r, blk.pos_weight[0].weight, blk.alpha if baseop == "cos_x" else None,
                         C // groups, 1.0)
    assert rel_err(core.cpu().numpy(), ref.numpy()) < TOL


def test_tselk_block_vs_oracle():
    """Detection twin: Linear(3,C) with the first C/2 columns tiled twice, r=3, spconv-layout indices."""
    import link_amd as la
    torch.manual_seed(3)
    C, n, stride = 32, 6000, 7
    blk = la.TSELKBlock(C, C, baseop="cos").cuda().eval()
    coords = s_uniform(n, grid=80, seed=9)
    gen = torch.Generator().manual_seed(4)
    feats = torch.randn(n, C, generator=gen)
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref_core = O.elk_core_torch(feats, coords, params, stride, 3, "cos", 1, variant="det", agg=O.aggregate_c)
    indices = coords[:, [3, 2, 1, 0]].contiguous().cuda()          # (batch, z, y, x)
    sct = la.SparseConvTensor(feats.cuda(), indices, spatial_shape=[80, 80, 80], batch_size=1)
    cap = {}
    h = blk.local_mix.register_forward_hook(lambda mod, i, o: cap.__setitem__("local", o.F))
    with torch.no_grad():
        out = blk(sct, stride)
    h.remove()
    assert isinstance(out, la.SparseConvTensor) and torch.equal(out.indices, indices)
    local = torch.nn.functional.layer_norm(cap["local"].cpu(), (C,), params["norm_local.weight"],
                                           params["norm_local.bias"], 1e-6)
    ref = torch.relu(ref_core + local)
    assert rel_err(out.features.cpu().numpy(), ref.numpy()) < TOL


@pytest.mark.parametrize("dense", [False, True])                     # general layout / dense-cell layout (opt-in)
def test_tselk_wrong_spatial_shape_contract(dense):
    """spatial_shape smaller than the data (caller error): voxels outside it are dropped by the index.  The
    contract: their core rows are zeros -- never uninitialised memory -- rows of the other
    voxels are finite, and the status word reports it (ElkCorePlan.check / BlockIndex.M raise)."""
    import link_amd as la
    import link_amd._lib as L
    torch.manual_seed(3)
    C, stride, grid, n = 32, 7, 80, 6000
    blk = la.TSELKBlock(C, C, baseop="cos").cuda().eval()
    blk.dense_layout = dense           # caller-supplied bounds use the general layout unless the module opts in
    coords = s_uniform(n, grid=grid, seed=9)
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(4)).cuda()
    indices = coords[:, [3, 2, 1, 0]].contiguous().cuda()
    half = grid // 2
    # the block grid covers whole blocks: a voxel is dropped when its BLOCK lies beyond the last block of the shape
    outside = (torch.div(coords[:, :3], stride, rounding_mode="floor") > (half - 1) // stride).any(1)
    assert outside.any() and (~outside).any()
    cap = {}
    orig = blk._core
    blk._core = lambda *a, **k: cap.setdefault("core", orig(*a, **k))
    for _ in range(2):          # twice: a recycled allocation holds the previous (non-zero) result
        cap.clear()
        sct = la.SparseConvTensor(feats, indices, spatial_shape=[half, half, half], batch_size=1)
        with torch.no_grad():
            blk(sct, stride)
        core = cap["core"].cpu()
        assert torch.isfinite(core).all()
        assert (core[outside] == 0).all()
        assert (core[~outside].abs().sum(1) > 0).all()
    plans = [p for p in blk.__dict__.get("_dc_plans", {}).values() if p is not None]
    assert bool(plans) == dense
    if plans:
        with pytest.raises(L.LinkAmdError):
            plans[0].check()


def test_core_on_coordinates_beyond_the_dense_grid_limit():
    """Two clusters 40 000 cells apart in every axis: the dense block grid would need > 2^28 cells, so the block
    core takes the reference algorithm on the op kernels (hash / unique / query) -- same result as the oracle,
    forward and backward."""
    import link_amd as la
    torch.manual_seed(5)
    C, s, r = 16, 3, 2
    blk = la.ELKBlock(C, C, groups=1, baseop="cos_x").cuda()
    a = s_uniform(1500, grid=24, seed=1)
    b = s_uniform(1500, grid=24, seed=2)
    b[:, :3] += 40000
    coords = torch.cat([a, b]).contiguous()
    feats = torch.randn(3000, C, generator=torch.Generator().manual_seed(3))
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, "cos_x", 1, agg=O.aggregate_c)
    st = la.SparseTensor(feats.cuda(), coords.cuda(), 1)
    with torch.no_grad():
        core = blk.eval()._core(st, s, r, blk.pos_weight[0].weight, blk.alpha, C, 1.0)
    assert rel_err(core.cpu().numpy(), ref.numpy()) < TOL
    f = feats.cuda().requires_grad_(True)
    out = blk.train()._core(la.SparseTensor(f, coords.cuda(), 1), s, r, blk.pos_weight[0].weight, blk.alpha, C, 1.0)
    assert rel_err(out.detach().cpu().numpy(), ref.numpy()) < TOL
    out.square().sum().backward()
    assert torch.isfinite(f.grad).all() and f.grad.abs().sum() > 0


def test_cfg2_full_size_core():
    """BASELINE cfg2: N=100k, C=64, cos g=2, r=3, s=7 through the fused core vs the oracle (a few s)."""
    import link_amd as la
    torch.manual_seed(2)
    blk = la.ELKBlock(64, 64, groups=2, baseop="cos").cuda().eval()
    coords = s_uniform(100_000)
    gen = torch.Generator().manual_seed(1)
    feats = torch.randn(100_000, 64, generator=gen)
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, 7, 3, "cos", 2, agg=O.aggregate_c)
    st = la.SparseTensor(feats.cuda(), coords.cuda(), 1)
    with torch.no_grad():
        core = blk._core(st, 7, 3, blk.pos_weight[0].weight, None, 32, 1.0)
        core2 = blk._core(st, 7, 3, blk.pos_weight[0].weight, None, 32, 1.0)
    assert rel_err(core.cpu().numpy(), ref.numpy()) < TOL
    assert torch.equal(core, core2)                     # deterministic


@pytest.mark.parametrize("stride,baseop,groups,s,r", [(1, "cos", 2, 14, 3), (2, "cos_x", 1, 6, 2)])
def test_core_on_lidar_like_frame(stride, baseop, groups, s, r):
    """Surface-like sparse frame (large dense grid, few occupied cells, many voxels per block): exercises
    the multi-tile look-back scan, the cooperative large-block modulate mode and the zero-row gather."""
    import link_amd as la
    from helpers import lidar_like
    torch.manual_seed(4)
    C = 64
    coords = torch.from_numpy(lidar_like(40000, seed=3, stride=stride))
    n = coords.shape[0]
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop, variant="encoder").cuda().eval()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(5))
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, baseop, groups, variant="encoder", tensor_stride=stride,
                           agg=O.aggregate_c)
    st = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
    with torch.no_grad():
        core = blk._core(st, s, r, blk.pos_weight[0].weight, blk.alpha if baseop == "cos_x" else None,
                         C // groups, float(stride) if baseop == "cos_x" else 1.0)
    idx = la.link_index_of(st, s)
    assert idx.M > 0
    if stride == 1:
        assert n / idx.M > 4                    # large blocks: the cooperative modulate mode is what ran
    assert rel_err(core.cpu().numpy(), ref.numpy()) < TOL


def test_core_multi_frame_batch_and_repeatability():
    """Two frames in one SparseTensor (batch column 0/1): blocks never mix frames (the batch index is part
    of the block key, utils.py:45); per-frame results equal the frames run alone; 20 repeated runs are
    bitwise identical (write-through stores + L2 state must not leak between launches)."""
    import link_amd as la
    torch.manual_seed(6)
    C, s, r = 64, 7, 3
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    c0, c1 = s_uniform(7000, grid=64, seed=21), s_uniform(9000, grid=64, seed=22, batch=1)
    coords = torch.cat([c0, c1], 0)
    perm = torch.randperm(coords.shape[0], generator=torch.Generator().manual_seed(1))
    coords = coords[perm].contiguous()                      # frames interleaved
    feats = torch.randn(coords.shape[0], C, generator=torch.Generator().manual_seed(2))

    def core(f, c):
        st = la.SparseTensor(f.cuda(), c.cuda(), 1)
        with torch.no_grad():
            return blk._core(st, s, r, blk.pos_weight[0].weight, None, C // 2, 1.0)
    both = core(feats, coords)
    for b in (0, 1):
        sel = coords[:, 3] == b
        alone = core(feats[sel].contiguous(), coords[sel].contiguous())
        assert rel_err(both[sel.cuda()].cpu().numpy(), alone.cpu().numpy()) < 2e-6
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    ref = O.elk_core_torch(feats, coords, params, s, r, "cos", 2, agg=O.aggregate_c)
    assert rel_err(both.cpu().numpy(), ref.numpy()) < TOL
    for _ in range(20):
        assert torch.equal(core(feats, coords), both)
