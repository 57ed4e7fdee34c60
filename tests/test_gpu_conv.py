"""Row N1: the stride-1 submanifold convolution kernel (link_subm_conv_forward
behind link_amd.Conv3d / link_amd.elk.subm_conv) against the oracle's restatement of the reference's
CPU branch (oracle.subm_conv_torch, itself pinned on the reference's local_mix goldens)."""
import numpy as np
import pytest
import torch

from helpers import golden_files, lidar_like, load_golden, rel_err, s_uniform

pytestmark = pytest.mark.gpu


def _frame(kind, n, stride):
    if kind == "lidar":
        c = torch.from_numpy(lidar_like(n, seed=4, stride=stride))
    else:
        c = s_uniform(n, grid=24, seed=6)              # dense enough that most 3^3 neighbours exist
        c[:, :3] *= stride
    return c


@pytest.mark.parametrize("C,kind,n,stride", [
    (64, "lidar", 20000, 1), (64, "dense", 5000, 1), (16, "dense", 3001, 2), (128, "dense", 2000, 1),
    (32, "lidar", 9000, 2), (48, "dense", 777, 1), (64, "dense", 7, 1),
    (8, "dense", 3000, 1), (24, "dense", 1500, 1),                     # lane=channel kernel
])
def test_subm_conv_forward_vs_oracle(C, kind, n, stride):
    import link_amd as la
    from oracle import link_oracle as lo
    coords = _frame(kind, n, stride)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(n, C, generator=g)
    conv = la.Conv3d(C, C, kernel_size=3).cuda()
    st = la.SparseTensor(feats.cuda(), coords.cuda(), stride)
    with torch.no_grad():
        out = conv(st).F
    ref = lo.subm_conv_torch(feats.double(), coords, conv.kernel.detach().cpu().double(), stride)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5
    # the neighbour table is the reference's kernel map (bit-exact integer work)
    nbr, order = st.kmaps[("link_conv_nbr", st.C.data_ptr(), n, st.s, (3, 3, 3))]
    assert order is not None and sorted(order.tensor().cpu().tolist()) == list(range(n))     # built on first use
    assert np.array_equal(nbr.cpu().numpy(), lo.conv_neighbor_table(coords.numpy(), 3, stride))


def test_subm_conv_rectangular_widths():
    from link_amd.elk import subm_conv
    from oracle import link_oracle as lo
    coords = s_uniform(2000, grid=20, seed=1)
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(2000, 12, generator=g)
    kernel = torch.randn(27, 12, 40, generator=g) * 0.1
    nbr = torch.from_numpy(lo.conv_neighbor_table(coords.numpy(), 3, 1)).cuda()
    out = subm_conv(feats.cuda(), kernel.cuda(), nbr)
    ref = lo.subm_conv_torch(feats.double(), coords, kernel.double(), 1)
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("C,kind", [(64, "lidar"), (16, "dense"), (8, "dense"), (48, "dense"), (80, "dense")])
def test_subm_conv_gradients_vs_oracle_autograd(C, kind):
    import link_amd as la
    from oracle import link_oracle as lo
    coords = _frame(kind, 6000 if kind == "lidar" else 2500, 1)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(8)
    feats = torch.randn(n, C, generator=g)
    gout = torch.randn(n, C, generator=g)
    conv = la.Conv3d(C, C, kernel_size=3).cuda()
    f = feats.cuda().requires_grad_(True)
    out = conv(la.SparseTensor(f, coords.cuda(), 1)).F
    out.backward(gout.cuda())
    fr = feats.double().requires_grad_(True)
    kr = conv.kernel.detach().cpu().double().requires_grad_(True)
    lo.subm_conv_torch(fr, coords, kr, 1).backward(gout.double())
    assert rel_err(f.grad.cpu().numpy(), fr.grad.numpy()) < 1e-5
    assert rel_err(conv.kernel.grad.cpu().numpy(), kr.grad.numpy()) < 1e-4


def test_elkblock_forward_full_block_vs_oracle():
    """R_block: whole ELKBlock.forward (local_mix on the HIP conv kernel + fused R_core + add/ReLU)."""
    import link_amd as la
    from oracle import link_oracle as lo
    import torch.nn.functional as TF
    C, s, r = 64, 7, 3
    coords = torch.from_numpy(lidar_like(15000, seed=9))
    n = coords.shape[0]
    torch.manual_seed(0)
    blk = la.ELKBlock(C, C, groups=2, baseop="cos").cuda().eval()
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(2))
    with torch.no_grad():
        out = blk(la.SparseTensor(feats.cuda(), coords.cuda(), 1), s, r).F
    sd = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    core = lo.elk_core_torch(feats, coords, sd, s, r, "cos", 2)
    local = lo.subm_conv_torch(feats, coords, sd["local_mix.0.kernel"], 1)
    ref = torch.relu(core + TF.layer_norm(local, (C,), sd["norm_local.weight"], sd["norm_local.bias"], 1e-6))
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-4


@pytest.mark.parametrize("C,baseop,groups", [(64, "cos", 2), (8, "sin", 2), (32, "cos_x", 1)])
def test_fused_epilogue_equals_module_by_module(C, baseop, groups):
    """Row N2: the conv kernel's fused LayerNorm + add + ReLU tail against the same block run module by
    module (a forward hook on local_mix switches the block to that path)."""
    import link_amd as la
    coords = torch.from_numpy(lidar_like(12000, seed=5)).cuda()
    n = coords.shape[0]
    torch.manual_seed(1)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().eval()
    with torch.no_grad():
        blk.norm_local.weight.uniform_(0.5, 1.5); blk.norm_local.bias.uniform_(-0.5, 0.5)
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(2)).cuda()
    s, r = (7, 3) if baseop != "cos_x" else (3, 2)
    with torch.no_grad():
        fused = blk(la.SparseTensor(feats.clone(), coords, 1), s, r).F
        seen = []
        h = blk.local_mix.register_forward_hook(lambda m, i, o: seen.append(o.F))
        try:
            plain = blk(la.SparseTensor(feats.clone(), coords, 1), s, r).F
        finally:
            h.remove()
    assert len(seen) == 1                           # the hooked run went module by module
    assert rel_err(fused.cpu().numpy(), plain.cpu().numpy()) < 2e-6
    assert float(fused.min()) >= 0.0


@pytest.mark.parametrize("C,baseop,groups,s,r", [(64, "cos", 2, 7, 3), (32, "cos_x", 1, 3, 2), (16, "sin", 2, 5, 3)])
def test_elkblock_training_step_vs_fp64_oracle(C, baseop, groups, s, r):
    """Whole ELKBlock forward+backward in training mode (fused core backward, HIP conv + its gradients,
    fused tail backward) against fp64 autograd over the oracle restatements."""
    import link_amd as la
    from oracle import link_oracle as lo
    import torch.nn.functional as TF
    # 0.2 m voxels: coordinates stay below ~500, so theta keeps the fp32 precision the 1e-4 gate assumes
    coords = torch.from_numpy(lidar_like(9000, seed=11, voxel=0.2))
    n = coords.shape[0]
    torch.manual_seed(3)
    blk = la.ELKBlock(C, C, groups=groups, baseop=baseop).cuda().train()
    with torch.no_grad():
        for nm, p in blk.named_parameters():
            if "norm" in nm or nm.endswith("pre_mix.1.weight") or nm.endswith("pre_mix.1.bias"):
                p.add_(0.1 * torch.randn_like(p))
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(2))
    gout = torch.randn(n, C, generator=torch.Generator().manual_seed(4))
    f = feats.cuda().requires_grad_(True)
    out = blk(la.SparseTensor(f, coords.cuda(), 1), s, r).F
    out.backward(gout.cuda())
    # fp64 oracle
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in blk.state_dict().items()}
    fr = feats.double().requires_grad_(True)
    core = lo.elk_core_torch(fr, coords, sd, s, r, baseop, groups)
    local = lo.subm_conv_torch(fr, coords, sd["local_mix.0.kernel"], 1)
    z = core + TF.layer_norm(local, (C,), sd["norm_local.weight"], sd["norm_local.bias"], 1e-6)
    # ReLU is discontinuous: differentiate the oracle with the GPU run's activation mask (the two masks may
    # only disagree where the pre-activation is within rounding of zero), else one flipped element moves
    # sums of random-sign gradients by percents
    mask = (out.detach().cpu() > 0)
    flips = (z.detach() > 0) != mask
    assert float(z.detach().abs()[flips].max()) < 1e-3 if bool(flips.any()) else True
    ref = z * mask.double()
    ref.backward(gout.double())
    # forward: the 1e-4 gate is against the fp32 restatement (theta of LiDAR-sized coordinates carries
    # ~1e-4 of fp32 rounding that an fp64 oracle does not have); fp64 bounds the gradients
    sd32 = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    z32 = lo.elk_core_torch(feats, coords, sd32, s, r, baseop, groups) + TF.layer_norm(
        lo.subm_conv_torch(feats, coords, sd32["local_mix.0.kernel"], 1), (C,), sd32["norm_local.weight"],
        sd32["norm_local.bias"], 1e-6)
    assert rel_err(out.detach().cpu().numpy(), torch.relu(z32).numpy()) < 1e-4
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 5e-4
    assert rel_err(f.grad.cpu().numpy(), fr.grad.numpy()) < 5e-4
    for nm, p in blk.named_parameters():
        assert p.grad is not None, nm
        assert rel_err(p.grad.cpu().numpy(), sd[nm].grad.numpy()) < 5e-4, nm


def _chain(la, g):
    c1 = la.Conv3d(8, 16, 3).cuda(); c2 = la.Conv3d(16, 16, 2, stride=2).cuda()
    c3 = la.Conv3d(16, 24, 3).cuda(); c4 = la.Conv3d(24, 8, 2, stride=2, transposed=True).cuda()
    with torch.no_grad():
        for c, k in ((c1, "k1"), (c2, "k2"), (c3, "k3"), (c4, "k4")):
            c.kernel.copy_(torch.from_numpy(g[k]))
    return c1, c2, c3, c4


@pytest.mark.parametrize("name", golden_files("g_stridedconv_*.npz"))
def test_strided_and_transposed_conv_vs_reference(name):
    """The reference's own spnn.Conv3d chain (k3 s1 -> k2 s2 down -> k3 at stride 2 -> k2 s2 transposed):
    coordinates and their order bit-exact, features within 1e-5 (batch > 1 fixtures: coordinates from the
    reference, features against the oracle -- the reference CPU neighbour hash is defective there)."""
    import link_amd as la
    from oracle import link_oracle as lo
    g = load_golden(name)
    c1, c2, c3, c4 = _chain(la, g)
    x0 = la.SparseTensor(torch.from_numpy(g["feats"]).cuda(), torch.from_numpy(g["coords"]).cuda(), 1)
    x0.cmaps.setdefault(x0.stride, x0.coords)
    with torch.no_grad():
        x1 = c1(x0); x2 = c2(x1); x3 = c3(x2); x4 = c4(x3)
    assert x2.s == (2, 2, 2) and x3.s == (2, 2, 2) and x4.s == (1, 1, 1)
    assert np.array_equal(x2.C.cpu().numpy(), g["x2_C"]) and np.array_equal(x4.C.cpu().numpy(), g["x4_C"])
    assert x2.kmaps is x0.kmaps and x4.cmaps is x0.cmaps
    t = lambda a: torch.from_numpy(a)
    if g["meta"]["features_valid"]:
        ref = {k: g[k] for k in ("x1_F", "x2_F", "x3_F", "x4_F")}
    else:
        r1 = lo.subm_conv_torch(t(g["feats"]), g["coords"], t(g["k1"]), 1)
        down = lo.strided_conv_table(g["coords"], g["x2_C"], 2, 1)
        r2 = lo.gather_conv_torch(r1, down, t(g["k2"]))
        r3 = lo.subm_conv_torch(r2, g["x2_C"], t(g["k3"]), 2)
        r4 = lo.gather_conv_torch(r3, None, t(g["k4"]), n_out=g["coords"].shape[0], transposed_of=down)
        ref = {"x1_F": r1.numpy(), "x2_F": r2.numpy(), "x3_F": r3.numpy(), "x4_F": r4.numpy()}
    for k, x in (("x1_F", x1), ("x2_F", x2), ("x3_F", x3), ("x4_F", x4)):
        assert rel_err(x.F.cpu().numpy(), ref[k]) < 1e-5, k


def test_strided_chain_gradients_and_mfma_widths():
    """Down / transposed / rectangular convolutions at MFMA widths (32 -> 64 -> 64 -> 32): forward and all
    gradients against fp64 autograd over the oracle restatements."""
    import link_amd as la
    from oracle import link_oracle as lo
    coords = torch.from_numpy(lidar_like(8000, seed=6, voxel=0.2))
    n = coords.shape[0]
    g = torch.Generator().manual_seed(5)
    feats = torch.randn(n, 32, generator=g)
    torch.manual_seed(1)
    c1 = la.Conv3d(32, 64, 3).cuda(); c2 = la.Conv3d(64, 64, 2, stride=2, bias=True).cuda()
    c3 = la.Conv3d(64, 32, 2, stride=2, transposed=True).cuda()
    f = feats.cuda().requires_grad_(True)
    x0 = la.SparseTensor(f, coords.cuda(), 1); x0.cmaps.setdefault(x0.stride, x0.coords)
    x3 = c3(c2(c1(x0)))
    gout = torch.randn(n, 32, generator=g)
    x3.F.backward(gout.cuda())
    fr = feats.double().requires_grad_(True)
    k1, k2, k3 = [c.kernel.detach().cpu().double().requires_grad_(True) for c in (c1, c2, c3)]
    b2 = c2.bias.detach().cpu().double().requires_grad_(True)
    cc = lo.downsample_coords(coords.numpy(), 2, 1)
    down = lo.strided_conv_table(coords.numpy(), cc, 2, 1)
    r = lo.gather_conv_torch(lo.gather_conv_torch(lo.subm_conv_torch(fr, coords, k1, 1), down, k2) + b2, None, k3,
                             n_out=n, transposed_of=down)
    r.backward(gout.double())
    assert rel_err(x3.F.detach().cpu().numpy(), r.detach().numpy()) < 1e-5
    assert rel_err(f.grad.cpu().numpy(), fr.grad.numpy()) < 1e-5
    for c, k in ((c1, k1), (c2, k2), (c3, k3)):
        assert rel_err(c.kernel.grad.cpu().numpy(), k.grad.numpy()) < 1e-4
    assert rel_err(c2.bias.grad.cpu().numpy(), b2.grad.numpy()) < 1e-5


# ------------------------------------------------------------------------------------------------
# pair-list form (link_conv_pairs_gemm + link_conv_pairs_sum): the sparse-frame kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,kind,n", [
    (64, 64, "uniform", 50000), (64, 64, "lidar", 20000), (16, 16, "dense", 3001), (32, 64, "lidar", 9000),
    (128, 128, "dense", 1500), (64, 32, "uniform", 7000), (128, 64, "lidar", 4000), (64, 64, "dense", 7),
])
def test_pair_form_vs_oracle_and_table_form(cin, cout, kind, n):
    """Both forms of the convolution against the oracle restatement, on sparse (S-uniform: ~1.1 neighbours per
    voxel), LiDAR-like (~6) and dense (~20) neighbourhoods; the two forms agree to rounding and each is bitwise
    reproducible."""
    import link_amd as la
    from link_amd.elk import subm_conv
    from oracle import link_oracle as lo
    coords = s_uniform(n, grid=96, seed=5) if kind == "uniform" else _frame(kind, n, 1)
    n = coords.shape[0]
    feats = torch.randn(n, cin, generator=torch.Generator().manual_seed(3))
    conv = la.Conv3d(cin, cout, kernel_size=3).cuda()
    st = la.SparseTensor(feats.cuda(), coords.cuda(), 1)
    nbr, order = conv._neighbor_table(st)
    w = conv.kernel.detach()
    a = subm_conv(st.F, w, nbr, order, form="pairs")
    a2 = subm_conv(st.F, w, nbr, order, form="pairs")
    b = subm_conv(st.F, w, nbr, order, form="table")
    assert torch.equal(a, a2)
    ref = lo.subm_conv_torch(feats.double(), coords, w.cpu().double(), 1)
    assert rel_err(a.cpu().numpy(), ref.numpy()) < 1e-5
    assert rel_err(b.cpu().numpy(), ref.numpy()) < 1e-5
    plan = nbr._link_pairs
    assert plan.direct and plan.rows_pad % 128 == 0
    # the plan is the reference's kernel map regrouped: as many pairs as the table has neighbours
    assert plan.pairs + n == int((nbr >= 0).sum().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,cout,kind,n", [(64, 64, "lidar", 20000), (32, 64, "uniform", 9000), (128, 128, "dense", 1500),
                                             (4, 16, "lidar", 6000)])
def test_pair_form_half_rows(cin, cout, kind, n, dtype, monkeypatch):
    """fp16 / bf16 feature rows at the boundary (the reference's AMP contract, nn/functional/conv.py:18).
    (a) link_conv_*_io (fp32 weights): the kernels widen on load and round once on store, so the result equals the
    fp32 path's output rounded to the row type, bit for bit, with and without the fused epilogue.
    (b) link_conv_*_amp (the default for half rows): the weights are rounded to the row type as custom_fwd does and
    the products run on the 16-bit matrix cores -- exact products, fp32 accumulation in another order: equal to
    the fp32 path on the rounded weights up to one rounding of the row type; with 16-bit contribution rows (the
    default for fp16, as the reference's half mm) one more rounding per neighbour.  The table form returns the same dtype."""
    import link_amd as la
    import link_amd.elk as E
    from link_amd.elk import subm_conv, subm_conv_ln_add_relu
    coords = s_uniform(n, grid=96, seed=5) if kind == "uniform" else _frame(kind, n, 1)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(8)
    feats = torch.randn(n, cin, generator=g).to(dtype).cuda()
    conv = la.Conv3d(cin, cout, kernel_size=3).cuda()
    st = la.SparseTensor(feats.float(), coords.cuda(), 1)
    nbr, order = conv._neighbor_table(st)
    w = conv.kernel.detach()
    wr = w.to(dtype).float()
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    lw, lb = torch.randn(cout, generator=g).cuda(), torch.randn(cout, generator=g).cuda()
    add = torch.randn(n, cout, generator=g).to(dtype).cuda()
    monkeypatch.setattr(E, "SPLIT_MFMA", False)        # the fp32 comparison runs use the fp32 matrix instruction, like the _io kernels
    for amp, c16 in ((False, False), (True, False), (True, True)):     # c16: contribution rows in the row type too
        monkeypatch.setattr(E, "AMP_MFMA", amp)
        monkeypatch.setattr(E, "AMP_CONTRIB16", (dtype,) if c16 else ())
        a = subm_conv(feats, w, nbr, order, form="pairs")
        ref = subm_conv(feats.float(), wr if amp else w, nbr, order, form="pairs")
        assert a.dtype == dtype and ref.dtype == torch.float32
        if amp:
            assert float((a.float() - ref).abs().max()) <= (3 if c16 else 1) * ulp * float(ref.abs().max())
            assert rel_err(a.float().cpu().numpy(), ref.cpu().numpy()) < (2 if c16 else 1) * ulp
        else:
            assert torch.equal(a, ref.to(dtype))
        for affine in (False, True):
            a = subm_conv_ln_add_relu(feats, w, nbr, order, lw, lb, 1e-6, add, relu=True, form="pairs", affine=affine)
            ref = subm_conv_ln_add_relu(feats.float(), wr if amp else w, nbr, order, lw, lb, 1e-6, add.float(), relu=True,
                                        form="pairs", affine=affine)
            assert a.dtype == dtype
            if amp:
                assert rel_err(a.float().cpu().numpy(), ref.cpu().numpy()) < (2 if c16 else 1) * ulp
            else:
                assert torch.equal(a, ref.to(dtype))
    b = subm_conv(feats, w, nbr, order, form="table")
    ref = subm_conv(feats.float(), w, nbr, order, form="pairs")
    assert b.dtype == dtype and rel_err(b.float().cpu().numpy(), ref.cpu().numpy()) < ulp


def test_pair_form_strided_tables_and_tail():
    """Tables that are not submanifold (k2-s2 down-sampling and its transpose: no identity rows) and the
    LayerNorm + add + ReLU epilogue, pair form vs table form."""
    import link_amd as la
    from link_amd.elk import subm_conv, subm_conv_ln_add_relu
    coords = torch.from_numpy(lidar_like(30000, seed=2))
    n = coords.shape[0]
    C = 64
    feats = torch.randn(n, C, generator=torch.Generator().manual_seed(1)).cuda()
    st = la.SparseTensor(feats, coords.cuda(), 1)
    down = la.Conv3d(C, C, kernel_size=2, stride=2).cuda()
    km = down._strided_map(st)
    w = down.kernel.detach()
    for table, src in ((km.nbr_down, feats), (km.nbr_up, torch.randn(km.nbr_down.shape[0], C, device="cuda"))):
        a = subm_conv(src, w, table, None, form="pairs")
        b = subm_conv(src, w, table, None, form="table")
        assert not table._link_pairs.direct
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5
    conv = la.Conv3d(C, C, kernel_size=3).cuda()
    nbr, order = conv._neighbor_table(st)
    lw, lb = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
    add = torch.randn(n, C, device="cuda")
    a = subm_conv_ln_add_relu(feats, conv.kernel.detach(), nbr, order, lw, lb, 1e-6, add, relu=True, form="pairs")
    b = subm_conv_ln_add_relu(feats, conv.kernel.detach(), nbr, order, lw, lb, 1e-6, add, relu=True, form="table")
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 2e-5
    a0 = subm_conv_ln_add_relu(feats, conv.kernel.detach(), nbr, order, lw, lb, 1e-6, None, relu=False, form="pairs")
    ref = torch.nn.functional.layer_norm(subm_conv(feats, conv.kernel.detach(), nbr, order, form="table"), (C,), lw, lb, 1e-6)
    assert rel_err(a0.cpu().numpy(), ref.cpu().numpy()) < 2e-5


# ------------------------------------------------------------------------------------------------
# functional / backend forms (torchsparse.nn.functional.conv3d, spdownsample; backend.convolution_*_cuda)
# ------------------------------------------------------------------------------------------------
def test_functional_conv3d_and_spdownsample():
    import link_amd as la
    import link_amd.functional as F
    from oracle import link_oracle as lo
    coords = torch.from_numpy(lidar_like(9000, seed=7))
    n = coords.shape[0]
    feats = torch.randn(n, 32, generator=torch.Generator().manual_seed(1)).cuda()
    mod = la.Conv3d(32, 32, 3, bias=True).cuda()
    with torch.no_grad():
        a = mod(la.SparseTensor(feats, coords.cuda(), 1))
        b = F.conv3d(la.SparseTensor(feats, coords.cuda(), 1), mod.kernel, 3, mod.bias)
    assert torch.equal(a.F, b.F) and torch.equal(a.C, b.C) and a.s == b.s
    down = la.Conv3d(32, 64, 2, stride=2).cuda()
    up = la.Conv3d(64, 32, 2, stride=2, transposed=True).cuda()
    with torch.no_grad():
        x = la.SparseTensor(feats, coords.cuda(), 1)
        x.cmaps.setdefault(x.s, x.C)                    # a stride-1 layer would have registered them (conv.py:144)
        d1 = down(x)
        u1 = up(d1)
        y = la.SparseTensor(feats, coords.cuda(), 1)
        y.cmaps.setdefault(y.s, y.C)
        d2 = F.conv3d(y, down.kernel, 2, None, stride=2)
        u2 = F.conv3d(d2, up.kernel, 2, None, stride=2, transposed=True)
    assert torch.equal(d1.F, d2.F) and torch.equal(d1.C, d2.C) and d2.s == (2, 2, 2)
    assert torch.equal(u1.F, u2.F) and torch.equal(u2.C, coords.cuda())
    # spdownsample: the k2-s2 branch is the module's coordinate set; the general branch (k3-s2) by brute force
    c2 = F.spdownsample(coords.cuda(), 2, 2, 1)
    assert torch.equal(c2, d1.C)
    assert np.array_equal(c2.cpu().numpy(), lo.downsample_coords(coords.numpy(), 2, 1))
    small = s_uniform(400, grid=12, seed=2).cuda()
    c3 = F.spdownsample(small, 2, 3, 1).cpu().numpy()
    cn = small.cpu().numpy()
    cand = set()
    lo3 = cn[:, :3].min(0)
    for row in cn:
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dz in (-1, 0, 1):
                    p = row[:3] + (dx, dy, dz)
                    if (p % 2 == 0).all() and (p >= lo3).all():
                        cand.add((int(row[3]), int(p[0]), int(p[1]), int(p[2])))
    ref = np.array(sorted(cand), dtype=np.int32)[:, [1, 2, 3, 0]]
    assert np.array_equal(c3, ref)


def test_backend_convolution_names_with_reference_kmap_layout():
    """torchsparse.backend.convolution_forward_cuda / convolution_backward_cuda with the reference's kernel map
    (pairs grouped by offset + per-offset sizes on the host, conv.py:109-122), against a torch loop over offsets."""
    import link_amd as la
    import link_amd.backend as B
    coords = torch.from_numpy(lidar_like(8000, seed=9)).cuda()
    n, cin, cout = coords.shape[0], 32, 64
    conv = la.Conv3d(cin, cout, 3).cuda()
    st = la.SparseTensor(torch.randn(n, cin, device="cuda"), coords, 1)
    nbr, _ = conv._neighbor_table(st)
    res = nbr.t().contiguous()                                  # [K, N_out] like sphashquery's result
    nbsizes = (res != -1).sum(1)
    nz = torch.nonzero(res != -1)
    nbmaps = torch.stack([res[nz[:, 0], nz[:, 1]].long(), nz[:, 1]], 1).int()   # (in, out) grouped by offset
    w = conv.kernel.detach()
    out = torch.zeros(n, cout, device="cuda")
    B.convolution_forward_cuda(st.F, out, w, nbmaps, nbsizes.cpu().int(), False)
    ref = torch.zeros(n, cout, device="cuda", dtype=torch.float64)
    cur = 0
    for k in range(27):
        m = nbmaps[cur:cur + int(nbsizes[k])].long()
        ref.index_add_(0, m[:, 1], st.F[m[:, 0]].double() @ w[k].double())
        cur += int(nbsizes[k])
    assert rel_err(out.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    g = torch.randn(n, cout, device="cuda")
    gi, gw = torch.zeros(n, cin, device="cuda"), torch.zeros_like(w)
    B.convolution_backward_cuda(st.F, gi, g, w, gw, nbmaps, nbsizes.cpu().int(), False)
    gi_ref = torch.zeros(n, cin, device="cuda", dtype=torch.float64)
    gw_ref = torch.zeros(27, cin, cout, device="cuda", dtype=torch.float64)
    cur = 0
    for k in range(27):
        m = nbmaps[cur:cur + int(nbsizes[k])].long()
        gi_ref.index_add_(0, m[:, 0], g[m[:, 1]].double() @ w[k].double().t())
        gw_ref[k] = st.F[m[:, 0]].double().t() @ g[m[:, 1]].double()
        cur += int(nbsizes[k])
    assert rel_err(gi.cpu().numpy(), gi_ref.cpu().numpy()) < 1e-5
    assert rel_err(gw.cpu().numpy(), gw_ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("cin,cout,kind", [(128, 128, "lidar"), (32, 64, "lidar"), (64, 128, "uniform"), (128, 64, "dense"),
                                           (4, 64, "lidar"), (5, 16, "uniform")])     # first layers: rows padded to 16 channels
def test_pair_list_weight_gradient_wide_and_rectangular(cin, cout, kind):
    """link_conv_pairs_wgrad (widths the table weight-gradient kernel does not take) against fp64 per-offset GEMMs:
    submanifold table (centre offset as identity granules) and a strided table; bitwise reproducible."""
    import link_amd as la
    from link_amd.elk import _conv_weight_grad
    coords = s_uniform(6000, grid=40, seed=5) if kind == "uniform" else _frame(kind, 6000, 1)
    n = coords.shape[0]
    conv = la.Conv3d(cin, cout, 3).cuda()
    st = la.SparseTensor(torch.randn(n, cin, device="cuda"), coords.cuda(), 1)
    nbr, _ = conv._neighbor_table(st)
    g = torch.randn(n, cout, device="cuda")
    a = _conv_weight_grad(st.F, g, nbr, (27, cin, cout))
    b = _conv_weight_grad(st.F, g, nbr, (27, cin, cout))
    assert torch.equal(a, b)
    pad = torch.cat([st.F.double(), torch.zeros(1, cin, device="cuda", dtype=torch.float64)])
    idx = torch.where(nbr < 0, torch.full_like(nbr, n), nbr).long()
    ref = torch.stack([pad[idx[:, k]].t() @ g.double() for k in range(27)])
    assert rel_err(a.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    down = la.Conv3d(cin, cout, 2, stride=2).cuda()
    km = down._strided_map(st)
    gd = torch.randn(km.nbr_down.shape[0], cout, device="cuda")
    a = _conv_weight_grad(st.F, gd, km.nbr_down, (8, cin, cout))
    idx = torch.where(km.nbr_down < 0, torch.full_like(km.nbr_down, n), km.nbr_down).long()
    ref = torch.stack([pad[idx[:, k]].t() @ gd.double() for k in range(8)])
    assert rel_err(a.cpu().numpy(), ref.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("cin,cout,kind,n", [(64, 64, "lidar", 20000), (128, 128, "dense", 1500), (32, 64, "uniform", 9000), (128, 64, "lidar", 4000)])
def test_pair_gemm_fp16_split_of_fp32_rows(cin, cout, kind, n, monkeypatch):
    """link_conv_pairs_gemm_split (fp32 rows and weights as fp16 hi + lo pairs on the f16 matrix cores; inference forms)
    against the fp32-instruction kernel: same result to fp32 rounding; values outside the fp16 range -- a row entry
    of 1e5, a weight of 4e4 -- take the fp32 instruction inside the kernel and stay exact."""
    import link_amd as la
    import link_amd.elk as E
    from link_amd.elk import subm_conv_ln_add_relu
    coords = s_uniform(n, grid=96, seed=5) if kind == "uniform" else _frame(kind, n, 1)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(11)
    feats = torch.randn(n, cin, generator=g).cuda()
    conv = la.Conv3d(cin, cout, kernel_size=3).cuda()
    nbr, order = conv._neighbor_table(la.SparseTensor(feats, coords.cuda(), 1))
    sc, sh = torch.rand(cout, generator=g).cuda() + 0.5, torch.randn(cout, generator=g).cuda()

    def run(f, w, split):
        monkeypatch.setattr(E, "SPLIT_MFMA", split)
        return subm_conv_ln_add_relu(f, w, nbr, order, sc, sh, 0.0, None, relu=False, form="pairs", affine=True)

    w = conv.kernel.detach().clone()
    a, b = run(feats, w, True), run(feats, w, False)
    assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-6
    assert torch.equal(a, run(feats, w, True))
    big_f = feats.clone(); big_f[n // 2, 3] = 1.0e5                      # one row outside the range: its wave falls back
    assert rel_err(run(big_f, w, True).cpu().numpy(), run(big_f, w, False).cpu().numpy()) < 1e-6
    big_w = w.clone(); big_w[5, 1, 2] = 4.0e4                            # a weight outside the range: every wave falls back
    assert rel_err(run(feats, big_w, True).cpu().numpy(), run(feats, big_w, False).cpu().numpy()) < 1e-6


@pytest.mark.parametrize("C,ks", [(16, 5), (32, 5), (64, 5), (16, 7)])
def test_odd_kernels_larger_than_3_run_the_table_kernel(C, ks):
    """Conv3d advertises odd cubic kernels at stride 1; at the widths the pair-list kernels take (16/32/64/128) a 5^3 or
    7^3 neighbourhood has more offsets than the pair plan's 64-bit per-voxel mask holds, so these must fall through to
    the table kernel (forward, input gradient, weight gradient) instead of raising."""
    import link_amd as la
    from oracle import link_oracle as lo
    coords = s_uniform(1500, grid=16, seed=8)
    n = coords.shape[0]
    g = torch.Generator().manual_seed(2)
    feats = torch.randn(n, C, generator=g)
    conv = la.Conv3d(C, C, kernel_size=ks).cuda()
    assert conv.kernel.shape[0] == ks ** 3
    f = feats.cuda().requires_grad_(True)
    out = conv(la.SparseTensor(f, coords.cuda(), 1)).F
    fr = feats.double().requires_grad_(True)
    kr = conv.kernel.detach().cpu().double().requires_grad_(True)
    ref = lo.subm_conv_torch(fr, coords, kr, 1)
    assert rel_err(out.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    go = torch.randn(n, C, generator=g)
    out.backward(go.cuda())
    ref.backward(go.double())
    assert rel_err(f.grad.cpu().numpy(), fr.grad.numpy()) < 1e-5
    assert rel_err(conv.kernel.grad.cpu().numpy(), kr.grad.numpy()) < 1e-5
    with torch.no_grad():                                # the fused inference tail takes the same route
        blk_out = conv(la.SparseTensor(feats.cuda(), coords.cuda(), 1)).F
    assert rel_err(blk_out.cpu().numpy(), ref.detach().numpy()) < 1e-5


@pytest.mark.parametrize("subm", [True, False])
def test_pair_plan_laid_out_on_the_device_matches_the_host_layout(subm):
    """link_pair_plan_layout (no host round trip) against the host-side layout of the same table: same pairs, same rows per
    output, same convolution result; the counts that arrive later agree; finalize() trims to the exact plan."""
    import link_amd as la
    from link_amd import elk
    torch.manual_seed(11)
    n, c = 30000, 32
    coords = s_uniform(n, grid=64, seed=5).cuda()
    st = la.SparseTensor(torch.randn(n, c).cuda(), coords, 1)
    nbr, order = elk.neighbor_table_of(st, (3, 3, 3))
    if not subm:                                       # a gather table between two site sets: drop the identity column's meaning
        nbr = nbr[torch.randperm(n, device=nbr.device)].contiguous()
    host = elk._PairPlan(nbr, None)
    dev = elk._PairPlan(nbr, subm)
    assert host.exact and not dev.exact and host.direct == dev.direct == subm
    w = torch.randn(27, c, c).cuda() * 0.1
    outs = []
    for plan in (host, dev):
        out = torch.empty((n, c), device="cuda")
        outs.append(elk._conv_pairs(plan, st.F, w, c, c, out).clone())
    assert torch.equal(outs[0], outs[1])
    torch.cuda.synchronize()
    assert abs(dev.density - host.density) < 1e-12 and dev.pairs == host.pairs
    assert dev.rows_launch == host.rows_pad
    dev.finalize()
    assert dev.exact and dev.rows_pad == host.rows_pad
    assert torch.equal(dev.wg_k[: host.wg_k.numel()], host.wg_k)
    assert torch.equal(dev.pair_in[: host.rows_pad], host.pair_in) and torch.equal(dev.pair_out[: host.rows_pad], host.pair_out)
    assert torch.equal(dev.ext_start, host.ext_start) and torch.equal(dev.ext_list[: host.pairs], host.ext_list[: host.pairs])


def test_pair_plan_on_device_rejects_a_wrong_structural_claim():
    import link_amd as la
    from link_amd import _lib as L, elk
    n = 5000
    coords = s_uniform(n, grid=40, seed=6).cuda()
    st = la.SparseTensor(torch.randn(n, 16).cuda(), coords, 1)
    nbr, _ = elk.neighbor_table_of(st, (3, 3, 3))
    shuffled = nbr[torch.randperm(n, device=nbr.device)].contiguous()        # centre column no longer the identity
    plan = elk._PairPlan(shuffled, True)
    with pytest.raises(L.LinkAmdError):
        plan.finalize()


def test_a_bad_pair_plan_is_reported_once_and_does_not_poison_later_plans():
    """the verdict of a device-laid-out plan whose table was bad used to stay in the pending list and re-raise from
    every later plan -- of unrelated, valid tables -- while the bad neighbour tensor lived.  Now: reported once from the poll, the
    bad plan itself keeps raising whenever IT is used, plans of valid tables are unaffected."""
    import link_amd as la
    from link_amd import _lib as L, elk
    n = 5000
    coords = s_uniform(n, grid=40, seed=7).cuda()
    st = la.SparseTensor(torch.randn(n, 16).cuda(), coords, 1)
    nbr, _ = elk.neighbor_table_of(st, (3, 3, 3))
    shuffled = nbr[torch.randperm(n, device=nbr.device)].contiguous()
    bad = elk._PairPlan(shuffled, True)                  # a wrong structural claim; nobody asks this plan anything
    assert not bad.exact
    torch.cuda.synchronize()                             # its counts have arrived
    with pytest.raises(L.LinkAmdError):
        elk._PairPlan(nbr, True)                         # the next device-laid-out plan polls: the verdict surfaces HERE, once
    good = elk._PairPlan(nbr, True)                      # ... and not again: the bad plan has left the pending list
    good2 = elk._PairPlan(nbr, True)
    torch.cuda.synchronize()
    assert good.finalize().exact and good2.density > 0
    with pytest.raises(L.LinkAmdError):                  # the bad plan itself stays bad
        bad.finalize()
    with pytest.raises(L.LinkAmdError):
        _ = bad.density
    assert all(r() is not bad for r in elk._PENDING_PLANS)


@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (16, 32), (32, 16)])
def test_resident_weights_kernel_exact_and_split(cin, cout):
    """Narrow layers (conv.hip: k_subm_conv_resident, every W_k in LDS): the plain forward on the exact f32 instruction and the
    fused inference entry on the fp16-split products, against a float64 restatement -- LayerNorm and folded-affine
    epilogues with addend + ReLU, a table with fewer taps than 27, rows beyond the fp16 range (exact fallback), and the
    pair-list form on the same inputs."""
    import link_amd as la
    from link_amd import elk
    torch.manual_seed(cin + cout)
    coords = torch.from_numpy(lidar_like(30000, seed=9, stride=1))
    n = coords.shape[0]
    feats = torch.randn(n, cin, generator=torch.Generator().manual_seed(1))
    feats[::113] *= 3e5                                 # beyond 2^15: those tiles take the exact instruction
    st = la.SparseTensor(feats.cuda(), coords.cuda(), 1)
    nbr, _ = elk.neighbor_table_of(st, (3, 3, 3))
    w = (torch.randn(27, cin, cout, generator=torch.Generator().manual_seed(2)) * 0.2).cuda()
    assert elk._resident_form(cin, cout, 27, False, "auto")

    def ref64(table, wk):
        f64 = torch.cat([feats.double(), torch.zeros(1, cin, dtype=torch.float64)], 0)
        idx = table.cpu().long()
        idx = torch.where(idx >= 0, idx, torch.full_like(idx, n))
        return torch.einsum("nkc,kcd->nd", f64[idx], wk.cpu().double())

    ref = ref64(nbr, w)
    plain = elk.subm_conv(st.F, w, nbr, None)           # exact f32 products
    assert rel_err(plain.cpu().numpy(), ref.numpy()) < 2e-6
    pairs = elk.subm_conv(st.F, w, nbr, None, form="pairs")
    assert rel_err(pairs.cpu().numpy(), ref.numpy()) < 2e-6
    # fused inference entry (fp16-split products): folded affine + addend + ReLU ...
    sc, sh = (torch.rand(cout) + 0.5).cuda(), torch.randn(cout).cuda()
    add = torch.randn(n, cout).cuda()
    got = elk.subm_conv_ln_add_relu(st.F, w, nbr, None, sc, sh, 0.0, add, relu=True, affine=True)
    want = torch.relu(add.cpu().double() + ref * sc.cpu().double() + sh.cpu().double())
    assert rel_err(got.cpu().numpy(), want.numpy()) < 5e-6
    # ... and LayerNorm (the LinK block's tail)
    got = elk.subm_conv_ln_add_relu(st.F, w, nbr, None, sc, sh, 1e-6, add, relu=False)
    want = add.cpu().double() + torch.nn.functional.layer_norm(ref, (cout,), sc.cpu().double(), sh.cpu().double(), 1e-6)
    assert rel_err(got.cpu().numpy(), want.numpy()) < 2e-5
    # a table with 3 taps (the backbone's extra_conv shape)
    t3 = nbr[:, [4, 13, 22]].contiguous()
    got3 = elk.subm_conv(st.F, w[:3].contiguous(), t3, None)
    assert rel_err(got3.cpu().numpy(), ref64(t3, w[:3]).numpy()) < 2e-6
    again = elk.subm_conv_ln_add_relu(st.F, w, nbr, None, sc, sh, 1e-6, add, relu=False)
    assert torch.equal(again, got)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cin,cout", [(16, 16), (32, 32), (16, 32)])
def test_resident_weights_kernel_half_rows(cin, cout, dtype):
    """AMP form (16-bit rows and weights, fp32 accumulation): against the float64 result on the SAME rounded rows and
    weights (accumulation error only), and against the pair-list AMP form (which rounds every offset's product first)."""
    import link_amd as la
    from link_amd import elk
    torch.manual_seed(cin * 3 + cout)
    coords = torch.from_numpy(lidar_like(20000, seed=10, stride=1))
    n = coords.shape[0]
    feats = torch.randn(n, cin, generator=torch.Generator().manual_seed(1)).to(dtype)
    st = la.SparseTensor(feats.cuda(), coords.cuda(), 1)
    nbr, _ = elk.neighbor_table_of(st, (3, 3, 3))
    w = (torch.randn(27, cin, cout, generator=torch.Generator().manual_seed(2)) * 0.2).cuda()
    wr = w.to(dtype).double().cpu()                    # the AMP contract rounds the kernel with the features
    f64 = torch.cat([feats.double(), torch.zeros(1, cin, dtype=torch.float64)], 0)
    idx = nbr.cpu().long()
    idx = torch.where(idx >= 0, idx, torch.full_like(idx, n))
    ref = torch.einsum("nkc,kcd->nd", f64[idx], wr)
    ulp = 2.0 ** -10 if dtype == torch.float16 else 2.0 ** -7
    got = elk.subm_conv(st.F, w, nbr, None)
    assert got.dtype == dtype and rel_err(got.float().cpu().numpy(), ref.numpy()) < ulp
    sc, sh = (torch.rand(cout) + 0.5).cuda(), torch.randn(cout).cuda()
    add = torch.randn(n, cout).to(dtype).cuda()
    got = elk.subm_conv_ln_add_relu(st.F, w, nbr, None, sc, sh, 0.0, add, relu=True, affine=True)
    want = torch.relu(add.double().cpu() + ref * sc.cpu().double() + sh.cpu().double())
    assert got.dtype == dtype and rel_err(got.float().cpu().numpy(), want.numpy()) < ulp
    pair = elk.subm_conv_ln_add_relu(st.F, w, nbr, None, sc, sh, 0.0, add, relu=True, affine=True, form="pairs")
    assert rel_err(got.float().cpu().numpy(), pair.float().cpu().numpy()) < 4 * ulp
    got_ln = elk.subm_conv_ln_add_relu(st.F, w, nbr, None, sc, sh, 1e-6, add, relu=False)
    want = add.double().cpu() + torch.nn.functional.layer_norm(ref, (cout,), sc.cpu().double(), sh.cpu().double(), 1e-6)
    assert rel_err(got_ln.float().cpu().numpy(), want.numpy()) < 2 * ulp
