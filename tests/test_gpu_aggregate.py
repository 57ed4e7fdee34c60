"""GPU parity of voxel_to_aux / aux_to_voxel (the reference's L2 surface) vs golden fixtures and the
oracle, forward and backward.  Tolerance: BASELINE.json's 1e-4 rel on fp32 features (we assert much
tighter); indices bit-exact."""
import numpy as np
import pytest
import torch

from helpers import golden_files, load_golden, rel_err, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("name", golden_files("g_agg_*.npz"))
def test_surface_vs_golden(name):
    import link_amd as la
    g = load_golden(name)
    s, r = g["meta"]["s"], g["meta"]["r"]
    large = la.SparseTensor(dev(g["feats"]), dev(g["coords"]), 1)
    large.cmaps["marker"] = 1
    small, idx, counts = la.voxel_to_aux(large, s)
    assert small.cmaps is large.cmaps and small.kmaps is large.kmaps           # utils.py:55-56
    assert small.s == (s, s, s)
    assert idx.dtype == torch.int64 and counts.dtype == torch.int32 and small.C.dtype == torch.int32
    assert np.array_equal(small.C.cpu().numpy(), g["small_c"])
    assert np.array_equal(idx.cpu().numpy(), g["idx_query"])
    assert np.array_equal(counts.cpu().numpy(), g["counts"])
    assert rel_err(small.F.cpu().numpy(), g["aux_f"]) < 1e-6
    out = (la.aux_to_voxel(small, large, idx, counts, r) if r != 3 or "b2" in name
           else la.small_to_large_v2(small, large, idx, counts))
    assert out is large                                                         # utils.py:82-84 in place
    assert rel_err(large.F.cpu().numpy(), g["out"]) < 1e-5
    if r == 3:        # round 6: the same call through the reference's COMPILED devoxelize_forward_cpu (8-wide slices, make_golden.py)
        assert "reference compiled ops" in g["meta"]["devoxelize"]
        assert rel_err(large.F.cpu().numpy(), g["out_refcpu"]) < 1e-5


def test_block_mean_bit_exact_vs_oracle():
    """Same operation order as voxelize_cpu.cpp (divide, then add in ascending voxel id) -> identical."""
    import link_amd as la
    g = load_golden("g_agg_a_s3_r2_w16.npz")
    large = la.SparseTensor(dev(g["feats"]), dev(g["coords"]), 1)
    small, _, _ = la.voxel_to_aux(large, 3)
    assert np.array_equal(small.F.cpu().numpy(), g["aux_f"])


@pytest.mark.parametrize("s,r,W", [(3, 2, 24), (7, 3, 16), (2, 3, 129), (5, 2, 256)])
def test_surface_grads_vs_oracle(s, r, W):
    import link_amd as la
    coords = s_uniform(3000, grid=40, seed=s + r)
    rng = np.random.default_rng(W)
    x = rng.standard_normal((3000, W)).astype(np.float32)
    gout = rng.standard_normal((3000, W)).astype(np.float32)
    xt = torch.from_numpy(x).requires_grad_(True)
    ref = O.aggregate_torch(xt, coords, s, r)
    ref.backward(torch.from_numpy(gout))
    xd = dev(x).requires_grad_(True)
    large = la.SparseTensor(xd, coords.cuda(), 1)
    small, idx, counts = la.voxel_to_aux(large, s)
    la.aux_to_voxel(small, large, idx, counts, r)
    assert rel_err(large.F.detach().cpu().numpy(), ref.detach().numpy()) < 1e-5
    large.F.backward(dev(gout))
    assert rel_err(xd.grad.cpu().numpy(), xt.grad.numpy()) < 1e-5


def test_foreign_inputs_and_generic_path(monkeypatch):
    """aux_to_voxel with a small_x that did not come from voxel_to_aux, and the beyond-dense-grid
    generic path (reference algorithm on the HIP op kernels), must give the same answers."""
    import link_amd as la
    import link_amd.index as li
    g = load_golden("g_agg_neg_s3_r3_w8.npz")
    large = la.SparseTensor(dev(g["feats"]), dev(g["coords"]), 1)
    foreign = la.SparseTensor(dev(g["aux_f"]), dev(g["small_c"]), 3)
    la.aux_to_voxel(foreign, large, dev(g["idx_query"]), dev(g["counts"]), 3)
    assert rel_err(large.F.cpu().numpy(), g["out"]) < 1e-5
    monkeypatch.setattr(li, "MAX_CELLS", 8)
    large = la.SparseTensor(dev(g["feats"]).requires_grad_(True), dev(g["coords"]), 1)
    small, idx, counts = la.voxel_to_aux(large, 3)
    assert small._link_index is None
    assert np.array_equal(small.C.cpu().numpy(), g["small_c"])
    assert np.array_equal(idx.cpu().numpy(), g["idx_query"])
    assert np.array_equal(counts.cpu().numpy(), g["counts"])
    la.aux_to_voxel(small, large, idx, counts, 3)
    assert rel_err(large.F.detach().cpu().numpy(), g["out"]) < 1e-5
    large.F.sum().backward()      # generic path is differentiable too


def test_full_size_properties():
    """BASELINE cfg2 size (N=100k, W=128, s=7, r=3): size-independent properties."""
    import link_amd as la
    coords = s_uniform(100_000).cuda()
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(100_000, 128, generator=gen).cuda()

    def agg(t):
        large = la.SparseTensor(t.clone(), coords, 1)
        small, idx, counts = la.voxel_to_aux(large, 7)
        return la.aux_to_voxel(small, large, idx, counts, 3).F, small, idx, counts
    y, small, idx, counts = agg(x)
    assert small.C.shape[0] == 43334 and int(counts.sum()) == 100_000
    # constant field is a fixed point of a mean
    one, *_ = agg(torch.ones_like(x))
    assert torch.allclose(one, torch.ones_like(one), rtol=0, atol=2e-6)
    # linearity
    z, *_ = agg(2.5 * x + 1.0)
    assert rel_err((2.5 * y + 1.0).cpu().numpy(), z.cpu().numpy()) < 1e-5
    # voxels of one block receive identical rows
    first = torch.zeros(43334, dtype=torch.long, device="cuda").scatter_(0, idx, torch.arange(100_000, device="cuda"))
    assert torch.equal(y, y[first[idx]])
    # determinism: bitwise identical on re-run
    y2, *_ = agg(x)
    assert torch.equal(y, y2)
    # mean over ALL voxels weighted by neighbourhood size is conserved: sum_i y_i*D_i/n_i ... cheap check
    # against the scalar C oracle on a channel slice
    ref = O.aggregate(x[:, :4].cpu().numpy(), coords.cpu().numpy(), 7, 3)
    assert rel_err(y[:, :4].cpu().numpy(), ref) < 1e-5


def test_upsample_voxel():
    """utils.py:327-340 restated with numpy: parent lookup by coordinate, -1 -> last row."""
    import link_amd as la
    rng = np.random.default_rng(0)
    fine = np.unique(rng.integers(0, 40, (3000, 3)), axis=0).astype(np.int32)
    fine = np.concatenate([fine, np.zeros((fine.shape[0], 1), np.int32)], 1)
    stride = 4
    parents = np.unique(fine[:, :3] // stride, axis=0)
    keep = parents[rng.random(parents.shape[0]) < 0.8]                 # some parents absent
    coarse = np.concatenate([keep * stride, np.zeros((keep.shape[0], 1), np.int32)], 1).astype(np.int32)
    cf = rng.standard_normal((coarse.shape[0], 12)).astype(np.float32)
    x = la.SparseTensor(dev(cf), dev(coarse), stride)
    ref_x = la.SparseTensor(torch.zeros(fine.shape[0], 1, device="cuda"), dev(fine), 1)
    out = la.upsample_voxel(x, ref_x)
    lut = {tuple(r): i for i, r in enumerate(keep)}
    idx = np.array([lut.get(tuple(r), -1) for r in fine[:, :3] // stride])
    assert (idx == -1).any() and (idx >= 0).any()
    assert np.array_equal(out.F.cpu().numpy(), cf[idx])
    assert out.s == (1, 1, 1) and out.C is ref_x.C
