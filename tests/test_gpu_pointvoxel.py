"""Row N4: link_amd.initial_voxelize / point_to_voxel / voxel_to_point on the HIP
ops against the reference's own outputs (tests/golden/g_pointvoxel_*.npz) -- integer results bit-exact,
features within 1e-5 -- and the same reference functions UNMODIFIED-in-spirit through module aliasing."""
import numpy as np
import pytest
import torch

from helpers import golden_files, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", golden_files("g_pointvoxel_*.npz"))
def test_pointvoxel_vs_reference(name):
    import link_amd as la
    g = load_golden(name)
    m = g["meta"]
    z = la.PointTensor(torch.from_numpy(g["feats"]).cuda(), torch.from_numpy(g["points"]).cuda())
    st = la.initial_voxelize(z, m["init_res"], m["after_res"])
    assert np.array_equal(st.C.cpu().numpy(), g["vox_C"])
    assert np.array_equal(z.additional_features["idx_query"][1].cpu().numpy(), g["idx_query"])
    assert np.array_equal(z.additional_features["counts"][1].cpu().numpy(), g["counts"])
    assert np.array_equal(z.C.cpu().numpy(), g["z_C"])
    assert rel_err(st.F.cpu().numpy(), g["vox_F"]) < 1e-5
    assert st.cmaps[st.stride] is st.coords

    z.F = torch.from_numpy(g["p2v_feats_in"]).cuda()
    v = la.point_to_voxel(st, z)
    assert rel_err(v.F.cpu().numpy(), g["p2v_F"]) < 1e-5
    assert v.C is st.C and v.cmaps is st.cmaps and st.s in z.additional_features["idx_query"]

    if "v2p1_F" not in g:
        return
    x1 = la.SparseTensor(torch.from_numpy(g["v2p1_F_in"]).cuda(), st.C, 1)
    p1 = la.voxel_to_point(x1, z)
    assert np.array_equal(z.idx_query[x1.s].cpu().numpy(), g["v2p1_idx"])
    assert rel_err(z.weights[x1.s].cpu().numpy(), g["v2p1_w"]) < 1e-5
    assert rel_err(p1.F.cpu().numpy(), g["v2p1_F"]) < 1e-5
    x2 = la.SparseTensor(torch.from_numpy(g["v2p2_F_in"]).cuda(), torch.from_numpy(g["v2p2_C"]).cuda(), 2)
    p2 = la.voxel_to_point(x2, z)
    assert np.array_equal(z.idx_query[x2.s].cpu().numpy(), g["v2p2_idx"])
    assert rel_err(p2.F.cpu().numpy(), g["v2p2_F"]) < 1e-5
    # cached branch: a second call reuses idx/weights and gives the same features
    assert torch.equal(la.voxel_to_point(x2, z).F, p2.F)
    zn = la.PointTensor(z.F, z.C)
    pn = la.voxel_to_point(la.SparseTensor(x1.F, st.C, 1), zn, nearest=True)
    assert rel_err(pn.F.cpu().numpy(), g["v2p1_nearest_F"]) < 1e-5


def test_voxel_to_point_gradient():
    """Trilinear devoxelisation is differentiable wrt the voxel features (spdevoxelize backward kernel)."""
    import link_amd as la
    g = load_golden("g_pointvoxel_a.npz")
    z = la.PointTensor(torch.from_numpy(g["feats"]).cuda(), torch.from_numpy(g["points"]).cuda())
    st = la.initial_voxelize(z, 1, 1)
    f = torch.from_numpy(g["v2p1_F_in"]).cuda().requires_grad_(True)
    p = la.voxel_to_point(la.SparseTensor(f, st.C, 1), z)
    go = torch.randn_like(p.F)
    p.F.backward(go)
    idx, w = z.idx_query[(1, 1, 1)].long(), z.weights[(1, 1, 1)]
    ref = torch.zeros_like(f).index_add_(0, idx.clamp(min=0).reshape(-1),
                                         ((w * (idx >= 0))[..., None] * go[:, None, :]).reshape(-1, f.shape[1]))
    assert rel_err(f.grad.cpu().numpy(), ref.cpu().numpy()) < 1e-5


def test_dense_grid_path_against_the_hash_path(monkeypatch):
    """The native forms (one dense-grid index over the points; dense cell table look-ups) against the reference algorithm on
    the op kernels, which point sets beyond the dense-grid limit still take (pointvoxel.py: GridTooLarge): same voxel set in
    the same (hash) order, same point -> voxel maps and counts bit for bit, same corner tables; features to rounding; the
    gradient of the voxel means reaches the point features through the indexed kernel.  Negative coordinates, 2 batch items,
    a full-size frame."""
    import link_amd as la
    from link_amd import pointvoxel as PV
    from link_amd.index import GridTooLarge
    g = torch.Generator().manual_seed(11)
    P = 120000
    pts = torch.cat([(torch.rand(P, 3, generator=g) - 0.4) * torch.tensor([90.0, 70.0, 12.0]), torch.randint(0, 2, (P, 1), generator=g).float()], 1).cuda()
    feats = torch.randn(P, 9, generator=g).cuda()

    def run(force_hash):
        with monkeypatch.context() as mp:
            if force_hash:
                def boom(*a, **k):
                    raise GridTooLarge("forced")
                mp.setattr(PV, "BlockIndex", boom)
                mp.setattr(PV, "foreign_neighbor_map", boom)
            f = feats.clone().requires_grad_(True)
            z = la.PointTensor(f, pts.clone())
            st = la.initial_voxelize(z, 1.0, 0.5)
            st.F.square().sum().backward()
            v = la.point_to_voxel(st, z)
            x2 = la.SparseTensor(torch.ones(st.C.shape[0], 4, device="cuda"), st.C, 1)
            p = la.voxel_to_point(x2, z)
            return st, z, v, p, f.grad
    a, b = run(False), run(True)
    assert torch.equal(a[0].C, b[0].C) and a[0].C.shape[0] > 50000
    for key in ("idx_query", "counts"):
        assert torch.equal(a[1].additional_features[key][1], b[1].additional_features[key][1]), key
    assert rel_err(a[0].F.detach().cpu().numpy(), b[0].F.detach().cpu().numpy()) < 1e-5
    assert rel_err(a[4].cpu().numpy(), b[4].cpu().numpy()) < 1e-5
    assert torch.equal(a[1].idx_query[(1, 1, 1)], b[1].idx_query[(1, 1, 1)])
    assert rel_err(a[2].F.detach().cpu().numpy(), b[2].F.detach().cpu().numpy()) < 1e-5
    assert rel_err(a[3].F.detach().cpu().numpy(), b[3].F.detach().cpu().numpy()) < 1e-5


@pytest.mark.parametrize("scale", [1, 2, 4])
def test_trilinear_weights_kernel_against_the_composition(scale):
    """link_ti_weights (one kernel) against the elementwise composition of devoxelize.py:10-48 (run on the CPU copies):
    absent corners, negative coordinates, the scale-dependent cell origin."""
    import link_amd as la
    g = torch.Generator().manual_seed(scale)
    P = 20000
    pts = torch.cat([(torch.rand(P, 3, generator=g) - 0.5) * 60.0, torch.zeros(P, 1)], 1)
    idx = torch.randint(-1, 500, (8, P), generator=g)
    idx[:, :5] = -1                                        # points with no corner at all: 0 / 1e-8
    ref = la.calc_ti_weights(pts, idx, scale=scale)
    got = la.calc_ti_weights(pts.cuda(), idx.cuda(), scale=scale)
    assert got.is_cuda and got.shape == (8, P)
    assert float((got.cpu() - ref).abs().max()) < 2e-6
