"""GPU parity of the torchsparse.backend drop-ins (include/link_amd.h section A) vs the oracle and
the reference-generated golden fixtures.  Integer ops bit-exact; fp32 within 1e-4 rel (tighter here)."""
import numpy as np
import pytest
import torch

from helpers import load_golden, rel_err
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_hash_golden_bit_exact():
    import link_amd as la
    g = load_golden("g_hash.npz")
    assert np.array_equal(la.sphash(dev(g["kat_coords"])).cpu().numpy(), g["kat_hash"])
    assert np.array_equal(la.sphash(dev(g["rnd_coords"])).cpu().numpy(), g["rnd_hash"])
    for r, key in ((3, "khash_r3"), (2, "khash_r2")):
        off = la.get_kernel_offsets(r, device="cuda")
        out = la.sphash(dev(g["single_coords"]), off)
        assert out.shape == (r ** 3, 4096) and out.dtype == torch.int64
        assert np.array_equal(out.cpu().numpy(), g[key])
    # batch > 0: CUDA semantics (own batch index), i.e. the oracle without the CPU defect
    out = la.sphash(dev(g["rnd_coords"]), la.get_kernel_offsets(3, device="cuda")).cpu().numpy()
    assert np.array_equal(out, O.sphash_offsets(g["rnd_coords"], O.get_kernel_offsets(3)))


def test_hash_edge_cases():
    import link_amd as la
    e = torch.empty((0, 4), dtype=torch.int32, device="cuda")
    assert la.sphash(e).shape == (0,)
    assert la.sphash(e, la.get_kernel_offsets(3, device="cuda")).shape == (27, 0)
    ext = np.array([[2 ** 31 - 1, -2 ** 31, 0, 5], [-1, -1, -1, -1]], np.int32)
    assert np.array_equal(la.sphash(dev(ext)).cpu().numpy(), O.sphash(ext))
    with pytest.raises(AssertionError):
        la.sphash(torch.zeros((3, 4), dtype=torch.int64, device="cuda"))
    with pytest.raises(Exception):
        la.sphash(torch.zeros((3, 4), dtype=torch.int32))     # CPU tensor: no fallback


@pytest.mark.parametrize("n,n1", [(0, 5), (1, 1), (1000, 3000), (50000, 200000)])
def test_hash_query(n, n1):
    import link_amd as la
    rng = np.random.default_rng(n + n1)
    ref = rng.integers(-2 ** 40, 2 ** 40, n).astype(np.int64)
    if n > 10:
        ref[n // 2:] = ref[: n - n // 2]            # lots of duplicates: first one must win
        ref[3] = 0                                   # the reference's reserved key is an ordinary key here
    q = np.concatenate([rng.choice(ref, n1 // 2) if n else np.zeros(0, np.int64),
                        rng.integers(-2 ** 40, 2 ** 40, n1 - n1 // 2)]).astype(np.int64)
    out = la.sphashquery(dev(q).view(-1, 1), dev(ref))
    assert out.shape == (q.shape[0], 1)
    assert np.array_equal(out.cpu().numpy().reshape(-1), O.sphashquery(q, ref))


def test_hash_query_golden():
    import link_amd as la
    g = load_golden("g_hash.npz")
    out = la.sphashquery(dev(g["query_q"]), dev(g["query_ref"])).cpu().numpy()
    assert out.tolist() == [0, 3, -1, -1] == g["query_out"].tolist()


def test_count():
    import link_amd as la
    g = load_golden("g_hash.npz")
    assert np.array_equal(la.spcount(dev(g["count_idx"]), 7).cpu().numpy(), g["count_out"])
    rng = np.random.default_rng(1)
    idx = rng.integers(-3, 1000, 100000).astype(np.int32)
    assert np.array_equal(la.spcount(dev(idx), 1000).cpu().numpy(), O.spcount(idx, 1000))


@pytest.mark.parametrize("n,c,n1", [(5000, 24, 300), (3000, 129, 700), (10, 1, 3), (20000, 256, 5000)])
def test_voxelize(n, c, n1):
    import link_amd as la
    rng = np.random.default_rng(c)
    idx = rng.integers(0, n1, n).astype(np.int32)
    counts = O.spcount(idx, n1)
    feats = rng.standard_normal((n, c)).astype(np.float32)
    f = dev(feats).requires_grad_(True)
    out = la.spvoxelize(f, dev(idx).long(), dev(counts))
    assert rel_err(out.detach().cpu().numpy(), O.spvoxelize_fwd(feats, idx, counts)) < 1e-5
    top = rng.standard_normal((n1, c)).astype(np.float32)
    out.backward(dev(top))
    assert np.array_equal(f.grad.cpu().numpy(), O.spvoxelize_bwd(top, idx, counts, n))


@pytest.mark.parametrize("nq,n,c,r", [(700, 300, 24, 2), (2000, 900, 129, 3), (100, 50, 65, 5), (5, 5, 1, 1)])
def test_devoxelize(nq, n, c, r):
    import link_amd as la
    rng = np.random.default_rng(nq)
    K = r ** 3
    ind = rng.integers(-1, n, (nq, K)).astype(np.int32)
    w = rng.random((nq, K)).astype(np.float32)
    feat = rng.standard_normal((n, c)).astype(np.float32)
    f = dev(feat).requires_grad_(True)
    out = la.spdevoxelize(f, dev(ind), dev(w), r)
    ref = O.spdevoxelize_fwd(feat, ind, w)
    assert rel_err(out.detach().cpu().numpy(), ref) < 1e-6
    top = rng.standard_normal((nq, c)).astype(np.float32)
    out.backward(dev(top))
    assert rel_err(f.grad.cpu().numpy(), O.spdevoxelize_bwd(top, ind, w, n)) < 1e-5


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16, torch.float64])
def test_backend_dtype_parity(dt):
    """The backend accepts the dtypes the reference dispatches and returns the caller's dtype
    (accumulation is always fp32)."""
    from link_amd import backend as B
    rng = np.random.default_rng(3)
    n, c, n1 = 2000, 32, 200
    idx = rng.integers(0, n1, n).astype(np.int32)
    counts = O.spcount(idx, n1)
    feats = rng.standard_normal((n, c)).astype(np.float32)
    out = B.voxelize_forward_cuda(dev(feats).to(dt), dev(idx), dev(counts))
    assert out.dtype == dt
    ref = O.spvoxelize_fwd(dev(feats).to(dt).float().cpu().numpy(), idx, counts)
    tol = {torch.float16: 2e-3, torch.bfloat16: 2e-2, torch.float64: 1e-6}[dt]
    assert rel_err(out.float().cpu().numpy(), ref) < tol
    ind = rng.integers(-1, n1, (300, 8)).astype(np.int32)
    w = rng.random((300, 8)).astype(np.float32)
    dv = B.devoxelize_forward_cuda(out, dev(ind), dev(w).to(dt), 2)
    assert dv.dtype == dt and dv.shape == (300, c)
