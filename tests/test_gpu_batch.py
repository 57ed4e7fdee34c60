"""GPU parity of the batch entry point (include/link_amd.h section H: link_elk_core_dense_forward_batch, csrc/dense_batch.hip):
R_core of a batch of independent frames as one slot-insert grid + two persistent, queue-fed kernels.  Everything here goes through
the C ABI (ElkCoreBatch -> ctypes) and is compared BIT FOR BIT with the per-frame path (ElkCorePlan.run), which the other test files
pin on the oracle and the reference fixtures (linkunet.py:132,151-162,178 + utils.py:44-84 semantics; BASELINE.json configs[3])."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests.helpers import rel_err, s_uniform  # noqa: E402

pytestmark = pytest.mark.gpu
BOUNDS = ((0, 0, 0, 0), (255, 255, 255, 0))


def _block(c, baseop, dev):
    import link_amd as la
    torch.manual_seed(2)
    return la.ELKBlock(c, c, groups=2, baseop=baseop).to(dev).eval()


def _bind(obj, blk):
    obj.bind(blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None, blk.norm.weight,
             blk.norm.bias)
    return obj


def _frames(k, n, c, dev, seed0=0, ragged=False):
    out = []
    for i in range(k):
        ni = n - (137 * i if ragged else 0)
        g = torch.Generator().manual_seed(100 + seed0 + i)
        out.append((torch.randn(ni, c, generator=g).to(dev), s_uniform(ni, seed=seed0 + i).to(dev)))
    return out


@pytest.mark.parametrize("baseop,r,s", [("cos", 3, 7), ("sin", 3, 7), ("cos", 2, 7), ("cos", 3, 5)])
def test_batch_equals_per_frame_plans_bitwise(baseop, r, s):
    """8 frames of 20 k voxels (ragged sizes) in one call == ElkCorePlan.run per frame, bit for bit; status words clean; a second
    call on the same arenas (buffers reused: the call is ordered behind the first) reproduces it."""
    import link_amd as la
    dev = torch.device("cuda:0")
    C, N, K = 64, 20000, 8
    blk = _block(C, baseop, dev)
    frames = _frames(K, N, C, dev, ragged=True)
    # (k1_form = 0: the cell-range form of the pre_mix kernel, whose body the batch's K1 role runs -- the matrix-core sums form a
    # single plan picks for one frame alone adds a cell's voxels in another association: 2e-7 apart on sparse frames)
    plan = _bind(la.ElkCorePlan(N, C, baseop, C // 2, r, s, BOUNDS, dev, layout="dense", k1_form=0), blk)
    ref = [plan.run(f, co).clone() for f, co in frames]
    batch = _bind(la.ElkCoreBatch(K, N, C, baseop, C // 2, r, s, BOUNDS, dev), blk)
    for _ in range(2):
        outs = [o.clone() for o in batch.run([f for f, _ in frames], [co for _, co in frames])]
        batch.check()
        for i in range(K):
            assert outs[i].shape == ref[i].shape
            assert torch.equal(outs[i], ref[i]), (i, rel_err(outs[i].cpu().numpy(), ref[i].cpu().numpy()))


def test_batch_full_size_cfg2_vs_oracle_and_plans():
    """cfg2-sized frames (100 k voxels, C = 64, cos, r = 3, s = 7): 6 frames in one call against the oracle (1e-4 rel, north_star) on
    frame 0 and bit for bit against the per-frame plans on all; fewer frames than arenas; caller-provided result tensors."""
    import numpy as np
    import link_amd as la
    from oracle import link_oracle as O
    dev = torch.device("cuda:0")
    C, N, K = 64, 100000, 6
    blk = _block(C, "cos", dev)
    frames = _frames(K, N, C, dev)
    plan = _bind(la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, BOUNDS, dev, layout="dense", k1_form=0), blk)
    ref = [plan.run(f, co).clone() for f, co in frames]
    batch = _bind(la.ElkCoreBatch(8, N, C, "cos", C // 2, 3, 7, BOUNDS, dev), blk)
    dst = [torch.full((N, C), float("nan"), device=dev) for _ in range(K)]
    outs = batch.run([f for f, _ in frames], [co for _, co in frames], outs=dst)
    batch.check()
    for i in range(K):
        assert outs[i].data_ptr() == dst[i].data_ptr()
        assert torch.equal(outs[i], ref[i]), i
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    core = O.elk_core_torch(frames[0][0].cpu(), frames[0][1].cpu(), params, 7, 3, "cos", 2, agg=O.aggregate_c)
    assert rel_err(outs[0].cpu().numpy(), core.numpy()) < 1e-4
    assert np.isfinite(outs[K - 1].cpu().numpy()).all()


def test_two_batches_in_flight_share_one_context():
    """Two ElkCoreBatch objects on ONE context, alternated on two streams (the bench's arrangement): calls on disjoint arenas overlap
    on the device, calls on the same arenas are ordered; 6 rounds, every result bit-equal to the per-frame path."""
    import link_amd as la
    dev = torch.device("cuda:0")
    C, N, K = 64, 30000, 4
    blk = _block(C, "cos", dev)
    plan = _bind(la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, BOUNDS, dev, layout="dense", k1_form=0), blk)
    sets = [_frames(K, N, C, dev, seed0=10 * j) for j in range(2)]
    refs = [[plan.run(f, co).clone() for f, co in fs] for fs in sets]
    b0 = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, BOUNDS, dev), blk)
    b1 = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, BOUNDS, dev, share=b0), blk)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for rnd in range(6):
        j = rnd & 1
        with torch.cuda.stream(streams[j]):
            outs = (b0, b1)[j].run([f for f, _ in sets[j]], [co for _, co in sets[j]])
            got = [o.clone() for o in outs]              # stream-ordered behind the batch
        streams[j].synchronize()
        for i in range(K):
            assert torch.equal(got[i], refs[j][i]), (rnd, i)
    b0.check()
    b1.check()


def test_batch_reports_a_dropped_voxel_and_refuses_what_it_does_not_serve():
    import link_amd as la
    from link_amd import _lib as L
    dev = torch.device("cuda:0")
    C, N = 64, 5000
    blk = _block(C, "cos", dev)
    with pytest.raises(L.LinkAmdError):
        la.ElkCoreBatch(2, N, 32, "cos", 16, 3, 7, BOUNDS, dev)                  # C = 32: not this entry point
    with pytest.raises(L.LinkAmdError):
        la.ElkCoreBatch(2, N, C, "cos_x", 64, 3, 7, BOUNDS, dev)                # three-part rows: not this entry point
    batch = _bind(la.ElkCoreBatch(2, N, C, "cos", C // 2, 3, 7, BOUNDS, dev), blk)
    frames = _frames(2, N, C, dev)
    bad = frames[1][1].clone()
    bad[7, 0] = 300                                                               # outside the plan's bounds (0 .. 255)
    batch.run([frames[0][0], frames[1][0]], [frames[0][1], bad])
    with pytest.raises(L.LinkAmdError, match="outside the plan's bounds"):
        batch.check()


def test_batch_longer_than_one_launch_set():
    """More frames than one launch set holds (50 > 48: the frame table travels as a kernel argument): the call runs two launch sets
    back to back on the same context; every frame bit-equal to the per-frame path.  Small slot capacity keeps the arenas small."""
    import link_amd as la
    dev = torch.device("cuda:0")
    C, N, K = 64, 4000, 50
    blk = _block(C, "cos", dev)
    frames = [(torch.randn(N, C, generator=torch.Generator().manual_seed(150 + i)).to(dev), s_uniform(N, grid=64, seed=50 + i).to(dev))
              for i in range(K)]                                                 # unique coordinates inside 64^3
    bounds = ((0, 0, 0, 0), (63, 63, 63, 0))
    nmax = max(co.shape[0] for _, co in frames)
    plan = _bind(la.ElkCorePlan(nmax, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k1_form=0, slot_cap=64), blk)
    ref = [plan.run(f, co).clone() for f, co in frames]
    batch = _bind(la.ElkCoreBatch(K, nmax, C, "cos", C // 2, 3, 7, bounds, dev, slot_cap=64), blk)
    outs = batch.run([f for f, _ in frames], [co for _, co in frames])
    batch.check()
    assert all(torch.equal(outs[i], ref[i]) for i in range(K))


def test_submit_then_join_from_one_stream_keeps_two_calls_in_flight():
    """link_dc_batch_submit / link_dc_batch_join: call s + 1 submitted before call s is joined, all from ONE stream, two batch objects on
    one context; rows bit-equal to the per-frame path; a consumer kernel queued on the stream after join(s) sees call s complete; a
    ticket that does not exist is refused."""
    import link_amd as la
    from link_amd import _lib as L
    dev = torch.device("cuda:0")
    C, N, K = 64, 20000, 6
    blk = _block(C, "cos", dev)
    bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
    frames = _frames(2 * K, N, C, dev, seed0=70)
    plan = _bind(la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k1_form=0), blk)
    ref = [plan.run(f, co).clone() for f, co in frames]
    a = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, bounds, dev), blk)
    b = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, bounds, dev, share=a), blk)
    sets = [([f for f, _ in frames[:K]], [co for _, co in frames[:K]]), ([f for f, _ in frames[K:]], [co for _, co in frames[K:]])]
    copies, prev = [], None
    for s in range(6):                                                       # a, b, a, b, ... : s + 1 submitted before s is joined
        obj, (fs, cs) = (a, b)[s % 2], sets[s % 2]
        res, tk = obj.submit(fs, cs)
        if prev is not None:
            a.join(prev[1])
            copies.append((prev[2], [r.clone() for r in prev[0]]))           # clones are queued on the stream behind the join
        prev = (res, tk, s % 2)
    a.join(prev[1])
    copies.append((prev[2], [r.clone() for r in prev[0]]))
    torch.cuda.synchronize()
    a.check()
    for which, rows in copies:
        assert all(torch.equal(rows[i], ref[which * K + i]) for i in range(K))
    assert L.lib().link_dc_batch_join(a._ctx, 10 ** 6, L.current_stream_handle()) == L.LINK_ERR_ARG


def test_calibrate_keeps_a_working_context_and_the_rows():
    """ElkCoreBatch.calibrate: a few contexts measured on the batch's own arenas, the fastest kept (for both objects of a pair); the rows
    afterwards are still the per-frame path's, bit for bit; a shared (non-owning) batch refuses to calibrate."""
    import link_amd as la
    dev = torch.device("cuda:0")
    C, N, K = 64, 12000, 4
    blk = _block(C, "cos", dev)
    bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
    frames = _frames(K, N, C, dev, seed0=90)
    plan = _bind(la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k1_form=0), blk)
    ref = [plan.run(f, co).clone() for f, co in frames]
    a = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, bounds, dev), blk)
    b = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, bounds, dev, share=a), blk)
    fs, cs = [f for f, _ in frames], [co for _, co in frames]
    rates = a.calibrate(fs, cs, partner=b, tries=3, calls=6)
    assert len(rates) == 3 and all(r > 0 for r in rates) and b._ctx.value == a._ctx.value
    for obj in (a, b):
        outs = obj.run(fs, cs)
        torch.cuda.synchronize()
        obj.check()
        assert all(torch.equal(outs[i], ref[i]) for i in range(K))
    from link_amd import _lib as L
    with pytest.raises(L.LinkAmdError):
        b.calibrate(fs, cs)


def test_streams_on_own_queues_and_the_queue_test():
    """link_streams_share_queue: a stream against itself is one queue by definition (>= 150 us: the stamp kernel sits behind the spin
    kernel and its event record); streams_on_own_queues hands out streams that pass the test pair by pair."""
    import ctypes
    from link_amd import _lib as L
    from link_amd.parallel import streams_on_own_queues
    dev = torch.device("cuda:0")
    s = torch.cuda.Stream(device=dev)
    d = ctypes.c_double(0.0)
    assert L.lib().link_streams_share_queue(s.cuda_stream, s.cuda_stream, ctypes.byref(d)) == 0 and d.value >= 140.0
    kept, rec = streams_on_own_queues(3, dev)
    assert len(kept) == 3 and len({k.cuda_stream for k in kept}) == 3
    if rec["candidates_passed_over"] < 9:
        assert max(rec["max_delay_us_to_earlier_kept_stream"]) < 75.0
    for i in range(3):
        for j in range(i + 1, 3):
            assert L.lib().link_streams_share_queue(kept[i].cuda_stream, kept[j].cuda_stream, ctypes.byref(d)) == 0
            assert d.value < 75.0 or rec["candidates_passed_over"] >= 9


@pytest.mark.parametrize("K", [1, 48, 49, 193])
def test_batch_sizes_around_the_launch_set_and_the_ring(K):
    """One frame; exactly one launch set (48); one frame more; five sets in ONE call (193 frames: the ring of four sync areas wraps inside
    the call) -- every frame bit-equal to the per-frame path, and the same again on a second call (arenas and ring slots reused)."""
    import link_amd as la
    dev = torch.device("cuda:0")
    C, N = 64, 1500
    blk = _block(C, "cos", dev)
    bounds = ((0, 0, 0, 0), (63, 63, 63, 0))
    frames = [(torch.randn(N - 7 * (i % 5), C, generator=torch.Generator().manual_seed(300 + i)).to(dev),
               s_uniform(N - 7 * (i % 5), grid=64, seed=200 + i).to(dev)) for i in range(K)]
    plan = _bind(la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k1_form=0, slot_cap=64), blk)
    ref = [plan.run(f, co).clone() for f, co in frames]
    batch = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, bounds, dev, slot_cap=64), blk)
    for _ in range(2):
        outs = batch.run([f for f, _ in frames], [co for _, co in frames])
        torch.cuda.synchronize()
        batch.check()
        assert all(torch.equal(outs[i], ref[i]) for i in range(K))
        for o in outs:
            o.zero_()


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
def test_batch_half_rows_equal_the_per_frame_plan_bitwise(dt):
    """fp16 / bf16 rows at the kernel boundary (the reference's AMP contract; fp32 inside): the batch call's rows are those of
    ElkCorePlan.run on the same half rows, bit for bit, and within the half tolerance of the fp32 rows; mixed row types are refused."""
    import link_amd as la
    from link_amd import _lib as L
    dev = torch.device("cuda:0")
    C, N, K = 64, 20000, 5
    blk = _block(C, "cos", dev)
    bounds = ((0, 0, 0, 0), (255, 255, 255, 0))
    frames = _frames(K, N, C, dev, seed0=400, ragged=True)
    plan = _bind(la.ElkCorePlan(N, C, "cos", C // 2, 3, 7, bounds, dev, layout="dense", k1_form=0), blk)
    ref32 = [plan.run(f, co).clone() for f, co in frames]
    ref = [plan.run(f.to(dt), co).clone() for f, co in frames]
    batch = _bind(la.ElkCoreBatch(K, N, C, "cos", C // 2, 3, 7, bounds, dev), blk)
    outs = batch.run([f.to(dt) for f, _ in frames], [co for _, co in frames])
    torch.cuda.synchronize()
    batch.check()
    for i in range(K):
        assert outs[i].dtype == dt and torch.equal(outs[i], ref[i])
        assert rel_err(outs[i].float().cpu().numpy(), ref32[i].cpu().numpy()) < (4e-3 if dt == torch.float16 else 3e-2)
    with pytest.raises((L.LinkAmdError, AssertionError)):
        batch.run([frames[0][0].to(dt), frames[1][0]], [frames[0][1], frames[1][1]])

