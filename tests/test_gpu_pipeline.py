"""The three-frame step kernel (include/link_amd.h: link_elk_core_dense_step3; link_amd/csrc/dense_step3_impl.h) behind
ElkCorePipeline: one launch runs the slot insert, the fused pre_mix kernel and the gather + de-modulate kernel of three consecutive
frames.  Every frame's result must be that of a step on its own -- against the CPU oracle (reference call path
modules.py:180-191) and, bit for bit, against ElkCorePlan with the same kernels."""
import numpy as np
import pytest
import torch

from helpers import rel_err, s_uniform
from oracle import link_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _block(la, baseop, seed):
    torch.manual_seed(seed)
    blk = la.ELKBlock(64, 64, groups=2, baseop=baseop).cuda().eval()
    with torch.no_grad():
        for nme, p in blk.named_parameters():
            if "norm" in nme or "pre_mix.1" in nme:
                p.add_(0.2 * torch.randn_like(p))
    return blk


def _par(blk):
    return (blk.pre_mix[0].weight, blk.pre_mix[1].weight, blk.pre_mix[1].bias, blk.pos_weight[0].weight, None,
            blk.norm.weight, blk.norm.bias)


@pytest.mark.parametrize("baseop,r,s,extent", [("cos", 3, 7, 96), ("sin", 2, 5, 64), ("cos", 2, 7, 128)])
def test_pipeline_frames_vs_oracle_and_single_frame_plan(baseop, r, s, extent):
    """Seven frames of different sizes through one pipeline (fill, steady state, drain): each against the oracle within 1e-4 of
    the row scale, and bitwise against a step of ElkCorePlan(k1_form=2) with the same launch geometry on the same frame."""
    import link_amd as la
    blk = _block(la, baseop, 5 + r)
    params = {k: v.detach().cpu() for k, v in blk.state_dict().items()}
    bounds = ((0, 0, 0, 0), (extent - 1, extent - 1, extent - 1, 0))
    # (sin: a voxel alone in its neighbourhood gives sin(theta - theta) = 0 before the LayerNorm -- rounding noise normalised
    # to unit scale, in the reference too -- so the near-empty frames go to the cos cases only)
    sizes = [20000, 7000, 20000, 1, 13001, 64, 19999] if baseop == "cos" else [20000, 7000, 20000, 13001, 9000, 19999]
    frames = [(torch.randn(n, 64, generator=torch.Generator().manual_seed(40 + i)), s_uniform(n, grid=extent, seed=50 + i))
              for i, n in enumerate(sizes)]
    n_cap = max(sizes)
    # the same launch geometry on both sides: where a wave's range of cells starts decides where a crowded cell is cut into
    # tiles, i.e. the order its rows are summed in
    geo = dict(k1_wgs=512, k2_zsplit=2)
    pipe = la.ElkCorePipeline(n_cap, 64, baseop, 32, r, s, bounds, "cuda", **geo).bind(*_par(blk))
    plan = la.ElkCorePlan(n_cap, 64, baseop, 32, r, s, bounds, "cuda", layout="dense", k1_form=2, **geo).bind(*_par(blk))
    dev_frames = [(f.cuda(), c.cuda()) for f, c in frames]
    got = []
    for i, (f, c) in enumerate(dev_frames):
        res = pipe.push(f, c)
        assert (res is None) == (i < 2)
        if res is not None:
            got.append(res.clone())
    got += [t.clone() for t in pipe.flush()]
    assert pipe.flush() == []
    pipe.check()
    assert [g.shape[0] for g in got] == sizes
    for i, ((f, c), (fd, cd), g) in enumerate(zip(frames, dev_frames, got)):
        one = plan.run(fd, cd)
        assert torch.equal(g, one), (i, float((g - one).abs().max()), int(((g - one).abs().amax(1) > 0).sum()))
        ref = O.elk_core_torch(f, c, params, s, r, baseop, 2, agg=O.aggregate_c).numpy()
        assert rel_err(g.cpu().numpy(), ref) < TOL, (i, g.shape[0], rel_err(g.cpu().numpy(), ref))


def test_pipeline_half_rows_and_caller_buffers():
    """fp16 feature rows in, fp16 rows out into buffers the caller owns; bitwise against the single-frame plan on fp16 rows."""
    import link_amd as la
    from link_amd import _lib as L
    blk = _block(la, "cos", 9)
    bounds = ((0, 0, 0, 0), (95, 95, 95, 0))
    n = 15000
    geo = dict(k1_wgs=512, k2_zsplit=2)
    pipe = la.ElkCorePipeline(n, 64, "cos", 32, 3, 7, bounds, "cuda", **geo).bind(*_par(blk))
    plan = la.ElkCorePlan(n, 64, "cos", 32, 3, 7, bounds, "cuda", layout="dense", k1_form=2, **geo).bind(*_par(blk))
    frames = [(torch.randn(n - 100 * i, 64, generator=torch.Generator().manual_seed(i)).cuda().half(),
               s_uniform(n - 100 * i, grid=96, seed=i).cuda()) for i in range(4)]
    outs = [torch.full((f.shape[0], 64), float("nan"), dtype=torch.float16, device="cuda") for f, _ in frames]
    done = []
    for (f, c), o in zip(frames, outs):
        r = pipe.push(f, c, out=o)
        if r is not None:
            done.append(r)
    done += pipe.flush()
    assert len(done) == 4 and all(d.data_ptr() == o.data_ptr() for d, o in zip(done, outs))
    for (f, c), o in zip(frames, outs):
        assert torch.equal(o, plan.run(f, c))
    with pytest.raises(L.LinkAmdError):                 # a float32 frame while fp16 frames are in flight
        pipe.push(frames[0][0], frames[0][1])
        pipe.push(frames[0][0].float(), frames[0][1])
    pipe.flush()


def test_pipeline_reports_voxels_outside_its_bounds():
    import link_amd as la
    from link_amd import _lib as L
    blk = _block(la, "cos", 3)
    bounds = ((0, 0, 0, 0), (63, 63, 63, 0))
    pipe = la.ElkCorePipeline(5000, 64, "cos", 32, 3, 7, bounds, "cuda").bind(*_par(blk))
    f = torch.randn(5000, 64).cuda()
    c = s_uniform(5000, grid=64, seed=1).cuda()
    bad = c.clone()
    bad[17, 0] = 300
    pipe.push(f, c)
    pipe.push(f, bad)
    pipe.push(f, c)
    pipe.flush()
    with pytest.raises(L.LinkAmdError):
        pipe.check()


def test_pipeline_rejects_what_the_step_kernel_does_not_serve():
    import link_amd as la
    from link_amd import _lib as L
    bounds = ((0, 0, 0, 0), (63, 63, 63, 0))
    for kw in (dict(c=32, cg=16, baseop="cos"), dict(c=64, cg=64, baseop="cos"), dict(c=64, cg=32, baseop="cos_x")):
        with pytest.raises(L.LinkAmdError):
            la.ElkCorePipeline(1000, kw["c"], kw["baseop"], kw["cg"], 3, 7, bounds, "cuda")
