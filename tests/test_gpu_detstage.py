"""Row N3: one stage of the detection backbone SpMiddleResNetFHDELKv3 (link_amd.ELKv3Stage,
scn.py:477-494,586-590) on the SparseConvTensor shim.

Oracle: the submanifold convolutions as torch's dense conv3d on the densified grid, read back at the active sites
(the definition of a submanifold convolution), torch BatchNorm1d in eval mode, and the TSELKBlock through the
oracle's restatement (pinned on the reference's ELKBlock goldens).  spconv itself is an external wheel of the
reference and cannot run here: the weight layout ([Cout, kz, ky, kx, Cin]) and tap order are spconv 2.x's documented
ones, parity with the wheel is unpinned."""
import numpy as np
import pytest
import torch

from helpers import rel_err, s_uniform

pytestmark = pytest.mark.gpu


def dense_subm(feats, indices, shape, weight, bias):
    """Submanifold 3^3 convolution by definition: dense cross-correlation of the zero-filled grid, sampled at the
    active sites.  indices (b, z, y, x); weight [Cout, kz, ky, kx, Cin]."""
    b = int(indices[:, 0].max()) + 1
    d, h, w = shape
    vol = torch.zeros(b, feats.shape[1], d, h, w, dtype=torch.float64)
    bi, zi, yi, xi = [indices[:, k].long() for k in range(4)]
    vol[bi, :, zi, yi, xi] = feats.double()
    out = torch.nn.functional.conv3d(vol, weight.double().permute(0, 4, 1, 2, 3), None if bias is None else bias.double(),
                                     padding=1)
    return out[bi, :, zi, yi, xi]


def oracle_stage(stage, feats, indices, shape, block_sz):
    from oracle import link_oracle as O
    sd = {k: v.detach().cpu() for k, v in stage.state_dict().items()}

    def conv(prefix, x):
        return dense_subm(x, indices, shape, sd[prefix + ".weight"], sd.get(prefix + ".bias"))

    def bn(prefix, x):
        eps = 1e-3
        return (x - sd[prefix + ".running_mean"].double()) / torch.sqrt(sd[prefix + ".running_var"].double() + eps) \
            * sd[prefix + ".weight"].double() + sd[prefix + ".bias"].double()

    x = feats.double()
    for i in range(2):
        h = torch.relu(bn(f"conv.{i}.bn1", conv(f"conv.{i}.conv1", x)))
        x = torch.relu(bn(f"conv.{i}.bn2", conv(f"conv.{i}.conv2", h)) + x)
    x_conv = bn("conv_tail.1", conv("conv_tail.0", x))
    # TSELKBlock (ts_elk.py:144-230): relu(core + LayerNorm(local_mix(x)))
    coords = indices[:, [3, 2, 1, 0]].contiguous()
    elk = {k[len("elk."):]: v for k, v in sd.items() if k.startswith("elk.")}
    core = O.elk_core_torch(feats, coords, elk, block_sz, 3, "cos", 1, variant="det", agg=O.aggregate_c).double()
    local = O.subm_conv_torch(feats.double(), coords, elk["local_mix.0.kernel"].double(), 1)
    c = feats.shape[1]
    local = torch.nn.functional.layer_norm(local, (c,), elk["norm_local.weight"].double(), elk["norm_local.bias"].double(), 1e-6)
    e = torch.relu(core + local)
    x_lk = bn("elk_tail.1", conv("elk_tail.0", e))
    return torch.relu(x_conv + x_lk)


def _stage(planes, seed):
    import link_amd as la
    torch.manual_seed(seed)
    stage = la.ELKv3Stage(planes).cuda()
    for m in stage.modules():                      # non-trivial BatchNorm statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.5, 0.5)
            m.running_var.uniform_(0.5, 2.0)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.uniform_(-0.3, 0.3)
    return stage.eval()


@pytest.mark.parametrize("planes,n,grid", [(16, 6000, 40), (32, 4000, 24), (64, 3000, 28)])
def test_stage_forward_vs_dense_oracle(planes, n, grid):
    import link_amd as la
    stage = _stage(planes, 5)
    coords = s_uniform(n, grid=grid, seed=11)
    coords[:, 2] = coords[:, 2] % 16                  # flat in z like the detection grids
    coords = torch.unique(coords, dim=0)
    n = coords.shape[0]
    shape = [16, grid, grid]                          # (z, y, x)
    indices = coords[:, [3, 2, 1, 0]].contiguous().int()
    feats = torch.randn(n, planes, generator=torch.Generator().manual_seed(2))
    sct = la.SparseConvTensor(feats.cuda(), indices.cuda(), shape, 1)
    with torch.no_grad():
        out = stage(sct)
        out2 = stage(sct)
    assert torch.equal(out.indices.cpu(), indices) and torch.equal(out.features, out2.features)
    ref = oracle_stage(stage, feats, indices, shape, stage.block_sz)
    assert rel_err(out.features.cpu().numpy(), ref.numpy()) < 2e-5
    # module-by-module execution (what runs with grad enabled) gives the same features
    stage.train(False)
    with torch.enable_grad():
        f = feats.cuda().requires_grad_(True)
        o3 = stage(la.SparseConvTensor(f, indices.cuda(), shape, 1))
        o3.features.square().sum().backward()
    assert rel_err(o3.features.detach().cpu().numpy(), ref.numpy()) < 2e-5
    assert f.grad is not None and torch.isfinite(f.grad).all()
    assert stage.conv[0].conv1.weight.grad is not None and stage.elk_tail[0].weight.grad.abs().sum() > 0


def test_submconv_weight_layout_and_state_dict_mapping():
    """Single SubMConv3d against the dense definition with an asymmetric kernel (catches a tap-order or transpose
    mix-up), and loading a backbone-style state_dict (keys conv2.*, conv2_tail.*, elk2.*, elk2_tail.*)."""
    import link_amd as la
    torch.manual_seed(0)
    conv = la.SubMConv3d(16, 32, 3, bias=True).cuda()
    coords = s_uniform(2500, grid=14, seed=3)
    indices = coords[:, [3, 2, 1, 0]].contiguous().int()
    feats = torch.randn(2500, 16)
    sct = la.SparseConvTensor(feats.cuda(), indices.cuda(), [14, 14, 14], 1)
    with torch.no_grad():
        out = conv(sct).features
    ref = dense_subm(feats, indices, [14, 14, 14], conv.weight.detach().cpu(), conv.bias.detach().cpu())
    assert rel_err(out.cpu().numpy(), ref.numpy()) < 1e-5
    src = _stage(32, 1)
    sd = {}
    for k, v in src.state_dict().items():
        head, rest = k.split(".", 1)
        sd[{"conv": "conv2", "conv_tail": "conv2_tail", "elk": "elk2", "elk_tail": "elk2_tail"}[head] + "." + rest] = v
    sd["conv3.0.conv1.weight"] = torch.zeros(1)       # other stages' keys are ignored
    dst = la.ELKv3Stage(32).cuda().eval()
    dst.load_backbone_stage(sd, 2)
    for (k1, v1), (k2, v2) in zip(src.state_dict().items(), dst.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
