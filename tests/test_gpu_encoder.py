"""cfg3 of BASELINE.json in miniature: the LinK encoder stages (stem + stages of down-conv, residual blocks,
ELKBlock branch) assembled from link_amd modules, forward and backward, against the same graph on the CPU
from the oracle restatements.  A deep ReLU network cannot be compared element by element (one activation
flipping between two fp32 evaluations moves a gradient entry by O(1)), so forward outputs are gated by the
max-norm 1e-4 metric per stage and gradients by relative L2 error."""
import numpy as np
import pytest
import torch

from helpers import lidar_like, rel_err
import link_encoder as LE

pytestmark = pytest.mark.gpu


def _l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("baseop,groups,s,r,c,n_stages", [("cos_x", 1, 3, 2, 32, 3), ("cos", 2, 3, 3, 16, 2)])
def test_encoder_stages_forward_backward_vs_oracle(baseop, groups, s, r, c, n_stages):
    import link_amd as la
    from oracle import link_oracle as lo
    coords = torch.from_numpy(lidar_like(12000, seed=21, voxel=0.2))
    n = coords.shape[0]
    feats = torch.rand(n, 4, generator=torch.Generator().manual_seed(1))
    torch.manual_seed(5)
    net = LE.build_stages(la, 4, c, baseop, groups, n_stages).cuda().train()
    f = feats.cuda().requires_grad_(True)
    outs = net(la.SparseTensor(f, coords.cuda(), 1), s, r)
    loss = sum(o.F.square().mean() for o in outs)
    loss.backward()

    sd = {k: v.detach().cpu().double().requires_grad_(v.dtype.is_floating_point) if v.dtype.is_floating_point
          else v.cpu() for k, v in net.state_dict().items()}
    fr = feats.double().requires_grad_(True)
    ref = LE.oracle_stages(lo, sd, fr, coords, s, r, baseop, groups, n_stages, c)
    sum(o.square().mean() for o, _ in ref).backward()

    for i, (o, (ro, rc)) in enumerate(zip(outs, ref)):
        assert np.array_equal(o.C.cpu().numpy(), rc), f"stage {i} coordinates"
        assert o.s == (2 ** (i + 1),) * 3
        assert rel_err(o.F.detach().cpu().numpy(), ro.detach().numpy()) < 2e-4, f"stage {i} output"
    assert _l2(f.grad.cpu().numpy(), fr.grad.numpy()) < 2e-3
    checked = 0
    for name, p in net.named_parameters():
        if p.grad is None:
            continue
        assert sd[name].grad is not None, name
        assert _l2(p.grad.cpu().numpy(), sd[name].grad.numpy()) < 5e-3, name
        checked += 1
    assert checked > 40
