"""tests/link_encoder.py -- the encoder stages of the LinK segmentation networks (cfg3 of BASELINE.json:
"full LinK cos_x:(2x3)^3 encoder fwd+bwd") assembled from link_amd modules, plus the same graph on the CPU
from the oracle restatements.  Test infrastructure: the reference's graph is
segmentation/core/models/semantic_kitti/linkencoder.py:186-290 (modules) and :342-368 (forward):

    stem (2 x [Conv3d k3, BN, ReLU]) -> 4 x [ down = Conv3d k2 s2 + BN + ReLU;
                                              x = tail(2 ResidualBlocks(down));  lk = elk_tail(ELKBlock(down, ts*s, r));
                                              x.F = ReLU(x.F + lk.F) ]

ResidualBlock = linkencoder.py:61-92, BasicConvolutionBlock = :23-39.  spnn.BatchNorm is BatchNorm1d on the
feature rows (torchsparse/nn/modules/norm.py), training-mode statistics in both implementations.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as TF


def build_stages(la, cin=4, c=64, baseop="cos_x", groups=1, n_stages=4):
    """GPU module built from link_amd.Conv3d / link_amd.ELKBlock (+ torch BatchNorm1d / ReLU on the rows)."""

    def on_rows(mod, x):
        y = la.SparseTensor(mod(x.F), x.C, x.s)
        y.cmaps, y.kmaps = x.cmaps, x.kmaps
        return y

    class ConvBN(nn.Module):
        def __init__(self, inc, outc, ks=3, stride=1, relu=True):
            super().__init__()
            self.conv, self.bn, self.relu = la.Conv3d(inc, outc, ks, stride=stride), nn.BatchNorm1d(outc), relu

        def forward(self, x):
            y = on_rows(self.bn, self.conv(x))
            return on_rows(torch.relu, y) if self.relu else y

    class Residual(nn.Module):
        def __init__(self, inc, outc):
            super().__init__()
            self.a, self.b = ConvBN(inc, outc, 3, relu=True), ConvBN(outc, outc, 3, relu=False)
            self.short = None if inc == outc else ConvBN(inc, outc, 1, relu=False)

        def forward(self, x):
            y = self.b(self.a(x))
            sc = x if self.short is None else self.short(x)
            out = la.SparseTensor(torch.relu(y.F + sc.F), y.C, y.s)
            out.cmaps, out.kmaps = x.cmaps, x.kmaps
            return out

    class Stages(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = nn.Sequential(ConvBN(cin, c), ConvBN(c, c))
            self.down = nn.ModuleList([ConvBN(c, c, 2, stride=2) for _ in range(n_stages)])
            self.res = nn.ModuleList([nn.Sequential(Residual(c, c), Residual(c, c)) for _ in range(n_stages)])
            self.tail = nn.ModuleList([ConvBN(c, c, 3, relu=False) for _ in range(n_stages)])
            self.elk = nn.ModuleList([la.ELKBlock(c, c, groups, baseop=baseop, variant="encoder") for _ in range(n_stages)])
            self.elk_tail = nn.ModuleList([ConvBN(c, c, 3, relu=False) for _ in range(n_stages)])

        def forward(self, x, s, r):
            x.cmaps.setdefault(x.stride, x.coords)
            x = self.stem(x)
            outs = []
            for i in range(n_stages):
                d = self.down[i](x)
                y = self.tail[i](self.res[i](d))
                lk = self.elk_tail[i](self.elk[i](d, d.s[0] * s, r))     # ELKBlock overwrites d.F (reference contract)
                x = la.SparseTensor(torch.relu(y.F + lk.F), y.C, y.s)
                x.cmaps, x.kmaps = d.cmaps, d.kmaps
                outs.append(x)
            return outs

    return Stages()


def oracle_stages(lo, sd, feats, coords, s, r, baseop="cos_x", groups=1, n_stages=4, c=64):
    """The same graph on the CPU from the oracle restatements; `sd` = the GPU module's state_dict moved to the
    CPU in the dtype to compute in (tensors may require grad)."""

    def bn(x, pre):
        return TF.batch_norm(x, None, None, sd[pre + ".bn.weight"], sd[pre + ".bn.bias"], True, 0.1, 1e-5)

    def conv3(x, cc, ts, pre):                      # ConvBN with a k3 / k1 stride-1 conv
        k = sd[pre + ".conv.kernel"]
        y = x @ k if k.ndim == 2 else lo.subm_conv_torch(x, cc, k, ts)
        return bn(y, pre)

    def residual(x, cc, ts, pre):
        y = conv3(torch.relu(conv3(x, cc, ts, pre + ".a")), cc, ts, pre + ".b")
        sc = conv3(x, cc, ts, pre + ".short") if (pre + ".short.conv.kernel") in sd else x
        return torch.relu(y + sc)

    cc = coords.numpy() if hasattr(coords, "numpy") else np.asarray(coords)
    ts = 1
    x = torch.relu(conv3(torch.relu(conv3(feats, cc, ts, "stem.0")), cc, ts, "stem.1"))
    outs = []
    for i in range(n_stages):
        coarse = lo.downsample_coords(cc, 2, ts)
        table = lo.strided_conv_table(cc, coarse, 2, ts)
        d = torch.relu(bn(lo.gather_conv_torch(x, table, sd[f"down.{i}.conv.kernel"]), f"down.{i}"))
        cc, ts = coarse, ts * 2
        y = conv3(residual(residual(d, cc, ts, f"res.{i}.0"), cc, ts, f"res.{i}.1"), cc, ts, f"tail.{i}")
        blk = {k[len(f"elk.{i}."):]: v for k, v in sd.items() if k.startswith(f"elk.{i}.")}
        core = lo.elk_core_torch(d, torch.from_numpy(cc), blk, ts * s, r, baseop, groups, variant="encoder", tensor_stride=ts)
        local = lo.subm_conv_torch(d, cc, blk["local_mix.0.kernel"], ts)
        e = torch.relu(core + TF.layer_norm(local, (c,), blk["norm_local.weight"], blk["norm_local.bias"], 1e-6))
        lk = conv3(e, cc, ts, f"elk_tail.{i}")
        x = torch.relu(y + lk)
        outs.append((x, cc))
    return outs


from harness.networks import build_reference_shaped_encoder  # noqa: E402,F401  (moved into the package: bench.py uses it)
