"""link_amd/detstage.py -- one stage of the detection backbone SpMiddleResNetFHDELKv3 (row N3 of SURVEY.md 8f).

Reference: detection/det3d/models/backbones/scn.py:452-626.  A stage K of that backbone is

    x_conv = convK_tail(convK(x))            convK = 2 x SparseBasicBlock (SubMConv3d 3^3 + BN + ReLU +
                                             SubMConv3d 3^3 + BN, + identity, ReLU; scn.py:64-107),
                                             convK_tail = SubMConv3d 3^3 + BN          (scn.py:482-489)
    x_lk   = elkK_tail(elkK(x, block_sz=7))  elkK = TSELKBlock (ts_elk.py:110-230), elkK_tail = SubMConv3d + BN
    x      = ReLU(x_conv.features + x_lk.features)                                      (scn.py:586-590)

on spconv tensors.  spconv is an external wheel of the reference (not in /root/reference, not installed here),
so this module carries its own `SubMConv3d` with spconv 2.x's parameter layout (weight [Cout, kz, ky, kx, Cin],
bias [Cout]; indices (batch, z, y, x)) and the standard submanifold semantics: outputs at the input's active
sites only, out[p] = sum_{a,b,c} W[:, a, b, c, :] . in[p + (a-1, b-1, c-1)] (cross-correlation, as
torch.nn.functional.conv3d on the densified input).  tests/test_gpu_detstage.py pins exactly that against
torch's dense conv3d; parity with the spconv wheel itself is UNPINNED (it cannot run here) and says so.

Inference runs every convolution on the HIP kernels with its BatchNorm (folded to scale / shift together with
the convolution's bias), the residual add and the ReLU in the convolution's finish phase
(link_subm_conv_ln_add_relu / link_conv_centre_sum with flag bit 1): six launches-with-epilogue + the fused
TSELKBlock per stage, no elementwise pass over [N, C] in between (row N2: "BN+ReLU of the elkK_tail convs").
With grad enabled the modules run one by one (torch BatchNorm in training mode needs batch statistics).
The k3-s2 SparseConv3d between the stages (it creates new active sites) is not part of this module.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from .elk import (SparseConvTensor, TSELKBlock, _SubmConv, fold_batchnorm, neighbor_table_of, spconv2ts, subm_conv,
                  subm_conv_ln_add_relu)
from .utils import get_kernel_offsets

__all__ = ["SubMConv3d", "SparseBasicBlock", "ELKv3Stage"]


def _replace_feature(sct, feats):
    out = SparseConvTensor(feats, sct.indices, sct.spatial_shape, sct.batch_size, sct.grid, sct.voxel_num,
                           sct.indice_dict, sct.benchmark)
    out.benchmark_record = sct.benchmark_record
    return out


def _site_table(sct):
    """(neighbour table int32[N,27] in link_amd's offset order, spatial tile order) of the tensor's active sites:
    the same cached table the TSELKBlock's local_mix convolution uses (all submanifold convolutions of a stage
    share it, as spconv's indice_key does)."""
    st, _ = spconv2ts(sct)
    return neighbor_table_of(st, (3, 3, 3))


class SubMConv3d(nn.Module):
    """3^3 submanifold convolution with spconv 2.x's parameter layout (see the module docstring)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, bias: bool = True,
                 indice_key: Optional[str] = None):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("SubMConv3d: 3^3 kernels (what the ELKv3 backbone uses)")
        self.in_channels, self.out_channels, self.indice_key = in_channels, out_channels, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, 3, in_channels))
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=5 ** 0.5)
        if bias:
            bound = 1.0 / (27 * in_channels) ** 0.5
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._kio = None

    def kernel_kio(self) -> torch.Tensor:
        """Weights as [27, Cin, Cout] in link_amd's kernel-offset order (offset k = (dx, dy, dz) of
        get_kernel_offsets(3) <-> spconv tap (a, b, c) = (dz+1, dy+1, dx+1)); cached per weight version."""
        ver = (self.weight._version, self.weight.device)
        if self._kio is None or self._kio[0] != ver or torch.is_grad_enabled() and self.weight.requires_grad:
            offs = get_kernel_offsets(3, device="cpu").tolist()
            a = torch.tensor([o[2] + 1 for o in offs], device=self.weight.device)
            b = torch.tensor([o[1] + 1 for o in offs], device=self.weight.device)
            c = torch.tensor([o[0] + 1 for o in offs], device=self.weight.device)
            kio = self.weight[:, a, b, c, :].permute(1, 2, 0).contiguous()        # [27, Cin, Cout]
            if torch.is_grad_enabled() and self.weight.requires_grad:
                return kio
            self._kio = (ver, kio.detach())
        return self._kio[1]

    def forward(self, sct):
        nbr, order = _site_table(sct)
        w = self.kernel_kio()
        if torch.is_grad_enabled() and (sct.features.requires_grad or self.weight.requires_grad):
            out = _SubmConv.apply(sct.features.float(), w, nbr, order)
        else:
            out = subm_conv(sct.features, w, nbr, order)
        if self.bias is not None:
            out = out + self.bias
        return _replace_feature(sct, out)


def _fold(conv: SubMConv3d, bn: nn.BatchNorm1d):
    """(scale, shift) of BatchNorm (running statistics) applied to conv + bias: y = conv * scale + shift."""
    return fold_batchnorm(bn, conv.bias)


def _conv_bn(conv: SubMConv3d, bn: nn.BatchNorm1d, feats, nbr, order, addend=None, relu=False):
    sc, sh = _fold(conv, bn)
    return subm_conv_ln_add_relu(feats, conv.kernel_kio(), nbr, order, sc, sh, 0.0, addend, relu=relu, affine=True)


class SparseBasicBlock(nn.Module):
    """scn.py:64-107 (stride 1, no downsample): relu(bn2(conv2(relu(bn1(conv1(x))))) + x)."""

    def __init__(self, planes: int, eps: float = 1e-3, momentum: float = 0.01, indice_key: Optional[str] = None):
        super().__init__()
        self.conv1 = SubMConv3d(planes, planes, 3, bias=True, indice_key=indice_key)
        self.bn1 = nn.BatchNorm1d(planes, eps=eps, momentum=momentum)
        self.relu = nn.ReLU()
        self.conv2 = SubMConv3d(planes, planes, 3, bias=True, indice_key=indice_key)
        self.bn2 = nn.BatchNorm1d(planes, eps=eps, momentum=momentum)

    def forward(self, sct):
        out = self.conv1(sct)
        out = _replace_feature(out, self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        out = _replace_feature(out, self.bn2(out.features))
        return _replace_feature(out, self.relu(out.features + sct.features))

    def fused(self, feats, nbr, order):
        h = _conv_bn(self.conv1, self.bn1, feats, nbr, order, relu=True)
        return _conv_bn(self.conv2, self.bn2, h, nbr, order, addend=feats, relu=True)


class ELKv3Stage(nn.Module):
    """Stage K of SpMiddleResNetFHDELKv3 (scn.py:477-494 constructor, :586-590 forward); attribute names
    `conv`, `conv_tail`, `elk`, `elk_tail` stand for the backbone's convK, convK_tail, elkK, elkK_tail
    (`load_backbone_stage` maps a backbone state_dict onto them)."""

    def __init__(self, planes: int, block_sz: int = 7, eps: float = 1e-3, momentum: float = 0.01, baseop: str = "cos"):
        super().__init__()
        self.planes, self.block_sz = planes, block_sz
        self.conv = nn.Sequential(SparseBasicBlock(planes, eps, momentum), SparseBasicBlock(planes, eps, momentum))
        self.conv_tail = nn.Sequential(SubMConv3d(planes, planes, 3, bias=False), nn.BatchNorm1d(planes, eps=eps, momentum=momentum))
        self.elk = TSELKBlock(planes, planes, baseop=baseop)
        self.elk_tail = nn.Sequential(SubMConv3d(planes, planes, 3, bias=False), nn.BatchNorm1d(planes, eps=eps, momentum=momentum))
        self.act = nn.ReLU(inplace=True)

    def load_backbone_stage(self, state_dict, k: int, strict: bool = True):
        """Load stage k (1..4) of a SpMiddleResNetFHDELKv3 state_dict (keys conv{k}.*, conv{k}_tail.*, elk{k}.*,
        elk{k}_tail.*)."""
        ren = {f"conv{k}.": "conv.", f"conv{k}_tail.": "conv_tail.", f"elk{k}.": "elk.", f"elk{k}_tail.": "elk_tail."}
        sd = {}
        for key, val in state_dict.items():
            for old, new in ren.items():
                if key.startswith(old):
                    sd[new + key[len(old):]] = val
        return self.load_state_dict(sd, strict=strict)

    def _modules_path(self, sct):
        x_conv = self.conv(sct)
        t = self.conv_tail[0](x_conv)
        x_conv = _replace_feature(t, self.conv_tail[1](t.features))
        x_lk = self.elk(sct, self.block_sz)
        t = self.elk_tail[0](x_lk)
        x_lk = _replace_feature(t, self.elk_tail[1](t.features))
        return _replace_feature(x_conv, self.act(x_conv.features + x_lk.features))

    def forward(self, sct):
        feats = sct.features
        needs_grad = torch.is_grad_enabled() and (feats.requires_grad or any(p.requires_grad for p in self.parameters()))
        if needs_grad or self.training or not feats.is_cuda:
            if not feats.is_cuda:
                raise L.LinkAmdError("ELKv3Stage needs GPU tensors (HIP path; no CPU fallback)")
            return self._modules_path(sct)
        io_dtype = feats.dtype
        nbr, order = _site_table(sct)
        x = feats.float().contiguous()
        for blk in self.conv:
            x = blk.fused(x, nbr, order)
        x_conv = _conv_bn(self.conv_tail[0], self.conv_tail[1], x, nbr, order)
        x_lk = self.elk(_replace_feature(sct, feats.float()), self.block_sz).features
        out = _conv_bn(self.elk_tail[0], self.elk_tail[1], x_lk, nbr, order, addend=x_conv, relu=True)
        return _replace_feature(sct, out if io_dtype == torch.float32 else out.to(io_dtype))
