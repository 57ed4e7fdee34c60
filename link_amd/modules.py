"""link_amd/modules.py -- the row-wise module wrappers the reference networks take from `torchsparse.nn`
(torchsparse/nn/modules/{norm,activation}.py, nn/utils/apply.py): a torch module applied to the feature
rows of a SparseTensor, coordinates / stride / cmaps / kmaps carried over by reference."""
from __future__ import annotations

from typing import Callable

import torch
from torch import nn

from .tensor import SparseTensor

__all__ = ["fapply", "BatchNorm", "ReLU", "LeakyReLU", "fuse_for_inference"]


def fapply(input: SparseTensor, fn: Callable[..., torch.Tensor], *args, **kwargs) -> SparseTensor:
    out = SparseTensor(fn(input.feats, *args, **kwargs), input.coords, input.stride)
    out.cmaps, out.kmaps = input.cmaps, input.kmaps
    return out


class BatchNorm(nn.BatchNorm1d):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class ReLU(nn.ReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class LeakyReLU(nn.LeakyReLU):
    def forward(self, input: SparseTensor) -> SparseTensor:
        return fapply(input, super().forward)


class _FusedSequential(nn.Sequential):
    """nn.Sequential whose [Conv3d, BatchNorm(, ReLU)] runs execute as one launch at inference (see
    fuse_for_inference).  Same children, same state_dict keys; with grad enabled or in training mode it is the
    plain nn.Sequential (BatchNorm then needs batch statistics and the autograd path)."""

    def forward(self, input):
        mods = list(self)
        if torch.is_grad_enabled() or self.training or not isinstance(input, SparseTensor) or not input.F.is_cuda:
            for m in mods:
                input = m(input)
            return input
        from .elk import fold_batchnorm
        i = 0
        while i < len(mods):
            g = self._link_groups.get(i)
            if g is None:
                input = mods[i](input)
                i += 1
                continue
            conv, bn, relu, nxt = mods[i], mods[g[0]], g[1], g[2]
            sc, sh = fold_batchnorm(bn, conv.bias)
            input = conv.forward_affine(input, sc, sh, relu)
            i = nxt
        return input


def fuse_for_inference(model: nn.Module) -> nn.Module:
    """Rewrite, in place, every nn.Sequential of `model` so that runs of link_amd.Conv3d -> BatchNorm
    (-> ReLU) execute as ONE kernel launch at inference: the BatchNorm's running statistics and affine (and the
    convolution's bias) are folded into a per-channel scale / shift applied in the convolution's finish phase
    (include/link_amd.h: flag bit 1 of link_subm_conv_ln_add_relu / link_conv_centre_sum / link_conv_pairs_sum).
    That is the shape of the reference's BasicConvolutionBlock / ResidualBlock.net / stageK_tail / elkK_tail
    (linkunet.py:18-92,217-224), so it applies to the reference's own network classes built on the aliased
    surface.  Children, parameter names and state_dict keys are untouched (only the container's class changes);
    training mode and autograd run the modules one by one as before.  Returns `model`."""
    from .elk import Conv3d
    for mod in list(model.modules()):
        if not isinstance(mod, nn.Sequential) or isinstance(mod, _FusedSequential):
            continue
        kids = list(mod)
        groups, i = {}, 0
        while i < len(kids):
            if isinstance(kids[i], Conv3d) and i + 1 < len(kids) and isinstance(kids[i + 1], nn.BatchNorm1d):
                relu = i + 2 < len(kids) and isinstance(kids[i + 2], nn.ReLU)
                groups[i] = (i + 1, relu, i + (3 if relu else 2))
                i += 3 if relu else 2
            else:
                i += 1
        if groups and type(mod) is nn.Sequential:
            mod.__class__ = _FusedSequential
            mod._link_groups = groups
    return model
