"""link_amd/modules.py -- the row-wise module wrappers the reference networks take from `torchsparse.nn`
(torchsparse/nn/modules/{norm,activation}.py, nn/utils/apply.py): a torch module applied to the feature
rows of a SparseTensor, coordinates / stride / cmaps / kmaps carried over by reference."""
from __future__ import annotations

from typing import Callable

import torch
from torch import nn

from .tensor import SparseTensor

__all__ = ["fapply", "BatchNorm", "ReLU", "LeakyReLU"]


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
