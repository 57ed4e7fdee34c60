"""link_amd/tensor.py -- the sparse data model the LinK surface is expressed in.

Mirrors the attribute surface of torchsparse v1.4 `SparseTensor` / `PointTensor`
(/root/reference/segmentation/torchsparse-u/torchsparse/tensor.py:10-112): `feats [N,C]`,
`coords [N,4] int32 (x,y,z,batch)`, `stride` 3-tuple, `cmaps` / `kmaps` caches that are SHARED BY
REFERENCE between tensors derived from one another, plus the short aliases F / C / s.
"""
from __future__ import annotations

from typing import Any, Dict, Tuple, Union

import torch

from .utils import make_ntuple

__all__ = ["SparseTensor", "PointTensor", "cat"]


class SparseTensor:
    def __init__(self, feats: torch.Tensor, coords: torch.Tensor,
                 stride: Union[int, Tuple[int, ...]] = 1) -> None:
        self.feats = feats
        self.coords = coords
        self._stride = make_ntuple(stride, ndim=3)
        self.cmaps: Dict[Tuple[int, ...], torch.Tensor] = {}
        self.kmaps: Dict[Tuple[Any, ...], Any] = {}

    # -- reference attribute names -----------------------------------------------------------
    @property
    def stride(self) -> Tuple[int, ...]:
        return self._stride

    @stride.setter
    def stride(self, value) -> None:
        self._stride = make_ntuple(value, ndim=3)

    F = property(lambda self: self.feats, lambda self, v: setattr(self, "feats", v))
    C = property(lambda self: self.coords, lambda self, v: setattr(self, "coords", v))
    s = property(lambda self: self._stride, lambda self, v: setattr(self, "stride", v))

    # -- device / graph helpers (in place, returning self, as the reference does) ---------------
    def _map(self, fn) -> "SparseTensor":
        self.feats, self.coords = fn(self.feats), fn(self.coords)
        return self

    def cuda(self) -> "SparseTensor":
        return self._map(lambda t: t.cuda())

    def detach(self) -> "SparseTensor":
        return self._map(lambda t: t.detach())

    def to(self, device, non_blocking: bool = True) -> "SparseTensor":
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def __add__(self, other: "SparseTensor") -> "SparseTensor":
        out = SparseTensor(self.feats + other.feats, self.coords, self._stride)
        out.cmaps, out.kmaps = self.cmaps, self.kmaps
        return out

    def __repr__(self) -> str:
        return (f"SparseTensor(N={self.feats.shape[0]}, C={self.feats.shape[1] if self.feats.ndim > 1 else '-'}, "
                f"stride={self._stride}, device={self.feats.device})")


class PointTensor:
    def __init__(self, feats, coords, idx_query=None, weights=None):
        self.F = feats
        self.C = coords
        self.idx_query = {} if idx_query is None else idx_query
        self.weights = {} if weights is None else weights
        self.additional_features = {"idx_query": {}, "counts": {}}

    def _map(self, fn) -> "PointTensor":
        self.F, self.C = fn(self.F), fn(self.C)
        return self

    def cuda(self):
        return self._map(lambda t: t.cuda())

    def detach(self):
        return self._map(lambda t: t.detach())

    def to(self, device, non_blocking=True):
        return self._map(lambda t: t.to(device, non_blocking=non_blocking))

    def __add__(self, other):
        out = PointTensor(self.F + other.F, self.C, self.idx_query, self.weights)
        out.additional_features = self.additional_features
        return out


def cat(inputs) -> SparseTensor:
    """torchsparse.cat (operators.py:10-17): concatenate features of tensors sharing coordinates."""
    first = inputs[0]
    out = SparseTensor(torch.cat([t.feats for t in inputs], dim=1), first.coords, first.stride)
    out.cmaps, out.kmaps = first.cmaps, first.kmaps
    return out
