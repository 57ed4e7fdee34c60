"""link_amd/utils.py -- small host helpers with the reference's names.

make_ntuple        torchsparse/utils/utils.py:9-20
get_kernel_offsets torchsparse/nn/utils/kernel.py:11-32
"""
from __future__ import annotations

from functools import lru_cache
from typing import Tuple, Union

import torch

__all__ = ["make_ntuple", "get_kernel_offsets"]


def make_ntuple(x, ndim: int) -> Tuple[int, ...]:
    if isinstance(x, torch.Tensor):
        x = [int(v) for v in x.reshape(-1).tolist()]
    if isinstance(x, int):
        return (x,) * ndim
    x = tuple(x)
    assert len(x) == ndim, x
    return x


@lru_cache(maxsize=64)
def _offsets_list(size, stride, dilation):
    axes = []
    for k in range(3):
        lo, hi = -size[k] // 2 + 1, size[k] // 2 + 1      # python floor division, as the reference
        axes.append([v * stride[k] * dilation[k] for v in range(lo, hi)])
    vol = size[0] * size[1] * size[2]
    if vol % 2 == 1:    # odd volume: x fastest, z outermost (MinkowskiEngine-compatible weight layout)
        return tuple((x, y, z) for z in axes[2] for y in axes[1] for x in axes[0])
    return tuple((x, y, z) for x in axes[0] for y in axes[1] for z in axes[2])


def get_kernel_offsets(size: Union[int, Tuple[int, ...]], stride: Union[int, Tuple[int, ...]] = 1,
                       dilation: Union[int, Tuple[int, ...]] = 1, device="cpu") -> torch.Tensor:
    """int32[K,3] kernel offsets in the reference's order (host list is cached; one small H2D copy)."""
    offs = _offsets_list(make_ntuple(size, 3), make_ntuple(stride, 3), make_ntuple(dilation, 3))
    return torch.tensor(offs, dtype=torch.int32, device=device).reshape(-1, 3)
