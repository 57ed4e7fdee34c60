"""link_amd/functional.py -- the `torchsparse.nn.functional` names on the LinK path.

sphash / sphashquery / spcount / spvoxelize / spdevoxelize with the reference's signatures and
dtype conventions (torchsparse/nn/functional/{hash,query,count,voxelize,devoxelize}.py), dispatching
to the HIP library only.  The reference wraps voxelize/devoxelize in `custom_fwd(cast_inputs=half)`;
this path computes in fp32 (half/bf16 inputs are up-cast, results returned in fp32) -- fp32
accumulation is what the parity gate is defined on.
"""
from __future__ import annotations

from typing import Optional

import torch
from torch.autograd import Function

from . import backend

__all__ = ["sphash", "sphashquery", "spcount", "spvoxelize", "spdevoxelize", "calc_ti_weights", "conv3d", "spdownsample"]


def sphash(coords: torch.Tensor, offsets: Optional[torch.Tensor] = None) -> torch.Tensor:
    """hash.py:10-37.  int32[N,4] -> int64[N]; with offsets int32[K,3] -> int64[K,N]."""
    assert coords.dtype == torch.int, coords.dtype
    assert coords.ndim == 2 and coords.shape[1] == 4, coords.shape
    if offsets is None:
        return backend.hash_cuda(coords)
    assert offsets.dtype == torch.int, offsets.dtype
    assert offsets.ndim == 2 and offsets.shape[1] == 3, offsets.shape
    return backend.kernel_hash_cuda(coords, offsets.to(coords.device))


def sphashquery(queries: torch.Tensor, references: torch.Tensor) -> torch.Tensor:
    """query.py:8-33.  Position of each query hash in `references` (first duplicate wins) or -1;
    the query tensor's shape is preserved."""
    shape = queries.shape
    indices = torch.arange(references.numel(), device=queries.device, dtype=torch.long)
    out = backend.hash_query_cuda(queries.contiguous().view(-1), references.contiguous().view(-1), indices)
    return (out - 1).view(shape)


def spcount(coords: torch.Tensor, num) -> torch.Tensor:
    """count.py:8-16.  int32[N] -> int32[num]."""
    return backend.count_cuda(coords.contiguous(), int(num))


class VoxelizeFunction(Function):
    """voxelize.py:10-52."""

    @staticmethod
    def forward(ctx, feats, coords, counts):
        feats = feats.contiguous().float()
        coords = coords.contiguous().int()
        counts = counts.contiguous().int()
        out = backend.voxelize_forward_cuda(feats, coords, counts)
        ctx.for_backwards = (coords, counts, feats.shape[0])
        return out

    @staticmethod
    def backward(ctx, grad_output):
        coords, counts, n = ctx.for_backwards
        g = backend.voxelize_backward_cuda(grad_output.contiguous().float(), coords, counts, n)
        return g, None, None


class DevoxelizeFunction(Function):
    """devoxelize.py:51-93 (with LinK's `r`)."""

    @staticmethod
    def forward(ctx, feats, coords, weights, r):
        feats = feats.contiguous().float()
        coords = coords.contiguous().int()
        weights = weights.contiguous().float()
        out = backend.devoxelize_forward_cuda(feats, coords, weights, r)
        ctx.for_backwards = (coords, weights, feats.shape[0], r)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        coords, weights, n, r = ctx.for_backwards
        g = backend.devoxelize_backward_cuda(grad_output.contiguous().float(), coords, weights, n, r)
        return g, None, None, None


def spvoxelize(feats: torch.Tensor, coords: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    return VoxelizeFunction.apply(feats, coords, counts)


def spdevoxelize(feats: torch.Tensor, coords: torch.Tensor, weights: torch.Tensor, r: int = 2) -> torch.Tensor:
    return DevoxelizeFunction.apply(feats, coords, weights, r)


def calc_ti_weights(coords: torch.Tensor, idx_query: torch.Tensor, scale: float = 1) -> torch.Tensor:
    """Trilinear interpolation weights [8, P] of the corner voxels (corner k = 4*dx + 2*dy + dz, the order of
    get_kernel_offsets(2)), zero for absent corners, renormalised to sum to one (+1e-8)."""
    if coords.is_cuda and coords.shape[1] == 4 and idx_query.dim() == 2 and idx_query.shape[0] == 8:
        # one HIP kernel (link_ti_weights) instead of ~30 elementwise launches over [8, P] temporaries
        from . import _lib as L
        pts = coords.detach().float().contiguous()
        idx = idx_query.long().contiguous()
        w = torch.empty((8, pts.shape[0]), dtype=torch.float32, device=coords.device)
        L.check(L.lib().link_ti_weights(pts.data_ptr(), idx.data_ptr(), pts.shape[0], float(scale), w.data_ptr(),
                                        L.current_stream_handle()), "link_ti_weights")
        return w
    with torch.no_grad():
        p = coords[:, :3]
        lo = (torch.floor(p / scale) * scale if scale != 1 else torch.floor(p)).float()
        hi = lo + scale
        near, far = hi - p, p - lo                       # weight of the low / high corner per axis
        cols = []
        for dx in (0, 1):
            for dy in (0, 1):
                for dz in (0, 1):
                    wx = far[:, 0] if dx else near[:, 0]
                    wy = far[:, 1] if dy else near[:, 1]
                    wz = far[:, 2] if dz else near[:, 2]
                    cols.append(wx * wy * wz)
        w = torch.stack(cols, 0).contiguous()
        if scale != 1:
            w /= scale ** 3
        w[idx_query == -1] = 0
        w /= w.sum(0) + 1e-8
    return w


def conv3d(*args, **kwargs):
    """torchsparse.nn.functional.conv3d (nn/functional/conv.py:83) -- see link_amd.elk.conv3d."""
    from .elk import conv3d as _impl
    return _impl(*args, **kwargs)


def spdownsample(*args, **kwargs):
    """torchsparse.nn.functional.spdownsample (nn/functional/downsample.py:11) -- see link_amd.elk.spdownsample."""
    from .elk import spdownsample as _impl
    return _impl(*args, **kwargs)
