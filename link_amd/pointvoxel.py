"""link_amd/pointvoxel.py -- row N4 of SURVEY.md section 8f: the point <-> voxel helpers the LinK
segmentation models call at their entry and exit (`initial_voxelize(pt, 1, 1)` is the first line of
both networks: linkunet.py:402-403, linkencoder.py:399-400), on the HIP op kernels.

  calc_ti_weights   torchsparse/nn/functional/devoxelize.py:10-48   trilinear weights of the 8 corner voxels
  initial_voxelize  segmentation/core/models/utils.py:234-254       points -> voxel means
  point_to_voxel    utils.py:259-281                                 point features -> existing voxel set
  voxel_to_point    utils.py:286-324                                 trilinear (or nearest) voxel -> point

Same signatures, same side effects on the PointTensor caches (`additional_features['idx_query'|'counts']`,
`idx_query`, `weights`, the overwrite of `z.C` by initial_voxelize), same row order of the produced voxel set
(ascending coordinate HASH: the reference numbers voxels by `torch.unique(sphash(...))`).
"""
from __future__ import annotations

import torch

from . import functional as F
from .tensor import PointTensor, SparseTensor
from .utils import get_kernel_offsets

from .functional import calc_ti_weights

__all__ = ["calc_ti_weights", "initial_voxelize", "point_to_voxel", "voxel_to_point"]


def _voxel_keys(pc: torch.Tensor, stride: int) -> torch.Tensor:
    """int32 [P,4] voxel coordinate (multiple of `stride`) + batch of float point coordinates."""
    return torch.cat([torch.floor(pc[:, :3] / stride).int() * stride, pc[:, -1].int().view(-1, 1)], 1)


def initial_voxelize(z: PointTensor, init_res, after_res) -> SparseTensor:
    scaled = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
    cell = torch.floor(scaled)
    pc_hash = F.sphash(cell.int())
    vox_hash = torch.unique(pc_hash)                     # voxel numbering = ascending hash
    idx_query = F.sphashquery(pc_hash, vox_hash)
    counts = F.spcount(idx_query.int(), len(vox_hash))
    vox_coords = torch.round(F.spvoxelize(cell, idx_query, counts)).int()   # mean of identical cells = the cell
    vox_feats = F.spvoxelize(z.F, idx_query, counts)
    out = SparseTensor(vox_feats, vox_coords, 1)
    out.cmaps.setdefault(out.stride, out.coords)
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    z.C = scaled
    return out


def point_to_voxel(x: SparseTensor, z: PointTensor) -> SparseTensor:
    cache = z.additional_features
    if cache is None or cache.get("idx_query") is None or cache["idx_query"].get(x.s) is None:
        idx_query = F.sphashquery(F.sphash(_voxel_keys(z.C, x.s[0])), F.sphash(x.C))
        counts = F.spcount(idx_query.int(), x.C.shape[0])
        cache["idx_query"][x.s] = idx_query
        cache["counts"][x.s] = counts
    else:
        idx_query, counts = cache["idx_query"][x.s], cache["counts"][x.s]
    out = SparseTensor(F.spvoxelize(z.F, idx_query, counts), x.C, x.s)
    out.cmaps, out.kmaps = x.cmaps, x.kmaps
    return out


def voxel_to_point(x: SparseTensor, z: PointTensor, nearest: bool = False) -> PointTensor:
    cached = (z.idx_query is not None and z.weights is not None and z.idx_query.get(x.s) is not None
              and z.weights.get(x.s) is not None)
    if not cached:
        corners = get_kernel_offsets(2, x.s, 1, device=z.F.device)
        idx_query = F.sphashquery(F.sphash(_voxel_keys(z.C, x.s[0]), corners), F.sphash(x.C.to(z.F.device)))
        weights = calc_ti_weights(z.C, idx_query, scale=x.s[0]).transpose(0, 1).contiguous()
        idx_query = idx_query.transpose(0, 1).contiguous()
        if nearest:
            weights[:, 1:] = 0.0
            idx_query[:, 1:] = -1
        z.idx_query[x.s] = idx_query
        z.weights[x.s] = weights
    out = PointTensor(F.spdevoxelize(x.F, z.idx_query[x.s], z.weights[x.s]), z.C, idx_query=z.idx_query,
                      weights=z.weights)
    out.additional_features = z.additional_features
    return out
