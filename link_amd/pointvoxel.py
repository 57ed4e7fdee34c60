"""link_amd/pointvoxel.py -- row N4 of SURVEY.md section 8f: the point <-> voxel helpers the LinK
segmentation models call at their entry and exit (`initial_voxelize(pt, 1, 1)` is the first line of
both networks: linkunet.py:402-403, linkencoder.py:399-400).

  calc_ti_weights   torchsparse/nn/functional/devoxelize.py:10-48   trilinear weights of the 8 corner voxels
  initial_voxelize  segmentation/core/models/utils.py:234-254       points -> voxel means
  point_to_voxel    utils.py:259-281                                 point features -> existing voxel set
  voxel_to_point    utils.py:286-324                                 trilinear (or nearest) voxel -> point

Same signatures, same side effects on the PointTensor caches (`additional_features['idx_query'|'counts']`,
`idx_query`, `weights`, the overwrite of `z.C` by initial_voxelize), same row order of the produced voxel set
(ascending coordinate HASH: the reference numbers voxels by `torch.unique(sphash(...))`).

How they run here (round 3; the reference chains sphash -> torch.unique -> sphashquery -> spcount -> spvoxelize
over all P points):

  * the voxel set of a point cloud is the block structure of its integer cells for block edge 1, so
    initial_voxelize builds ONE dense-grid index over the points (`BlockIndex(cells, 1)`: unique cells, the point ->
    cell map, counts and the id-sorted member lists -- section B of the C ABI, no hash of a point, no sort of P keys)
    and takes the voxel means from the indexed block-mean kernel.  Only the M unique voxels are hashed and sorted, to
    put them into the reference's numbering;
  * point -> voxel look-ups against an existing voxel set (point_to_voxel: the point's own voxel; voxel_to_point: its
    8 corner voxels) address a dense cell table of the voxel set directly (`foreign_neighbor_map`: table built by M
    scattered writes, P x K reads, undone by M scattered zeros) instead of building and probing a hash table.

Point sets whose bounding box exceeds the dense-grid limit (index.GridTooLarge) take the hash-based op kernels, as
the aggregation path does (aggregate._generic_voxel_to_aux).
"""
from __future__ import annotations

import torch

from . import functional as F
from .functional import calc_ti_weights
from .index import BlockIndex, GridTooLarge, foreign_neighbor_map
from .tensor import PointTensor, SparseTensor
from .utils import get_kernel_offsets

__all__ = ["calc_ti_weights", "initial_voxelize", "point_to_voxel", "voxel_to_point"]


def _voxel_keys(pc: torch.Tensor, stride: int) -> torch.Tensor:
    """int32 [P,4] voxel coordinate (multiple of `stride`) + batch of float point coordinates."""
    return torch.cat([torch.floor(pc[:, :3] / stride).int() * stride, pc[:, -1].int().view(-1, 1)], 1)


def _bounds_of(x: SparseTensor):
    """Bounding box of x.C if some builder left it on the tensor's shared cache (checked ones only)."""
    key = ("link_bounds", x.C.data_ptr(), x.C.shape[0])
    if x.cmaps.get(("link_bounds_unchecked", x.C.data_ptr(), x.C.shape[0])):
        return None
    return x.cmaps.get(key)


def _lookup(keys: torch.Tensor, x: SparseTensor, r: int) -> torch.Tensor:
    """int64 [P, r^3]: row of x.C holding keys[p] + offset_k * x.stride (offset order of get_kernel_offsets(r)), -1
    if absent.  Dense cell table of x.C; hash table beyond the dense-grid limit."""
    try:
        st = int(x.s[0])
        if st > 1:
            # coordinates at tensor stride st are multiples of st: look up on the grid of SITES (coordinates / st, edge 1, step 1)
            # -- st^3 times fewer cells than an edge-1 grid over the raw coordinates, which hit the dense-grid limit (or a 1 GiB
            # table) at stride 4-8 on full LiDAR extents.  The divided rows of x.C are cached with the tensor's maps.
            ckey = ("link_site_rows", x.C.data_ptr(), x.C.shape[0], st)
            ent = x.cmaps.get(ckey)
            if ent is None:
                div = torch.tensor([st, st, st, 1], dtype=torch.int32, device=x.C.device)
                b = _bounds_of(x)
                sb = None if b is None else (tuple(int(v) // st for v in b[0][:3]) + (int(b[0][3]),),
                                             tuple(int(v) // st for v in b[1][:3]) + (int(b[1][3]),))
                ent = x.cmaps[ckey] = (torch.div(x.C, div, rounding_mode="floor").contiguous(), div, sb)
            rows, div, sb = ent
            return foreign_neighbor_map(torch.div(keys, div, rounding_mode="floor").contiguous(), r, step=1, table_rows=rows,
                                        bounds=sb).long()
        return foreign_neighbor_map(keys.contiguous(), r, step=st, table_rows=x.C, bounds=_bounds_of(x)).long()
    except GridTooLarge:
        if r == 1:
            return F.sphashquery(F.sphash(keys), F.sphash(x.C)).view(-1, 1)
        offs = get_kernel_offsets(r, x.s, 1, device=keys.device)
        return F.sphashquery(F.sphash(keys, offs), F.sphash(x.C.to(keys.device))).transpose(0, 1).contiguous()


def _voxelize_by_index(cells: torch.Tensor, feats: torch.Tensor):
    """(vox_feats, vox_coords, idx_query, counts) of utils.py:240-247 from one dense-grid index over the points."""
    from .aggregate import _BlockMean
    index = BlockIndex(cells, 1)                              # GridTooLarge: the caller's hash path
    m = index.M                                               # the one host round trip (sizes the voxel set)
    ucoords = index.blk_coords[:m]                            # unique cells, grid order
    order = torch.argsort(F.sphash(ucoords))                  # the reference's numbering: ascending hash of the M voxels
    rank = torch.empty(m, dtype=torch.long, device=cells.device)
    rank[order] = torch.arange(m, device=cells.device)
    idx_query = rank[index.idx_query]
    counts = index.counts[order].contiguous()
    vox_feats = _BlockMean.apply(feats, index)[order]         # means in id order inside a voxel: the same bits every run
    return vox_feats, ucoords[order].contiguous(), idx_query, counts, index


def initial_voxelize(z: PointTensor, init_res, after_res) -> SparseTensor:
    scaled = torch.cat([(z.C[:, :3] * init_res) / after_res, z.C[:, -1].view(-1, 1)], 1)
    cell = torch.floor(scaled)
    cells = cell.int().contiguous()
    out = None
    if cells.is_cuda and cells.shape[0] > 0:
        try:
            vox_feats, vox_coords, idx_query, counts, index = _voxelize_by_index(cells, z.F)
            out = SparseTensor(vox_feats, vox_coords, 1)
            # the voxel set's bounding box is the points': later map builders need not measure it
            out.cmaps.setdefault(("link_bounds", vox_coords.data_ptr(), vox_coords.shape[0]), index.bounds)
        except GridTooLarge:
            out = None
    if out is None:                                           # reference algorithm on the op kernels
        pc_hash = F.sphash(cells)
        vox_hash = torch.unique(pc_hash)                     # voxel numbering = ascending hash
        idx_query = F.sphashquery(pc_hash, vox_hash)
        counts = F.spcount(idx_query.int(), len(vox_hash))
        vox_coords = torch.round(F.spvoxelize(cell, idx_query, counts)).int()   # mean of identical cells = the cell
        out = SparseTensor(F.spvoxelize(z.F, idx_query, counts), vox_coords, 1)
    out.cmaps.setdefault(out.stride, out.coords)
    z.additional_features["idx_query"][1] = idx_query
    z.additional_features["counts"][1] = counts
    z.C = scaled
    return out


def point_to_voxel(x: SparseTensor, z: PointTensor) -> SparseTensor:
    cache = z.additional_features
    if cache is None or cache.get("idx_query") is None or cache["idx_query"].get(x.s) is None:
        idx_query = _lookup(_voxel_keys(z.C, x.s[0]), x, 1).view(-1)
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
        idx_query = _lookup(_voxel_keys(z.C, x.s[0]), x, 2)                      # [P, 8], corner k = 4 dx + 2 dy + dz
        weights = calc_ti_weights(z.C, idx_query.transpose(0, 1).contiguous(), scale=x.s[0]).transpose(0, 1).contiguous()
        if nearest:
            weights[:, 1:] = 0.0
            idx_query[:, 1:] = -1
        z.idx_query[x.s] = idx_query
        z.weights[x.s] = weights
    out = PointTensor(F.spdevoxelize(x.F, z.idx_query[x.s], z.weights[x.s]), z.C, idx_query=z.idx_query,
                      weights=z.weights)
    out.additional_features = z.additional_features
    return out
