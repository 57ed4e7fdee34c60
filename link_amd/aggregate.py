"""link_amd/aggregate.py -- LinK's block aggregation behind the reference's Python surface.

Drop-in replacements (same names, arguments, return tuples, dtypes and in-place behaviour) for
  voxel_to_aux(large_x, s) / aux_to_voxel(small_x, large_x, idx, counts, r=2)
      /root/reference/segmentation/core/models/utils.py:44-84
  large_to_small(large_x, stride) / small_to_large_v2(small_x, large_x, idx, counts)   (r = 3)
      /root/reference/detection/det3d/models/utils/ts_elk.py:68-107
computed by the HIP library: the index comes from BlockIndex (dense grid, bit-exact with the
reference's hash/unique/query chain), features from deterministic segmented reductions.  Both are
differentiable (custom autograd Functions with HIP backward kernels).

When the frame's block grid would exceed the dense-grid limit (index.MAX_CELLS) the same results are
produced by the generic path that mirrors the reference step by step on the HIP op kernels
(sphash / torch.unique / sphashquery / spcount / spvoxelize / spdevoxelize).
"""
from __future__ import annotations

import ctypes

import torch
from torch.autograd import Function

from . import _lib as L
from . import functional as F
from .index import BlockIndex, GridTooLarge, foreign_neighbor_map
from .tensor import SparseTensor
from .utils import get_kernel_offsets

__all__ = ["voxel_to_aux", "aux_to_voxel", "large_to_small", "small_to_large_v2", "link_index_of",
           "upsample_voxel"]


def _st():
    return L.current_stream_handle()


class _BlockMean(Function):
    """Indexed spvoxelize: forward link_block_mean, backward link_voxelize_backward."""

    @staticmethod
    def forward(ctx, feats, index: BlockIndex):
        feats = feats.contiguous().float()
        n, c = feats.shape
        m = index.M
        out = torch.empty((m, c), dtype=torch.float32, device=feats.device)
        L.check(L.lib().link_block_mean(feats.data_ptr(), index.perm.data_ptr(), index.blk_start.data_ptr(),
                                        index.hdr.data_ptr(), n, c, m, out.data_ptr(), _st()), "link_block_mean")
        ctx.index = index
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        index = ctx.index
        g = g.contiguous().float()
        c = g.shape[1]
        out = torch.empty((ctx.n, c), dtype=torch.float32, device=g.device)
        L.check(L.lib().link_voxelize_backward(g.data_ptr(), index.vox_blk.data_ptr(),
                                               index.counts_buf.data_ptr(), ctx.n, c, out.data_ptr(), _st()),
                "link_voxelize_backward")
        return out, None


class _AuxToVoxel(Function):
    """aux_to_voxel feature half: forward link_aux_to_voxel_forward, backward ..._backward."""

    @staticmethod
    def forward(ctx, small_f, counts, nbr, idx, index, r):
        small_f = small_f.contiguous().float()
        m, c = small_f.shape
        n = idx.shape[0]
        k = r ** 3
        dev = small_f.device
        denom = torch.empty(m, dtype=torch.float32, device=dev)
        # rows of voxels the index dropped (outside caller-supplied bounds, hdr[STATUS]) are never written: zeros then
        out = (torch.empty if getattr(index, "rows_checked", True) else torch.zeros)((n, c), dtype=torch.float32, device=dev)
        if (c % 8 == 0 or c % 12 == 0) and r <= 3 and c <= 512 and m > 0 and (m + 1) * (c + 1) * 4 < 2 ** 32:
            # dense-grid form: the fused path's block-gather kernel (no neighbour table needed), which also hands each
            # block's row to the block's voxels (`new_feat[idx]`, utils.py:84, without new_feat)
            S = torch.empty((m + 1) * (c + 1), dtype=torch.float32, device=dev)
            L.check(L.lib().link_aux_to_voxel_forward_scatter(
                small_f.data_ptr(), counts.data_ptr(), index.blk_coords.data_ptr(), index.cell_blk.data_ptr(),
                ctypes.byref(index.grid), index.hdr.data_ptr(), index.blk_start.data_ptr(), index.perm.data_ptr(), n, m, c, r,
                S.data_ptr(), denom.data_ptr(), out.data_ptr(), _st()), "link_aux_to_voxel_forward_scatter")
        else:
            new_feat = torch.empty((m, c), dtype=torch.float32, device=dev)
            nbr = index.neighbor_map(r)
            k = nbr.shape[1]
            L.check(L.lib().link_aux_to_voxel_forward(small_f.data_ptr(), counts.data_ptr(), nbr.data_ptr(),
                                                      idx.data_ptr(), n, m, c, k, new_feat.data_ptr(),
                                                      denom.data_ptr(), out.data_ptr(), _st()),
                    "link_aux_to_voxel_forward")
        ctx.index, ctx.r = index, r
        ctx.save_for_backward(counts, denom)
        ctx.shape = (n, m, c, k)
        return out

    @staticmethod
    def backward(ctx, g):
        index, r = ctx.index, ctx.r
        counts, denom = ctx.saved_tensors
        n, m, c, k = ctx.shape
        g = g.contiguous().float()
        nbr_t = index.neighbor_map(r, transpose=True)
        g_new = torch.empty((m, c), dtype=torch.float32, device=g.device)
        g_small = torch.empty((m, c), dtype=torch.float32, device=g.device)
        L.check(L.lib().link_aux_to_voxel_backward(g.data_ptr(), index.perm.data_ptr(),
                                                   index.blk_start.data_ptr(), counts.data_ptr(),
                                                   nbr_t.data_ptr(), denom.data_ptr(), n, m, c, k,
                                                   g_new.data_ptr(), g_small.data_ptr(), _st()),
                "link_aux_to_voxel_backward")
        return g_small, None, None, None, None, None


def link_index_of(st: SparseTensor, s: int) -> BlockIndex:
    """BlockIndex of `st.C` for block edge `s`, cached on the tensor's shared kmaps dict (the same
    place the reference caches its convolution kernel maps, nn/functional/conv.py:103,122) and keyed
    by the coords storage so a derived tensor with the same coordinates reuses it."""
    key = ("link_block_index", st.C.data_ptr(), st.C.shape[0], int(s))
    idx = st.kmaps.get(key)
    if idx is None:
        bounds = st.cmaps.get(("link_bounds", st.C.data_ptr(), st.C.shape[0]))
        idx = BlockIndex(st.C, int(s), bounds=bounds)
        # bounds computed from these coordinates (coords_bounds) cannot drop a voxel; bounds from the caller's
        # metadata can (hdr[STATUS]) -- consumers then zero-fill their output rows
        idx.rows_checked = bounds is None or not st.cmaps.get(("link_bounds_unchecked", st.C.data_ptr(), st.C.shape[0]))
        st.cmaps.setdefault(("link_bounds", st.C.data_ptr(), st.C.shape[0]), idx.bounds)
        st.kmaps[key] = idx
    return idx


def _generic_voxel_to_aux(large_x, s):
    """Reference algorithm verbatim on the HIP op kernels (used only beyond the dense-grid limit)."""
    x_C = torch.cat([torch.div(large_x.C[:, :3], s, rounding_mode="floor").int(), large_x.C[:, 3:]], dim=1)
    large_hash = F.sphash(x_C)
    small_C = torch.unique(x_C, dim=0)
    small_hash = F.sphash(small_C)
    idx_query = F.sphashquery(large_hash, small_hash)
    counts = F.spcount(idx_query.int(), len(small_hash))
    feat = F.spvoxelize(large_x.F, idx_query, counts)
    return feat, small_C, idx_query, counts, None


def voxel_to_aux(large_x: SparseTensor, s: int):
    """utils.py:44-58.  Returns (small_x, idx_query int64[N], counts int32[M]); small_x.F are the
    block MEANS, small_x.C the sorted unique block coordinates, small_x.s = s; cmaps/kmaps shared."""
    s = int(s)
    try:
        index = link_index_of(large_x, s)
        feat = _BlockMean.apply(large_x.F, index)
        small_C, idx_query, counts = index.block_coords, index.idx_query, index.counts
    except GridTooLarge:
        feat, small_C, idx_query, counts, index = _generic_voxel_to_aux(large_x, s)
    small_x = SparseTensor(feat, small_C, s)
    small_x.cmaps = large_x.cmaps
    small_x.kmaps = large_x.kmaps
    small_x._link_index = index
    return small_x, idx_query, counts


def aux_to_voxel(small_x: SparseTensor, large_x: SparseTensor, idx: torch.Tensor, counts: torch.Tensor,
                 r: int = 2) -> SparseTensor:
    """utils.py:61-84.  Overwrites large_x.F with the mean over the r^3 neighbour blocks and returns
    the same large_x object."""
    r = int(r)
    index = getattr(small_x, "_link_index", None)
    if index is not None and idx is index.idx_query and small_x.C.data_ptr() == index.blk_coords.data_ptr():
        large_x.F = _AuxToVoxel.apply(small_x.F, counts.contiguous().int(), None, idx, index, r)
        return large_x
    # foreign inputs: reference algorithm on the op kernels (neighbour map from the dense table when the
    # rows fit, else hash queries)
    try:
        idx_query = foreign_neighbor_map(small_x.C, r).long()
    except GridTooLarge:
        offsets = get_kernel_offsets(r, 1, 1, device=large_x.F.device)
        idx_query = F.sphashquery(F.sphash(small_x.C, offsets), F.sphash(small_x.C)).transpose(0, 1).contiguous()
    f = torch.cat([small_x.F, torch.ones_like(small_x.F[:, :1])], dim=1) * counts.unsqueeze(dim=-1)
    weights = (idx_query != -1).float()
    new_feat = F.spdevoxelize(f, idx_query, weights, r)
    new_feat = new_feat[:, :-1] / new_feat[:, -1:]
    large_x.F = new_feat[idx]
    return large_x


def large_to_small(large_x: SparseTensor, stride: int):
    """detection/det3d/models/utils/ts_elk.py:68-81 (identical to voxel_to_aux)."""
    return voxel_to_aux(large_x, stride)


def small_to_large_v2(small_x: SparseTensor, large_x: SparseTensor, idx: torch.Tensor, counts: torch.Tensor):
    """ts_elk.py:84-107: aux_to_voxel with the neighbourhood hard-wired to 3^3."""
    return aux_to_voxel(small_x, large_x, idx, counts, 3)


def upsample_voxel(x: SparseTensor, ref_x: SparseTensor) -> SparseTensor:
    """/root/reference/segmentation/core/models/utils.py:327-340: give every fine voxel of `ref_x` the
    features of the coarse voxel of `x` that contains it (parent = floor(coord / x.stride)); a fine voxel
    whose parent is absent gets x.F[-1] exactly as the reference's `x.F[idx_query]` with idx -1 does."""
    stride = x.s[0]
    x_C = torch.cat([torch.div(x.C[:, :3], stride, rounding_mode="floor").int(), x.C[:, 3:]], dim=1)
    ref_x_C = torch.cat([torch.div(ref_x.C[:, :3], stride, rounding_mode="floor").int(), ref_x.C[:, 3:]], dim=1)
    # the fine voxel's parent among the coarse voxels: one look-up per fine voxel in a dense cell table of the coarse set
    # (hash table beyond the dense-grid limit)
    bounds = None
    cb = x.cmaps.get(("link_bounds", x.C.data_ptr(), x.C.shape[0]))
    if cb is not None and not x.cmaps.get(("link_bounds_unchecked", x.C.data_ptr(), x.C.shape[0])):
        bounds = (tuple(v // stride for v in cb[0][:3]) + (cb[0][3],), tuple(v // stride for v in cb[1][:3]) + (cb[1][3],))
    try:
        idx_query = foreign_neighbor_map(ref_x_C.contiguous(), 1, table_rows=x_C.contiguous(), bounds=bounds).view(-1).long()
    except GridTooLarge:
        idx_query = F.sphashquery(F.sphash(ref_x_C), F.sphash(x_C))
    new_tensor = SparseTensor(x.F[idx_query], ref_x.C, ref_x.s)
    new_tensor.cmaps.setdefault(new_tensor.stride, new_tensor.coords)
    return new_tensor
