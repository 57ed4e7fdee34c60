"""oracle/link_oracle.py -- CPU oracle for the LinK hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (link_amd/) never does; it is HIP-only and fails loudly without its extension.

Two layers:

* ctypes wrappers over oracle/link_oracle.c (plain C restatement of the reference's native ops,
  each C function cites the reference file:line it follows);
* a torch-CPU restatement of the Python layer of the path (voxel_to_aux / aux_to_voxel /
  ELKBlock / TSELKBlock), written so that it is differentiable where the reference is, for the
  gradient oracle ("Backward: torch.autograd over the torch restatement").

Reference citations are relative to /root/reference/.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liblink_oracle.so")
_SO_OMP = os.path.join(_HERE, "liblink_oracle_omp.so")
_SRC = os.path.join(_HERE, "link_oracle.c")
_libs = {False: None, True: None}
_omp = False


def build(force: bool = False) -> str:
    """Compile oracle/link_oracle.c with gcc (seconds): the scalar checker and its OpenMP twin (same source,
    -fopenmp -DLINK_ORACLE_OMP: pragmas at the reference's loop placement; bench.py's all-cores baseline)."""
    for so, extra in ((_SO, []), (_SO_OMP, ["-fopenmp", "-DLINK_ORACLE_OMP"])):
        if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(_SRC):
            subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared"] + extra + ["-o", so, _SRC, "-lm"])
    return _SO


def set_omp(flag: bool) -> None:
    """Route the native-op layer through the OpenMP build (bench.py's cpu_baseline leg only)."""
    global _omp
    _omp = bool(flag)


def lib() -> ctypes.CDLL:
    _lib = _libs[_omp]
    if _lib is None:
        build()
        _lib = _libs[_omp] = ctypes.CDLL(_SO_OMP if _omp else _SO)
        i64, p = ctypes.c_int64, ctypes.c_void_p
        _lib.oracle_hash.argtypes = [i64, p, p]
        _lib.oracle_kernel_hash.argtypes = [i64, i64, p, p, p, ctypes.c_int]
        _lib.oracle_hash_query.argtypes = [i64, p, i64, p, p, p]
        _lib.oracle_hash_query.restype = ctypes.c_int
        _lib.oracle_count.argtypes = [i64, p, ctypes.c_int32, p]
        _lib.oracle_voxelize_fwd.argtypes = [i64, i64, p, p, p, i64, p]
        _lib.oracle_voxelize_bwd.argtypes = [i64, i64, p, p, p, p]
        _lib.oracle_devoxelize_fwd.argtypes = [i64, i64, i64, p, p, p, p]
        _lib.oracle_devoxelize_bwd.argtypes = [i64, i64, i64, i64, p, p, p, p]
        _lib.oracle_block_coords.argtypes = [i64, p, ctypes.c_int32, p]
        _lib.oracle_unique_rows.argtypes = [i64, p, p]
        _lib.oracle_unique_rows.restype = i64
        _lib.oracle_voxel_to_aux_index.argtypes = [i64, p, ctypes.c_int32, p, p, p]
        _lib.oracle_voxel_to_aux_index.restype = i64
        _lib.oracle_neighbor_index.argtypes = [i64, p, i64, p, p]
        _lib.oracle_neighbor_index.restype = ctypes.c_int
        _lib.oracle_aux_to_voxel_feats.argtypes = [i64, i64, i64, i64, p, p, p, p, p]
        _lib.oracle_aux_to_voxel_feats.restype = ctypes.c_int
    return _lib


def _p(a: np.ndarray) -> int:
    return a.ctypes.data


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a), dtype=dtype)


# ------------------------------------------------------------------------------ native-op layer
def sphash(coords) -> np.ndarray:
    """torchsparse/nn/functional/hash.py:10-24 -> backend/hash/hash_cpu.cpp:7-18."""
    c = _c(coords, np.int32)
    assert c.ndim == 2 and c.shape[1] == 4
    out = np.empty(c.shape[0], np.int64)
    lib().oracle_hash(c.shape[0], _p(c), _p(out))
    return out


def sphash_offsets(coords, offsets, cpu_batch_bug: bool = False) -> np.ndarray:
    """hash.py:25-37 -> backend/hash/hash_cuda.cu:27-55; returns int64[K,N]."""
    c, o = _c(coords, np.int32), _c(offsets, np.int32)
    assert c.ndim == 2 and c.shape[1] == 4 and o.ndim == 2 and o.shape[1] == 3
    out = np.empty((o.shape[0], c.shape[0]), np.int64)
    lib().oracle_kernel_hash(c.shape[0], o.shape[0], _p(c), _p(o), _p(out), int(cpu_batch_bug))
    return out


def hash_query_backend(query, target, target_idx) -> np.ndarray:
    """backend/others/query_cpu.cpp:12-37: value = idx+1, 0 = miss."""
    q, t, ti = _c(query, np.int64).reshape(-1), _c(target, np.int64), _c(target_idx, np.int64)
    out = np.empty(q.shape[0], np.int64)
    rc = lib().oracle_hash_query(q.shape[0], _p(q), t.shape[0], _p(t), _p(ti), _p(out))
    assert rc == 0
    return out


def sphashquery(queries, references) -> np.ndarray:
    """torchsparse/nn/functional/query.py:8-33: index of the query hash in references or -1."""
    q = _c(queries, np.int64)
    r = _c(references, np.int64)
    out = hash_query_backend(q.reshape(-1), r, np.arange(r.shape[0], dtype=np.int64)) - 1
    return out.reshape(q.shape)


def spcount(idx, num: int) -> np.ndarray:
    """torchsparse/nn/functional/count.py:8-16 -> backend/others/count_cpu.cpp:7-23."""
    i = _c(idx, np.int32)
    out = np.empty(int(num), np.int32)
    lib().oracle_count(i.shape[0], _p(i), int(num), _p(out))
    return out


def spvoxelize_fwd(feats, idx, counts) -> np.ndarray:
    """voxelize.py:10-32 -> backend/voxelize/voxelize_cpu.cpp:7-25."""
    f, i, cn = _c(feats, np.float32), _c(idx, np.int32), _c(counts, np.int32)
    out = np.empty((cn.shape[0], f.shape[1]), np.float32)
    lib().oracle_voxelize_fwd(f.shape[0], f.shape[1], _p(f), _p(i), _p(cn), cn.shape[0], _p(out))
    return out


def spvoxelize_bwd(top, idx, counts, n: int) -> np.ndarray:
    """voxelize.py:34-52 -> backend/voxelize/voxelize_cpu.cpp:27-43."""
    t, i, cn = _c(top, np.float32), _c(idx, np.int32), _c(counts, np.int32)
    out = np.empty((int(n), t.shape[1]), np.float32)
    lib().oracle_voxelize_bwd(int(n), t.shape[1], _p(t), _p(i), _p(cn), _p(out))
    return out


def spdevoxelize_fwd(feats, ind, weights) -> np.ndarray:
    """devoxelize.py:51-73 -> backend/devoxelize/devoxelize_cuda.cu:11-34 (K = ind.shape[1])."""
    f, i, w = _c(feats, np.float32), _c(ind, np.int32), _c(weights, np.float32)
    out = np.empty((i.shape[0], f.shape[1]), np.float32)
    lib().oracle_devoxelize_fwd(i.shape[0], f.shape[1], i.shape[1], _p(i), _p(w), _p(f), _p(out))
    return out


def spdevoxelize_bwd(top, ind, weights, n: int) -> np.ndarray:
    """devoxelize.py:75-93 -> backend/devoxelize/devoxelize_cuda.cu:37-59."""
    t, i, w = _c(top, np.float32), _c(ind, np.int32), _c(weights, np.float32)
    out = np.empty((int(n), t.shape[1]), np.float32)
    lib().oracle_devoxelize_bwd(i.shape[0], int(n), t.shape[1], i.shape[1], _p(i), _p(w), _p(t), _p(out))
    return out


def unique_rows(rows) -> np.ndarray:
    """torch.unique(x, dim=0) at segmentation/core/models/utils.py:47."""
    r = _c(rows, np.int32)
    out = np.empty_like(r)
    m = lib().oracle_unique_rows(r.shape[0], _p(r), _p(out))
    return out[:m].copy()


def block_coords(coords, s: int) -> np.ndarray:
    """segmentation/core/models/utils.py:45."""
    c = _c(coords, np.int32)
    out = np.empty_like(c)
    lib().oracle_block_coords(c.shape[0], _p(c), int(s), _p(out))
    return out


def get_kernel_offsets(size: int) -> np.ndarray:
    """torchsparse/nn/utils/kernel.py:11-32 with stride = dilation = 1 (as aux_to_voxel calls it,
    segmentation/core/models/utils.py:65).  Odd volume: x fastest; even volume: z fastest."""
    ax = np.arange(-size // 2 + 1, size // 2 + 1)
    if (size ** 3) % 2 == 1:
        offs = [[x, y, z] for z in ax for y in ax for x in ax]
    else:
        offs = [[x, y, z] for x in ax for y in ax for z in ax]
    return np.asarray(offs, dtype=np.int32).reshape(-1, 3)


# ------------------------------------------------------------------------------ aggregation layer
def voxel_to_aux_index(coords, s: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Index half of voxel_to_aux (segmentation/core/models/utils.py:44-51).
    Returns (small_C int32[M,4], idx_query int64[N], counts int32[M])."""
    c = _c(coords, np.int32)
    n = c.shape[0]
    small = np.empty((max(n, 1), 4), np.int32)
    idx = np.empty(max(n, 1), np.int64)
    cnt = np.empty(max(n, 1), np.int32)
    m = lib().oracle_voxel_to_aux_index(n, _p(c), int(s), _p(small), _p(idx), _p(cnt))
    assert m >= 0
    return small[:m].copy(), idx[:n].copy(), cnt[:m].copy()


def neighbor_index(small_c, r: int) -> np.ndarray:
    """Neighbour map int32[M, r^3] (segmentation/core/models/utils.py:65-73)."""
    sc = _c(small_c, np.int32)
    off = get_kernel_offsets(r)
    out = np.empty((sc.shape[0], off.shape[0]), np.int32)
    rc = lib().oracle_neighbor_index(sc.shape[0], _p(sc), off.shape[0], _p(off), _p(out))
    assert rc == 0
    return out


def voxel_to_aux(feats, coords, s: int):
    """segmentation/core/models/utils.py:44-58 (== detection/.../ts_elk.py:68-81).
    Returns (small_F [M,W] block means, small_C, idx_query, counts)."""
    small_c, idx, counts = voxel_to_aux_index(coords, s)
    small_f = spvoxelize_fwd(feats, idx.astype(np.int32), counts)     # utils.py:52
    return small_f, small_c, idx, counts


def aux_to_voxel(small_f, small_c, idx, counts, r: int = 2) -> np.ndarray:
    """segmentation/core/models/utils.py:61-84 (== ts_elk.py:84-107 with r=3).  Returns F[N,W]."""
    sf, cn, ix = _c(small_f, np.float32), _c(counts, np.int32), _c(idx, np.int64)
    nbr = neighbor_index(small_c, r)
    out = np.empty((ix.shape[0], sf.shape[1]), np.float32)
    rc = lib().oracle_aux_to_voxel_feats(ix.shape[0], sf.shape[0], sf.shape[1], nbr.shape[1],
                                         _p(sf), _p(cn), _p(nbr), _p(ix), _p(out))
    assert rc == 0
    return out


def aggregate(feats, coords, s: int, r: int) -> np.ndarray:
    """voxel_to_aux followed by aux_to_voxel: X[N,W] -> mean of X over the r^3 neighbour blocks."""
    sf, sc, idx, cnt = voxel_to_aux(feats, coords, s)
    return aux_to_voxel(sf, sc, idx, cnt, r)


# ------------------------------------------------------------------------------ block layer (torch)
def theta_torch(coords_t, pos_weight, baseop: str, groups: int, alpha=None, variant: str = "unet",
                tensor_stride: int = 1):
    """theta = pos_weight(coords.float()) with the variant-specific tiling.

    unet   : segmentation/core/models/semantic_kitti/linkunet.py:137-138,151-152,164-165
             Linear(3, C/g) then repeat([1,g]) for 'sin'/'cos'; cos_x multiplies by alpha, no tiling.
    encoder: linkencoder.py:165 -- cos_x feeds coords/stride.
    det    : detection/det3d/models/utils/ts_elk.py:167-168 -- Linear(3, C), first C/2 columns
             tiled twice ('cos'); 'sin' uses all C columns untiled (:156).
    """
    import torch
    xyz = coords_t[:, :3].to(pos_weight.dtype)     # .float() in the reference; fp64 weights -> fp64 check runs
    if variant == "encoder" and baseop == "cos_x":
        xyz = xyz / tensor_stride
    th = torch.nn.functional.linear(xyz, pos_weight)
    if variant == "det":
        if baseop == "cos":
            th = th[:, : pos_weight.shape[0] // 2].repeat([1, 2])
        return th
    if baseop == "cos_x":
        return th * alpha
    return th.repeat([1, groups])


def elk_core_torch(feats, coords, params: dict, s: int, r: int, baseop: str = "cos", groups: int = 1,
                   variant: str = "unet", tensor_stride: int = 1, agg=None):
    """R_core of ELKBlock.forward: pre_mix -> theta -> modulate ->
    voxel_to_aux -> aux_to_voxel -> demodulate -> norm, i.e. linkunet.py:132,135-176,178
    (ts_elk.py:152-172,224 for variant='det') WITHOUT local_mix and the final add/ReLU.

    `agg(X, coords, s, r)` computes the aggregation; default is the differentiable torch
    restatement below.  params keys follow the reference state_dict: 'pre_mix.0.weight',
    'pre_mix.1.weight', 'pre_mix.1.bias', 'pos_weight.0.weight', 'alpha', 'norm.weight', 'norm.bias'.
    """
    import torch
    import torch.nn.functional as TF
    C = feats.shape[1]
    fin = TF.layer_norm(TF.linear(feats, params["pre_mix.0.weight"]), (C,),
                        params["pre_mix.1.weight"], params["pre_mix.1.bias"], 1e-6)
    th = theta_torch(coords, params["pos_weight.0.weight"], baseop, groups, params.get("alpha"),
                     variant, tensor_stride)
    sin, cos = torch.sin(th), torch.cos(th)
    if agg is None:
        agg = aggregate_torch
    if baseop == "sin":
        x = torch.cat([fin * sin, fin * cos], dim=1).contiguous()
        v = agg(x, coords, s, r)
        new = v[:, :C] * cos - v[:, C:] * sin
    elif baseop == "cos":
        x = torch.cat([fin * cos, fin * sin], dim=1).contiguous()
        v = agg(x, coords, s, r)
        new = v[:, :C] * cos + v[:, C:] * sin
    elif baseop == "cos_x":
        lin = fin * th
        x = torch.cat([fin * cos, fin * sin, lin], dim=1).contiguous()
        v = agg(x, coords, s, r)
        new = v[:, :C] * cos + v[:, C:2 * C] * sin + (v[:, 2 * C:] - lin)
    else:
        raise ValueError(baseop)
    return TF.layer_norm(new, (C,), params["norm.weight"], params["norm.bias"], 1e-6)


def aggregate_torch(x, coords, s: int, r: int):
    """Differentiable torch restatement of voxel_to_aux + aux_to_voxel (utils.py:44-84) using the
    C oracle for the integer index structures and index_add / gather for the features."""
    import torch
    small_c, idx, counts = voxel_to_aux_index(coords.numpy(), s)
    nbr = torch.from_numpy(neighbor_index(small_c, r).astype(np.int64))
    idx_t = torch.from_numpy(idx)
    cnt = torch.from_numpy(counts).to(x.dtype)
    m = small_c.shape[0]
    mean = torch.zeros(m, x.shape[1], dtype=x.dtype).index_add_(0, idx_t, x / cnt[idx_t][:, None])
    f = torch.cat([mean, torch.ones_like(mean[:, :1])], dim=1) * cnt[:, None]      # utils.py:75-76
    w = (nbr != -1).to(x.dtype)                                                     # utils.py:77-78
    new = (f[nbr.clamp(min=0)] * w[..., None]).sum(1)                               # devoxelize_cuda.cu:21-33
    new = new[:, :-1] / new[:, -1:]                                                 # utils.py:80
    return new[idx_t]                                                               # utils.py:82


def conv_neighbor_table(coords, kernel_size: int = 3, tensor_stride: int = 1) -> np.ndarray:
    """Kernel map of a stride-1 sparse convolution as a per-output table int32[N, K]: row of
    coords[i] + offset_k * tensor_stride, -1 absent (torchsparse/nn/functional/conv.py:103-113:
    offsets = get_kernel_offsets(kernel_size, stride=input.stride); queries = sphash(coords, offsets);
    results = sphashquery(queries, sphash(coords)) -> [K, N])."""
    c = _c(coords, np.int32)
    off = get_kernel_offsets(kernel_size) * int(tensor_stride)
    res = sphashquery(sphash_offsets(c, off), sphash(c))             # [K, N]
    return np.ascontiguousarray(res.T).astype(np.int32)


def subm_conv_torch(feats, coords, kernel, tensor_stride: int = 1):
    """Stride-1 sparse convolution forward as the reference's CPU branch computes it
    (conv.py:47-61: per kernel offset `output[out_map] += input[in_map] @ weight[k]`), differentiable."""
    import torch
    k = kernel.shape[0]
    ks = round(k ** (1 / 3))
    nbr = torch.from_numpy(conv_neighbor_table(coords.numpy() if hasattr(coords, "numpy") else coords, ks,
                                               tensor_stride).astype(np.int64))
    out = torch.zeros(feats.shape[0], kernel.shape[2], dtype=feats.dtype)
    for j in range(k):
        out_map = torch.nonzero(nbr[:, j] >= 0).view(-1)
        if out_map.numel():
            out = out.index_add(0, out_map, feats[nbr[out_map, j]] @ kernel[j])
    return out


def aggregate_c(x, coords, s: int, r: int):
    """Same as aggregate_torch but through the scalar C oracle (no autograd)."""
    import torch
    return torch.from_numpy(aggregate(x.detach().numpy(), coords.numpy(), s, r))


# ------------------------------------------------------------------------------ point <-> voxel (row N4)
def calc_ti_weights_np(points, idx_query, scale=1):
    """torchsparse/nn/functional/devoxelize.py:10-48: trilinear weights [8, P]; idx_query int64[8, P]."""
    p = np.asarray(points, np.float32)[:, :3]
    pf = (np.floor(p / np.float32(scale)) * np.float32(scale)).astype(np.float32) if scale != 1 else np.floor(p)
    pc = pf + np.float32(scale)
    near, far = pc - p, p - pf
    rows = []
    for dx in (0, 1):
        for dy in (0, 1):
            for dz in (0, 1):
                rows.append((far[:, 0] if dx else near[:, 0]) * (far[:, 1] if dy else near[:, 1])
                            * (far[:, 2] if dz else near[:, 2]))
    w = np.stack(rows, 0).astype(np.float32)
    if scale != 1:
        w /= np.float32(scale ** 3)
    w[np.asarray(idx_query) == -1] = 0
    w /= w.sum(0) + np.float32(1e-8)
    return w


def _voxel_keys_np(points, stride):
    pts = np.asarray(points, np.float32)
    return np.concatenate([(np.floor(pts[:, :3] / np.float32(stride)).astype(np.int32) * stride),
                           pts[:, 3:].astype(np.int32)], 1).astype(np.int32)


def initial_voxelize_np(points, feats, init_res, after_res):
    """segmentation/core/models/utils.py:234-254.  Returns (vox_F, vox_C int32, idx_query int64, counts
    int32, z_C) -- voxels numbered by ascending coordinate hash (torch.unique of the hashes)."""
    pts = np.asarray(points, np.float32)
    scaled = np.concatenate([(pts[:, :3] * np.float32(init_res)) / np.float32(after_res), pts[:, 3:]], 1).astype(np.float32)
    cell = np.floor(scaled)
    pc_hash = sphash(cell.astype(np.int32))
    vox_hash = np.unique(pc_hash)
    idx = sphashquery(pc_hash, vox_hash)
    counts = spcount(idx.astype(np.int32), len(vox_hash))
    vox_c = np.round(spvoxelize_fwd(cell, idx.astype(np.int32), counts)).astype(np.int32)
    vox_f = spvoxelize_fwd(np.asarray(feats, np.float32), idx.astype(np.int32), counts)
    return vox_f, vox_c, idx, counts, scaled


def point_to_voxel_np(points_scaled, feats, vox_c, stride=1):
    """utils.py:259-281 (uncached branch)."""
    idx = sphashquery(sphash(_voxel_keys_np(points_scaled, stride)), sphash(vox_c))
    counts = spcount(idx.astype(np.int32), vox_c.shape[0])
    return spvoxelize_fwd(np.asarray(feats, np.float32), idx.astype(np.int32), counts), idx, counts


def voxel_to_point_np(vox_f, vox_c, stride, points_scaled, nearest=False):
    """utils.py:286-324 (uncached branch).  Returns (point feats, idx_query int64[P,8], weights fp32[P,8])."""
    off = get_kernel_offsets(2) * int(stride)
    idx = sphashquery(sphash_offsets(_voxel_keys_np(points_scaled, stride), off), sphash(vox_c))      # [8, P]
    w = calc_ti_weights_np(points_scaled, idx, stride).T.copy()
    idx = np.ascontiguousarray(idx.T)
    if nearest:
        w[:, 1:] = 0.0
        idx[:, 1:] = -1
    return spdevoxelize_fwd(np.asarray(vox_f, np.float32), idx.astype(np.int32), w), idx, w



# ------------------------------------------------------------------------------ strided sparse conv (row N1)
def downsample_coords(coords, stride: int = 2, tensor_stride: int = 1) -> np.ndarray:
    """torchsparse/nn/functional/downsample.py:26-49 for kernel_size == stride: floor to multiples of
    stride*tensor_stride, unique rows ordered by (batch, x, y, z)."""
    c = _c(coords, np.int32).copy()
    ss = int(stride) * int(tensor_stride)
    c[:, :3] = np.floor_divide(c[:, :3], ss) * ss
    u = np.unique(c[:, [3, 0, 1, 2]], axis=0)
    return np.ascontiguousarray(u[:, [1, 2, 3, 0]]).astype(np.int32)


def strided_conv_table(in_coords, out_coords, kernel_size: int = 2, tensor_stride: int = 1) -> np.ndarray:
    """conv.py:105-113 for a down-sampling convolution: int32[N_out, K], entry = input row at
    out_coords[j] + offset_k * tensor_stride (offsets of get_kernel_offsets(kernel_size, stride=input.stride))."""
    off = get_kernel_offsets(kernel_size) * int(tensor_stride)
    res = sphashquery(sphash_offsets(_c(out_coords, np.int32), off), sphash(_c(in_coords, np.int32)))   # [K, N_out]
    return np.ascontiguousarray(res.T).astype(np.int32)


def gather_conv_torch(feats, table, kernel, n_out=None, transposed_of=None):
    """conv.py:47-61 (`output[out_map] += input[in_map] @ weight[k]`).  `table` int[N_out,K] per-output table.
    transposed_of: the down-conv's table [N_coarse,K] -- computes the TRANSPOSED convolution (in/out maps
    swapped, conv.py:56-57) producing n_out fine rows from coarse `feats`."""
    import torch
    k = kernel.shape[0]
    if transposed_of is not None:
        t = torch.from_numpy(np.asarray(transposed_of).astype(np.int64))
        out = torch.zeros(int(n_out), kernel.shape[2], dtype=feats.dtype)
        for j in range(k):
            rows = torch.nonzero(t[:, j] >= 0).view(-1)                     # coarse rows having a fine child at offset j
            if rows.numel():
                out = out.index_add(0, t[rows, j], feats[rows] @ kernel[j])
        return out
    t = torch.from_numpy(np.asarray(table).astype(np.int64))
    out = torch.zeros(t.shape[0], kernel.shape[2], dtype=feats.dtype)
    for j in range(k):
        rows = torch.nonzero(t[:, j] >= 0).view(-1)
        if rows.numel():
            out = out.index_add(0, rows, feats[t[rows, j]] @ kernel[j])
    return out
