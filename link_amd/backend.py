"""link_amd/backend.py -- the `torchsparse.backend` surface on top of liblink_amd.so.

Same function names, argument order and return conventions as the reference's pybind11 module
(/root/reference/segmentation/torchsparse-u/torchsparse/backend/pybind_cuda.cpp:18-39) for the
functions on the LinK path; tensors in, freshly allocated tensors out, launched on torch's current
HIP stream.  There are no `_cpu` twins: CPU tensors raise (the product has no CPU path).
"""
from __future__ import annotations

import torch

from . import _lib as L


def _stream() -> int:
    return L.current_stream_handle()


def _need_gpu(*ts):
    for t in ts:
        if t.device.type != "cuda":
            raise L.LinkAmdError("link_amd operates on GPU tensors only (HIP path; no CPU fallback); got "
                                 f"a tensor on {t.device}")


def _p(t):
    return t.data_ptr() if t is not None else None


def hash_cuda(idx: torch.Tensor) -> torch.Tensor:
    """hash_cuda(idx[N,4] i32) -> i64[N]  (backend/hash/hash_cuda.cu:67-73)."""
    _need_gpu(idx)
    if idx.dtype != torch.int32 or idx.ndim != 2 or idx.shape[1] != 4:
        raise ValueError(f"hash_cuda expects int32[N,4], got {idx.dtype} {tuple(idx.shape)}")
    idx = idx.contiguous()
    out = torch.empty(idx.shape[0], dtype=torch.int64, device=idx.device)
    L.check(L.lib().link_hash(_p(idx), idx.shape[0], _p(out), _stream()), "hash_cuda")
    return out


def kernel_hash_cuda(idx: torch.Tensor, kernel_offset: torch.Tensor) -> torch.Tensor:
    """kernel_hash_cuda(idx[N,4] i32, off[K,3] i32) -> i64[K,N]  (hash_cuda.cu:75-84)."""
    _need_gpu(idx, kernel_offset)
    if idx.dtype != torch.int32 or idx.ndim != 2 or idx.shape[1] != 4:
        raise ValueError(f"kernel_hash_cuda expects int32[N,4], got {idx.dtype} {tuple(idx.shape)}")
    if kernel_offset.dtype != torch.int32 or kernel_offset.ndim != 2 or kernel_offset.shape[1] != 3:
        raise ValueError("kernel_hash_cuda expects int32[K,3] offsets")
    idx, kernel_offset = idx.contiguous(), kernel_offset.contiguous()
    out = torch.empty((kernel_offset.shape[0], idx.shape[0]), dtype=torch.int64, device=idx.device)
    L.check(L.lib().link_kernel_hash(_p(idx), idx.shape[0], _p(kernel_offset), kernel_offset.shape[0],
                                     _p(out), _stream()), "kernel_hash_cuda")
    return out


def hash_query_cuda(hash_query: torch.Tensor, hash_target: torch.Tensor,
                    idx_target: torch.Tensor) -> torch.Tensor:
    """hash_query_cuda(q i64[n1], tgt i64[n], tgt_idx i64[n]) -> i64[n1]: idx+1 or 0
    (backend/others/query_cuda.cu:9-58)."""
    _need_gpu(hash_query, hash_target, idx_target)
    for t in (hash_query, hash_target, idx_target):
        if t.dtype != torch.int64:
            raise ValueError("hash_query_cuda expects int64 tensors")
    q, t, ti = hash_query.contiguous().view(-1), hash_target.contiguous(), idx_target.contiguous()
    out = torch.empty(q.shape[0], dtype=torch.int64, device=q.device)
    nbytes = L.lib().link_hash_query_workspace_bytes(t.shape[0])
    ws = torch.empty(nbytes, dtype=torch.uint8, device=q.device)
    L.check(L.lib().link_hash_query(_p(q), q.shape[0], _p(t), _p(ti), t.shape[0], _p(out), _p(ws), nbytes,
                                    _stream()), "hash_query_cuda")
    return out


def count_cuda(idx: torch.Tensor, s: int) -> torch.Tensor:
    """count_cuda(idx i32[N], s) -> i32[s]  (backend/others/count_cuda.cu:21-31)."""
    _need_gpu(idx)
    if idx.dtype != torch.int32:
        raise ValueError("count_cuda expects int32 indices")
    idx = idx.contiguous()
    out = torch.empty(int(s), dtype=torch.int32, device=idx.device)
    L.check(L.lib().link_count(_p(idx), idx.numel(), _p(out), int(s), _stream()), "count_cuda")
    return out


_FLOAT_TYPES = (torch.float32, torch.float16, torch.bfloat16, torch.float64)


def _f32(t, name):
    """Features may arrive as half / bfloat16 / double (the reference dispatches
    AT_DISPATCH_FLOATING_TYPES_AND_HALF, voxelize_cuda.cu:52): computed in fp32 here, results are
    returned in the caller's dtype."""
    if t.dtype not in _FLOAT_TYPES:
        raise ValueError(f"{name}: floating-point features expected, got {t.dtype}")
    return t.contiguous().float()


def voxelize_forward_cuda(inputs: torch.Tensor, idx: torch.Tensor, counts: torch.Tensor) -> torch.Tensor:
    """voxelize_forward_cuda(in fp[N,c], idx i32[N], counts i32[N1]) -> fp[N1,c]
    (backend/voxelize/voxelize_cuda.cu:44-61)."""
    _need_gpu(inputs, idx, counts)
    dt = inputs.dtype
    inputs = _f32(inputs, "voxelize_forward_cuda")
    idx, counts = idx.contiguous(), counts.contiguous()
    if idx.dtype != torch.int32 or counts.dtype != torch.int32:
        raise ValueError("voxelize_forward_cuda expects int32 idx/counts")
    n, c = inputs.shape
    n1 = counts.shape[0]
    out = torch.empty((n1, c), dtype=torch.float32, device=inputs.device)
    L.check(L.lib().link_voxelize_forward(_p(inputs), _p(idx), _p(counts), n, c, n1, _p(out), _stream()),
            "voxelize_forward_cuda")
    return out.to(dt)


def voxelize_backward_cuda(top_grad: torch.Tensor, idx: torch.Tensor, counts: torch.Tensor,
                           N: int) -> torch.Tensor:
    """voxelize_backward_cuda(top fp[N1,c], idx, counts, N) -> fp[N,c]  (voxelize_cuda.cu:63-80)."""
    _need_gpu(top_grad, idx, counts)
    dt = top_grad.dtype
    top_grad = _f32(top_grad, "voxelize_backward_cuda")
    idx, counts = idx.contiguous(), counts.contiguous()
    c = top_grad.shape[1]
    out = torch.empty((int(N), c), dtype=torch.float32, device=top_grad.device)
    L.check(L.lib().link_voxelize_backward(_p(top_grad), _p(idx), _p(counts), int(N), c, _p(out), _stream()),
            "voxelize_backward_cuda")
    return out.to(dt)


def devoxelize_forward_cuda(feat: torch.Tensor, indices: torch.Tensor, weight: torch.Tensor,
                            r: int) -> torch.Tensor:
    """devoxelize_forward_cuda(feat fp[n,c], ind i32[N,r^3], w fp[N,r^3], r) -> fp[N,c]
    (backend/devoxelize/devoxelize_cuda.cu:63-81)."""
    _need_gpu(feat, indices, weight)
    dt = feat.dtype
    feat, weight = _f32(feat, "devoxelize_forward_cuda"), _f32(weight, "devoxelize_forward_cuda")
    indices = indices.contiguous()
    if indices.dtype != torch.int32 or indices.ndim != 2 or indices.shape[1] != int(r) ** 3:
        raise ValueError(f"devoxelize_forward_cuda expects int32[N,{int(r) ** 3}] indices")
    nq, k = indices.shape
    c = feat.shape[1]
    out = torch.empty((nq, c), dtype=torch.float32, device=feat.device)
    L.check(L.lib().link_devoxelize_forward(_p(feat), _p(indices), _p(weight), nq, c, k, _p(out), _stream()),
            "devoxelize_forward_cuda")
    return out.to(dt)


def devoxelize_backward_cuda(top_grad: torch.Tensor, indices: torch.Tensor, weight: torch.Tensor,
                             n: int, r: int) -> torch.Tensor:
    """devoxelize_backward_cuda(top fp[N,c], ind, w, n, r) -> fp[n,c]  (devoxelize_cuda.cu:85-101)."""
    _need_gpu(top_grad, indices, weight)
    dt = top_grad.dtype
    top_grad, weight = _f32(top_grad, "devoxelize_backward_cuda"), _f32(weight, "devoxelize_backward_cuda")
    indices = indices.contiguous()
    nq, k = indices.shape
    c = top_grad.shape[1]
    out = torch.empty((int(n), c), dtype=torch.float32, device=top_grad.device)
    L.check(L.lib().link_devoxelize_backward(_p(top_grad), _p(indices), _p(weight), nq, int(n), c, k,
                                             _p(out), _stream()), "devoxelize_backward_cuda")
    return out.to(dt)


__all__ = ["hash_cuda", "kernel_hash_cuda", "hash_query_cuda", "count_cuda", "voxelize_forward_cuda",
           "voxelize_backward_cuda", "devoxelize_forward_cuda", "devoxelize_backward_cuda"]


# ------------------------------------------------------------------------------------------------
# convolution_{forward,backward}_cuda (pybind_cuda.cpp:19-21; convolution_cuda.cu:53-278)
# ------------------------------------------------------------------------------------------------
def _tables_of(neighbor_map: torch.Tensor, neighbor_offset: torch.Tensor, n_in: int, n_out: int, transpose: bool):
    """The reference's kernel map -- pairs (in, out) int32[L, 2] grouped by kernel offset, sizes int32[K] on the
    host (nn/functional/conv.py:109-122) -- as link_amd's per-output tables: fwd[out, k] = in, back[in, k] = out
    (cached on the map tensor; `transpose` swaps the roles of the two columns, convolution_cuda.cu:117-131)."""
    key = "_link_tables_t" if transpose else "_link_tables"
    hit = getattr(neighbor_map, key, None)
    if hit is None:
        sizes = neighbor_offset.to(device=neighbor_map.device, dtype=torch.int64)
        k = sizes.numel()
        kk = torch.repeat_interleave(torch.arange(k, device=neighbor_map.device), sizes)
        src = neighbor_map[:, 1 if transpose else 0].long()
        dst = neighbor_map[:, 0 if transpose else 1].long()
        fwd = torch.full((n_out, k), -1, dtype=torch.int32, device=neighbor_map.device)
        fwd[dst, kk] = src.int()
        back = torch.full((n_in, k), -1, dtype=torch.int32, device=neighbor_map.device)
        back[src, kk] = dst.int()
        hit = (fwd, back)
        setattr(neighbor_map, key, hit)
    return hit


def convolution_forward_cuda(in_feat: torch.Tensor, out_feat: torch.Tensor, kernel: torch.Tensor,
                             neighbor_map: torch.Tensor, neighbor_offset: torch.Tensor, transpose: bool) -> None:
    """out_feat += sum_k in_feat[in_k] @ kernel[k] scattered to out_k: the reference's gather / GEMM / scatter-add
    loop as ONE pass of the table or pair-list kernel (include/link_amd.h section D)."""
    from .elk import subm_conv
    _need_gpu(in_feat, out_feat, kernel, neighbor_map)
    if in_feat.shape[1] != kernel.shape[1]:
        raise ValueError("Input feature size and kernel size mismatch")          # convolution_cuda.cu:57-59
    fwd, _ = _tables_of(neighbor_map, neighbor_offset, in_feat.shape[0], out_feat.shape[0], bool(transpose))
    out_feat += subm_conv(in_feat, kernel, fwd).to(out_feat.dtype)


def convolution_backward_cuda(in_feat: torch.Tensor, grad_in_feat: torch.Tensor, grad_out_feat: torch.Tensor,
                              kernel: torch.Tensor, grad_kernel: torch.Tensor, neighbor_map: torch.Tensor,
                              neighbor_offset: torch.Tensor, transpose: bool) -> None:
    """grad_in_feat += grad_out[out_k] @ kernel[k]^T scattered to in_k; grad_kernel[k] += in[in_k]^T @ grad_out[out_k]
    (convolution_cuda.cu:167-278)."""
    from .elk import _conv_weight_grad, subm_conv
    _need_gpu(in_feat, grad_in_feat, grad_out_feat, kernel, grad_kernel, neighbor_map)
    fwd, back = _tables_of(neighbor_map, neighbor_offset, in_feat.shape[0], grad_out_feat.shape[0], bool(transpose))
    g = grad_out_feat.contiguous().float()
    grad_in_feat += subm_conv(g, kernel.detach().transpose(1, 2).contiguous(), back).to(grad_in_feat.dtype)
    grad_kernel += _conv_weight_grad(in_feat, g, fwd, tuple(kernel.shape)).to(grad_kernel.dtype)
