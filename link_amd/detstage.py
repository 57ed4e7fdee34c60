"""link_amd/detstage.py -- one stage of the detection backbone SpMiddleResNetFHDELKv3 (row N3 of SURVEY.md 8f).

Reference: detection/det3d/models/backbones/scn.py:452-626.  A stage K of that backbone is

    x_conv = convK_tail(convK(x))            convK = 2 x SparseBasicBlock (SubMConv3d 3^3 + BN + ReLU +
                                             SubMConv3d 3^3 + BN, + identity, ReLU; scn.py:64-107),
                                             convK_tail = SubMConv3d 3^3 + BN          (scn.py:482-489)
    x_lk   = elkK_tail(elkK(x, block_sz=7))  elkK = TSELKBlock (ts_elk.py:110-230), elkK_tail = SubMConv3d + BN
    x      = ReLU(x_conv.features + x_lk.features)                                      (scn.py:586-590)

on spconv tensors.  spconv is an external wheel of the reference (not in /root/reference, not installed here),
so this module carries its own `SubMConv3d` with spconv 2.x's parameter layout (weight [Cout, kz, ky, kx, Cin],
bias [Cout]; indices (batch, z, y, x)) and the standard submanifold semantics: outputs at the input's active
sites only, out[p] = sum_{a,b,c} W[:, a, b, c, :] . in[p + (a-1, b-1, c-1)] (cross-correlation, as
torch.nn.functional.conv3d on the densified input).  tests/test_gpu_detstage.py pins exactly that against
torch's dense conv3d; parity with the spconv wheel itself is UNPINNED (it cannot run here) and says so.

Inference runs every convolution on the HIP kernels with its BatchNorm (folded to scale / shift together with
the convolution's bias), the residual add and the ReLU in the convolution's finish phase
(link_subm_conv_ln_add_relu / link_conv_centre_sum with flag bit 1): six launches-with-epilogue + the fused
TSELKBlock per stage, no elementwise pass over [N, C] in between (row N2: "BN+ReLU of the elkK_tail convs").
With grad enabled the modules run one by one (torch BatchNorm in training mode needs batch statistics).
The k3-s2 SparseConv3d between the stages (it creates new active sites) is not part of this module.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch
import torch.nn as nn

from . import _lib as L
from .elk import (SparseConvTensor, TSELKBlock, _SubmConv, _pair_plan, fold_batchnorm, neighbor_table_of, spconv2ts,
                  subm_conv, subm_conv_ln_add_relu)
from .utils import get_kernel_offsets

__all__ = ["SubMConv3d", "SparseConv3d", "SparseBasicBlock", "ELKv3Stage", "SpMiddleResNetFHDELKv3", "to_dense"]


def _replace_feature(sct, feats):
    out = SparseConvTensor(feats, sct.indices, sct.spatial_shape, sct.batch_size, sct.grid, sct.voxel_num,
                           sct.indice_dict, sct.benchmark)
    out.benchmark_record = sct.benchmark_record
    return out


def _plan_ahead(nbr, c: int, dtype) -> None:
    """Build the stage's pair plan ahead of its convolutions -- unless they run the resident-weights kernel, which reads the
    table itself (16 / 32 channels)."""
    from .elk import _resident_form
    if not _resident_form(c, c, nbr.shape[1], dtype != torch.float32, "auto"):
        _pair_plan(nbr, c, c)


def _site_table(sct):
    """(neighbour table int32[N,27] in link_amd's offset order, spatial tile order) of the tensor's active sites:
    the same cached table the TSELKBlock's local_mix convolution uses (all submanifold convolutions of a stage
    share it, as spconv's indice_key does)."""
    st, _ = spconv2ts(sct)
    return neighbor_table_of(st, (3, 3, 3))


class SubMConv3d(nn.Module):
    """3^3 submanifold convolution with spconv 2.x's parameter layout (see the module docstring)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, bias: bool = True,
                 indice_key: Optional[str] = None):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("SubMConv3d: 3^3 kernels (what the ELKv3 backbone uses)")
        self.in_channels, self.out_channels, self.indice_key = in_channels, out_channels, indice_key
        self.weight = nn.Parameter(torch.empty(out_channels, 3, 3, 3, in_channels))
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=5 ** 0.5)
        if bias:
            bound = 1.0 / (27 * in_channels) ** 0.5
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._kio = None

    def kernel_kio(self) -> torch.Tensor:
        """Weights as [27, Cin, Cout] in link_amd's kernel-offset order (offset k = (dx, dy, dz) of
        get_kernel_offsets(3) <-> spconv tap (a, b, c) = (dz+1, dy+1, dx+1)); cached per weight version."""
        ver = (self.weight._version, self.weight.device)
        if self._kio is None or self._kio[0] != ver or torch.is_grad_enabled() and self.weight.requires_grad:
            offs = get_kernel_offsets(3, device="cpu").tolist()
            a = torch.tensor([o[2] + 1 for o in offs], device=self.weight.device)
            b = torch.tensor([o[1] + 1 for o in offs], device=self.weight.device)
            c = torch.tensor([o[0] + 1 for o in offs], device=self.weight.device)
            kio = self.weight[:, a, b, c, :].permute(1, 2, 0).contiguous()        # [27, Cin, Cout]
            if torch.is_grad_enabled() and self.weight.requires_grad:
                return kio
            self._kio = (ver, kio.detach())
        return self._kio[1]

    def forward(self, sct):
        nbr, order = _site_table(sct)
        w = self.kernel_kio()
        if torch.is_grad_enabled() and (sct.features.requires_grad or self.weight.requires_grad):
            out = _SubmConv.apply(sct.features.float(), w, nbr, order)
        else:
            out = subm_conv(sct.features, w, nbr, order)
        if self.bias is not None:
            out = out + self.bias
        return _replace_feature(sct, out)


def _fold(conv: SubMConv3d, bn: nn.BatchNorm1d):
    """(scale, shift) of BatchNorm (running statistics) applied to conv + bias: y = conv * scale + shift."""
    return fold_batchnorm(bn, conv.bias)


def _conv_bn(conv: SubMConv3d, bn: nn.BatchNorm1d, feats, nbr, order, addend=None, relu=False):
    sc, sh = _fold(conv, bn)
    return subm_conv_ln_add_relu(feats, conv.kernel_kio(), nbr, order, sc, sh, 0.0, addend, relu=relu, affine=True)


class SparseBasicBlock(nn.Module):
    """scn.py:64-107 (stride 1, no downsample): relu(bn2(conv2(relu(bn1(conv1(x))))) + x)."""

    def __init__(self, planes: int, eps: float = 1e-3, momentum: float = 0.01, indice_key: Optional[str] = None):
        super().__init__()
        self.conv1 = SubMConv3d(planes, planes, 3, bias=True, indice_key=indice_key)
        self.bn1 = nn.BatchNorm1d(planes, eps=eps, momentum=momentum)
        self.relu = nn.ReLU()
        self.conv2 = SubMConv3d(planes, planes, 3, bias=True, indice_key=indice_key)
        self.bn2 = nn.BatchNorm1d(planes, eps=eps, momentum=momentum)

    def forward(self, sct):
        out = self.conv1(sct)
        out = _replace_feature(out, self.relu(self.bn1(out.features)))
        out = self.conv2(out)
        out = _replace_feature(out, self.bn2(out.features))
        return _replace_feature(out, self.relu(out.features + sct.features))

    def fused(self, feats, nbr, order):
        h = _conv_bn(self.conv1, self.bn1, feats, nbr, order, relu=True)
        return _conv_bn(self.conv2, self.bn2, h, nbr, order, addend=feats, relu=True)


def _needs_modules(mod: nn.Module, feats: torch.Tensor) -> bool:
    """True when the modules must run one by one (autograd, or training-mode BatchNorm statistics)."""
    if not feats.is_cuda:
        raise L.LinkAmdError("the detection stages need GPU tensors (HIP path; no CPU fallback)")
    return mod.training or (torch.is_grad_enabled() and (feats.requires_grad or any(p.requires_grad for p in mod.parameters())))


def _seq_conv_bn(seq, sct, relu: bool):
    """[conv, BatchNorm(, ReLU)] module by module."""
    t = seq[0](sct)
    f = seq[1](t.features)
    return _replace_feature(t, torch.relu(f) if relu else f)


def _stage_modules(conv, conv_tail, elk, elk_tail, act, sct, block_sz):
    """scn.py:586-590 one module at a time (the differentiable path)."""
    x_conv = _seq_conv_bn(conv_tail, conv(sct), False)
    x_lk = _seq_conv_bn(elk_tail, elk(sct, block_sz), False)
    return _replace_feature(x_conv, act(x_conv.features + x_lk.features))


def _stage_fused(conv, conv_tail, elk, elk_tail, sct, block_sz):
    """scn.py:586-590 at inference: six convolutions with BatchNorm / residual / ReLU in their finish phase + the
    fused TSELKBlock, one neighbour table for all of them."""
    feats = sct.features
    io_dtype = feats.dtype
    nbr, order = _site_table(sct)
    x = feats.contiguous()                             # fp16 / bf16 rows stay 16-bit through every convolution (AMP form)
    for blk in conv:
        x = blk.fused(x, nbr, order)
    x_conv = _conv_bn(conv_tail[0], conv_tail[1], x, nbr, order)
    x_lk = elk(_replace_feature(sct, feats.float()), block_sz).features      # the block's general layout is fp32
    out = _conv_bn(elk_tail[0], elk_tail[1], x_lk if io_dtype == torch.float32 else x_lk.to(io_dtype), nbr, order,
                   addend=x_conv, relu=True)
    return _replace_feature(sct, out)


class ELKv3Stage(nn.Module):
    """Stage K of SpMiddleResNetFHDELKv3 (scn.py:477-494 constructor, :586-590 forward); attribute names
    `conv`, `conv_tail`, `elk`, `elk_tail` stand for the backbone's convK, convK_tail, elkK, elkK_tail
    (`load_backbone_stage` maps a backbone state_dict onto them)."""

    def __init__(self, planes: int, block_sz: int = 7, eps: float = 1e-3, momentum: float = 0.01, baseop: str = "cos"):
        super().__init__()
        self.planes, self.block_sz = planes, block_sz
        self.conv = nn.Sequential(SparseBasicBlock(planes, eps, momentum), SparseBasicBlock(planes, eps, momentum))
        self.conv_tail = nn.Sequential(SubMConv3d(planes, planes, 3, bias=False), nn.BatchNorm1d(planes, eps=eps, momentum=momentum))
        self.elk = TSELKBlock(planes, planes, baseop=baseop)
        self.elk_tail = nn.Sequential(SubMConv3d(planes, planes, 3, bias=False), nn.BatchNorm1d(planes, eps=eps, momentum=momentum))
        self.act = nn.ReLU(inplace=True)

    def load_backbone_stage(self, state_dict, k: int, strict: bool = True):
        """Load stage k (1..4) of a SpMiddleResNetFHDELKv3 state_dict (keys conv{k}.*, conv{k}_tail.*, elk{k}.*,
        elk{k}_tail.*)."""
        ren = {f"conv{k}.": "conv.", f"conv{k}_tail.": "conv_tail.", f"elk{k}.": "elk.", f"elk{k}_tail.": "elk_tail."}
        sd = {}
        for key, val in state_dict.items():
            for old, new in ren.items():
                if key.startswith(old):
                    sd[new + key[len(old):]] = val
        return self.load_state_dict(sd, strict=strict)

    def _modules_path(self, sct):
        return _stage_modules(self.conv, self.conv_tail, self.elk, self.elk_tail, self.act, sct, self.block_sz)

    def forward(self, sct):
        if _needs_modules(self, sct.features):
            return self._modules_path(sct)
        return _stage_fused(self.conv, self.conv_tail, self.elk, self.elk_tail, sct, self.block_sz)


# ------------------------------------------------------------------------------------------------
# the rest of the backbone's sparse half: strided convolutions between the stages, extra_conv, dense()
# ------------------------------------------------------------------------------------------------
def _ntuple3(v):
    return tuple(int(x) for x in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


class SparseConv3d(nn.Module):
    """Regular (site-creating) sparse convolution with spconv 2.x's layout, for the forms the backbone uses
    (scn.py:496-502,518-524,540-546,562-568): kernel 3 or 1 per axis, stride 2 or 1 per axis, per-axis padding,
    axes in (z, y, x) order.  Output sites are all positions o inside the output shape
    floor((in + 2 pad - k) / stride) + 1 that have at least one active input under the kernel,
    out[o] = sum_taps W[:, a, b, c, :] . in[o * stride - pad + (a, b, c)].  The kernel map is a per-output table
    (dense cell table of the input sites), cached in the tensor's indice_dict; contraction on the table / pair-list
    convolution kernels with the BatchNorm + ReLU of the surrounding SparseSequential folded in at inference."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding=0, bias=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = _ntuple3(kernel_size), _ntuple3(stride), _ntuple3(padding)
        if any(k not in (1, 3) for k in self.kernel_size) or any(s not in (1, 2) for s in self.stride):
            raise NotImplementedError("SparseConv3d: kernel 1 or 3 and stride 1 or 2 per axis")
        kz, ky, kx = self.kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, kz, ky, kx, in_channels))
        nn.init.kaiming_uniform_(self.weight.view(out_channels, -1), a=5 ** 0.5)
        if bias:
            bound = 1.0 / (kz * ky * kx * in_channels) ** 0.5
            self.bias = nn.Parameter(torch.empty(out_channels).uniform_(-bound, bound))
        else:
            self.register_parameter("bias", None)
        self._kio = None

    def _taps(self):
        return [(a, b, c) for a in range(self.kernel_size[0]) for b in range(self.kernel_size[1])
                for c in range(self.kernel_size[2])]

    # A strided convolution's table is used once per frame: building its pair plan (two passes + a host round trip)
    # costs more than the pair-list kernels save over the table kernel.  "auto" / "pairs" select as subm_conv does.
    form = "table"

    def kernel_kio(self) -> torch.Tensor:
        """[taps, Cin, Cout], taps in (a, b, c) row-major order (the table's column order)."""
        ver = (self.weight._version, self.weight.device)
        live = torch.is_grad_enabled() and self.weight.requires_grad
        if self._kio is None or self._kio[0] != ver or live:
            kio = self.weight.reshape(self.out_channels, -1, self.in_channels).permute(1, 2, 0).contiguous()
            if live:
                return kio
            self._kio = (ver, kio.detach())
        return self._kio[1]

    def out_shape(self, shape):
        return [(int(shape[d]) + 2 * self.padding[d] - self.kernel_size[d]) // self.stride[d] + 1 for d in range(3)]

    def _map(self, sct):
        """(output indices int32[M,4] (b,z,y,x) sorted, table int32[M, taps] of input rows, transposed-direction
        table for the input gradient), cached per (input coordinate set, geometry)."""
        key = ("link_sparse_conv", sct.indices.data_ptr(), sct.indices.shape[0], self.kernel_size, self.stride, self.padding)
        hit = sct.indice_dict.get(key)
        if hit is not None:
            return hit[0], hit[1], hit[2]
        dev = sct.indices.device
        oshape = self.out_shape(sct.spatial_shape)
        k, s, p = self.kernel_size, self.stride, self.padding
        n = sct.indices.shape[0]
        lib, stream = L.lib(), L.current_stream_handle()
        cst = self.__dict__.setdefault("_const", {}).get(dev)
        if cst is None:                                  # small constant tensors: one H2D each, once per device
            offs = get_kernel_offsets(3, device="cpu").tolist()
            col = {(o3[2] + 1, o3[1] + 1, o3[0] + 1): j for j, o3 in enumerate(offs)}      # window position (a,b,c)
            sel = [col[(a if k[0] == 3 else 1, b if k[1] == 3 else 1, c if k[2] == 3 else 1)] for a, b, c in self._taps()]
            cst = self._const[dev] = (torch.tensor(sel, device=dev),
                                      torch.tensor([s[2], s[1], s[0], 1], dtype=torch.int32, device=dev),
                                      torch.tensor([(1 if k[2] == 3 else 0) - p[2], (1 if k[1] == 3 else 0) - p[1],
                                                    (1 if k[0] == 3 else 0) - p[0], 0], dtype=torch.int32, device=dev))
        sel_t, mul_t, add_t = cst
        # output sites: candidates of every input (HIP), then their sorted unique rows through the dense cell
        # grid (link_index_cells; rows of -1 = invalid combinations, dropped there); columns are (b, z, y, x), so the
        # index's lexicographic row order is the site order
        i3 = ctypes.c_int32 * 3
        ka, sa, pa, oa = i3(*k), i3(*s), i3(*p), i3(*oshape)
        ncomb = int(lib.link_conv_out_candidate_count(ka, sa))
        ind = sct.indices.contiguous()
        cand = torch.empty((max(n * ncomb, 1), 4), dtype=torch.int32, device=dev)
        L.check(lib.link_conv_out_candidates(ind.data_ptr(), n, ka, sa, pa, oa, cand.data_ptr(), stream), "link_conv_out_candidates")
        from .index import foreign_neighbor_map, unique_cells
        hi = (int(sct.batch_size) - 1, oshape[0] - 1, oshape[1] - 1, oshape[2] - 1)
        sites, hdr = unique_cells(cand[: n * ncomb], ((0, 0, 0, 0), hi))
        m = int(hdr[L.HDR_M].item())                     # the one host round trip of this map
        out_ind = sites[:m]
        # table[j, t] = input row at out_ind[j] * s - p + tap_t, straight from the (b, z, y, x) rows: the input sites go into the
        # workspace's cell table (kept all zero between uses), one kernel writes the table, the sites are taken out again
        shp = [int(v) for v in sct.spatial_shape]
        from .index import MAX_CELLS, _workspace
        cells = int(sct.batch_size) * shp[0] * shp[1] * shp[2]
        out_ind = out_ind.contiguous()
        if cells <= MAX_CELLS:                           # the persistent workspace table is capped at 1 GiB (index.MAX_CELLS)
            ws = _workspace(dev)
            site = ws.cell_table(cells)
            shp_a = i3(*shp)
            table = torch.empty((m, len(self._taps())), dtype=torch.int32, device=dev)
            try:
                L.check(lib.link_conv_site_table(ind.data_ptr(), n, shp_a, int(sct.batch_size), site.data_ptr(), 0, stream), "link_conv_site_table")
                L.check(lib.link_conv_gather_table(out_ind.data_ptr(), m, ka, sa, pa, shp_a, int(sct.batch_size), site.data_ptr(),
                                                   table.data_ptr(), stream), "link_conv_gather_table")
                L.check(lib.link_conv_site_table(ind.data_ptr(), n, shp_a, int(sct.batch_size), site.data_ptr(), 1, stream), "link_conv_site_table")
            except Exception:
                ws.drop_cell_table()                     # never leave a dirty table behind for the next map
                raise
        else:
            # A 3-wide window per axis around `base`: kernel 3 -> base = o*s - p + 1 (taps at -1, 0, +1); kernel 1 -> its tap is the centre
            rows = out_ind[:, [3, 2, 1, 0]] * mul_t + add_t                                    # (x, y, z, b) window centres
            in_xyzb = ind[:, [3, 2, 1, 0]].contiguous()
            full = foreign_neighbor_map(rows.contiguous(), 3, table_rows=in_xyzb,
                                        bounds=((0, 0, 0, 0), (shp[2] - 1, shp[1] - 1, shp[0] - 1, int(sct.batch_size) - 1)))
            table = full[:, sel_t].contiguous()
        table._link_subm = False                         # structural mark for the pair plan: a gather table between two site sets
        back = None                                      # transposed direction: built on the first backward
        hit = sct.indice_dict[key] = [out_ind.contiguous(), table, back, sct.indices]
        sct.indice_dict[("link_unique", hit[0].data_ptr(), hit[0].shape[0])] = True     # sorted unique cells: unique by construction
        return hit[0], hit[1], hit[2]

    def _out_tensor(self, sct, out_ind, feats):
        out = SparseConvTensor(feats, out_ind, self.out_shape(sct.spatial_shape), sct.batch_size, None, None,
                               sct.indice_dict, sct.benchmark)
        return out

    def forward(self, sct):
        out_ind, table, back = self._map(sct)
        w = self.kernel_kio()
        if torch.is_grad_enabled() and (sct.features.requires_grad or self.weight.requires_grad):
            from .elk import _GatherConv
            if back is None:                             # back[i, t] = output row that reads input i through tap t
                key = ("link_sparse_conv", sct.indices.data_ptr(), sct.indices.shape[0], self.kernel_size, self.stride, self.padding)
                jj, tt = torch.nonzero(table >= 0, as_tuple=True)
                back = torch.full((sct.indices.shape[0], table.shape[1]), -1, dtype=torch.int32, device=table.device)
                back[table[jj, tt].long(), tt] = jj.int()
                sct.indice_dict[key][2] = back
            out = _GatherConv.apply(sct.features.float(), w, table, back)
        else:
            out = subm_conv(sct.features, w, table, None)
        if self.bias is not None:
            out = out + self.bias
        return self._out_tensor(sct, out_ind, out)

    def fused(self, sct, bn, relu=True):
        out_ind, table, _ = self._map(sct)
        sc, sh = fold_batchnorm(bn, self.bias)
        out = subm_conv_ln_add_relu(sct.features, self.kernel_kio(), table, None, sc, sh, 0.0, None, relu=relu, affine=True,
                                    form=self.form)
        return self._out_tensor(sct, out_ind, out)


def to_dense(sct) -> torch.Tensor:
    """spconv's SparseConvTensor.dense(): [B, C, D, H, W] with zeros at inactive sites."""
    d, h, w = [int(v) for v in sct.spatial_shape]
    ind = sct.indices.long()
    out = torch.zeros((int(sct.batch_size), d, h, w, sct.features.shape[1]), dtype=sct.features.dtype,
                      device=sct.features.device)
    out[ind[:, 0], ind[:, 1], ind[:, 2], ind[:, 3]] = sct.features
    return out.permute(0, 4, 1, 2, 3).contiguous()


MAP_STREAM = True     # build the kernel maps of a frame on a side stream (see _MapAhead)


class _MapAhead:
    """Kernel-map construction ahead of the compute, on its own HIP stream.

    The maps of a frame depend on its coordinates only, but every one of them costs a host round trip (the pair
    counts / the number of output sites size the next allocation -- the same numbers the reference's nbsizes holds
    on the host, nn/functional/conv.py:114-116).  On the compute stream each of those round trips would wait for
    all the convolutions queued before it.  Inside `with maps:` the current stream is a side stream that never
    waits for the compute stream after the frame has begun, so the host builds scale k+1's maps while the GPU is
    still computing scale k; leaving the block makes the compute stream wait for the side stream.  Rules that
    keep this race-free: every tensor a map is built FROM is either the caller's `coors` (the side stream waits
    for the compute stream once, when the frame begins) or was produced inside an earlier `with maps:` block; map
    tensors are allocated from the side stream's pool and die with the frame's indice_dict, and the next frame
    begins with the side stream waiting for the compute stream again, so their memory is not reused early."""

    _streams = {}

    def __init__(self, device):
        self.main = torch.cuda.current_stream(device)
        key = (device.index if device.index is not None else torch.cuda.current_device())
        side = _MapAhead._streams.get(key)
        if side is None:
            side = _MapAhead._streams[key] = torch.cuda.Stream(device=device)
        self.side = side
        side.wait_stream(self.main)
        self._ctx = None

    def __enter__(self):
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        self.main.wait_stream(self.side)
        return False


class SpMiddleResNetFHDELKv3(nn.Module):
    """The sparse half of the detection backbone (scn.py:452-626), attribute names as the reference's
    (conv_input, conv{k}, conv{k}_tail, elk{k}, elk{k}_tail, act{k}, down{k}, extra_conv), on the SparseConvTensor
    shim.  forward(voxel_features, coors, batch_size, input_shape) returns (BEV tensor [B, 128 * D, H, W],
    {'conv1'..'conv4': SparseConvTensor}) exactly as scn.py:570-626.  Inference: every convolution carries its
    BatchNorm / residual / ReLU in its finish phase; training / autograd: module by module."""

    def __init__(self, num_input_features=128, norm_cfg=None, name="SpMiddleResNetFHDELKv3", **kwargs):
        super().__init__()
        self.name = name
        eps, mom = (norm_cfg or {}).get("eps", 1e-3), (norm_cfg or {}).get("momentum", 0.01)
        bn = lambda c: nn.BatchNorm1d(c, eps=eps, momentum=mom)
        self.planes = [16, 32, 64, 128]
        self.block_sz = 7
        p = self.planes
        self.conv_input = nn.Sequential(SubMConv3d(num_input_features, p[0], 3, bias=False), bn(p[0]), nn.ReLU(inplace=True))
        pads = {2: 1, 3: 1, 4: [0, 1, 1]}
        for k in (1, 2, 3, 4):
            c = p[k - 1]
            if k > 1:
                setattr(self, f"down{k}", nn.Sequential(SparseConv3d(p[k - 2], c, 3, 2, padding=pads[k], bias=False), bn(c),
                                                        nn.ReLU(inplace=True)))
            setattr(self, f"conv{k}", nn.Sequential(SparseBasicBlock(c, eps, mom), SparseBasicBlock(c, eps, mom)))
            setattr(self, f"conv{k}_tail", nn.Sequential(SubMConv3d(c, c, 3, bias=False), bn(c)))
            setattr(self, f"elk{k}", TSELKBlock(c, c))
            setattr(self, f"elk{k}_tail", nn.Sequential(SubMConv3d(c, c, 3, bias=False), bn(c)))
            setattr(self, f"act{k}", nn.ReLU(inplace=True))
        self.extra_conv = nn.Sequential(SparseConv3d(p[3], p[3], (3, 1, 1), (2, 1, 1), bias=False), bn(p[3]), nn.ReLU())

    def forward(self, voxel_features, coors, batch_size, input_shape, indice_dict=None):
        """`indice_dict` (not in the reference's signature): a dict kept by the caller to reuse the kernel maps of a
        coordinate set across calls (benchmarks of the kernels alone; a static scene).  Default: maps are built
        per call, as the reference does."""
        sparse_shape = [int(v) for v in list(input_shape)[::-1]]
        sparse_shape[0] += 1                                           # scn.py:573
        x = SparseConvTensor(voxel_features, coors.int(), sparse_shape, batch_size, indice_dict=indice_dict)
        fused = not _needs_modules(self, voxel_features)
        maps = _MapAhead(x.features.device) if fused and MAP_STREAM else None
        if fused:
            if maps is not None:
                with maps:                                             # scale 1: site table + its pair plan
                    nbr, order = _site_table(x)
                    _plan_ahead(nbr, 16, x.features.dtype)
            nbr, order = _site_table(x)
            sc, sh = fold_batchnorm(self.conv_input[1], self.conv_input[0].bias)
            x = _replace_feature(x, subm_conv_ln_add_relu(x.features, self.conv_input[0].kernel_kio(), nbr, order, sc, sh,
                                                          0.0, None, relu=True, affine=True))
        else:
            x = _seq_conv_bn(self.conv_input, x, True)
        scales = {}
        for k in (1, 2, 3, 4):
            if k > 1:
                down = getattr(self, f"down{k}")
                if maps is not None:
                    with maps:                                         # scale k: output sites + gather table of the strided
                        out_ind, table, _ = down[0]._map(x)            # convolution, then the new sites' table and the plans
                        if down[0].form != "table":
                            _pair_plan(table, down[0].in_channels, down[0].out_channels)
                        nbr, _ = _site_table(down[0]._out_tensor(x, out_ind, None))
                        _plan_ahead(nbr, down[0].out_channels, x.features.dtype)
                x = down[0].fused(x, down[1], relu=True) if fused else _seq_conv_bn(down, x, True)
            parts = [getattr(self, f"{n}{k}{s}") for n, s in (("conv", ""), ("conv", "_tail"), ("elk", ""), ("elk", "_tail"))]
            if fused:
                x = _stage_fused(*parts, x, self.block_sz)
            else:
                x = _stage_modules(*parts, getattr(self, f"act{k}"), x, self.block_sz)
            scales[f"conv{k}"] = x
        if maps is not None:
            with maps:
                _, table, _ = self.extra_conv[0]._map(x)
                if self.extra_conv[0].form != "table":
                    _pair_plan(table, self.extra_conv[0].in_channels, self.extra_conv[0].out_channels)
        ret = self.extra_conv[0].fused(x, self.extra_conv[1], relu=True) if fused else _seq_conv_bn(self.extra_conv, x, True)
        ret = to_dense(ret)
        n, c, d, h, w = ret.shape
        return ret.view(n, c * d, h, w), scales
