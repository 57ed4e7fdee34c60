"""link_amd/elk.py -- the LinK block modules behind the reference's names.

  ELKBlock(inc, outc, groups=1, baseop='cos_x').forward(st, s, r)
      /root/reference/segmentation/core/models/semantic_kitti/linkunet.py:94-185
      (variant='encoder' reproduces linkencoder.py:165: cos_x feeds coords / tensor stride)
  TSELKBlock(inc, outc, baseop='cos').forward(sct, stride) / .forward_(st, stride)
      /root/reference/detection/det3d/models/utils/ts_elk.py:110-230
  spconv2ts / ts2spconv                                   ts_elk.py:10-59

Parameter names and shapes match the reference state_dict (alpha [1,C/g], pos_weight.0.weight,
pre_mix.0.weight, pre_mix.1.{weight,bias}, local_mix.0.kernel [27,C,C], norm_local.*, norm.*), so
released checkpoints load.  forward() overwrites `st.F` of its INPUT object and returns that object
(linkunet.py:158,183-185).

Two execution paths, both HIP:
  * inference (no grad): the fused R_core -- link_premix_ln -> link_modulate_block_sum ->
    link_gather_demod_ln (include/link_amd.h section C), no host sync when the index is cached or
    bounds are known;
  * training (grad enabled): pre_mix Linear and the two LayerNorms through torch autograd, everything
    between them (theta, modulate, block sums, r^3 gather, de-modulate) through _ElkMid, whose
    forward and backward are three HIP kernels each (deterministic, no atomics).  Widths that are
    not a multiple of 4, or r > 3, use the op-by-op composition elk_core_autograd (torch elementwise
    ops + the HIP autograd Functions of aggregate.py).
"""
from __future__ import annotations

import ctypes
import weakref
import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import _lib as L
from .aggregate import _AuxToVoxel, _BlockMean, link_index_of
from .functional import sphash, sphashquery
from .index import BlockIndex, GridTooLarge, foreign_neighbor_map
from .tensor import SparseTensor
from .utils import get_kernel_offsets, make_ntuple

__all__ = ["ELKBlock", "TSELKBlock", "Conv3d", "spconv2ts", "ts2spconv", "SparseConvTensor",
           "elk_core_fused", "elk_core_autograd", "elk_core_train", "ElkCorePlan", "subm_conv", "subm_conv_ln_add_relu",
           "invalidate_derived_weights", "release_block_driver"]

_OPS = {"cos": L.OP_COS, "sin": L.OP_SIN, "cos_x": L.OP_COSX}
_IO_DTYPES = {torch.float32: L.IO_F32, torch.float16: L.IO_F16, torch.bfloat16: L.IO_BF16}


def _st():
    return L.current_stream_handle()


# ------------------------------------------------------------------------------------------------
# fused R_core (inference)
# ------------------------------------------------------------------------------------------------
LEAN_FORM = True      # inference R_core of coordinate sets without a block index: the three-launch lean form (tests flip it)
LEAN_SECOND_VISIT_INDEX = True   # a coordinate set seen again gets a block index and moves to the tile form (_core_lean)
TILE_FORM = True      # inference R_core on the general layout: the two-launch tile form where its widths / r apply (tests flip it)


def elk_core_fused(feats: torch.Tensor, coords: torch.Tensor, index: BlockIndex, w_pre: torch.Tensor,
                   pre_ln_w: torch.Tensor, pre_ln_b: torch.Tensor, w_pos: torch.Tensor,
                   alpha: Optional[torch.Tensor], ln_w: torch.Tensor, ln_b: torch.Tensor, baseop: str,
                   cg: int, r: int, coord_div: float = 1.0, eps: float = 1e-6,
                   m_cap: Optional[int] = None) -> torch.Tensor:
    """R_core of ELKBlock.forward: pre_mix -> theta -> modulate -> block
    pre-aggregation -> r^3 neighbour sum -> de-modulate -> norm.  2 kernel launches (tile form; 4 for other widths), no
    host sync.  w_pos fp32[cg,3]; channel j uses theta[j % cg]."""
    n, c = feats.shape
    dev = feats.device
    tiles = TILE_FORM and c in (16, 32, 64, 128) and r in (2, 3) and n > 0
    # fp16 / bf16 rows (autocast) go through the tile form as they are and come back in their type; the four-kernel form is fp32
    io = _IO_DTYPES[feats.dtype] if (tiles and feats.dtype in _IO_DTYPES) else L.IO_F32
    feats = feats.contiguous() if io != L.IO_F32 else feats.contiguous().float()
    op = _OPS[baseop]
    parts = 3 if op == L.OP_COSX else 2
    if m_cap is None:
        m_cap = index._m if index._m is not None else n
    m_cap = max(m_cap, 1)
    out = _alloc_out(index, (n, c), dev)
    desc = L.LinkElkDesc(op, c, cg, r, float(coord_div), float(eps))
    lib, st = L.lib(), _st()
    w_pos = w_pos.contiguous().float()
    al = alpha.contiguous().float().view(-1) if alpha is not None else None
    # parameters are read as fp32 whatever the module was cast to
    w_pre, pre_ln_w, pre_ln_b, ln_w, ln_b = (t.contiguous().float() for t in (w_pre, pre_ln_w, pre_ln_b, ln_w, ln_b))
    if tiles:
        # tile form: two launches (pre_mix + modulate + block sums on the matrix cores; neighbour sum + de-modulate + norm)
        s_bytes = int(lib.link_elk_tiles_table_bytes(ctypes.byref(desc), n, m_cap))
        if 0 < s_bytes < 2 ** 32 and n * c * 4 < 2 ** 32:
            if io != L.IO_F32:
                out = _alloc_out(index, (n, c), dev, feats.dtype)
            S = torch.empty((s_bytes + 3) // 4, dtype=torch.float32, device=dev)
            fin = torch.empty((n, c), dtype=torch.float32, device=dev) if op == L.OP_COSX else None
            fin_p = fin.data_ptr() if fin is not None else None
            L.check(lib.link_elk_premix_modsum_tiles_io(feats.data_ptr(), io, index.vox_sorted.data_ptr(), index.pos_blk.data_ptr(),
                                                        index.blk_start.data_ptr(), index.hdr.data_ptr(),
                                                        w_pre.contiguous().data_ptr(), pre_ln_w.data_ptr(), pre_ln_b.data_ptr(),
                                                        w_pos.data_ptr(), al.data_ptr() if al is not None else None,
                                                        ctypes.byref(desc), n, m_cap, S.data_ptr(), s_bytes, fin_p, st),
                    "link_elk_premix_modsum_tiles_io")
            L.check(lib.link_elk_gather_demod_tiles_io(S.data_ptr(), fin_p, index.vox_sorted.data_ptr(), index.pos_blk.data_ptr(),
                                                       index.blk_coords.data_ptr(), index.cell_blk.data_ptr(),
                                                       ctypes.byref(index.grid), index.hdr.data_ptr(), w_pos.data_ptr(),
                                                       al.data_ptr() if al is not None else None, ln_w.data_ptr(), ln_b.data_ptr(),
                                                       ctypes.byref(desc), n, m_cap, out.data_ptr(), io, st),
                    "link_elk_gather_demod_tiles_io")
            return out
        if io != L.IO_F32:                                   # table beyond 32-bit offsets: the fp32 four-kernel form below
            feats = feats.float()
    fin = torch.empty((n, c), dtype=torch.float32, device=dev)
    S = torch.empty((m_cap + 1) * (parts * c + 1), dtype=torch.float32, device=dev)
    L.check(lib.link_premix_ln(feats.data_ptr(), w_pre.contiguous().data_ptr(), pre_ln_w.data_ptr(),
                               pre_ln_b.data_ptr(), n, c, float(eps), fin.data_ptr(), st), "link_premix_ln")
    L.check(lib.link_modulate_block_sum(fin.data_ptr(), index.vox_sorted.data_ptr(), w_pos.data_ptr(),
                                        al.data_ptr() if al is not None else None,
                                        index.blk_start.data_ptr(), index.hdr.data_ptr(), ctypes.byref(desc),
                                        n, m_cap, S.data_ptr(), st), "link_modulate_block_sum")
    # the block-gather kernel addresses the table with 32-bit byte offsets: beyond 4 GiB (C = 256 above ~1.4 M
    # voxels when M is not known) the one-kernel gather + de-modulate with 64-bit row addressing takes over
    if c % 4 == 0 and r <= 3 and (m_cap + 1) * (parts * c + 1) * 4 < 2 ** 32:
        A = torch.empty((m_cap, parts * c), dtype=torch.float32, device=dev)
        L.check(lib.link_block_gather(S.data_ptr(), index.blk_coords.data_ptr(),
                                      index.cell_blk.data_ptr(), ctypes.byref(index.grid), index.hdr.data_ptr(),
                                      ctypes.byref(desc), m_cap, A.data_ptr(), st), "link_block_gather")
        L.check(lib.link_voxel_demod_ln(A.data_ptr(), fin.data_ptr(), index.vox_sorted.data_ptr(),
                                        index.pos_blk.data_ptr(), w_pos.data_ptr(),
                                        al.data_ptr() if al is not None else None, ln_w.data_ptr(),
                                        ln_b.data_ptr(), index.hdr.data_ptr(), ctypes.byref(desc), n,
                                        out.data_ptr(), st), "link_voxel_demod_ln")
        return out
    L.check(lib.link_gather_demod_ln(S.data_ptr(), fin.data_ptr(), index.vox_sorted.data_ptr(), w_pos.data_ptr(),
                                     al.data_ptr() if al is not None else None, ln_w.data_ptr(),
                                     ln_b.data_ptr(), index.blk_start.data_ptr(),
                                     index.blk_coords.data_ptr(), index.cell_blk.data_ptr(),
                                     ctypes.byref(index.grid), index.hdr.data_ptr(), ctypes.byref(desc), n,
                                     m_cap, out.data_ptr(), st), "link_gather_demod_ln")
    return out


def _alloc_out(index: BlockIndex, shape, dev, dtype=torch.float32) -> torch.Tensor:
    """Output rows of a core call.  The kernels write the rows of every *indexed* voxel; a voxel outside
    caller-supplied bounds is dropped by the index (hdr[STATUS] bit 0), so when the bounds were not derived
    from the coordinates themselves the rows start as zeros instead of uninitialised memory."""
    if getattr(index, "rows_checked", True):
        return torch.empty(shape, dtype=dtype, device=dev)
    return torch.zeros(shape, dtype=dtype, device=dev)


class ElkCorePlan:
    """Preallocated arena for repeated R_core steps (what bench.py times and what a serving loop would
    hold per stream): every buffer is allocated once for frames of <= n_cap voxels inside fixed
    coordinate `bounds`, so one step = ONE FFI call with no allocation and no host sync.

    Two layouts, chosen here on the host from what is known before the frame arrives:
      * dense-cell (include/link_amd.h section E): block table indexed by grid cell, the index is a
        per-cell slot list -- 3 launches (slot insert; fused pre_mix + modulate + cell sums; fused box
        sum + de-modulate).  Taken when the padded grid has
        at most `dense_ratio` x n_cap cells (a mostly occupied grid: cfg1/cfg2), C in {16,32,64,128},
        r in {2,3}.  Slot capacity s^3 covers every frame with unique coordinates; duplicates beyond
        it are dropped and reported by blocks().
      * general (sections B + C): count/scan/place index + block-id table -- 8 launches; any grid.
    `run(..., build_index=False)` reuses the index of the previous call (same coords): the "warm"
    number of SURVEY.md section 8d."""

    def __init__(self, n_cap: int, c: int, baseop: str, cg: int, r: int, s: int, bounds, device,
                 coord_div: float = 1.0, eps: float = 1e-6, layout: str = "auto", dense_ratio: float = 4.0,
                 frames_in_flight: int = 1, slot_cap: int = 0, block_order: str = "auto",
                 **tuning):
        self.n_cap, self.c, self.baseop, self.cg, self.r, self.s = n_cap, c, baseop, cg, r, int(s)
        self.frames_in_flight = max(1, int(frames_in_flight))
        if block_order not in ("auto", "first", "cell"):
            raise ValueError(f"block_order must be auto|first|cell, got {block_order!r}")
        # general layout: "first" numbers the blocks by a scan over the voxels (link_index_build_first) instead of over every cell
        # of the grid; "cell" keeps the reference's order of small_x.C in the plan's blk_coords / counts (link_index_build).
        # A/B on one box over the eight LiDAR stage frames (tools/lidar_core.py, ORDER=cell|first): the voxel scan wins where the
        # grid has many more cells than the frame has voxels (S-kitti stage 1: 982k cells, 59k voxels: 60.4 against 67.2 us with
        # the index rebuilt) and loses 2-7 us everywhere else (cells <= 5 x voxels) -- "auto" draws the line at 8 x n_cap
        self.block_order = block_order
        self._tuning = dict(tuning)
        self._tiles_opt = bool(self._tuning.pop("tiles", True))     # general layout: the two-launch tile form where it applies
        self._lean_cs = self._tuning.pop("lean_cs", None)           # lean form: channel-split first launch (None: by frame size)
        self._lean_pm = self._tuning.pop("lean_pm", None)           # lean form: pre_mix inside launch 2, no X matrix (None: by frame size)
        self.grid = L.grid_from_bounds(bounds[0], bounds[1], int(s))
        self.desc = L.LinkElkDesc(_OPS[baseop], c, cg, r, float(coord_div), float(eps))
        self.parts = 3 if baseop == "cos_x" else 2
        self.device = device
        if layout not in ("auto", "dense", "general", "lean"):
            raise ValueError(f"layout must be auto|dense|general|lean, got {layout!r}")
        self.lean = False
        if layout == "lean":
            if self._tuning:                             # the lean form has no per-plan launch geometry: say so instead of ignoring it
                raise L.LinkAmdError(f"ElkCorePlan(layout='lean'): unknown keyword(s) {sorted(self._tuning)} "
                                     "(it takes lean_cs / lean_pm; launch geometry belongs to the dense-cell layout)")
            self._init_lean(int(slot_cap))
            return
        try:
            dcg = L.dc_grid_from(self.grid, int(slot_cap) if slot_cap else 0) if layout != "general" else None
        except L.LinkAmdError:
            if layout == "dense":
                raise
            dcg = None
        ok = dcg is not None and self._dense_supported(dcg, n_cap, c, r, self.parts)
        if layout == "dense" and not ok:
            raise L.LinkAmdError("ElkCorePlan(layout='dense'): width / r / grid size not supported by the "
                                 "dense-cell path (include/link_amd.h section E)")
        self.dense = ok and (layout == "dense" or (layout == "auto" and dcg.vp <= dense_ratio * max(n_cap, 1)))
        self.sparse = False                              # (the sparse-cell layout of round 4 was removed in round 6: docs/experiments.md section 6)
        self.dcg = dcg if self.dense else None
        if layout == "auto" and not self.dense and slot_cap and self.lean_auto(n_cap, c, baseop, r, self.s, bounds, int(slot_cap)):
            # LiDAR-shaped frame of moderate size whose blocks hold at most `slot_cap` voxels: three launches with the index
            # rebuilt instead of six (profiles/r04_v3_lidar_stages.jsonl)
            self._init_lean(int(slot_cap))
            return
        self.out = torch.empty((n_cap, c), dtype=torch.float32, device=device)
        self.fin = torch.empty((n_cap, c), dtype=torch.float32, device=device)
        self.hdr = torch.zeros(L.HDR_WORDS, dtype=torch.int32, device=device)
        if self.dense:
            self._init_dense()
        else:
            self._init_general()

    @staticmethod
    def _dense_supported(dcg, n_cap: int, c: int, r: int, parts: int) -> bool:
        # 32-bit byte offsets in every table, and a slot arena (vp * k records of 16 B) of at most 1 GiB
        return (c in (16, 32, 64, 128) and r in (2, 3) and (dcg.vp + 1) * parts * c * 4 < 2 ** 32
                and n_cap * c * 4 < 2 ** 32 and dcg.vp * dcg.k * 16 <= 2 ** 30)

    @classmethod
    def would_be_dense(cls, n_cap: int, c: int, baseop: str, r: int, s: int, bounds, dense_ratio: float = 4.0) -> bool:
        """What layout='auto' would choose, without allocating anything."""
        try:
            dcg = L.dc_grid_from(L.grid_from_bounds(bounds[0], bounds[1], int(s)))
        except L.LinkAmdError:
            return False
        return cls._dense_supported(dcg, n_cap, c, r, 3 if baseop == "cos_x" else 2) and dcg.vp <= dense_ratio * max(n_cap, 1)

    def _init_dense(self):
        g, dev, n_cap, c = self.dcg, self.device, self.n_cap, self.c
        i32 = dict(dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        vp, w = int(g.vp), self.parts * c
        self.cnt = torch.zeros(vp, **i32)                              # self-cleaning
        self.slots = torch.empty((vp * int(g.k), 4), **i32)           # touched only where voxels land
        self.vrec = torch.empty((n_cap, 4), **i32)
        self.vcell = torch.empty(n_cap, **i32)
        self.cell_n = torch.zeros(vp, **i32)
        self.S = torch.zeros((vp + 1, w), **f32)                       # border rows stay zero
        self.A = torch.zeros((vp + 1, w), **f32)
        b = self.buf = L.LinkDcBuffers()
        b.cnt, b.slots, b.vrec, b.vcell = self.cnt.data_ptr(), self.slots.data_ptr(), self.vrec.data_ptr(), self.vcell.data_ptr()
        b.cell_n, b.hdr, b.fin = self.cell_n.data_ptr(), self.hdr.data_ptr(), self.fin.data_ptr()
        b.S, b.A, b.out = self.S.data_ptr(), self.A.data_ptr(), self.out.data_ptr()
        self.sid = torch.empty(vp * max(int(g.k), 8), **i32)              # voxel ids per cell (tile form of the fused pre_mix kernel)
        b.sid = self.sid.data_ptr()
        self._fn = L.lib().link_elk_core_dense_forward
        self.m_cap = vp
        self.set_tuning(**self._tuning)

    LEAN_MAX_BYTES = 3 << 30

    @classmethod
    def lean_supported(cls, n_cap: int, c: int, baseop: str, r: int, s: int, bounds, slot_cap: int = 0) -> bool:
        """Whether ElkCorePlan(layout='lean') takes this block: widths / r of the fused kernels, blocks of at most 352 voxels,
        fewer than 2^27 grid cells, and tables (addressed by cell, touched only where voxels land) within LEAN_MAX_BYTES."""
        try:
            grid = L.grid_from_bounds(bounds[0], bounds[1], int(s))
        except L.LinkAmdError:
            return False
        k = int(slot_cap) if slot_cap else int(s) ** 3
        w = (3 if baseop == "cos_x" else 2) * c
        v = grid.cells
        kch = (min(k, L.LEAN_KMAX) + L.LEAN_CHUNK - 1) // L.LEAN_CHUNK
        return (c in (16, 32, 64, 128) and r in (2, 3) and k <= L.LEAN_KMAX and v < 2 ** 27 and n_cap * c * 4 < 2 ** 32
                and v * kch * w * 4 + v * k * 4 + n_cap * (w * 4 + 512) <= cls.LEAN_MAX_BYTES)

    LEAN_AUTO_FLOATS = 8_000_000

    @classmethod
    def lean_auto(cls, n_cap: int, c: int, baseop: str, r: int, s: int, bounds, slot_cap: int) -> bool:
        """Where the lean form is the faster one with the index rebuilt (A/B over the eight LiDAR stage frames, round 4): frames
        whose scratch matrix X (n x P*C floats, written by launch 1 and read by launch 2) stays below ~32 MB -- 150k voxels at
        C = 16, 90k at C = 32, 40k at C = 64 cos_x (wins by 6-16 us there; the 59k-voxel C = 64 cos_x stage, 45 MB, loses 4); above that the tile form's single pass over the rows wins."""
        w = (3 if baseop == "cos_x" else 2) * c
        return n_cap * w <= cls.LEAN_AUTO_FLOATS and cls.lean_supported(n_cap, c, baseop, r, s, bounds, slot_cap)

    def _init_lean(self, slot_cap: int):
        """Lean form (link_elk_core_lean_forward, csrc/elk_lean_impl.h): three launches with the index rebuilt, tables addressed
        by grid cell.  `slot_cap` = the most voxels a block can hold (0: s^3, what unique coordinates at tensor stride 1 allow;
        a caller at tensor stride t passes (s / t)^3)."""
        dev, n_cap, c = self.device, self.n_cap, self.c
        k = int(slot_cap) if slot_cap else self.s ** 3
        w = self.parts * c
        v = self.grid.cells
        if not (c in (16, 32, 64, 128) and self.r in (2, 3) and 1 <= k <= L.LEAN_KMAX and v < 2 ** 27 and n_cap * c * 4 < 2 ** 32):
            raise L.LinkAmdError("ElkCorePlan(layout='lean'): needs C in {16,32,64,128}, r in {2,3}, slot capacity <= 352 and "
                                 "fewer than 2^27 grid cells (include/link_amd.h, link_elk_core_lean_forward)")
        kch = (k + L.LEAN_CHUNK - 1) // L.LEAN_CHUNK
        if v * kch * w * 4 + v * k * 4 + n_cap * (w * 4 + 512) > self.LEAN_MAX_BYTES:
            raise L.LinkAmdError("ElkCorePlan(layout='lean'): tables beyond ElkCorePlan.LEAN_MAX_BYTES")
        self.lean, self.dense, self.sparse, self.dcg, self.k = True, False, False, None, k
        if self._lean_cs is not None:
            self.desc.flags |= L.ELK_LEAN_CS if self._lean_cs else L.ELK_LEAN_NO_CS
        if self._lean_pm is not None:
            self.desc.flags |= L.ELK_LEAN_PM if self._lean_pm else L.ELK_LEAN_NO_PM
        i32 = dict(dtype=torch.int32, device=dev)
        self.out = torch.empty((n_cap, c), dtype=torch.float32, device=dev)
        self.hdr = torch.zeros(64, dtype=torch.int32, device=dev)         # 8 words of header; the rest for profiling builds (-DLEAN_DBG)
        # a counter per line on small grids (a few hundred cells of up to 343 voxels: the insert's atomics would serialise on a
        # handful of lines), packed tighter as the grid grows
        self.cnt_shift = 5 if v <= 32768 else (3 if v <= 262144 else 0)
        self.cnt2 = [torch.zeros(max(v, 1) << self.cnt_shift, **i32), torch.zeros(max(v, 1) << self.cnt_shift, **i32)]   # self-cleaning, alternating
        self.list = torch.empty(v * k, **i32)
        self.seg_cap = (n_cap // 64 + 16) // L.LEAN_SEGS * 64 + 64
        self.rec2 = torch.empty((L.LEAN_SEGS * self.seg_cap * L.LEAN_CHUNK, 4), **i32)
        self.occ2 = [torch.empty(L.LEAN_SEGS * self.seg_cap, **i32), torch.empty(L.LEAN_SEGS * self.seg_cap, **i32)]
        self.ctrl2 = [torch.zeros(256, **i32), torch.zeros(256, **i32)]
        self.X = torch.empty((n_cap, w), dtype=torch.float32, device=dev)
        self.S = torch.empty((v * kch, w), dtype=torch.float32, device=dev)            # touched only where voxels land
        self._cur, self._n_prev = 0, 0
        b = self.buf = L.LinkLeanBuffers()
        b.list, b.rec2, b.X, b.S = self.list.data_ptr(), self.rec2.data_ptr(), self.X.data_ptr(), self.S.data_ptr()
        b.hdr, b.out, b.seg_cap, b.k, b.cnt_shift = self.hdr.data_ptr(), self.out.data_ptr(), self.seg_cap, k, self.cnt_shift
        self.m_cap = n_cap

    def _run_lean(self, n: int, build_index: bool, st) -> int:
        cur, b = (self._cur ^ 1 if build_index else self._cur), self.buf
        b.cnt, b.cnt_prev = self.cnt2[cur].data_ptr(), self.cnt2[cur ^ 1].data_ptr()
        b.occ, b.occ_prev = self.occ2[cur].data_ptr(), self.occ2[cur ^ 1].data_ptr()
        b.ctrl, b.ctrl_prev = self.ctrl2[cur].data_ptr(), self.ctrl2[cur ^ 1].data_ptr()
        rc = L.lib().link_elk_core_lean_forward(ctypes.byref(b), ctypes.byref(self.grid), ctypes.byref(self.desc), n,
                                                int(self._n_prev) if build_index else 0, int(bool(build_index)), st)
        if build_index and rc == 0:
            # the alternating counters / marks advance only with a step that was actually launched: a call the library refused
            # (LINK_ERR_ARG: nothing ran) must leave "current" pointing at the last frame that was really indexed, or the next
            # insert would count on top of that frame's counters with nothing scheduled to clean them
            self._cur, self._n_prev = cur, n
        return rc

    def set_tuning(self, **kw):
        """Launch geometry / kernel selection of THIS plan (link_dc_tuning_t; nothing is process-global).  Defaults follow
        `frames_in_flight`: one frame alone spreads every kernel over two workgroups per CU (k1_wgs 512, z-segments by
        the tile count); with several frames in flight the kernels of different frames share the CUs, so each takes one
        workgroup per CU and the gather kernel 2 z-segments (fewer halo planes summed twice).  Keyword overrides:
        k1_wgs, k2_zsplit, k1_lds_pad, k2_lds_pad, k1_form (0 cell-range form, 1 tile form, 2 matrix-core sums form), k2_form, mode."""
        if not self.dense:
            if kw:
                raise L.LinkAmdError("ElkCorePlan.set_tuning: only the dense-cell layout has per-plan launch geometry "
                                     "(general layout: ElkCorePlan(..., tiles=False) selects the four-kernel form)")
            return self
        t = self.buf.tune
        multi = self.frames_in_flight > 1
        t.k1_wgs, t.k2_zsplit = (256, 2) if multi else (512, 0)
        t.k1_lds_pad = t.k2_lds_pad = t.k1_form = t.k2_form = t.mode = 0
        if not multi and not self.sparse and self.c in (32, 64) and self.baseop != "cos_x" and "k1_form" not in kw:
            # one frame alone: the matrix-core sums form at four waves per SIMD (4096 waves) -- 49.6-50.3 us per step against
            # 52.8-53.3 for the cell-range form (A/B on one box, round 4); with frames in flight the cell-range form at one
            # wave per SIMD leaves more of the CU to the other frames' kernels (35.9 against 37.9 us/frame)
            t.k1_form = 2
            if "k1_wgs" not in kw:
                t.k1_wgs = 1024
        for k, v in kw.items():
            if k not in ("k1_wgs", "k2_zsplit", "k1_lds_pad", "k2_lds_pad", "k1_form", "k2_form", "mode", "k1_dbg", "k2_dbg"):
                raise L.LinkAmdError(f"ElkCorePlan.set_tuning: unknown key {k!r}")
            setattr(t, k, v)
        if t.k1_form == 0 and multi and "k1_lds_pad" not in kw:
            t.k1_lds_pad = 2048      # cell-range form: one of its 80 KB workgroups + a gather workgroup of another frame per CU
        return self

    def _init_general(self):
        device, n_cap, c = self.device, self.n_cap, self.c
        v = self.grid.cells
        i32 = dict(dtype=torch.int32, device=device)
        f32 = dict(dtype=torch.float32, device=device)
        self.cell_counts = torch.zeros(max(v, 1), **i32)            # self-cleaning
        nbytes = L.lib().link_index_scratch_bytes(n_cap, v)
        self.scratch = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if self.block_order == "auto":
            self.block_order = "first" if v > 8 * max(n_cap, 1) else "cell"
        first = self.block_order == "first"
        self.cell_blk = torch.zeros(max(v, 1), **i32) if first else torch.empty(max(v, 1), **i32)
        self.cell_pair = torch.zeros(max(v, 1), dtype=torch.int64, device=device) if first else None      # self-cleaning
        self.vox_blk = torch.empty(n_cap, **i32)
        self.idx_query = torch.empty(n_cap, dtype=torch.int64, device=device)
        self.perm = torch.empty(n_cap, **i32)
        self.vox_sorted = torch.empty((n_cap, 4), **i32)
        self.pos_blk = torch.empty(n_cap, **i32)
        self.blk_start = torch.empty(n_cap + 1, **i32)
        self.blk_coords = torch.empty((n_cap, 4), **i32)
        self.counts = torch.empty(n_cap, **i32)
        # tile form (two launches, link_elk_premix_modsum_tiles + link_elk_gather_demod_tiles): the widths / r the fused
        # kernels are built for; its table carries scratch rows behind the counts
        self.tiles = self._tiles_opt and c in (16, 32, 64, 128) and self.r in (2, 3)
        s_bytes = (n_cap + 1) * (self.parts * c + 1) * 4
        if self.tiles:
            s_bytes = int(L.lib().link_elk_tiles_table_bytes(ctypes.byref(self.desc), n_cap, n_cap))
            self.tiles = 0 < s_bytes < 2 ** 32 and n_cap * c * 4 < 2 ** 32
        if self.tiles:
            self.desc.flags |= L.ELK_TILES
        else:
            s_bytes = (n_cap + 1) * (self.parts * c + 1) * 4
        self.S = torch.empty((s_bytes + 3) // 4, **f32)
        self.A = torch.empty((n_cap, self.parts * c), **f32) if not self.tiles else None
        b = self.buf = L.LinkElkBuffers()
        b.cell_counts, b.scratch, b.scratch_bytes = self.cell_counts.data_ptr(), self.scratch.data_ptr(), nbytes
        b.cell_blk, b.vox_blk, b.idx_query = self.cell_blk.data_ptr(), self.vox_blk.data_ptr(), self.idx_query.data_ptr()
        b.perm, b.blk_start, b.blk_coords = self.perm.data_ptr(), self.blk_start.data_ptr(), self.blk_coords.data_ptr()
        b.vox_sorted, b.pos_blk = self.vox_sorted.data_ptr(), self.pos_blk.data_ptr()
        b.A, b.s_bytes = (self.A.data_ptr() if self.A is not None else None), s_bytes
        b.counts, b.hdr = self.counts.data_ptr(), self.hdr.data_ptr()
        b.cell_pair = self.cell_pair.data_ptr() if first else None
        b.fin, b.S, b.out = self.fin.data_ptr(), self.S.data_ptr(), self.out.data_ptr()
        self._fn = L.lib().link_elk_core_forward
        self.m_cap = n_cap

    def bind(self, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b):
        """Bind the block's parameters (fp32, contiguous, on the plan's device)."""
        key = tuple((p.data_ptr(), p._version, tuple(p.shape), p.dtype) if p is not None else None
                    for p in (w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b))
        if self.__dict__.get("_bound") == key:              # same storage, unmodified since the last bind
            return self
        self._bound = key
        self._params = tuple(None if t is None else t.detach().contiguous().float().view(-1)
                             for t in (w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b))
        b = self.buf
        (b.w_pre, b.pre_ln_w, b.pre_ln_b, b.w_pos, b.alpha, b.ln_w, b.ln_b) = [
            None if t is None else t.data_ptr() for t in self._params]
        return self

    def run(self, feats: torch.Tensor, coords: torch.Tensor, build_index: bool = True,
            out: Optional[torch.Tensor] = None, stream: Optional[int] = None) -> torch.Tensor:
        """One R_core step.  `out` (fp32 [n, C], contiguous): write the result there instead of the plan's
        own buffer (what the module path does: the block's output tensor must outlive the plan).
        `stream`: raw hipStream_t (`torch.cuda.Stream.cuda_stream`) to launch on instead of torch's current stream -- a
        serving loop that keeps several frames in flight passes its streams here and skips torch's stream context
        (entering and leaving `torch.cuda.stream(...)` costs more host time than the step's three launches)."""
        n = feats.shape[0]
        assert n <= self.n_cap and feats.shape[1] == self.c and feats.dtype in _IO_DTYPES
        assert feats.is_contiguous() and coords.is_contiguous() and coords.dtype == torch.int32
        self.buf.feats, self.buf.coords = feats.data_ptr(), coords.data_ptr()
        own = self.out
        if feats.dtype != torch.float32:
            # fp16 / bf16 rows at the kernel boundary (AMP): the fused dense-cell kernels, or the tile form of the general layout
            if not (self.lean or (self.dense and self.c <= 64 and int(self.dcg.k) <= 352) or (not self.dense and getattr(self, "tiles", False))):
                raise L.LinkAmdError("ElkCorePlan: fp16/bf16 feature rows need the fused dense-cell kernels (dense layout, "
                                     "C <= 64, s^3 <= 352) or the tile form of the general layout (tiles=True)")
            own = self.__dict__.setdefault("_out_half", {}).get(feats.dtype)
            if own is None and out is None:
                own = self._out_half[feats.dtype] = torch.empty((self.n_cap, self.c), dtype=feats.dtype, device=self.device)
        self.buf.io_dtype = _IO_DTYPES[feats.dtype]
        if out is not None:
            assert out.shape == (n, self.c) and out.dtype == feats.dtype and out.is_contiguous()
        self.buf.out = (out if out is not None else own).data_ptr()
        st = L.current_stream_handle() if stream is None else int(stream)
        if self.lean:
            rc = self._run_lean(n, build_index, st)
        elif self.dense:
            rc = self._fn(ctypes.byref(self.buf), ctypes.byref(self.dcg), ctypes.byref(self.desc), n, int(build_index), st)
        else:
            rc = self._fn(ctypes.byref(self.buf), ctypes.byref(self.grid), ctypes.byref(self.desc), n,
                          min(self.m_cap, n), (2 if self.block_order == "first" else 1) if build_index else 0, st)
        if rc != 0:
            L.check(rc, "link_elk_core_dense_forward" if self.dense else "link_elk_core_forward")
        if L.DEBUG:
            self.check()
        return out if out is not None else own[:n]

    def _probe_fused(self) -> bool:
        """Whether the step's own insert can serve as the occupancy probe (the fused dense-cell kernels run this plan)."""
        return bool(self.dense and self.c <= 64 and int(self.dcg.k) <= 352 and (int(self.buf.tune.mode) or 7) & 1)

    def _probe(self, coords: torch.Tensor, n: int, stats: torch.Tensor) -> None:
        """Insert `coords` into this plan's cells; (voxels inside, occupied cells, fullest cell) go to stats i32[16][16] as 16
        partial slots (_probe_stats)."""
        self.buf.coords = coords.data_ptr()
        self._indexed = None
        L.check(L.lib().link_dc_index_probe(ctypes.byref(self.buf), ctypes.byref(self.dcg), n, stats.data_ptr(),
                                            L.current_stream_handle()), "link_dc_index_probe")

    def _unprobe(self) -> None:
        """Forget a probed frame: counters and status word back to zero (the slot lists need no cleaning)."""
        self.cnt.zero_()
        self.hdr.zero_()
        self._indexed = None

    def arena_bytes(self) -> int:
        """Device bytes this plan holds (its preallocated buffers)."""
        ts = [t for v in self.__dict__.values() for t in (v if isinstance(v, (list, tuple)) else (v,)) if isinstance(t, torch.Tensor)]
        return sum(t.numel() * t.element_size() for t in ts)

    def check(self) -> None:
        """Read the device status word of the last step (one 32-byte D2H sync) and raise if a voxel was
        dropped: outside the plan's bounds (its `out` row was not written), or in a cell whose slot list was
        full.  Runs after every step under LINK_AMD_DEBUG=1; INTEGRATION.md states the contract."""
        st = int(self.hdr[L.HDR_STATUS].item())
        if st != 0:
            raise L.LinkAmdError("ElkCorePlan: voxels outside the plan's bounds" if st & 1 else
                                 "ElkCorePlan: more voxels in a block than its slot list holds (duplicate coordinates)")

    def blocks(self) -> int:
        """M of the last indexed frame (D2H sync); raises if a voxel fell outside the plan's bounds (or, on
        the dense-cell layout, found its cell's slot list full: duplicate coordinates)."""
        self.check()
        h = self.hdr.tolist()
        if self.lean:
            return int((self.cnt2[self._cur] > 0).sum().item())
        if self.dense:
            return int((self.cell_n > 0).sum().item())
        return int(h[L.HDR_M])


class ElkCoreBatch:
    """R_core of a BATCH of independent frames in ONE call (include/link_amd.h section H, csrc/dense_batch.hip): the slot insert of
    every frame as one grid + two persistent kernels -- a pre_mix role that walks the frames (a frame's ranges of cells as soon as
    its insert has arrived) and a gather role that pulls (frame, tile) items off per-XCD queues (a tile as soon as the frame's
    pre_mix has arrived).  The stages of different frames overlap by construction; per-frame arrival counters replace the two
    stream-ordered launch boundaries every frame of `ElkCorePlan.run` pays.  What BASELINE.json configs[3] ("a batch of 8 independent
    frames") and the reference's collated batches (linkunet.py:132,151-162,178 over batch-indexed coordinates) are to this product.

    `frames` arenas (ElkCorePlan buffers, dense-cell layout) are allocated once; `run(feats_list, coords_list)` takes up to that many
    frames and returns their result rows (views of the arenas' `out` buffers, complete in stream order).  Results are bit for bit
    those of `ElkCorePlan.run` per frame.  C = 64, cg = 32, cos / sin, r in {2, 3}, fp32 / fp16 / bf16 rows (one type per call), coord_div = 1, no alpha, slot capacity
    <= 352 -- LinkAmdError otherwise (run such frames through ElkCorePlan).  Two batches in flight: two ElkCoreBatch objects sharing
    ONE context (`ElkCoreBatch(..., share=first)`), alternated with two streams -- the pre_mix role of the second batch starts
    under the gather role of the first.  `check()` raises if a voxel was dropped or a kernel's bounded wait gave up."""

    def __init__(self, frames: int, n_cap: int, c: int, baseop: str, cg: int, r: int, s: int, bounds, device, eps: float = 1e-6,
                 slot_cap: int = 0, share: Optional["ElkCoreBatch"] = None):
        if c != 64 or cg != 32 or baseop not in ("cos", "sin") or r not in (2, 3):
            raise L.LinkAmdError("ElkCoreBatch: C = 64, cg = 32, cos / sin, r in {2, 3} (include/link_amd.h section H)")
        if frames < 1:
            raise ValueError("ElkCoreBatch: at least one frame")
        self.plans = [ElkCorePlan(n_cap, c, baseop, cg, r, s, bounds, device, eps=eps, layout="dense", slot_cap=slot_cap)
                      for _ in range(frames)]
        if int(self.plans[0].dcg.k) > 352:
            raise L.LinkAmdError("ElkCoreBatch: slot capacity above 352")
        self.n_cap, self.c, self.device = n_cap, c, device
        self._bufs = (L.LinkDcBuffers * frames)()
        self._n = (ctypes.c_int64 * frames)()
        if share is not None:
            self._ctx, self._owner = share._ctx, share          # keep the owner alive
        else:
            h = ctypes.c_void_p()
            with torch.cuda.device(device):
                L.check(L.lib().link_dc_batch_create(ctypes.byref(h)), "link_dc_batch_create")
            self._ctx, self._owner = h, None
        self._fn = L.lib().link_elk_core_dense_forward_batch

    def __del__(self):
        try:
            if getattr(self, "_owner", None) is None and getattr(self, "_ctx", None):
                L.lib().link_dc_batch_destroy(self._ctx)
                self._ctx = None
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # A context's three streams land on hardware queues the runtime picks at creation, and how well the role kernels of consecutive calls
    # overlap depends on that placement (bench.py measures 37 vs 50 us / frame between two contexts on the same arenas).  A caller that
    # cares measures a few contexts and keeps the best: new_context / release_context / install_context / adopt_context move contexts
    # between batches without touching the arenas.
    def new_context(self) -> None:
        """Replace this batch's context by a fresh one (the old one is destroyed if this batch owned it)."""
        self.release_context(destroy=True)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            L.check(L.lib().link_dc_batch_create(ctypes.byref(h)), "link_dc_batch_create")
        self._ctx, self._owner = h, None

    def release_context(self, destroy: bool = False):
        """Detach the context and return its handle (None if it was shared from another batch); `destroy`: free it instead."""
        h, owned = getattr(self, "_ctx", None), getattr(self, "_owner", None) is None
        self._ctx, self._owner = None, None
        if h and owned and destroy:
            L.lib().link_dc_batch_destroy(h)
            return None
        return h if owned else None

    def install_context(self, handle) -> None:
        """Own `handle` (from release_context) from here on."""
        self.release_context(destroy=True)
        self._ctx, self._owner = handle, None

    def adopt_context(self, other: "ElkCoreBatch") -> None:
        """Share `other`'s context (as `share=other` does at construction)."""
        self.release_context(destroy=True)
        self._ctx, self._owner = other._ctx, other

    def calibrate(self, feats, coords, partner: Optional["ElkCoreBatch"] = None, tries: int = 4, calls: int = 12,
                  stream: Optional[int] = None):
        """Measure `tries` contexts on THESE arenas and keep the fastest (see the note above: a context's rate depends on the hardware
        queues the runtime hands its streams, 35 against 48 us per cfg2 frame, and nothing short of running the real kernels tells).
        Each try: `calls` calls of (feats, coords) -- alternating with `partner` (a second batch object: two calls in flight, submit(s + 1)
        before join(s)) when given -- twice, the second timing counts.  The winner becomes this batch's context (and the partner's).
        Returns the us per frame of every try; synchronises the device.  ~0.1 s for 48 cfg2 frames per call."""
        import time
        if getattr(self, "_owner", None) is not None:
            raise L.LinkAmdError("ElkCoreBatch.calibrate: call it on the batch that owns the context (the partner adopts the winner)")
        objs = [self] + ([partner] if partner is not None else [])
        st = L.current_stream_handle() if stream is None else int(stream)
        nfr = len(feats)
        rates, best = [], None
        for t in range(max(1, int(tries))):
            if t:
                self.new_context()
            if partner is not None:
                partner.adopt_context(self)
            for _ in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                prev = None
                for s in range(calls):
                    _, tk = objs[s % len(objs)].submit(feats, coords, stream=st)
                    if prev is not None:
                        self.join(prev, stream=st)
                    prev = tk
                self.join(prev, stream=st)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            rates.append(round(1e6 * dt / (calls * nfr), 2))
            self.check()
            if best is None or rates[-1] < best[0]:
                if best is not None and best[1]:
                    L.lib().link_dc_batch_destroy(best[1])
                best = (rates[-1], self.release_context())
            else:
                self.release_context(destroy=True)
        self.install_context(best[1])
        if partner is not None:
            partner.adopt_context(self)
        return rates

    def bind(self, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b):
        if alpha is not None:
            raise L.LinkAmdError("ElkCoreBatch: alpha is not supported")
        self.plans[0].bind(w_pre, pre_ln_w, pre_ln_b, w_pos, None, ln_w, ln_b)
        b0 = self.plans[0].buf
        for p in self.plans[1:]:                                 # ONE copy of the block's parameters for the whole batch
            p._params, p._bound = self.plans[0]._params, self.plans[0]._bound
            (p.buf.w_pre, p.buf.pre_ln_w, p.buf.pre_ln_b, p.buf.w_pos, p.buf.alpha, p.buf.ln_w, p.buf.ln_b) = (
                b0.w_pre, b0.pre_ln_w, b0.pre_ln_b, b0.w_pos, None, b0.ln_w, b0.ln_b)
        return self

    def _fill(self, feats, coords, outs):
        k = len(feats)
        assert 0 < k <= len(self.plans) and len(coords) == k and (outs is None or len(outs) == k)
        res = []
        dt = feats[0].dtype
        assert dt in _IO_DTYPES, "fp32, fp16 or bf16 rows"
        for i in range(k):
            f, co, p = feats[i], coords[i], self.plans[i]
            n = f.shape[0]
            assert 0 < n <= self.n_cap and f.shape[1] == self.c and f.dtype == dt and f.is_contiguous()     # one row type per call
            assert co.is_contiguous() and co.dtype == torch.int32 and co.shape[0] == n
            if outs is not None:
                dst = outs[i]
            elif dt == torch.float32:
                dst = p.out
            else:                                                    # half rows: a buffer of that type per arena (as ElkCorePlan.run keeps)
                half = p.__dict__.setdefault("_out_half", {})
                dst = half.get(dt)
                if dst is None:
                    dst = half[dt] = torch.empty((self.n_cap, self.c), dtype=dt, device=self.device)
            assert dst.dtype == dt and dst.is_contiguous()
            b = p.buf
            b.feats, b.coords, b.out, b.io_dtype = f.data_ptr(), co.data_ptr(), dst.data_ptr(), _IO_DTYPES[dt]
            self._bufs[i] = b
            self._n[i] = n
            res.append(dst[:n])
        return k, res

    def run(self, feats, coords, outs=None, stream: Optional[int] = None):
        """feats / coords: sequences of [n_i, C] fp32 rows and [n_i, 4] int32 coordinates (at most `frames` of them); `outs`:
        optional result tensors.  Returns the list of result rows (complete in stream order)."""
        k, res = self._fill(feats, coords, outs)
        p0 = self.plans[0]
        st = L.current_stream_handle() if stream is None else int(stream)
        rc = self._fn(self._ctx, self._bufs, self._n, k, ctypes.byref(p0.dcg), ctypes.byref(p0.desc), st)
        if rc != 0:
            L.check(rc, "link_elk_core_dense_forward_batch")
        if L.DEBUG:
            self.check()
        return res

    def submit(self, feats, coords, outs=None, stream: Optional[int] = None):
        """`run` without the wait: enqueues the call behind what the stream holds and returns (result rows, ticket); the rows are NOT
        ordered against the stream until `join(ticket)`.  Several calls in flight from one stream: submit call s + 1 (on another
        batch object of the same context: its own arenas), then join call s -- the pre_mix role of s + 1 runs under the gather role of
        s.  (`run` on two objects from two streams does the same only while the runtime keeps those two streams on different hardware
        queues: link_dc_batch_submit in include/link_amd.h.)"""
        k, res = self._fill(feats, coords, outs)
        p0 = self.plans[0]
        st = L.current_stream_handle() if stream is None else int(stream)
        ticket = ctypes.c_int64(-1)
        rc = L.lib().link_dc_batch_submit(self._ctx, self._bufs, self._n, k, ctypes.byref(p0.dcg), ctypes.byref(p0.desc), st, ctypes.byref(ticket))
        if rc != 0:
            L.check(rc, "link_dc_batch_submit")
        return res, int(ticket.value)

    def join(self, ticket: int, stream: Optional[int] = None) -> None:
        """The stream waits for the rows of the call `ticket` names (a ticket of ANY batch object sharing this context)."""
        st = L.current_stream_handle() if stream is None else int(stream)
        rc = L.lib().link_dc_batch_join(self._ctx, int(ticket), st)
        if rc != 0:
            L.check(rc, "link_dc_batch_join")
        if L.DEBUG:
            self.check()

    def check(self) -> None:
        """Synchronises the context's streams; raises if a bounded wait inside a kernel gave up or a frame dropped a voxel."""
        st = (ctypes.c_int32 * 2)()
        rc = L.lib().link_dc_batch_status(self._ctx, st)
        if rc != 0:
            raise L.LinkAmdError(f"ElkCoreBatch: link_dc_batch_status = {rc} (error word {st[0]}: a persistent kernel's bounded "
                                 "wait gave up; the batch's rows are undefined)")
        for p in self.plans:
            p.check()

    def set_timing(self, on: bool) -> None:
        """Bracket the three launches of every launch set with timed events on their own streams (link_dc_batch_set_timing)."""
        L.check(L.lib().link_dc_batch_set_timing(self._ctx, 1 if on else 0), "link_dc_batch_set_timing")

    def kernel_times_us(self, ticket: int):
        """(insert, pre_mix, gather) brackets of the launch set `ticket` names, in us; waits for that set (one of the last four)."""
        out = (ctypes.c_float * 3)()
        L.check(L.lib().link_dc_batch_kernel_times(self._ctx, int(ticket), out), "link_dc_batch_kernel_times")
        return [1e3 * float(v) for v in out]

    def probe_streams(self, stream: Optional[int] = None):
        """Diagnostic (link_dc_batch_probe_streams): for the pairs (pre_mix -> gather), (pre_mix -> insert), (gather -> insert),
        (stream -> pre_mix), (stream -> gather), (stream -> insert) how long a kernel on the second stream is held up by a 150 us kernel
        + event record on the first: ~10 us = separate hardware queues, >= 150 = the two share one."""
        out = (ctypes.c_double * 12)()
        L.check(L.lib().link_dc_batch_probe_streams(self._ctx, stream if stream is not None else _st(), out), "link_dc_batch_probe_streams")
        return [round(float(v), 1) for v in out]

    def arena_bytes(self) -> int:
        return sum(p.arena_bytes() for p in self.plans)


# ------------------------------------------------------------------------------------------------
# differentiable R_core (training)
# ------------------------------------------------------------------------------------------------
def _aggregate(x: torch.Tensor, index: BlockIndex, r: int) -> torch.Tensor:
    mean = _BlockMean.apply(x, index)
    return _AuxToVoxel.apply(mean, index.counts, None, index.idx_query, index, r)


def elk_core_autograd(feats, coords, index, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b, baseop,
                      cg, r, coord_div=1.0, eps=1e-6):
    c = feats.shape[1]
    fin = TF.layer_norm(TF.linear(feats, w_pre), (c,), pre_ln_w, pre_ln_b, eps)
    xyz = coords[:, :3].float()
    if coord_div != 1.0:
        xyz = xyz / coord_div
    th = TF.linear(xyz, w_pos)
    if alpha is not None:
        th = th * alpha
    if c != cg:
        th = th.repeat([1, c // cg])
    sin, cos = torch.sin(th), torch.cos(th)
    if baseop == "sin":
        v = _aggregate(torch.cat([fin * sin, fin * cos], dim=1), index, r)
        new = v[:, :c] * cos - v[:, c:] * sin
    elif baseop == "cos":
        v = _aggregate(torch.cat([fin * cos, fin * sin], dim=1), index, r)
        new = v[:, :c] * cos + v[:, c:] * sin
    else:
        lin = fin * th
        v = _aggregate(torch.cat([fin * cos, fin * sin, lin], dim=1), index, r)
        new = v[:, :c] * cos + v[:, c:2 * c] * sin + (v[:, 2 * c:] - lin)
    return TF.layer_norm(new, (c,), ln_w, ln_b, eps)


class _ElkMid(torch.autograd.Function):
    """new = demodulate(aggregate(modulate(fin, theta)), theta) with hand-written forward AND backward
    (include/link_amd.h: link_elk_mid_forward / link_elk_mid_backward; 3 kernels each, no atomics).
    Replaces, for training, the reference's VoxelizeFunction / DevoxelizeFunction autograd nodes
    (voxelize.py:10-56, devoxelize.py:51-98) and the torch graph of sin/cos/mul/cat around them
    (linkunet.py:151-176)."""

    @staticmethod
    def forward(ctx, fin, w_pos, alpha, index: BlockIndex, op: int, cg: int, r: int, coord_div: float):
        n, c = fin.shape
        dev = fin.device
        fin = fin.contiguous().float()
        parts = 3 if op == L.OP_COSX else 2
        m_cap = max(index._m if index._m is not None else n, 1)
        desc = L.LinkElkDesc(op, c, cg, r, float(coord_div), 1e-6)
        w_pos_c = w_pos.detach().contiguous().float()
        al = alpha.detach().contiguous().float().view(-1) if alpha is not None else None
        S = torch.empty((m_cap + 1) * (parts * c + 1), dtype=torch.float32, device=dev)
        A = torch.empty((m_cap, parts * c), dtype=torch.float32, device=dev)
        den = torch.empty(m_cap, dtype=torch.float32, device=dev)
        out = _alloc_out(index, (n, c), dev)
        L.check(L.lib().link_elk_mid_forward(
            fin.data_ptr(), index.vox_sorted.data_ptr(), index.pos_blk.data_ptr(), index.blk_start.data_ptr(),
            index.blk_coords.data_ptr(), index.cell_blk.data_ptr(), ctypes.byref(index.grid),
            index.hdr.data_ptr(), w_pos_c.data_ptr(), al.data_ptr() if al is not None else None,
            ctypes.byref(desc), None, None, n, m_cap, S.data_ptr(), A.data_ptr(), den.data_ptr(), out.data_ptr(),
            _st()), "link_elk_mid_forward")
        ctx.save_for_backward(fin, A, den, w_pos_c, al)
        ctx.index, ctx.desc, ctx.m_cap, ctx.S = index, desc, m_cap, S      # S: reused as backward scratch
        ctx.alpha_shape = alpha.shape if alpha is not None else None
        ctx.wpos_shape = w_pos.shape
        return out

    @staticmethod
    def backward(ctx, g):
        fin, A, den, w_pos_c, al = ctx.saved_tensors
        index, desc, m_cap = ctx.index, ctx.desc, ctx.m_cap
        n, c = fin.shape
        cg = desc.cg
        dev = fin.device
        g = g.contiguous().float()
        rows = int(L.lib().link_elk_mid_partial_rows())
        gS = torch.empty_like(A)
        g_fin = torch.empty_like(fin)
        partials = torch.empty((rows, 4, c), dtype=torch.float32, device=dev)
        L.check(L.lib().link_elk_mid_backward(
            g.data_ptr(), fin.data_ptr(), A.data_ptr(), den.data_ptr(), index.vox_sorted.data_ptr(),
            index.pos_blk.data_ptr(), index.blk_start.data_ptr(), index.blk_coords.data_ptr(),
            index.cell_blk.data_ptr(), ctypes.byref(index.grid), index.hdr.data_ptr(), w_pos_c.data_ptr(),
            al.data_ptr() if al is not None else None, ctypes.byref(desc), n, m_cap, ctx.S.data_ptr(),
            gS.data_ptr(), g_fin.data_ptr(), partials.data_ptr(), _st()), "link_elk_mid_backward")
        tot = partials.sum(0).view(4, c // cg, cg).sum(1)        # theta is tiled: channel ch -> ch % cg
        g_wpos = tot[1:4].t().contiguous().view(ctx.wpos_shape)
        g_alpha = tot[0].view(ctx.alpha_shape) if ctx.alpha_shape is not None else None
        return g_fin, g_wpos, g_alpha, None, None, None, None, None


def _weight_grad(g_pre: torch.Tensor, feats: torch.Tensor) -> torch.Tensor:
    """g_pre^T @ feats ([C,N] x [N,C]).  One GEMM with K = N runs on a handful of workgroups; cutting
    N into 32 batches (bmm + a 32-term sum) keeps the library GEMM busy: 282 us -> 29 us on cfg2."""
    n = g_pre.shape[0]
    b = 32
    if n < b * 64:
        return g_pre.t() @ feats
    k = n // b
    out = torch.bmm(g_pre[: b * k].view(b, k, g_pre.shape[1]).transpose(1, 2),
                    feats[: b * k].view(b, k, feats.shape[1])).sum(0)
    if b * k < n:
        out = out + g_pre[b * k:].t() @ feats[b * k:]
    return out


class _ElkCoreTrain(torch.autograd.Function):
    """Whole R_core (pre_mix -> ... -> self.norm) with a hand-written backward: forward = the inference
    kernels (+ saved A, den, fin); backward = link_elk_out_ln_backward -> link_elk_mid_backward ->
    link_premix_ln_backward (six kernels), the weight gradient through one batched library GEMM."""

    @staticmethod
    def forward(ctx, feats, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b, index: BlockIndex, op: int,
                cg: int, r: int, coord_div: float, eps: float):
        n, c = feats.shape
        dev = feats.device
        lib, st = L.lib(), _st()
        feats = feats.detach().contiguous().float()
        f32 = lambda t: t.detach().contiguous().float()
        w_pre_c, pre_w, pre_b, w_pos_c, ln_w_c, ln_b_c = map(f32, (w_pre, pre_ln_w, pre_ln_b, w_pos, ln_w, ln_b))
        al = f32(alpha).view(-1) if alpha is not None else None
        parts = 3 if op == L.OP_COSX else 2
        m_cap = max(index._m if index._m is not None else n, 1)
        desc = L.LinkElkDesc(op, c, cg, r, float(coord_div), float(eps))
        fin = torch.empty((n, c), dtype=torch.float32, device=dev)
        S = torch.empty((m_cap + 1) * (parts * c + 1), dtype=torch.float32, device=dev)
        A = torch.empty((m_cap, parts * c), dtype=torch.float32, device=dev)
        den = torch.empty(m_cap, dtype=torch.float32, device=dev)
        out = _alloc_out(index, (n, c), dev)
        L.check(lib.link_premix_ln(feats.data_ptr(), w_pre_c.data_ptr(), pre_w.data_ptr(), pre_b.data_ptr(), n, c,
                                   float(eps), fin.data_ptr(), st), "link_premix_ln")
        L.check(lib.link_elk_mid_forward(
            fin.data_ptr(), index.vox_sorted.data_ptr(), index.pos_blk.data_ptr(), index.blk_start.data_ptr(),
            index.blk_coords.data_ptr(), index.cell_blk.data_ptr(), ctypes.byref(index.grid),
            index.hdr.data_ptr(), w_pos_c.data_ptr(), al.data_ptr() if al is not None else None,
            ctypes.byref(desc), ln_w_c.data_ptr(), ln_b_c.data_ptr(), n, m_cap, S.data_ptr(), A.data_ptr(),
            den.data_ptr(), out.data_ptr(), st), "link_elk_mid_forward")
        ctx.save_for_backward(feats, fin, A, den, w_pre_c, pre_w, w_pos_c, al, ln_w_c)
        ctx.index, ctx.desc, ctx.m_cap, ctx.S = index, desc, m_cap, S
        ctx.shapes = (w_pre.shape, pre_ln_w.shape, pre_ln_b.shape, w_pos.shape,
                      alpha.shape if alpha is not None else None, ln_w.shape, ln_b.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        feats, fin, A, den, w_pre_c, pre_w, w_pos_c, al, ln_w_c = ctx.saved_tensors
        index, desc, m_cap = ctx.index, ctx.desc, ctx.m_cap
        n, c = fin.shape
        cg, eps = desc.cg, desc.eps
        dev = fin.device
        lib, st = L.lib(), _st()
        g = g.contiguous().float()
        rows = int(lib.link_elk_mid_partial_rows())
        alp = al.data_ptr() if al is not None else None
        work = torch.empty((3, n, c), dtype=torch.float32, device=dev)      # g_new (later g_pre) | g_fin | g_feats
        g_new, g_fin, g_feats = work[0], work[1], work[2]
        part = torch.empty(rows * 8 * c + 8 * c, dtype=torch.float32, device=dev)
        part_o, part_m, part_p = part[: rows * 2 * c], part[rows * 2 * c: rows * 6 * c], part[rows * 6 * c: rows * 8 * c]
        tot = part[rows * 8 * c:]
        gS = torch.empty_like(A)
        L.check(lib.link_elk_out_ln_backward(
            g.data_ptr(), A.data_ptr(), fin.data_ptr(), index.vox_sorted.data_ptr(), index.pos_blk.data_ptr(),
            w_pos_c.data_ptr(), alp, ln_w_c.data_ptr(), index.hdr.data_ptr(), ctypes.byref(desc), n,
            g_new.data_ptr(), part_o.data_ptr(), st), "link_elk_out_ln_backward")
        L.check(lib.link_elk_mid_backward(
            g_new.data_ptr(), fin.data_ptr(), A.data_ptr(), den.data_ptr(), index.vox_sorted.data_ptr(),
            index.pos_blk.data_ptr(), index.blk_start.data_ptr(), index.blk_coords.data_ptr(),
            index.cell_blk.data_ptr(), ctypes.byref(index.grid), index.hdr.data_ptr(), w_pos_c.data_ptr(), alp,
            ctypes.byref(desc), n, m_cap, ctx.S.data_ptr(), gS.data_ptr(), g_fin.data_ptr(), part_m.data_ptr(),
            st), "link_elk_mid_backward")
        g_pre = g_new                                   # g_new is dead from here on: reuse its storage
        L.check(lib.link_premix_ln_backward(feats.data_ptr(), w_pre_c.data_ptr(), pre_w.data_ptr(),
                                            g_fin.data_ptr(), n, c, float(eps), g_pre.data_ptr(),
                                            g_feats.data_ptr(), part_p.data_ptr(), st), "link_premix_ln_backward")
        L.check(lib.link_sum_partials(part_o.data_ptr(), 2 * c, part_m.data_ptr(), 4 * c, part_p.data_ptr(), 2 * c,
                                      rows, tot.data_ptr(), st), "link_sum_partials")
        sh = ctx.shapes
        g_w_pre = _weight_grad(g_pre, feats).view(sh[0])
        th = tot[2 * c: 6 * c].view(4, c // cg, cg)              # theta is tiled: channel ch -> ch % cg
        th = th.sum(1) if c != cg else th[:, 0]
        g_wpos = th[1:4].t().contiguous().view(sh[3])
        g_alpha = th[0].view(sh[4]) if sh[4] is not None else None
        return (g_feats, g_w_pre, tot[6 * c: 7 * c].view(sh[1]), tot[7 * c: 8 * c].view(sh[2]), g_wpos, g_alpha,
                tot[0: c].view(sh[5]), tot[c: 2 * c].view(sh[6]), None, None, None, None, None, None)


def elk_core_train(feats, coords, index, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b, baseop,
                   cg, r, coord_div=1.0, eps=1e-6):
    """Differentiable R_core for training.  C % 16 == 0 (<= 128): everything hand-written
    (_ElkCoreTrain).  Other multiples of 4: pre_mix Linear + the two LayerNorms through torch autograd,
    the middle through _ElkMid."""
    c = feats.shape[1]
    if c % 16 == 0 and c <= 128:
        return _ElkCoreTrain.apply(feats, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, ln_w, ln_b, index,
                                   _OPS[baseop], cg, r, float(coord_div), float(eps))
    fin = TF.layer_norm(TF.linear(feats, w_pre), (c,), pre_ln_w, pre_ln_b, eps)
    new = _ElkMid.apply(fin, w_pos, alpha, index, _OPS[baseop], cg, r, float(coord_div))
    return TF.layer_norm(new, (c,), ln_w, ln_b, eps)


# ------------------------------------------------------------------------------------------------
# local 3^3 sparse convolution (row N1 of SURVEY.md section 8f, minimal form)
# ------------------------------------------------------------------------------------------------
class _StridedMap:
    """Kernel map of a kernel-2 stride-2 convolution: coarse coordinates + the per-output table of the
    down direction [N_coarse, 8] and of the transposed direction [N_fine, 8] (one parent per fine voxel)."""

    def __init__(self, out_coords, nbr_down, nbr_up):
        self.out_coords, self.nbr_down, self._nbr_up, self._n_in = out_coords, nbr_down, nbr_up, None

    @property
    def nbr_up(self):
        """[N_fine, K] with the one coarse row that reads fine row i through offset k (-1 elsewhere): the table of the
        transposed convolution and of the input gradient.  Built on first use (inference encoders never need it)."""
        if self._nbr_up is None:
            down = self.nbr_down
            jj, kk = torch.nonzero(down >= 0, as_tuple=True)
            up = torch.full((self._n_in, down.shape[1]), -1, dtype=torch.int32, device=down.device)
            up[down[jj, kk].long(), kk] = jj.int()
            self._nbr_up = up
        return self._nbr_up


TRUST_UNIQUE_INPUT_COORDS = True   # caller-supplied coordinate sets are taken as unique (torchsparse's contract), verified on the device


def mark_unique(cmaps: dict, coords: torch.Tensor) -> None:
    """Record in a tensor's cmaps that `coords` has unique rows BY CONSTRUCTION (output sites of a strided convolution, a voxeliser's
    output): neighbor_table_of may then hand its table to the pair planner as submanifold without relying on the caller's word."""
    cmaps[("link_unique", coords.data_ptr(), coords.shape[0])] = True


def neighbor_table_of(x: SparseTensor, kernel_size):
    """Per-output neighbour table of a stride-1 convolution over x's voxels: (int32[N, K], spatial tile order or
    None), cached on the tensor's kmaps (every convolution with this kernel size over these coordinates shares it)."""
    key = ("link_conv_nbr", x.C.data_ptr(), x.C.shape[0], x.s, tuple(kernel_size))
    nbr = x.kmaps.get(key)
    if nbr is None:
        # neighbour of voxel i at coords_i + offset*tensor_stride (conv.py:105-113: stride=input.stride)
        ts = int(x.s[0])
        bkey = ("link_bounds", x.C.data_ptr(), x.C.shape[0])
        bounds = x.cmaps.get(bkey)
        if bounds is None or x.cmaps.get(("link_bounds_unchecked", x.C.data_ptr(), x.C.shape[0])):
            # one bounding-box pass per coordinate set, shared with the block index; bounds that came from the
            # caller's metadata are not trusted here (the cell table is addressed with them)
            from .index import coords_bounds
            fresh = coords_bounds(x.C.contiguous())
            if bounds is None and x.C.is_contiguous():
                x.cmaps[bkey] = fresh
            bounds = fresh
        try:
            nbr = foreign_neighbor_map(x.C, kernel_size[0], step=ts, bounds=bounds)
        except GridTooLarge:
            offs = get_kernel_offsets(kernel_size, stride=x.s, device=x.F.device)
            nbr = sphashquery(sphash(x.C, offs), sphash(x.C)).t().contiguous().int()
        try:
            # structural mark for the pair plan: odd kernel over ONE coordinate set whose rows are unique -> the centre column is
            # the identity.  Uniqueness is structurally known for coordinate sets this package produced (mark_unique: the output
            # sites of strided convolutions); for coordinates handed in by the caller it is torchsparse's contract (sparse_quantize,
            # tensor.py) and TRUST_UNIQUE_INPUT_COORDS decides whether the claim is made -- it is then VERIFIED on the device
            # (rows whose centre neighbour is not the row itself are counted by link_pair_plan_build) and a wrong claim raises at
            # the next plan.  None = "unknown": the plan is laid out on the host from exact counts (one round trip).
            odd = all(int(k) % 2 == 1 for k in kernel_size)
            known = bool(x.cmaps.get(("link_unique", x.C.data_ptr(), x.C.shape[0])))
            nbr._link_subm = (True if (known or TRUST_UNIQUE_INPUT_COORDS) else None) if odd else False
        except AttributeError:
            pass
        nbr = (nbr, _TileOrder(x.C, 4 * ts))
        x.kmaps[key] = nbr
    return nbr


class _TileOrder:
    """Spatially coherent voxel order for the TABLE kernel's tile-level skipping: voxels grouped by (4*stride)^3
    blocks of the dense block grid (the LinK index with a small block edge).  Built on first use -- frames that
    run the pair-list kernels never need it."""

    def __init__(self, coords: torch.Tensor, edge: int):
        self._coords, self._edge, self._perm, self._done = coords, edge, None, False

    def tensor(self) -> Optional[torch.Tensor]:
        """The permutation int32[N] (None beyond the dense-grid limit)."""
        self.data_ptr()
        return self._perm

    def data_ptr(self):
        if not self._done:
            try:
                self._perm = BlockIndex(self._coords, self._edge, want_idx64=False).perm
            except GridTooLarge:
                self._perm = None
            self._done, self._coords = True, None
        return self._perm.data_ptr() if self._perm is not None else None


class Conv3d(nn.Module):
    """Sparse convolution with the reference's parameter layout (`kernel` [K, Cin, Cout], optional `bias`;
    torchsparse/nn/modules/conv.py:15-72; init U(+-1/sqrt(Cin*K)), or Cout*K when transposed) for the forms
    the LinK networks use (linkunet.py:40-92,109): odd cubic kernels at stride 1 (submanifold), and
    kernel 2 / stride 2 down-sampling and transposed up-sampling.  Kernel maps are per-output neighbour
    tables from the dense cell table (HIP) -- the same relation the reference builds with sphash ->
    sphashquery -> nonzero (nn/functional/conv.py:103-122) -- cached on the tensor's kmaps under the
    reference's key; contraction by the output-stationary MFMA kernel (include/link_amd.h section D)."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1,
                 dilation: int = 1, bias: bool = False, transposed: bool = False) -> None:
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = make_ntuple(kernel_size, 3)
        self.stride = make_ntuple(stride, 3)
        self.dilation, self.transposed = dilation, transposed
        if len(set(self.kernel_size)) != 1 or len(set(self.stride)) != 1:
            raise NotImplementedError("cubic kernels / isotropic strides only")
        ks, st = self.kernel_size[0], self.stride[0]
        if not ((st == 1 and ks % 2 == 1 and not transposed) or (st == 2 and ks == 2)):
            raise NotImplementedError("link_amd.Conv3d implements odd kernels at stride 1 and kernel 2 / stride 2 "
                                      "(down-sampling or transposed)")
        self.kernel_volume = ks ** 3
        if self.kernel_volume > 1:
            self.kernel = nn.Parameter(torch.zeros(self.kernel_volume, in_channels, out_channels))
        else:
            self.kernel = nn.Parameter(torch.zeros(in_channels, out_channels))
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_channels))
        else:
            self.register_parameter("bias", None)
        std = 1.0 / math.sqrt((out_channels if transposed else in_channels) * self.kernel_volume)
        self.kernel.data.uniform_(-std, std)
        if self.bias is not None:
            self.bias.data.uniform_(-std, std)

    def _neighbor_table(self, x: SparseTensor):
        """(int32[N, K] input row of every (output voxel, kernel offset), -1 absent; spatial voxel order or
        None), cached on the tensor's kmaps like the reference's kernel maps (nn/functional/conv.py:103,122)."""
        return neighbor_table_of(x, self.kernel_size)

    def _strided_map(self, x: SparseTensor) -> _StridedMap:
        """Down-sampling kernel map, cached under the reference's key (conv.py:103): output coordinates =
        unique(floor(c / (stride*ts)) * (stride*ts)) ordered by (batch, x, y, z) as spdownsample does
        (downsample.py:26-49), table[j,k] = input row at out_coords[j] + offset_k * ts."""
        key = (x.s, self.kernel_size, self.stride, self.dilation)
        km = x.kmaps.get(key)
        if km is None:
            ts = int(x.s[0])
            ss = ts * self.stride[0]
            # unique rows in (batch, x, y, z) order through ONE linear key (a 1-D sort instead of a 4-column one)
            bkey = ("link_bounds", x.C.data_ptr(), x.C.shape[0])
            bounds = x.cmaps.get(bkey)
            if bounds is None or x.cmaps.get(("link_bounds_unchecked", x.C.data_ptr(), x.C.shape[0])):
                from .index import coords_bounds
                bounds = coords_bounds(x.C.contiguous())
                if x.C.is_contiguous() and bkey not in x.cmaps:
                    x.cmaps[bkey] = bounds
            # sorted unique (batch, x', y', z') rows of the block coordinates through the dense cell grid (link_index_cells:
            # four launches and the one host round trip that sizes the output, instead of a 1-D key, torch.unique's device
            # sort with its own round trip, and the decode)
            from .index import unique_cells
            lo = [bounds[0][k] // ss for k in range(3)]
            hi = [bounds[1][k] // ss for k in range(3)]
            q = torch.div(x.C[:, :3], ss, rounding_mode="floor")
            rows = torch.cat([x.C[:, 3:4], q], 1).int().contiguous()
            out_c = None
            try:
                sites, hdr = unique_cells(rows, ((bounds[0][3], lo[0], lo[1], lo[2]), (bounds[1][3], hi[0], hi[1], hi[2])))
                m = int(hdr[L.HDR_M].item())
                out_c = torch.cat([sites[:m, 1:4] * ss, sites[:m, 0:1]], 1).contiguous()
            except GridTooLarge:
                pass
            if out_c is None:                           # grid beyond the dense-table limit: sort-based unique
                ext = [hi[k] - lo[k] + 1 for k in range(3)]
                ql = q.long()
                lin = (((x.C[:, 3].long() - bounds[0][3]) * ext[0] + (ql[:, 0] - lo[0])) * ext[1] + (ql[:, 1] - lo[1])) * ext[2] + (ql[:, 2] - lo[2])
                u = torch.unique(lin)
                oz = u % ext[2]; r1 = torch.div(u, ext[2], rounding_mode="floor")
                oy = r1 % ext[1]; r2 = torch.div(r1, ext[1], rounding_mode="floor")
                ox = r2 % ext[0]; ob = torch.div(r2, ext[0], rounding_mode="floor")
                out_c = torch.stack([(ox + lo[0]) * ss, (oy + lo[1]) * ss, (oz + lo[2]) * ss, ob + bounds[0][3]], 1).int().contiguous()
            mark_unique(x.cmaps, out_c)                    # unique rows by construction (sorted unique cells)
            # the coarse set's bounding box follows from the fine one: no second pass over the coordinates, no round trip
            x.cmaps.setdefault(("link_bounds", out_c.data_ptr(), out_c.shape[0]),
                               ((lo[0] * ss, lo[1] * ss, lo[2] * ss, bounds[0][3]), (hi[0] * ss, hi[1] * ss, hi[2] * ss, bounds[1][3])))
            try:
                down = foreign_neighbor_map(out_c, 2, step=ts, table_rows=x.C, bounds=bounds)
            except GridTooLarge:
                offs = get_kernel_offsets(self.kernel_size, stride=x.s, device=x.F.device)
                down = sphashquery(sphash(out_c, offs), sphash(x.C)).t().contiguous().int()
            try:
                down._link_subm = False                     # structural mark for the pair plan: a gather table between two site sets
            except AttributeError:
                pass
            km = _StridedMap(out_c, down, None)             # the transposed direction's table: built on first use
            km._n_in = x.C.shape[0]
            x.kmaps[key] = km
        return km

    def forward(self, x: SparseTensor) -> SparseTensor:
        feats = x.F
        if self.stride[0] == 1:
            if self.kernel_volume == 1:
                out = feats.matmul(self.kernel)
            else:
                nbr, order = self._neighbor_table(x)
                out = _SubmConv.apply(feats, self.kernel, nbr, order)
            coords, stride = x.C, x.s
        elif not self.transposed:
            km = self._strided_map(x)
            out = _GatherConv.apply(feats, self.kernel, km.nbr_down, km.nbr_up)
            coords, stride = km.out_coords, tuple(x.s[k] * self.stride[k] for k in range(3))
        else:
            stride = tuple(x.s[k] // self.stride[k] for k in range(3))
            km = x.kmaps[(stride, self.kernel_size, self.stride, self.dilation)]   # the matching down-conv's map
            out = _GatherConv.apply(feats, self.kernel, km.nbr_up, km.nbr_down)
            coords = x.cmaps[stride]
        if self.bias is not None:
            out = out + self.bias
        y = SparseTensor(out, coords, stride)
        y.cmaps, y.kmaps = x.cmaps, x.kmaps
        y.cmaps.setdefault(y.stride, y.coords)
        return y

    def forward_affine(self, x: SparseTensor, scale: torch.Tensor, shift: torch.Tensor, relu: bool) -> SparseTensor:
        """Inference only: relu?(conv(x) * scale + shift) in ONE launch -- the convolution with a following
        BatchNorm (running statistics, folded to scale / shift together with this module's bias by the caller)
        and ReLU in its finish phase (linkunet.py:23-92: BasicConvolutionBlock / ResidualBlock / *_tail)."""
        feats = x.F
        if self.stride[0] == 1:
            coords, stride = x.C, x.s
            table, order = (None, None) if self.kernel_volume == 1 else self._neighbor_table(x)
        elif not self.transposed:
            km = self._strided_map(x)
            table, order = km.nbr_down, None
            coords, stride = km.out_coords, tuple(x.s[k] * self.stride[k] for k in range(3))
        else:
            stride = tuple(x.s[k] // self.stride[k] for k in range(3))
            km = x.kmaps[(stride, self.kernel_size, self.stride, self.dilation)]
            table, order = km.nbr_up, None
            coords = x.cmaps[stride]
        if table is None:                               # 1x1x1: a dense GEMM on the rows
            out = torch.addcmul(shift, feats.float().matmul(self.kernel.detach()), scale)
            out = torch.relu_(out) if relu else out
        else:
            out = subm_conv_ln_add_relu(feats, self.kernel, table, order, scale, shift, 0.0, None, relu=relu, affine=True)
        y = SparseTensor(out, coords, stride)
        y.cmaps, y.kmaps = x.cmaps, x.kmaps
        y.cmaps.setdefault(y.stride, y.coords)
        return y


def conv3d(input: SparseTensor, weight: torch.Tensor, kernel_size, bias: Optional[torch.Tensor] = None, stride=1,
           dilation=1, transposed: bool = False) -> SparseTensor:
    """Functional form of the sparse convolution (torchsparse/nn/functional/conv.py:83-147: same arguments, same
    kmaps / cmaps contract) on the kernels behind link_amd.Conv3d; forms as the module's (odd kernels at stride 1,
    kernel 2 / stride 2 down and transposed)."""
    ks, st = make_ntuple(kernel_size, 3), make_ntuple(stride, 3)
    shell = Conv3d.__new__(Conv3d)                     # the module's forward with the caller's tensors as parameters
    nn.Module.__init__(shell)
    shell.kernel_size, shell.stride, shell.dilation, shell.transposed = ks, st, dilation, transposed
    if len(set(ks)) != 1 or len(set(st)) != 1 or not ((st[0] == 1 and ks[0] % 2 == 1 and not transposed) or
                                                      (st[0] == 2 and ks[0] == 2)):
        raise NotImplementedError("link_amd conv3d: odd cubic kernels at stride 1, kernel 2 / stride 2 (down or transposed)")
    shell.kernel_volume = ks[0] ** 3
    shell.in_channels, shell.out_channels = weight.shape[-2], weight.shape[-1]
    shell.__dict__["kernel"], shell.__dict__["bias"] = weight, bias
    return Conv3d.forward(shell, input)


def spdownsample(coords: torch.Tensor, stride=2, kernel_size=2, tensor_stride=1) -> torch.Tensor:
    """Output coordinates of a strided convolution (torchsparse/nn/functional/downsample.py:11-51): with
    stride in {1, kernel_size} per axis the coordinates are floored to multiples of stride * tensor_stride;
    otherwise every position input + offset that lies on the coarse lattice (and not below the inputs' minimum)
    is an output.  Rows unique, ordered by (batch, x, y, z)."""
    stride, ks, ts = make_ntuple(stride, 3), make_ntuple(kernel_size, 3), make_ntuple(tensor_stride, 3)
    ss = torch.tensor([stride[k] * ts[k] for k in range(3)], dtype=torch.int32, device=coords.device)
    if all(stride[k] in (1, ks[k]) for k in range(3)):
        c = coords.clone()
        c[:, :3] = torch.div(c[:, :3], ss, rounding_mode="floor") * ss
    else:
        offs = get_kernel_offsets(ks, ts, device=coords.device)
        lo = coords[:, :3].min(0, keepdim=True).values
        x = (coords[:, None, :3] + offs[None]).reshape(-1, 3)
        b = coords[:, 3:].repeat_interleave(offs.shape[0], 0)
        keep = ((x % ss == 0) & (x >= lo)).all(1)
        c = torch.cat([x, b], 1)[keep]
    return torch.unique(c[:, [3, 0, 1, 2]], dim=0)[:, [1, 2, 3, 0]].contiguous()


# a voxel with more neighbours than this runs on the output-stationary table kernel (conv.hip): measured
# cross-over of the two forms on MI355X (tools/convbench.py)
PAIR_DENSITY_MAX = 17.0
ASYNC_PAIR_PLANS = True      # lay pair plans out on the device when the table's structure is known (no host round trip)
ASYNC_PAIR_PLAN_MAX = 8_000_000   # ... up to this many table entries: the buffers are sized for every entry being a pair
                                  # (contribution rows at 64 channels: 256 B per entry), beyond it the exact host layout
ASYNC_PAIR_PLAN_MAX_BYTES = 1 << 30   # ... and up to this many bytes of capacity-sized contribution rows (entries x Cout x 4)
_PENDING_PLANS: list = []             # weak references to device-laid-out plans whose counts have not been looked at yet


def _poll_pending_plans() -> None:
    """Look at the counts of every device-laid-out plan whose kernels have finished: a table handed over as submanifold that
    is not one (duplicate coordinates) raises HERE -- at the next plan, the next frame at the latest -- instead of only when
    somebody happens to ask the plan for its density."""
    keep, err = [], None
    waiting = False                                      # plans are issued in stream order: behind the first one whose kernels are
    for ref in _PENDING_PLANS:                           # still running nothing is asked (an event query is ~5 us of host time,
        plan = ref()                                     # and a frame with every map rebuilt makes tens of plans)
        if plan is None or plan.exact or plan._density is not None or plan._error is not None:
            continue
        if not waiting and plan._ev.query():
            try:
                plan._arrived(False)                     # a bad table: the plan keeps the error (and raises it again whenever IT is
            except L.LinkAmdError as e:                  # used) and leaves the list either way -- the verdict is reported from here
                err = err or e                           # exactly once, never again by the plans of unrelated, valid tables
        else:
            waiting = True
            keep.append(ref)
    _PENDING_PLANS[:] = keep
    if err is not None:
        raise err
_DENSITY_SEEN: Dict[int, float] = {}      # kernel volume -> pairs per row of the last plan whose counts reached the host
_BBOX_STATS_INIT: Dict[torch.device, torch.Tensor] = {}   # (bbox init, zeroed occupancy counters) per device


def _probe_stats(slots) -> Tuple[int, int, int]:
    """(voxels inside the grid, occupied cells, fullest cell) from link_dc_index_probe's 16 partial slots of 16 ints."""
    return (sum(slots[0:256:16]), sum(slots[1:256:16]), max(slots[2:256:16]))
_PINNED: list = []


def _pinned_slot():
    """(8 ints of pinned host memory, slot, generation) from a small ring (allocating pinned memory per plan would cost
    more than the round trip it replaces).  A slot is handed out again after 512 later plans; a plan that looks at its
    counts only then sees the generation moved on and reads its device header instead (_pinned_current)."""
    if not _PINNED:
        _PINNED.extend([torch.empty((512, 8), dtype=torch.int32).pin_memory(), 0, [0] * 512])
    buf, i, gen = _PINNED
    _PINNED[1] = (i + 1) % buf.shape[0]
    gen[i] += 1
    return buf[i], i, gen[i]


def _pinned_current(slot: int, generation: int) -> bool:
    return _PINNED[2][slot] == generation


class _PairPlan:
    """Pair-list kernel map of one per-output neighbour table (include/link_amd.h: link_conv_pairs_*): the
    reference's nbmaps / nbsizes (nn/functional/conv.py:109-122) regrouped for one launch -- contribution rows
    grouped by kernel offset in 128-row granules (`pair_in`, `wg_k`), and the CSR list of every output voxel's
    rows in ascending offset order (`ext_start`, `ext_list`).  For a submanifold table (odd kernel, same
    coordinates in and out) the centre pairs are the identity and occupy rows [0, n)."""

    def __init__(self, nbr: torch.Tensor, subm: Optional[bool] = None, cout_hint: int = 64):
        """`subm`: what the caller knows about the table structurally -- True: a submanifold table (odd kernel, the same
        unique coordinates in and out, so the centre column is the identity), False: not one, None: unknown.  With a
        structural answer and ASYNC_PAIR_PLANS the plan is laid out on the device (link_pair_plan_layout) over capacity-
        sized buffers and nothing waits for the host; the counts arrive later in pinned memory (`density`, `finalize`)."""
        if (subm is not None and ASYNC_PAIR_PLANS and nbr.is_cuda and 0 < nbr.shape[0] * nbr.shape[1] <= ASYNC_PAIR_PLAN_MAX
                and nbr.shape[0] * nbr.shape[1] * max(int(cout_hint), 1) * 4 <= ASYNC_PAIR_PLAN_MAX_BYTES):
            self._init_async(nbr, bool(subm))
            return
        self.exact = True
        self._error = None
        n, kvol = nbr.shape
        dev = nbr.device
        centre = kvol // 2
        lib, st = L.lib(), _st()
        nbr = nbr.contiguous()
        i32 = dict(dtype=torch.int32, device=dev)
        # pass 1 on the device (per-workgroup counts), then ONE host round trip for what the layout needs: pairs per
        # kernel offset and whether the centre column is the identity (include/link_amd.h: link_pair_plan_count / _fill)
        nwg = (n + 255) // 256
        wg_counts = torch.empty((max(nwg, 1), kvol + 1), **i32)
        row_info = torch.empty(max(n, 1), **i32)
        L.check(lib.link_pair_plan_count(nbr.data_ptr(), n, kvol, wg_counts.data_ptr(), row_info.data_ptr(), st),
                "link_pair_plan_count")
        import numpy as _np
        per_wg = wg_counts[:nwg].cpu().numpy() if n else _np.zeros((0, kvol + 1), dtype=_np.int32)   # G x (kvol+1) ints
        host = per_wg.sum(0).tolist()
        direct = bool(kvol % 2 == 1 and n > 0 and host[kvol] == 0)
        cnt_k = host[:kvol]
        if direct:
            cnt_k[centre] = 0
        self.pairs = int(sum(cnt_k))
        self.n, self.kvol, self.direct = n, kvol, direct
        self._density = (self.pairs + (n if direct else 0)) / max(n, 1)
        _DENSITY_SEEN[kvol] = self._density
        base_k, gran, acc = [], [], 0
        for c in cnt_k:
            base_k.append(acc)
            gran.append((c + 127) // 128)
            acc += gran[-1] * 128
        self.rows_pad = acc
        ext_cnt = (row_info[:n] & 0xFFFF) - (row_info[:n] >> 16) if direct else row_info[:n] & 0xFFFF
        ext_start = torch.zeros(n + 1, **i32)
        torch.cumsum(ext_cnt, 0, out=ext_start[1:])
        pair_io = torch.full((2, max(self.rows_pad, 1)), -1, **i32)      # input row | output row of every pair
        pair_in, pair_out = pair_io[0], pair_io[1]
        ext_list = torch.empty(max(self.pairs, 1), **i32)
        pk = per_wg[:, :kvol].astype(_np.int32)
        wg_base = (_np.cumsum(pk, 0, dtype=_np.int32) - pk).reshape(-1)              # exclusive scan over workgroups
        nwk = wg_base.size
        gstart = _np.concatenate([[0], _np.cumsum(gran)]).astype(_np.int32)          # first granule of every offset
        meta = _np.concatenate([_np.asarray(base_k, dtype=_np.int32), wg_base, gstart,
                                _np.repeat(_np.arange(kvol, dtype=_np.int32), gran)])
        meta = torch.from_numpy(meta).to(dev)                      # base_k | wg_base | gran_start | wg_k: one H2D
        if self.pairs:
            L.check(lib.link_pair_plan_fill(nbr.data_ptr(), n, kvol, 1 if direct else 0, meta.data_ptr(),
                                            meta[kvol:].data_ptr(), ext_start.data_ptr(), pair_in.data_ptr(),
                                            pair_out.data_ptr(), ext_list.data_ptr(), st), "link_pair_plan_fill")
        self._meta = meta
        self.pair_in, self.pair_out, self.ext_start, self.ext_list = pair_in, pair_out, ext_start, ext_list
        self.gran_start, self.wg_k = meta[kvol + nwk: kvol + nwk + kvol + 1], meta[kvol + nwk + kvol + 1:]
        self._contrib: Dict[int, torch.Tensor] = {}

    def _init_async(self, nbr: torch.Tensor, direct: bool):
        _poll_pending_plans()                            # an earlier plan's verdict (duplicate coordinates) surfaces here at the latest
        n, kvol = nbr.shape
        lib, st = L.lib(), _st()
        nbr = nbr.contiguous()
        # one arena for everything the three kernels touch (count -> layout -> fill, one FFI call: link_pair_plan_build);
        # nothing in it needs initialising.  Capacity: every (voxel, offset) a pair, every offset's last granule partly filled
        # (link_pair_plan_arena: the piece offsets, shared with the one-call block driver of include/link_amd.h section G)
        offs = (ctypes.c_int64 * 11)()
        gran_cap = int(lib.link_pair_plan_arena(n, kvol, 1 if direct else 0, offs))
        arena = torch.empty(int(offs[10]), dtype=torch.int32, device=nbr.device)
        self._adopt(n, kvol, direct, arena, offs, gran_cap)
        meta = self._meta
        L.check(lib.link_pair_plan_build(nbr.data_ptr(), n, kvol, 1 if direct else 0, gran_cap, arena[int(offs[0]):].data_ptr(),
                                         arena[int(offs[1]):].data_ptr(), meta.data_ptr(), meta[kvol:].data_ptr(),
                                         self.gran_start.data_ptr(), arena[int(offs[3]):].data_ptr(), self.wg_k.data_ptr(),
                                         self._hdr.data_ptr(), self.ext_start.data_ptr(), self.pair_in.data_ptr(),
                                         self.pair_out.data_ptr(), self.ext_list.data_ptr(), st), "link_pair_plan_build")
        self._post_build()

    def _adopt(self, n: int, kvol: int, direct: bool, arena: torch.Tensor, offs, gran_cap: int) -> None:
        """Views into a capacity-sized arena laid out by link_pair_plan_arena."""
        nwg = (n + 255) // 256
        cap_pairs = n * (kvol - (1 if direct else 0))
        self.n, self.kvol, self.direct, self.exact = n, kvol, direct, False
        self.pairs, self.rows_pad, self._density = None, gran_cap * 128, None
        self._error = None
        o = [int(v) for v in offs]
        meta = arena[o[2]: o[2] + kvol + nwg * kvol + kvol + 1]
        self._meta, self._hdr, self._arena = meta, arena[o[5]: o[5] + 8], arena
        self.ext_start = arena[o[6]: o[6] + n + 1]
        self.pair_in, self.pair_out = arena[o[7]: o[7] + self.rows_pad], arena[o[8]: o[8] + self.rows_pad]
        self.ext_list = arena[o[9]: o[9] + max(cap_pairs, 1)]
        self.gran_start, self.wg_k = meta[kvol + nwg * kvol:], arena[o[4]: o[4] + gran_cap]
        self._contrib = {}

    def _post_build(self) -> None:
        # the counts travel to pinned memory behind the kernels; whoever asks first after they arrived checks them
        self._host, self._slot, self._gen = _pinned_slot()
        self._host.copy_(self._hdr, non_blocking=True)
        self._ev = torch.cuda.Event()
        self._ev.record()
        _PENDING_PLANS.append(weakref.ref(self))

    @classmethod
    def _from_arena(cls, nbr: torch.Tensor, arena: torch.Tensor, offs, gran_cap: int) -> "_PairPlan":
        """The plan of a submanifold table whose arena link_elk_block_forward has already filled (same layout, same kernels)."""
        self = cls.__new__(cls)
        self._adopt(nbr.shape[0], nbr.shape[1], True, arena, offs, gran_cap)
        self._post_build()
        return self

    def _arrived(self, wait: bool) -> bool:
        if self._error is not None:                      # a plan whose table was bad stays bad: every use of it says so
            raise self._error
        if self.exact or self._density is not None:
            return True
        if wait:
            self._ev.synchronize()
        elif not self._ev.query():
            return False
        src = self._host if _pinned_current(self._slot, self._gen) else self._hdr      # slot reused since: read the device copy
        pairs, rows, gran, misses, over = [int(v) for v in src[:5].tolist()]
        if over or (self.direct and misses):
            self._error = L.LinkAmdError(
                (f"pair plan of a [{self.n}, {self.kvol}] neighbour table: the table handed over as submanifold is not one "
                 f"(duplicate coordinates? {misses} rows whose centre neighbour is not the row itself)") if misses else
                f"pair plan of a [{self.n}, {self.kvol}] neighbour table: granule capacity exceeded")
            raise self._error
        self.pairs = pairs
        self._density = (pairs + (self.n if self.direct else 0)) / max(self.n, 1)
        self._exact_rows = rows
        _DENSITY_SEEN[self.kvol] = self._density
        self._contrib.clear()                            # capacity-sized contribution rows (every entry a pair): the next request
        return True                                      # allocates the exact size (contrib)

    @property
    def density(self) -> float:
        """Pairs per output row.  A device-laid-out plan whose counts have not arrived yet answers with the last density seen
        for this kernel volume (frames of a stream resemble each other; the pair-list kernels are correct at any density)."""
        if self._arrived(False):
            return self._density
        return _DENSITY_SEEN.get(self.kvol, 0.0)

    @property
    def rows_launch(self) -> int:
        """Contribution rows a GEMM launch has to cover: the exact count once it is known, the capacity before (the
        granules behind the last one return at once)."""
        if self.exact or not self._arrived(False):
            return self.rows_pad
        return self._exact_rows

    def finalize(self) -> "_PairPlan":
        """Wait for the counts of a device-laid-out plan and trim it to its exact size (what the weight-gradient kernel,
        whose launch covers every granule, needs)."""
        if not self.exact:
            self._arrived(True)
            self.rows_pad = self._exact_rows
            self.wg_k = self.wg_k[: max(self._exact_rows // 128, 1)]
            self._contrib.clear()
            self.exact = True
        return self

    def contrib(self, cout: int, dtype=torch.float32) -> torch.Tensor:
        rows = self.rows_launch                          # the exact count once it has arrived, the capacity before
        if dtype != torch.float32:
            buf = self._contrib.get((cout, dtype))
            if buf is None:
                buf = self._contrib[(cout, dtype)] = torch.empty((max(rows, 1), cout), dtype=dtype, device=self.pair_in.device)
            return buf
        buf = self._contrib.get(cout)
        if buf is None:
            if len(self._contrib) >= 2:
                self._contrib.clear()
            buf = self._contrib[cout] = torch.empty((max(rows, 1), cout), dtype=torch.float32, device=self.pair_in.device)
        return buf


def _pair_plan(nbr: torch.Tensor, cin: int, cout: int) -> Optional[_PairPlan]:
    """The table's pair plan when the pair-list kernels should run it (sparse neighbourhoods, supported
    widths), else None; built once per table and cached on the (kmaps-cached) table tensor.  Tables that carry their
    builder's structural mark (`_link_subm`: neighbor_table_of / the strided convolutions' gather tables) get their plan
    laid out on the device without a host round trip when no gradient is being recorded."""
    if not L.lib().link_conv_pairs_supported(cin, cout) or nbr.shape[1] > 64:
        return None                 # the pair plan holds one 64-bit offset mask per voxel: 5^3 / 7^3 kernels run the table kernel
    plan = getattr(nbr, "_link_pairs", False)
    if plan is False:
        plan = _PairPlan(nbr, None if torch.is_grad_enabled() else getattr(nbr, "_link_subm", None), cout_hint=cout)
        nbr._link_pairs = plan
    if torch.is_grad_enabled() and not plan.exact:
        plan.finalize()
    return plan if plan.density <= PAIR_DENSITY_MAX else None


def _pad_in_channels(f: torch.Tensor, w: torch.Tensor, kernel: torch.Tensor, cin: int, cout: int):
    """Few input channels (the networks' first layers: 4 or 5 point features) run the MFMA kernels with the rows
    zero-padded to 16 channels (the weight's extra rows are zero; cached per kernel version): (f, w, cin)."""
    cin_p = (cin + 15) // 16 * 16
    if cin_p == cin or cin > 16 or not L.lib().link_conv_pairs_supported(cin_p, cout):
        return f, w, cin
    hit = getattr(kernel, "_link_padded", None)
    ver = (kernel._version, kernel.data_ptr())
    if hit is None or hit[0] != ver:
        wp = torch.zeros((w.shape[0], cin_p, cout), dtype=torch.float32, device=w.device)
        wp[:, :cin] = w
        hit = (ver, wp)
        try:
            kernel._link_padded = hit
        except AttributeError:
            pass
    return TF.pad(f, (0, cin_p - cin)), hit[1], cin_p


AMP_CONTRIB16 = (torch.float16,)     # row types whose per-offset contribution rows are stored 16-bit too (as the reference's half mm does)
AMP_MFMA = True      # half rows: round the weights to the row type too (the reference's custom_fwd cast) and use the 16-bit matrix cores


def _amp_weights(w: torch.Tensor, dtype) -> torch.Tensor:
    """w [K, cin, cout] fp32 -> [K, cout, cin] in the row type, cached on the fp32 tensor (itself cached per
    parameter version by the callers: Conv3d.kernel, kernel_kio(), _pad_in_channels)."""
    hit = getattr(w, "_link_amp", None)
    ver = (w._version, w.data_ptr(), dtype)
    if hit is None or hit[0] != ver:
        hit = (ver, w.detach().float().transpose(1, 2).contiguous().to(dtype))
        try:
            w._link_amp = hit
        except AttributeError:
            pass
    return hit[1]


SPLIT_MFMA = True    # fp32 rows, inference forms: the pair GEMM as an fp16 hi/lo split on the f16 matrix cores


def _split_weights(w: torch.Tensor):
    """w [K, cin, cout] fp32 -> (fp16 [K, cout, 2 cin] = hi | lo per output channel, int32[1] flag: some |w| >= 2^15),
    cached on the long-lived fp32 tensor; no host synchronisation."""
    hit = getattr(w, "_link_split", None)
    ver = (w._version, w.data_ptr())
    if hit is None or hit[0] != ver:
        wt = w.detach().float().transpose(1, 2).contiguous()
        hi = wt.half()
        lo = (wt - hi.float()).half()
        hit = (ver, torch.cat([hi, lo], dim=2).contiguous(), (wt.abs().max() >= 32768.0).to(torch.int32).reshape(1))
        try:
            w._link_split = hit
        except AttributeError:
            pass
    return hit[1], hit[2]


def _conv_pairs(plan: _PairPlan, f, w, cin, cout, out, bias=None, ln=None, addend=None, relu=False, w_key=None, split=False):
    """f / addend / out rows share one dtype (fp32, fp16 or bf16: link_conv_*_io); everything else fp32.  Half rows
    with AMP_MFMA: weights rounded to the row type (cached on `w_key`, the caller's long-lived weight tensor)."""
    lib, st = L.lib(), _st()
    io = _IO_DTYPES[f.dtype]
    amp = AMP_MFMA and io != L.IO_F32
    c16 = amp and plan.direct and f.dtype in AMP_CONTRIB16      # 16-bit contribution rows (submanifold maps)
    contrib = plan.contrib(cout, f.dtype if c16 else torch.float32)
    cdt = io if c16 else L.IO_F32
    if amp:
        w = _amp_weights(w_key if w_key is not None and w_key.shape == w.shape else w, f.dtype)
        L.check(lib.link_conv_pairs_gemm_amp(f.data_ptr(), io, plan.pair_in.data_ptr(), plan.wg_k.data_ptr(), plan.rows_launch,
                                             w.data_ptr(), cin, cout, contrib.data_ptr(), cdt, st), "link_conv_pairs_gemm_amp")
    elif split and SPLIT_MFMA and io == L.IO_F32 and cin >= 32:
        ws, big = _split_weights(w_key if w_key is not None and w_key.shape == w.shape else w)
        L.check(lib.link_conv_pairs_gemm_split(f.data_ptr(), plan.pair_in.data_ptr(), plan.wg_k.data_ptr(), plan.rows_launch,
                                               ws.data_ptr(), w.data_ptr(), big.data_ptr(), cin, cout, contrib.data_ptr(), st),
                "link_conv_pairs_gemm_split")
    else:
        L.check(lib.link_conv_pairs_gemm_io(f.data_ptr(), io, plan.pair_in.data_ptr(), plan.wg_k.data_ptr(), plan.rows_launch,
                                            w.data_ptr(), cin, cout, contrib.data_ptr(), st), "link_conv_pairs_gemm")
    ln_w, ln_b, eps = ln if ln is not None else (None, None, 0.0)
    if plan.direct:
        centre_sum = lib.link_conv_centre_sum_amp if amp else lib.link_conv_centre_sum_io
        L.check(centre_sum(f.data_ptr(), w.data_ptr(), plan.kvol // 2, contrib.data_ptr(), *((cdt,) if amp else ()), plan.rows_pad,
                           plan.ext_start.data_ptr(), plan.ext_list.data_ptr(), plan.n, cin, cout,
                           bias.data_ptr() if bias is not None else None,
                           ln_w.data_ptr() if ln_w is not None else None,
                           ln_b.data_ptr() if ln_b is not None else None, float(eps),
                           addend.data_ptr() if addend is not None else None, int(relu),
                           out.data_ptr(), io, st), "link_conv_centre_sum")
        return out
    L.check(lib.link_conv_pairs_sum_io(contrib.data_ptr(), plan.ext_start.data_ptr(), plan.ext_list.data_ptr(), plan.n,
                                       0, cout,
                                       bias.data_ptr() if bias is not None else None,
                                       ln_w.data_ptr() if ln_w is not None else None,
                                       ln_b.data_ptr() if ln_b is not None else None, float(eps),
                                       addend.data_ptr() if addend is not None else None, int(relu),
                                       out.data_ptr(), io, st), "link_conv_pairs_sum")
    return out


def can_fold_batchnorm(bn: nn.Module) -> bool:
    """A BatchNorm can be folded into a convolution's finish phase when it normalises with running statistics."""
    return isinstance(bn, nn.BatchNorm1d) and bn.track_running_stats and bn.running_mean is not None and bn.running_var is not None


def invalidate_derived_weights(model: nn.Module) -> None:
    """Drop every cache derived from parameter VALUES (folded BatchNorm scale / shift, transposed / rounded / split
    convolution weights).  The caches are keyed on tensor versions and storage pointers, which writes through `.data`
    (EMA swaps, `param.data.copy_`) do not change: call this after such writes.  `load_state_dict` bumps versions and
    needs no call."""
    for m in model.modules():
        m.__dict__.pop("_link_fold", None)
        if "_kio" in m.__dict__:
            m._kio = None
        for p in m.parameters(recurse=False):
            for k in ("_link_amp", "_link_split", "_link_padded", "_link_flip"):
                p.__dict__.pop(k, None)


def fold_batchnorm(bn: nn.BatchNorm1d, conv_bias: Optional[torch.Tensor] = None):
    """(scale, shift) with bn(y + conv_bias) == y * scale + shift for a BatchNorm in inference mode (running
    statistics; affine optional); cached on the module and refreshed when any of its tensors changes (eight tiny
    launches otherwise, per call).  Writes through `.data` do not bump versions: see invalidate_derived_weights."""
    if not can_fold_batchnorm(bn):
        raise L.LinkAmdError("fold_batchnorm: the BatchNorm keeps no running statistics (track_running_stats=False)")
    ts = tuple(t for t in (bn.weight, bn.bias, bn.running_mean, bn.running_var) if t is not None)
    ver = tuple((t._version, t.data_ptr()) for t in ts) + \
        ((conv_bias._version, conv_bias.data_ptr()) if conv_bias is not None else ()) + (bn.running_mean.device,)
    hit = bn.__dict__.get("_link_fold")
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    sc = torch.rsqrt(bn.running_var.float() + bn.eps)
    if bn.weight is not None:
        sc = bn.weight.detach().float() * sc
    sh = -bn.running_mean.float() * sc
    if bn.bias is not None:
        sh = sh + bn.bias.detach().float()
    if conv_bias is not None:
        sh = sh + conv_bias.detach().float() * sc
    sc, sh = sc.contiguous(), sh.contiguous()
    bn.__dict__["_link_fold"] = (ver, sc, sh)
    return sc, sh


RESIDENT_FORM = True         # narrow fp32 layers (16 / 32 channels): the table kernel with every W_k resident in LDS (conv.hip)
RESIDENT_TILE_ORDER = False  # ... over the voxels in memory order (LiDAR frames arrive spatially coherent; no tile index to build)
_RESIDENT_SHAPES = ((16, 16), (32, 32), (16, 32), (32, 16))


def _resident_form(cin: int, cout: int, kvol: int, half: bool, form: str) -> bool:
    """fp32 rows: link_subm_conv_forward / _ln_add_relu take these shapes to the resident-weights kernel on their own;
    16-bit rows: link_subm_conv_resident_amp (weights rounded to the row type: the AMP contract, AMP_MFMA)."""
    return (RESIDENT_FORM and form == "auto" and kvol <= 27 and (cin, cout) in _RESIDENT_SHAPES and (not half or AMP_MFMA))


def _conv_resident_amp(f, w, w_key, nbr, order, cin, cout, ln=None, addend=None, flags=0):
    """16-bit rows through the AMP resident-weights kernel: out in the row type."""
    n, kvol = nbr.shape
    wt = _amp_weights(w_key if w_key is not None and w_key.shape == w.shape else w, f.dtype)
    out = torch.empty((n, cout), dtype=f.dtype, device=f.device)
    lw, lb, eps = ln if ln is not None else (None, None, 0.0)
    L.check(L.lib().link_subm_conv_resident_amp(
        f.data_ptr(), _IO_DTYPES[f.dtype], nbr.contiguous().data_ptr(), wt.data_ptr(),
        order.data_ptr() if order is not None else None, n, cin, cout, kvol,
        lw.data_ptr() if lw is not None else None, lb.data_ptr() if lb is not None else None, float(eps),
        addend.data_ptr() if addend is not None else None, int(flags), out.data_ptr(), _st()), "link_subm_conv_resident_amp")
    return out


def subm_conv(feats: torch.Tensor, kernel: torch.Tensor, nbr: torch.Tensor,
              order: Optional[torch.Tensor] = None, form: str = "auto") -> torch.Tensor:
    """out = sum_k feats[nbr[:,k]] @ kernel[k] (include/link_amd.h section D), no autograd.  `form`:
    "auto" picks the pair-list kernels on sparse neighbourhoods and the output-stationary table kernel on
    dense ones; "table" / "pairs" force one (tests, benches)."""
    if feats.device.type != "cuda":
        raise L.LinkAmdError("subm_conv needs GPU tensors (HIP path; no CPU fallback)")
    cin = feats.shape[1]
    kvol, cin2, cout = kernel.shape
    n = nbr.shape[0]                                   # output rows (== input rows for the submanifold form)
    assert cin2 == cin and nbr.shape == (n, kvol) and nbr.dtype == torch.int32
    half = feats.dtype in (torch.float16, torch.bfloat16)      # AMP rows: the pair-list kernels take them as they are
    f = feats.detach().contiguous()
    f = f if half else f.float()
    w = kernel.detach().contiguous().float()
    if form != "table" and w.ndim == 3 and cout < 16 and n > 0 and L.lib().link_conv_pairs_supported((cin + 15) // 16 * 16, 16):
        # few OUTPUT channels (the input gradient of a network's first layer): zero columns up to 16, slice afterwards
        wp = torch.zeros((kvol, cin, 16), dtype=torch.float32, device=w.device)
        wp[:, :, :cout] = w
        return subm_conv(f, wp, nbr, order, form)[:, :cout].contiguous()
    if form != "table" and w.ndim == 3:
        f, w, cin = _pad_in_channels(f, w, kernel, cin, cout)
    resident = _resident_form(cin, cout, kvol, half, form)
    plan = _pair_plan(nbr, cin, cout) if form != "table" and n > 0 and not resident else None
    if form == "pairs" and plan is None:
        plan = getattr(nbr, "_link_pairs", None)
        if plan is None:
            raise L.LinkAmdError(f"subm_conv(form='pairs'): widths {cin}->{cout} not supported by the pair-list kernels")
    if resident and not RESIDENT_TILE_ORDER:
        order = None
    if resident and half and n > 0:
        return _conv_resident_amp(f, w, kernel, nbr, order, cin, cout)
    if plan is not None:
        return _conv_pairs(plan, f, w, cin, cout, torch.empty((n, cout), dtype=f.dtype, device=feats.device), w_key=kernel)
    f = f.float()                                      # the table kernel is fp32: half rows are widened here
    out = torch.empty((n, cout), dtype=torch.float32, device=feats.device)
    L.check(L.lib().link_subm_conv_forward(f.data_ptr(), nbr.contiguous().data_ptr(), w.data_ptr(),
                                           order.data_ptr() if order is not None else None, n, cin, cout,
                                           kvol, out.data_ptr(), _st()), "link_subm_conv_forward")
    return out.to(feats.dtype) if half else out        # half rows in -> half rows out on either form


def subm_conv_ln_add_relu(feats: torch.Tensor, kernel: torch.Tensor, nbr: torch.Tensor,
                          order: Optional[torch.Tensor], ln_w: torch.Tensor, ln_b: torch.Tensor, eps: float,
                          addend: Optional[torch.Tensor], relu: bool = True, form: str = "auto",
                          affine: bool = False) -> torch.Tensor:
    """relu(addend + LayerNorm(subm_conv(feats))): the block's tail (linkunet.py:183) fused into the
    convolution's store phase (include/link_amd.h: link_subm_conv_ln_add_relu, or the epilogue of
    link_conv_pairs_sum on sparse neighbourhoods).  `affine=True`: ln_w / ln_b are a per-channel scale / shift
    (an inference BatchNorm folded with the convolution's bias) and no statistics are taken:
    relu(addend + conv * ln_w + ln_b) -- the BN / residual / ReLU epilogues of the detection stages.  No autograd."""
    cin = feats.shape[1]
    kvol, cin2, cout = kernel.shape
    n = nbr.shape[0]                                   # output rows (the table is per output row)
    assert cin2 == cin and nbr.shape == (n, kvol) and nbr.dtype == torch.int32
    assert addend is None or addend.shape == (n, cout)
    half = feats.dtype in (torch.float16, torch.bfloat16)      # AMP rows stay as they are on the pair-list kernels
    f = feats.detach().contiguous()
    f = f if half else f.float()
    w = kernel.detach().contiguous().float()
    lw, lb = ln_w.detach().contiguous().float(), ln_b.detach().contiguous().float()
    if form != "table":
        f, w, cin = _pad_in_channels(f, w, kernel, cin, cout)
    resident = _resident_form(cin, cout, kvol, half, form)
    plan = _pair_plan(nbr, cin, cout) if form != "table" and n > 0 and not resident else None
    if form == "pairs" and plan is None:
        plan = getattr(nbr, "_link_pairs", None)
    if resident and not RESIDENT_TILE_ORDER:
        order = None
    flags = (1 if relu else 0) | (2 if affine else 0)
    if resident and half and n > 0:
        add = addend.detach().contiguous().to(f.dtype) if addend is not None else None
        return _conv_resident_amp(f, w, kernel, nbr, order, cin, cout, ln=(lw, lb, eps), addend=add, flags=flags)
    if plan is not None:
        add = addend.detach().contiguous().to(f.dtype) if addend is not None else None
        return _conv_pairs(plan, f, w, cin, cout, torch.empty((n, cout), dtype=f.dtype, device=feats.device),
                           ln=(lw, lb, eps), addend=add, relu=flags, w_key=kernel, split=True)
    f = f.float()                                      # the table kernel is fp32: half rows are widened here
    add = addend.detach().contiguous().float() if addend is not None else None
    out = torch.empty((n, cout), dtype=torch.float32, device=feats.device)
    L.check(L.lib().link_subm_conv_ln_add_relu(
        f.data_ptr(), nbr.contiguous().data_ptr(), w.data_ptr(), order.data_ptr() if order is not None else None,
        n, cin, cout, kvol, lw.data_ptr(), lb.data_ptr(), float(eps), add.data_ptr() if add is not None else None,
        flags, out.data_ptr(), _st()), "link_subm_conv_ln_add_relu")
    return out.to(feats.dtype) if half else out


class _Tail(torch.autograd.Function):
    """relu(addend + LayerNorm(x)) with a hand-written backward (link_ln_add_relu_forward/backward)."""

    @staticmethod
    def forward(ctx, x, addend, ln_w, ln_b, eps):
        n, c = x.shape
        x = x.detach().contiguous().float()
        a = addend.detach().contiguous().float()
        w, b = ln_w.detach().contiguous().float(), ln_b.detach().contiguous().float()
        y = torch.empty_like(x)
        L.check(L.lib().link_ln_add_relu_forward(x.data_ptr(), a.data_ptr(), w.data_ptr(), b.data_ptr(), n, c,
                                                 float(eps), y.data_ptr(), _st()), "link_ln_add_relu_forward")
        ctx.save_for_backward(x, y, w)
        ctx.eps = float(eps)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y, w = ctx.saved_tensors
        n, c = x.shape
        lib = L.lib()
        rows = int(lib.link_elk_mid_partial_rows())
        g = g.contiguous().float()
        g_add, g_x = torch.empty_like(x), torch.empty_like(x)
        part = torch.empty(rows * 2 * c + 2 * c, dtype=torch.float32, device=x.device)
        L.check(lib.link_ln_add_relu_backward(g.data_ptr(), y.data_ptr(), x.data_ptr(), w.data_ptr(), n, c, ctx.eps,
                                              g_add.data_ptr(), g_x.data_ptr(), part.data_ptr(), _st()),
                "link_ln_add_relu_backward")
        tot = part[rows * 2 * c:]
        L.check(lib.link_sum_partials(part.data_ptr(), 2 * c, None, 0, None, 0, rows, tot.data_ptr(), _st()),
                "link_sum_partials")
        return g_x, g_add, tot[:c].view_as(w), tot[c:].view_as(w), None


WGRAD_TABLE_SQUARE = True      # square widths <= 64: the table weight-gradient kernel (else the pair list, like every other width)


def _conv_weight_grad(feats, g, nbr, kernel_shape):
    """g_w[k] = feats[nbr[:,k]]^T @ g: the MFMA kernel for square widths <= 64, else per offset gather +
    batched library GEMM."""
    kvol, cin, cout = kernel_shape
    n_out = nbr.shape[0]
    # square widths <= 64 keep the table kernel below (measured equal on the kernels, and it needs no second launch for
    # the centre); every other width pair the forward kernels take -- wide and rectangular layers -- runs the pair list
    square_small = WGRAD_TABLE_SQUARE and cin == cout and cin <= 64 and cin % 4 == 0
    if (n_out > 0 and len(kernel_shape) == 3 and cin < 16 and not square_small
            and L.lib().link_conv_pairs_supported(16, cout) and _pair_plan(nbr, 16, cout) is not None):
        # a network's first layer (4 or 5 point features): rows zero-padded to 16 channels as in the forward
        # (_pad_in_channels), the gradient of the padding rows dropped
        return _conv_weight_grad(TF.pad(feats.detach().float(), (0, 16 - cin)), g, nbr, (kvol, 16, cout))[:, :cin].contiguous()
    plan = _pair_plan(nbr, cin, cout) if (n_out > 0 and len(kernel_shape) == 3 and not square_small) else None
    if plan is not None:
        # pair-list form: one MFMA pass over the 128-pair granules + per-offset sums in granule order; the centre
        # offset of a submanifold table (identity pairs, not in the plan) is the plain feats^T . g
        lib = L.lib()
        plan.finalize()                                  # this launch covers every granule: a device-laid-out plan is trimmed first
        f = feats.detach().contiguous().float()
        gw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=g.device)
        n_dir = plan.n if plan.direct else 0
        part = torch.empty((max(plan.rows_pad // 128 + (n_dir + 127) // 128, 1), cin, cout), dtype=torch.float32, device=g.device)
        L.check(lib.link_conv_pairs_wgrad(f.data_ptr(), g.data_ptr(), plan.pair_in.data_ptr(), plan.pair_out.data_ptr(),
                                          plan.wg_k.data_ptr(), plan.gran_start.data_ptr(), plan.rows_pad, kvol, n_dir, cin, cout,
                                          part.data_ptr(), gw.data_ptr(), _st()), "link_conv_pairs_wgrad")
        return gw
    if cin == cout and cin <= 64 and cin % 4 == 0 and (WGRAD_TABLE_SQUARE or len(kernel_shape) != 3):
        lib = L.lib()
        nbr_t = getattr(nbr, "_link_t", None)          # transposed table cached on the (kmaps-cached) tensor
        if nbr_t is None:
            nbr_t = nbr.t().contiguous()
            nbr._link_t = nbr_t
        # the split: enough pieces to fill the chip on big frames, and the DENSE centre column of a submanifold table (every
        # voxel pairs with itself; the other offsets hold a fifth of that on LiDAR frames) cut four times finer -- its
        # workgroups set the kernel's time otherwise (284 -> ~80 us on the 113k-voxel stem of the cfg3 encoder)
        chunks = 16 if n_out < 16000 else (32 if n_out < 60000 else 64)
        dense_centre = bool(getattr(nbr, "_link_subm", False)) and kvol % 2 == 1 and kvol > 1 and n_out >= 4096
        extra = 3 * chunks if dense_centre else 0
        part = torch.empty((chunks * kvol + extra, cin, cout), dtype=torch.float32, device=g.device)
        f = feats.detach().contiguous().float()
        L.check(lib.link_subm_conv_wgrad_split(f.data_ptr(), g.data_ptr(), nbr_t.data_ptr(), n_out, cin, kvol, chunks, extra,
                                               part.data_ptr(), _st()), "link_subm_conv_wgrad_split")
        gw = torch.empty((kvol, cin, cout), dtype=torch.float32, device=g.device)
        L.check(lib.link_subm_conv_wgrad_reduce(part.data_ptr(), cin, kvol, chunks, extra, gw.data_ptr(), _st()), "link_subm_conv_wgrad_reduce")
        return gw
    padded = torch.cat([feats.detach().float(), feats.new_zeros(1, cin)], dim=0)
    idx = torch.where(nbr < 0, torch.full_like(nbr, feats.shape[0]), nbr).long()
    return torch.stack([_weight_grad(padded[idx[:, k]], g) for k in range(kvol)], 0)


class _GatherConv(torch.autograd.Function):
    """out[j] = sum_k feats[table[j,k]] @ kernel[k] for any per-output table (strided / transposed
    convolutions).  Input gradient: the same kernel on grad_out with kernel[k]^T and the opposite
    direction's table; weight gradient as for the submanifold form."""

    @staticmethod
    def forward(ctx, feats, kernel, table, table_back):
        ctx.save_for_backward(feats, kernel, table, table_back)
        return subm_conv(feats, kernel, table, None)

    @staticmethod
    def backward(ctx, g):
        feats, kernel, table, table_back = ctx.saved_tensors
        g = g.contiguous().float()
        g_feats = g_kernel = None
        if ctx.needs_input_grad[0]:
            g_feats = subm_conv(g, kernel.detach().transpose(1, 2).contiguous(), table_back, None)
        if ctx.needs_input_grad[1]:
            g_kernel = _conv_weight_grad(feats, g, table, kernel.shape)
        return g_feats, g_kernel, None, None


def _flipped_weights(kernel: torch.Tensor) -> torch.Tensor:
    """w'[k] = w[K-1-k]^T of _SubmConv's input gradient, cached ON the parameter object until it changes (`_version` moves with every
    in-place optimiser update): the flip + the transposing copy were two torch launches per convolution and backward call (450 + 450
    per 15 cfg3 training steps).  (Keyed by storage address it went stale when a freed model's storage was handed to the next one.)"""
    hit = kernel.__dict__.get("_link_flip")
    if hit is not None and hit[0] == kernel._version and hit[1].shape[0] == kernel.shape[0]:
        return hit[1]
    w = kernel.detach().flip(0).transpose(1, 2).contiguous()
    kernel.__dict__["_link_flip"] = (kernel._version, w)
    return w


class _SubmConv(torch.autograd.Function):
    """Differentiable stride-1 submanifold convolution on the HIP kernel.  Input gradient: the same
    kernel on grad_out with w'[k] = w[K-1-k]^T (odd kernel, same coordinates: nbr[v,k] = u  <=>
    nbr[u,K-1-k] = v).  Weight gradient: per offset gather + one batched library GEMM."""

    @staticmethod
    def forward(ctx, feats, kernel, nbr, order=None):
        ctx.save_for_backward(feats, kernel, nbr)
        ctx.order = order
        return subm_conv(feats, kernel, nbr, order)

    @staticmethod
    def backward(ctx, g):
        feats, kernel, nbr = ctx.saved_tensors
        g = g.contiguous().float()
        g_feats = g_kernel = None
        if ctx.needs_input_grad[0]:
            g_feats = subm_conv(g, _flipped_weights(kernel), nbr, ctx.order)
        if ctx.needs_input_grad[1]:
            g_kernel = _conv_weight_grad(feats, g, nbr, kernel.shape)
        return g_feats, g_kernel, None, None


# ------------------------------------------------------------------------------------------------
# modules
# ------------------------------------------------------------------------------------------------
BLOCK_DRIVER = True           # a coordinate set nothing is known about: the whole block as ONE host call (link_elk_block_forward)
_BLOCK_CTX: dict = {}         # (device index, host thread) -> link_block_ctx_t* (side stream, events, pinned scratch)
_BLOCK_CONTRIB: dict = {}     # (device, stream) -> contribution rows of the block driver's pair GEMM (scratch, grow-only)
BLOCK_DRIVER_CALLS = {"done": 0, "miss": 0}   # diagnostics (tests, tools): how the driver's calls ended


def _block_ctx(device):
    import threading
    key = (device.index if device.index is not None else torch.cuda.current_device(), threading.get_ident())
    ctx = _BLOCK_CTX.get(key)
    if ctx is None:
        h = ctypes.c_void_p()
        L.check(L.lib().link_block_ctx_create(ctypes.byref(h)), "link_block_ctx_create")
        ctx = _BLOCK_CTX[key] = h
    return ctx


def release_block_driver() -> None:
    """Destroy the block driver's contexts (side stream, events, scratch) and free its contribution-row scratch -- up to 1 GiB per
    (device, stream) that has run link_elk_block_forward ("persistent footprint")."""
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    for h in _BLOCK_CTX.values():
        L.lib().link_block_ctx_destroy(h)
    _BLOCK_CTX.clear()
    _BLOCK_CONTRIB.clear()


DENSE_MAX_MEAN, DENSE_MAX_CELL = 6.0, 24     # voxels per occupied block: mean and maximum the dense-cell kernels take
DENSE_PLAN_CACHE_BYTES = 4 << 30             # arenas the PROCESS keeps in its modules' plan caches, dense-cell and lean together
                                             # (ElkCorePlan.arena_bytes; round 5: one budget instead of one per cache and module)
LEAN_BOUNDS_PAD_BLOCKS = 4                   # lean plans are keyed by bounds padded to this many blocks: their tables are addressed by
                                             # cell and touched only where voxels land, so the padding costs address space, not time --
                                             # and the bounding box of a LiDAR stream moves by less than that from frame to frame


class _PlanCache(dict):
    """A module's cache of ElkCorePlan arenas (insertion order = age).  All caches of the process share ONE byte budget."""
    __slots__ = ("__weakref__",)


_PLAN_CACHES: list = []                      # weak references to every live _PlanCache


def _plan_cache_of(module, name: str) -> "_PlanCache":
    cache = module.__dict__.get(name)
    if cache is None:
        cache = module.__dict__[name] = _PlanCache()
        _PLAN_CACHES.append(weakref.ref(cache))
    return cache


def _plan_cache_admit(cache: "_PlanCache", key, plan) -> None:
    """Insert `plan` (or the None marker "no such plan for this key") and evict -- oldest first, this cache before the others --
    until the arenas of all caches fit DENSE_PLAN_CACHE_BYTES and this cache holds at most 8 entries."""
    live = [c for c in (r() for r in _PLAN_CACHES) if c is not None]
    _PLAN_CACHES[:] = [weakref.ref(c) for c in live]
    while len(cache) >= 8:
        cache.pop(next(iter(cache)))
    cache[key] = plan

    def total():
        return sum(p.arena_bytes() for c in live for p in c.values() if p is not None)
    for c in [cache] + [c for c in live if c is not cache]:
        while total() > DENSE_PLAN_CACHE_BYTES:
            victim = next((k for k in c if not (c is cache and k == key)), None)
            if victim is None:
                break
            c.pop(victim)


class _ELKBase(nn.Module):
    def _core_dense(self, st: SparseTensor, s_eff: int, r: int, w_pos, alpha, cg, coord_div):
        """Inference R_core on the dense-cell layout (ElkCorePlan), or None when the frame's block grid is
        too sparse / the width unsupported: then the general layout below runs.  Plans (arena + slot lists)
        are cached per module and grid; the slot index of a coordinate tensor is reused while that tensor
        lives (the reference recomputes everything per call, utils.py:45-51)."""
        feats, coords = st.F, st.C
        n, c = feats.shape
        if n == 0 or c not in (16, 32, 64, 128) or r not in (2, 3) or not feats.is_cuda:
            return None
        half = feats.dtype != torch.float32
        if half and (c > 64 or s_eff ** 3 > 352):            # half rows: fused kernels only
            return None
        # The dense-cell kernels are built for frames whose occupied cells hold a handful of voxels each (cfg1 / cfg2:
        # ~2); a cell is one wave's serial work there.  LiDAR-like frames (20+ voxels per occupied block, 100+
        # in the densest) belong to the general layout, whose kernels split big blocks over lanes.  The bounds
        # alone cannot tell the two apart, so the occupancy is measured once per coordinate set (cached in cmaps;
        # one small sync, like coords_bounds).  Caller-supplied bounds (TSELKBlock + spatial_shape) promise a
        # sync-free call, so those frames use the general layout unless the module opts in (dense_layout = True).
        if getattr(self, "dense_layout", None) is False:     # TSELKBlock: LiDAR detection frames, clumpy by nature
            return None
        okey = ("link_dense_ok", coords.data_ptr(), n, s_eff)
        ok = st.cmaps.get(okey)
        unchecked = st.cmaps.get(("link_bounds_unchecked", coords.data_ptr(), n))
        if ok is None and unchecked and not getattr(self, "dense_layout", False):
            ok = st.cmaps[okey] = False
        if ok is False:
            return None
        bkey = ("link_bounds", coords.data_ptr(), n)
        bounds = st.cmaps.get(bkey)
        n_cap = 1 << max(10, (n - 1).bit_length())
        sig = (feats.device, n_cap, c, self.baseop, cg, r, s_eff, float(coord_div))
        spec = None
        if bounds is None:
            last = self.__dict__.get("_dc_last")
            if ok is None and last is not None and last[0] == sig and last[1]._probe_fused():
                # a new coordinate set on a module that has just run the dense layout: bounding box and occupancy in ONE
                # round trip -- the frame is inserted into the last plan's grid on the guess that its (block-aligned)
                # bounds are that plan's again, which is what frames of a stream do; a wrong guess costs a zero-fill
                init = _BBOX_STATS_INIT.get(feats.device)
                if init is None:
                    init = _BBOX_STATS_INIT[feats.device] = torch.tensor([2 ** 31 - 1] * 4 + [-2 ** 31] * 4 + [0] * (8 + 256),
                                                                          dtype=torch.int32, device=feats.device)
                both = init.clone()
                cc = coords.contiguous()
                L.check(L.lib().link_coords_bbox(cc.data_ptr(), n, both.data_ptr(), _st()), "link_coords_bbox")
                last[1]._probe(cc, n, both[16:])                      # 64-byte aligned: the 16 slots sit on their own lines
                vals = both.tolist()
                bounds = st.cmaps[bkey] = (tuple(vals[:4]), tuple(vals[4:8]))
                spec = (last[1], _probe_stats(vals[16:]))
            else:
                from .index import coords_bounds
                bounds = st.cmaps[bkey] = coords_bounds(coords.contiguous())
        cache = _plan_cache_of(self, "_dc_plans")
        # plans are keyed by the bounds padded out to whole blocks (the grid they imply is the same): the exact extents of
        # real frames move from frame to frame, and every new key is a new arena (zero-filled tables, a slot arena of up
        # to 1 GiB) plus an occupancy probe.  Coarser padding would merge more frames but the gather kernel streams every
        # cell of the grid (cfg2: + 26 % cells at 4-block padding)
        q = int(s_eff)
        qbounds = (tuple((int(v) // q) * q for v in bounds[0][:3]) + (int(bounds[0][3]),),
                   tuple((int(v) // q) * q + q - 1 for v in bounds[1][:3]) + (int(bounds[1][3]),))
        key = (feats.device, n_cap, c, self.baseop, cg, r, s_eff, qbounds, float(coord_div))
        plan = cache.get(key, False)
        if plan is False:
            plan = None
            if ElkCorePlan.would_be_dense(n_cap, c, self.baseop, r, s_eff, qbounds):
                try:
                    plan = ElkCorePlan(n_cap, c, self.baseop, cg, r, s_eff, qbounds, feats.device, coord_div=coord_div,
                                       layout="dense")
                except L.LinkAmdError:
                    plan = None
            # arenas are large: the caches are capped by bytes, process-wide (and by entries, for the None markers)
            _plan_cache_admit(cache, key, plan)
        if spec is not None and spec[0] is not plan:
            spec[0]._unprobe()                                 # guessed the wrong grid: that plan forgets the frame
            spec = None
        if plan is None:
            st.cmaps[okey] = False
            return None
        inserted = False
        if ok is None:
            # first visit of this coordinate set: the step's own slot insert, in a variant that also counts the voxels
            # inside the grid, the occupied cells and the fullest cell (link_dc_index_probe); one host round trip, and the
            # step below starts behind the insert (build_index = 2)
            cc = coords.contiguous()
            if spec is not None:
                n_in, m, mx = spec[1]
                inserted = True
            elif plan._probe_fused():
                stats = torch.zeros(256, dtype=torch.int32, device=feats.device)
                plan._probe(cc, n, stats)
                n_in, m, mx = _probe_stats(stats.tolist())
                inserted = True
            else:                                              # C = 128 / long slot lists: the insert alone, then the counters
                L.check(L.lib().link_dc_index_ids(cc.data_ptr(), n, ctypes.byref(plan.dcg), plan.cnt.data_ptr(), plan.sid.data_ptr(),
                                                  plan.vcell.data_ptr(), plan.hdr.data_ptr(), _st()), "link_dc_index_ids")
                mx, m, n_in = torch.stack([plan.cnt.max(), (plan.cnt > 0).sum(), plan.cnt.sum()]).tolist()
                plan._unprobe()                                # the step below inserts again
            ok = st.cmaps[okey] = bool(m > 0 and n_in <= DENSE_MAX_MEAN * m and mx <= min(DENSE_MAX_CELL, int(plan.dcg.k)))
            if not ok:
                plan._unprobe()                                # nothing of this frame stays behind in the shared plan
                return None
            self.__dict__["_dc_last"] = (sig, plan)
        plan.bind(self.pre_mix[0].weight, self.pre_mix[1].weight, self.pre_mix[1].bias, w_pos, alpha,
                  self.norm.weight, self.norm.bias)
        ikey = (coords.data_ptr(), n, coords._version)
        # bounds taken from the caller's metadata (TSELKBlock: spatial_shape) are not verified without a sync:
        # rows of voxels outside them are never written, so they start as zeros (status word)
        alloc = torch.zeros if st.cmaps.get(("link_bounds_unchecked", coords.data_ptr(), n)) else torch.empty
        out = alloc((n, c), dtype=feats.dtype, device=feats.device)
        plan.run(feats.contiguous(), coords.contiguous(), build_index=2 if inserted else plan.__dict__.get("_indexed") != ikey, out=out)
        plan._indexed = ikey
        plan._keepalive = coords                             # the pointer in ikey stays valid while we hold it
        return out

    def _core_lean(self, st: SparseTensor, s_eff: int, r: int, w_pos, alpha, cg, coord_div):
        """Inference R_core through the lean form (ElkCorePlan(layout='lean'): three launches with the index rebuilt, nothing
        cached per coordinate set but the arena) for coordinate sets that have no block index yet -- every call, in a stream
        of LiDAR frames.  None when the form does not apply (then the general layout builds its index)."""
        feats, coords = st.F, st.C
        n, c = feats.shape
        if n == 0 or c not in (16, 32, 64, 128) or r not in (2, 3) or not feats.is_cuda or not LEAN_FORM:
            return None
        self._lean_poll_verdicts()                            # an EARLIER frame that overflowed a slot list is reported here (raises)
        if st.kmaps.get(("link_block_index", coords.data_ptr(), n, int(s_eff))) is not None:
            return None                                       # an index of these coordinates exists: the two tile launches
        # A coordinate set that COMES BACK (its maps were kept: a loop over one frame, the stages of a benchmark on warm maps)
        # is worth a block index: on a built index the tile form is the faster one on 7 of the 8 LiDAR stage frames (26-46 us
        # against the lean form's 32-66 with its lists reused; profiles/r05_*_lidar_stages.jsonl).  So the FIRST visit runs the
        # lean form (nothing to build: a stream of new frames never leaves it), the second visit builds the index and every
        # later one finds it above.  (Round 4 kept such sets on the lean form for bit-stability across visits; the forms agree
        # to ~1e-7 of max|out|, and a caller who needs the bits of the first visit can keep LEAN_SECOND_VISIT_INDEX = False.)
        vkey = ("link_lean_visits", coords.data_ptr(), n, int(s_eff), coords._version)
        if LEAN_SECOND_VISIT_INDEX and st.kmaps.get(vkey):
            return None
        ts = st.s[0] if isinstance(st.s, (tuple, list)) else st.s
        ts = max(int(ts), 1)
        # coordinates of a tensor at stride ts are multiples of ts (torchsparse/nn/functional/downsample.py:27-40), so a block
        # of edge s_eff holds at most (s_eff / ts)^3 of them.  A frame that breaks the promise (coordinates off the stride's
        # lattice, duplicate coordinates) overflows a slot list: the kernel drops the surplus voxels and raises status bit 1.
        # The status word of every step travels to pinned memory behind the kernels and is looked at on the next call (no
        # sync on this path): the broken promise raises LinkAmdError there, this (s_eff, stride) pair goes to the general
        # layout from then on, and the rows of dropped voxels are zeros, never uninitialised memory (`out` below).
        if (int(s_eff), ts) in self.__dict__.get("_lean_distrust", ()):
            return None
        k = (max(int(s_eff) // ts, 1)) ** 3 if int(s_eff) % ts == 0 else int(s_eff) ** 3
        n_cap = 1 << max(10, (n - 1).bit_length())
        if k > L.LEAN_KMAX or n * (3 if self.baseop == "cos_x" else 2) * c > ElkCorePlan.LEAN_AUTO_FLOATS:
            return None
        bkey = ("link_bounds", coords.data_ptr(), n)
        bounds = st.cmaps.get(bkey)
        if bounds is None:
            from .index import coords_bounds
            bounds = st.cmaps[bkey] = coords_bounds(coords.contiguous())
        # bounds measured on the frame move from frame to frame: pad them coarsely; bounds the caller declared (spatial_shape: fixed for
        # the network, and what lies beyond them must be dropped -- INTEGRATION.md) are taken block-exact
        unchecked = bool(st.cmaps.get(("link_bounds_unchecked", coords.data_ptr(), n)))
        q = int(s_eff) * (1 if unchecked else LEAN_BOUNDS_PAD_BLOCKS)
        qbounds = (tuple((int(v) // q) * q for v in bounds[0][:3]) + (int(bounds[0][3]),),
                   tuple((int(v) // q) * q + q - 1 for v in bounds[1][:3]) + (int(bounds[1][3]),))
        cache = _plan_cache_of(self, "_lean_plans")
        key = (feats.device, n_cap, c, self.baseop, cg, r, s_eff, qbounds, float(coord_div), k)
        plan = cache.get(key, False)
        if plan is False:
            plan = None
            if ElkCorePlan.lean_supported(n_cap, c, self.baseop, r, s_eff, qbounds, k):
                try:
                    plan = ElkCorePlan(n_cap, c, self.baseop, cg, r, s_eff, qbounds, feats.device, coord_div=coord_div,
                                       layout="lean", slot_cap=k)
                except L.LinkAmdError:
                    plan = None
            _plan_cache_admit(cache, key, plan)
        if plan is None:
            return None
        plan.bind(self.pre_mix[0].weight, self.pre_mix[1].weight, self.pre_mix[1].bias, w_pos, alpha,
                  self.norm.weight, self.norm.bias)
        # a coordinate set that comes back (its maps are kept: the stage of a loop over one frame) stays on this form, so that
        # a frame's rows do not depend on how often it was seen (rebuilt and reused lists give the same bits); the lists are
        # reused while the plan has seen nothing else in between
        ikey = (coords.data_ptr(), n, coords._version)
        out = torch.zeros((n, c), dtype=feats.dtype, device=feats.device)     # a dropped voxel's row is zero, not whatever was there
        rebuild = plan.__dict__.get("_indexed") != ikey
        plan.run(feats.contiguous(), coords.contiguous(), build_index=rebuild, out=out)
        plan._indexed = ikey
        plan._keepalive = coords                             # the pointer in ikey stays valid while we hold it
        if rebuild and k < int(s_eff) ** 3:
            # only a slot list SMALLER than what unique coordinates could fill rests on the stride promise; at k = s_eff^3
            # (tensor stride 1: the detection blocks) it cannot overflow without duplicate coordinates, and the ~20 us of host
            # time per block (pinned copy + event) stay off the sync-free detection path
            self._lean_post_verdict(plan, int(s_eff), ts, k)
        st.kmaps[vkey] = True
        return out

    def _lean_post_verdict(self, plan, s_eff: int, ts: int, k: int) -> None:
        """The status word of the step just issued goes to pinned memory behind its kernels (no sync); _lean_poll_verdicts looks."""
        host, slot, gen = _pinned_slot()
        host.copy_(plan.hdr[:8], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        pend = self.__dict__.setdefault("_lean_pending", [])
        pend.append((ev, host, slot, gen, s_eff, ts, k))
        if len(pend) > 8:                                    # bounded -- but no verdict is dropped unseen: the oldest is
            self._lean_poll_verdicts(wait_oldest=True)       # waited for (its kernels ran eight steps ago) and looked at
        if L.DEBUG:
            self.flush_lean_verdicts()

    def flush_lean_verdicts(self) -> None:
        """Wait for the status words of every lean-form step issued so far and raise if one of them dropped voxels (slot-list
        overflow).  The sync-free path reports such a frame at a LATER call on the same module; a caller that must know before it
        uses the rows -- the last frame of a run, a validation loop -- calls this (LINK_AMD_DEBUG=1 does after every step)."""
        pend = self.__dict__.get("_lean_pending")
        if pend:
            pend[-1][0].synchronize()
            self._lean_poll_verdicts()

    def _lean_poll_verdicts(self, wait_oldest: bool = False) -> None:
        pend = self.__dict__.get("_lean_pending")
        if not pend:
            return
        if wait_oldest:
            pend[0][0].synchronize()
        bad = None
        while pend and pend[0][0].query():
            ev, host, slot, gen, s_eff, ts, k = pend.pop(0)
            if _pinned_current(slot, gen) and int(host[L.HDR_STATUS]) & 2:
                bad = (s_eff, ts, k)
                self.__dict__.setdefault("_lean_distrust", set()).add((s_eff, ts))
        if bad is not None:
            raise L.LinkAmdError(
                f"{type(self).__name__}: an earlier frame held more than {bad[2]} voxels in a block of edge {bad[0]} at tensor stride "
                f"{bad[1]} -- coordinates that are not multiples of the tensor stride, or duplicate coordinates (torchsparse's own "
                "contract: tensor.py / downsample.py).  The surplus voxels of that frame were left out of the block sums and their "
                "output rows are zero; blocks of this edge and stride take the general layout from here on.")

    def _block_native(self, st: SparseTensor, conv, s_eff: int, r: int, w_pos, alpha, cg, coord_div) -> bool:
        """The whole block on a coordinate set nothing is known about yet, as ONE host call (include/link_amd.h section G,
        csrc/block.hip): the frame is tried on the module's last dense-cell plan -- bounding box + slot insert with occupancy
        counters, one round trip, then R_core on the caller's stream while the neighbour table, the pair plan and the pair GEMM
        of local_mix run on the driver's side stream, and the convolution's finish adds the two.  Returns False (having touched
        nothing but st.cmaps' bounds) when the call does not apply or the frame missed the plan: the per-stage path takes over.
        What the call built is registered where the per-stage path would have put it (cmaps / kmaps / the table's pair plan),
        so later blocks on the same coordinates find warm maps."""
        feats, coords = st.F, st.C
        n, c = feats.shape
        if (n == 0 or not feats.is_cuda or coords.dtype != torch.int32 or L.DEBUG or getattr(self, "dense_layout", None) is False
                or not (ASYNC_PAIR_PLANS and TRUST_UNIQUE_INPUT_COORDS) or conv.bias is not None):
            return False
        bkey, okey = ("link_bounds", coords.data_ptr(), n), ("link_dense_ok", coords.data_ptr(), n, s_eff)
        kkey = ("link_conv_nbr", coords.data_ptr(), n, st.s, tuple(conv.kernel_size))
        if bkey in st.cmaps or okey in st.cmaps or kkey in st.kmaps:
            return False                                   # something is known already: the per-stage path reuses it
        n_cap = 1 << max(10, (n - 1).bit_length())
        sig, plan = self._dc_last
        if sig != (feats.device, n_cap, c, self.baseop, cg, r, s_eff, float(coord_div)) or not plan._probe_fused():
            return False
        lib = L.lib()
        if (not lib.link_conv_pairs_supported(c, c) or _resident_form(c, c, 27, False, "auto") or n * 27 > ASYNC_PAIR_PLAN_MAX
                or n * 27 * c * 4 > ASYNC_PAIR_PLAN_MAX_BYTES or _DENSITY_SEEN.get(27, 0.0) > PAIR_DENSITY_MAX):
            return False
        kernel = conv.kernel
        if kernel.dtype != torch.float32 or not kernel.is_contiguous() or tuple(kernel.shape) != (27, c, c):
            return False
        dev = feats.device
        ctx = _block_ctx(dev)
        feats_c, cc = feats.contiguous(), coords.contiguous()
        plan.bind(self.pre_mix[0].weight, self.pre_mix[1].weight, self.pre_mix[1].bias, w_pos, alpha,
                  self.norm.weight, self.norm.bias)
        offs = (ctypes.c_int64 * 11)()
        gran_cap = int(lib.link_pair_plan_arena(n, 27, 1, offs))
        rows = gran_cap * 128
        ck = (dev, _st())
        contrib = _BLOCK_CONTRIB.get(ck)
        if contrib is None or contrib.numel() < rows * c:
            contrib = _BLOCK_CONTRIB[ck] = torch.empty(rows * c, dtype=torch.float32, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        core_out = torch.empty((n, c), dtype=torch.float32, device=dev)
        out = torch.empty((n, c), dtype=torch.float32, device=dev)
        nbr = torch.empty((n, 27), **i32)
        arena = torch.empty(int(offs[10]), **i32)
        nl = self.norm_local
        a = plan.__dict__.get("_blk_args")
        if a is None:
            a = plan._blk_args = L.LinkBlockArgs()
            a.buf, a.g, a.desc = ctypes.pointer(plan.buf), ctypes.pointer(plan.dcg), ctypes.pointer(plan.desc)
            a.mean_max, a.cell_max, a.subm, a.flags = int(DENSE_MAX_MEAN), int(DENSE_MAX_CELL), 1, 1
        b = plan.buf
        b.feats, b.coords, b.out, b.io_dtype = feats_c.data_ptr(), cc.data_ptr(), core_out.data_ptr(), L.IO_F32
        a.n, a.ts = n, int(st.s[0])
        a.nbr, a.pair_arena, a.pair_arena_words = nbr.data_ptr(), arena.data_ptr(), arena.numel()
        a.contrib, a.contrib_rows = contrib.data_ptr(), contrib.numel() // c
        a.w = kernel.data_ptr()
        if SPLIT_MFMA and c >= 32:
            ws, big = _split_weights(kernel)
            a.ws, a.w_big = ws.data_ptr(), big.data_ptr()
        else:
            a.ws = a.w_big = None
        nlw, nlb = nl.weight, nl.bias
        a.nl_w, a.nl_b, a.nl_eps, a.out = nlw.data_ptr(), nlb.data_ptr(), float(nl.eps), out.data_ptr()
        plan._indexed = None                               # the probe overwrites the plan's slot lists
        _poll_pending_plans()
        rc = lib.link_elk_block_forward(ctx, ctypes.byref(a), _st())
        if rc < 0:
            plan._unprobe()                                # (the driver takes the frame out of the plan on its error paths; belt and braces)
            L.check(rc, "link_elk_block_forward")
        bb = list(a.bbox)
        st.cmaps[bkey] = (tuple(bb[:4]), tuple(bb[4:]))
        if rc != L.BLOCK_DONE:
            BLOCK_DRIVER_CALLS["miss"] += 1
            if int(a.stats[3]) == 2:                       # inside the grid, occupancy too high: mean / maximum voxels per occupied cell do
                st.cmaps[okey] = False                     # not depend on the grid's extent -- the verdict stands, no second probe
            return False
        BLOCK_DRIVER_CALLS["done"] += 1
        st.cmaps[okey] = True
        plan._indexed = (coords.data_ptr(), n, coords._version)
        plan._keepalive = coords
        nbr._link_subm = True
        nbr._link_pairs = _PairPlan._from_arena(nbr, arena, offs, gran_cap)
        st.kmaps[kkey] = (nbr, _TileOrder(coords, 4 * int(st.s[0])))
        st.F = out
        return True

    def _core_generic(self, st: SparseTensor, s_eff: int, r: int, w_pos, alpha, cg, coord_div):
        """R_core as the reference writes it (linkunet.py:124-176), op by op on voxel_to_aux / aux_to_voxel:
        any grid extent, any width; differentiable through the ops' autograd Functions."""
        from .aggregate import aux_to_voxel, voxel_to_aux
        c = st.F.shape[1]
        fin = self.pre_mix(st.F.float())
        xyz = st.C[:, :3].float()
        if coord_div != 1.0:
            xyz = xyz / coord_div
        th = TF.linear(xyz, w_pos)
        if alpha is not None:
            th = th * alpha
        if c != cg:
            th = th.repeat([1, c // cg])
        sin, cos = torch.sin(th), torch.cos(th)
        if self.baseop == "sin":
            x = torch.cat([fin * sin, fin * cos], dim=1)
        elif self.baseop == "cos":
            x = torch.cat([fin * cos, fin * sin], dim=1)
        else:
            x = torch.cat([fin * cos, fin * sin, fin * th], dim=1)
        big = SparseTensor(x, st.C, st.s)
        big.cmaps, big.kmaps = st.cmaps, st.kmaps
        small, idx, counts = voxel_to_aux(big, s_eff)
        v = aux_to_voxel(small, big, idx, counts, r).F
        if self.baseop == "sin":
            new = v[:, :c] * cos - v[:, c:] * sin
        elif self.baseop == "cos":
            new = v[:, :c] * cos + v[:, c:] * sin
        else:
            new = v[:, :c] * cos + v[:, c:2 * c] * sin + (v[:, 2 * c:] - fin * th)
        return self.norm(new)

    def _core(self, st: SparseTensor, s_eff: int, r: int, w_pos, alpha, cg, coord_div):
        needs_grad0 = torch.is_grad_enabled() and (st.F.requires_grad or any(
            p.requires_grad for p in self.parameters()))
        if not needs_grad0 and st.F.dtype in _IO_DTYPES and st.C.dtype == torch.int32:
            out = self._core_dense(st, s_eff, r, w_pos, alpha, cg, coord_div)
            if out is None:
                out = self._core_lean(st, s_eff, r, w_pos, alpha, cg, coord_div)
            if out is not None:
                return out
        try:
            index = link_index_of(st, s_eff)
        except GridTooLarge:
            # coordinates spread over more cells than the dense block grid may hold: the reference algorithm on
            # the op kernels (hash / unique / query), as voxel_to_aux does on its own for such inputs
            return self._core_generic(st, s_eff, r, w_pos, alpha, cg, coord_div)
        args = (st.F, st.C, index, self.pre_mix[0].weight, self.pre_mix[1].weight, self.pre_mix[1].bias,
                w_pos, alpha, self.norm.weight, self.norm.bias, self.baseop, cg, r, coord_div, 1e-6)
        needs_grad = torch.is_grad_enabled() and (st.F.requires_grad or any(
            p.requires_grad for p in self.parameters()))
        if L.DEBUG:
            index.M                              # validates hdr[STATUS] (one D2H sync)
        if needs_grad:
            if st.F.shape[1] % 4 == 0 and r <= 3 and st.F.dtype == torch.float32:
                return elk_core_train(*args)
            return elk_core_autograd(*args)      # op-by-op composition: any width / r
        return elk_core_fused(*args)

    def _finish(self, st: SparseTensor, core_fn, core_args=None):
        """st.F = relu(core + norm_local(local_mix(st).F))  (linkunet.py:125,183 / ts_elk.py:146,228).
        Inference: the LayerNorm + add + ReLU run in the convolution kernel's store phase (row N2); with
        grad enabled, or when forward hooks observe local_mix / norm_local, the modules run one by one.
        A coordinate set nothing is known about yet takes the one-call block driver when it applies (_block_native)."""
        conv = self.local_mix[0]
        needs_grad = torch.is_grad_enabled() and (st.F.requires_grad or any(
            p.requires_grad for p in self.parameters()))
        hooked = any(m._forward_hooks or m._forward_pre_hooks
                     for m in (self.local_mix, conv, self.norm_local, self.activate))
        if (BLOCK_DRIVER and core_args is not None and not needs_grad and not hooked and conv.kernel_volume == 27
                and st.F.dtype == torch.float32 and self.__dict__.get("_dc_last") is not None
                and self._block_native(st, conv, *core_args)):
            return st
        if needs_grad or hooked or conv.kernel_volume == 1 or st.F.dtype != torch.float32:
            local = self.local_mix(st)
            new = core_fn()
            nl = self.norm_local
            if (needs_grad and not hooked and st.F.dtype == torch.float32 and new.shape[1] % 4 == 0
                    and new.shape[1] <= 256 and new.is_cuda):
                st.F = _Tail.apply(local.F, new, nl.weight, nl.bias, nl.eps)     # fused tail, fused backward
            else:
                st.F = self.activate(new + nl(local.F))
            return st
        new = core_fn()
        nbr, order = conv._neighbor_table(st)
        st.F = subm_conv_ln_add_relu(st.F, conv.kernel, nbr, order, self.norm_local.weight, self.norm_local.bias,
                                     self.norm_local.eps, new, relu=True)
        return st


class ELKBlock(_ELKBase):
    def __init__(self, inc, outc, groups=1, baseop="cos_x", variant="unet"):
        super().__init__()
        self.inc, self.outc, self.groups, self.baseop, self.variant = inc, outc, groups, baseop, variant
        assert inc % groups == 0
        assert baseop in ("cos", "sin", "cos_x")
        assert variant in ("unet", "encoder")
        if baseop == "cos_x":
            self.alpha = nn.Parameter(torch.ones(1, inc // groups).float(), requires_grad=True)
        self.pos_weight = nn.Sequential(nn.Linear(3, inc // groups, bias=False))
        self.pre_mix = nn.Sequential(nn.Linear(inc, inc, bias=False), nn.LayerNorm(inc, eps=1e-6))
        self.local_mix = nn.Sequential(Conv3d(inc, inc, kernel_size=3, dilation=1, stride=1))
        self.norm_local = nn.LayerNorm(inc, eps=1e-6)
        self.norm = nn.LayerNorm(inc, eps=1e-6)
        self.activate = nn.ReLU(True)

    def forward(self, st: SparseTensor, s, r):
        """st: SparseTensor; s: block edge (in st's coordinate units); r: neighbourhood edge in blocks."""
        if self.baseop == "cos_x" and self.groups != 1:
            # linkunet.py:165 does not tile theta for cos_x: only groups == 1 is shape-consistent
            raise ValueError("baseop='cos_x' requires groups == 1 (as in the reference configs)")
        alpha = self.alpha if self.baseop == "cos_x" else None
        coord_div = float(st.s[0]) if (self.variant == "encoder" and self.baseop == "cos_x") else 1.0
        cg = self.inc // self.groups
        return self._finish(st, lambda: self._core(st, int(s), int(r), self.pos_weight[0].weight, alpha, cg,
                                                   coord_div), (int(s), int(r), self.pos_weight[0].weight, alpha, cg, coord_div))


class SparseConvTensor:
    """Minimal stand-in for spconv.SparseConvTensor (the wheel is an external dependency of the
    reference, detection/INSTALL.md:56): features [N,C], indices int32 [N,4] = (batch, z, y, x)."""

    def __init__(self, features, indices, spatial_shape, batch_size, grid=None, voxel_num=None,
                 indice_dict=None, benchmark=False):
        self.features, self.indices = features, indices
        self.spatial_shape, self.batch_size = spatial_shape, batch_size
        self.grid, self.voxel_num = grid, voxel_num
        self.indice_dict = {} if indice_dict is None else indice_dict
        self.benchmark, self.benchmark_record = benchmark, {}


def spconv2ts(sct):
    """ts_elk.py:10-33: (batch,z,y,x) indices -> (x,y,z,batch) coords, stride 1; keep spconv metadata."""
    # the converted coordinates and the maps built on them are kept in the tensor's indice_dict (the dict spconv
    # itself shares between the tensors of one coordinate set), so the blocks and convolutions of a stage reuse
    # one neighbour table / block index instead of rebuilding them per call; the entry holds `indices` alive,
    # which keeps its address -- part of the key -- from being reused
    ind = sct.indices
    ent = None
    if isinstance(getattr(sct, "indice_dict", None), dict):
        key = ("link_ts", ind.data_ptr(), ind.shape[0], ind._version)
        ent = sct.indice_dict.get(key)
        if ent is None:
            ent = sct.indice_dict[key] = (ind[:, [3, 2, 1, 0]].contiguous(), ind, {}, {})
            if sct.indice_dict.get(("link_unique", ind.data_ptr(), ind.shape[0])):       # sites a strided convolution created
                mark_unique(ent[2], ent[0])
    coords = ent[0] if ent is not None else ind[:, [3, 2, 1, 0]].contiguous()
    st = SparseTensor(sct.features, coords, 1)
    if ent is not None:
        st.cmaps, st.kmaps = ent[2], ent[3]
    save = {k: getattr(sct, k, None) for k in ("batch_size", "benchmark", "benchmark_record", "grid",
                                               "indice_dict", "spatial_shape", "voxel_num")}
    save["_cls"] = type(sct)
    save["_ind"] = (coords, ind)
    return st, save


def ts2spconv(st: SparseTensor, sct_save: dict):
    """ts_elk.py:36-59."""
    src = sct_save.get("_ind")
    if src is not None and st.coords is src[0]:
        indices = src[1]                               # coordinates untouched: hand the caller's indices back
    else:
        indices = st.coords[:, [3, 2, 1, 0]].contiguous()
    cls = sct_save.get("_cls", SparseConvTensor)
    sct = cls(st.feats, indices, spatial_shape=sct_save["spatial_shape"], batch_size=sct_save["batch_size"],
              grid=sct_save["grid"], voxel_num=sct_save["voxel_num"], indice_dict=sct_save["indice_dict"],
              benchmark=sct_save["benchmark"])
    sct.benchmark_record = sct_save["benchmark_record"]
    return sct


class TSELKBlock(_ELKBase):
    """ts_elk.py:110-230 ('cos' and 'sin' base ops; the other branches of the reference are dead or
    reference an undefined self.alpha)."""

    dense_layout = False          # opt in per instance (True) for frames with small blocks; None = measure per frame

    def __init__(self, inc, outc, baseop="cos"):
        super().__init__()
        assert baseop in ("cos", "sin")
        self.inc, self.outc, self.baseop = inc, outc, baseop
        self.pre_mix = nn.Sequential(nn.Linear(inc, inc, bias=False), nn.LayerNorm(inc, eps=1e-6))
        self.local_mix = nn.Sequential(Conv3d(inc, inc, kernel_size=3, dilation=1, stride=1))
        self.pos_weight = nn.Sequential(nn.Linear(3, inc, bias=False))
        self.norm = nn.LayerNorm(inc, eps=1e-6)
        self.norm_local = nn.LayerNorm(inc, eps=1e-6)
        self.activate = nn.ReLU(True)

    def forward(self, sct, stride):
        st, save = spconv2ts(sct)
        bounds = None
        shape = save.get("spatial_shape")
        if shape is not None and save.get("batch_size") is not None:   # bounds for free: no bbox sync
            z, y, x = [int(v) for v in shape]
            bounds = ((0, 0, 0, 0), (x - 1, y - 1, z - 1, int(save["batch_size"]) - 1))
            if st.cmaps.setdefault(("link_bounds", st.C.data_ptr(), st.C.shape[0]), bounds) is bounds:
                st.cmaps[("link_bounds_unchecked", st.C.data_ptr(), st.C.shape[0])] = True
        return ts2spconv(self.forward_(st, stride), save)

    def forward_(self, st: SparseTensor, stride):
        if self.baseop == "cos":      # ts_elk.py:168: first C/2 columns, tiled twice
            w_pos, cg = self.pos_weight[0].weight[: self.inc // 2], self.inc // 2
        else:                          # 'sin' (ts_elk.py:155-156): all C columns, untiled
            w_pos, cg = self.pos_weight[0].weight, self.inc
        return self._finish(st, lambda: self._core(st, int(stride), 3, w_pos, None, cg, 1.0), (int(stride), 3, w_pos, None, cg, 1.0))
