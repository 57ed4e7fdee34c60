"""link_amd/index.py -- BlockIndex: the per-frame block structure of LinK's aggregation.

One BlockIndex replaces everything `voxel_to_aux` / `aux_to_voxel` recompute on every call in the
reference (segmentation/core/models/utils.py:45-51,65-73): block coordinates, the voxel->block map
(`idx_query`), per-block counts, and the r^3 neighbour lookup.  It is built by 4 kernel launches
without host synchronisation (link_index_build, include/link_amd.h section B); the only sync is the
optional read-back of M when a caller needs exactly-sized tensors (`.M`).

HBM layout (all int32 unless noted, N = voxels, V = cells of the dense block grid, M <= N blocks):
  cell_blk[V]  block id + 1 per cell (0 = empty)      vox_blk[N]      block of each voxel
  idx_query[N] int64 copy (reference dtype)           perm[N]         voxel ids grouped by block
  vox_sorted[N,4] (x,y,z,voxel id) in perm order: the stream the fused kernels read
  blk_start[N+1] segment starts                       blk_coords[N,4] rows [0,M) valid
  counts[N]    rows [0,M) valid                       hdr[8]          M, status, nvalid
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib as L

MAX_CELLS = 1 << 28   # dense-grid path limit (1 GiB of int32 cell table); beyond -> generic hash path


class _Workspace:
    """Per (device, stream) scratch: zero-initialised cell counters (self-cleaning) + index scratch."""

    def __init__(self, device):
        self.device = device
        self.cell_counts = None
        self.scratch = None

    def cell_table(self, v: int) -> torch.Tensor:
        """int32[>= v], all zero on entry of every user (each clears what it wrote: link_cell_table_clear).  Persistent: it grows
        to the largest grid this (device, stream) has seen, at most MAX_CELLS entries (1 GiB); `release_workspaces()` frees it."""
        if v > MAX_CELLS:
            raise GridTooLarge(f"cell table of {v} cells (> {MAX_CELLS})")
        t = self.__dict__.get("_cell_table")
        if t is None or t.numel() < v:
            t = self._cell_table = torch.zeros(max(v, 1 << 16), dtype=torch.int32, device=self.device)
        return t

    def drop_cell_table(self) -> None:
        """Forget the cell table (after a failed build / look-up / clear triple it may hold stale entries: the next user gets a
        freshly zeroed one)."""
        self.__dict__.pop("_cell_table", None)

    def ensure(self, n: int, v: int):
        if self.cell_counts is None or self.cell_counts.numel() < v:
            self.cell_counts = torch.zeros(max(v, 1 << 16), dtype=torch.int32, device=self.device)
        need = L.lib().link_index_scratch_bytes(n, v)
        if self.scratch is None or self.scratch.numel() < need:
            self.scratch = torch.empty(max(need, 1 << 16), dtype=torch.uint8, device=self.device)
        return self.cell_counts, self.scratch


_workspaces: Dict[Tuple[int, int], _Workspace] = {}


def release_workspaces(device=None) -> None:
    """Free the per-(device, stream) scratch this module keeps between calls (cell counters, index scratch and the cell table:
    up to 1 GiB each on a stream that has mapped a MAX_CELLS grid).  INTEGRATION.md "persistent footprint"."""
    for key in list(_workspaces):
        if device is None or key[0] == (device.index if getattr(device, "index", None) is not None else key[0]):
            del _workspaces[key]


def _workspace(device) -> _Workspace:
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           L.current_stream_handle())
    ws = _workspaces.get(key)
    if ws is None:
        ws = _workspaces[key] = _Workspace(device)
    return ws


_BBOX_INIT: Dict[torch.device, torch.Tensor] = {}
SPECULATE_GRID = True       # BlockIndex without bounds: built on the last grid of its block edge, box + M in one round trip
_SPEC_GRID: Dict = {}       # (device, block edge) -> bounds (padded to whole blocks) of the last index whose bounds were measured


def coords_bounds(coords: torch.Tensor) -> Tuple[Tuple[int, ...], Tuple[int, ...]]:
    """Inclusive (lo, hi) of int32 coords[N,4] via the HIP bbox kernel (one 32-byte D2H sync)."""
    init = _BBOX_INIT.get(coords.device)
    if init is None:                                   # one H2D per device; per call a device-side copy
        imax, imin = 2 ** 31 - 1, -2 ** 31
        init = _BBOX_INIT[coords.device] = torch.tensor([imax] * 4 + [imin] * 4, dtype=torch.int32, device=coords.device)
    bbox = init.clone()
    L.check(L.lib().link_coords_bbox(coords.data_ptr(), coords.shape[0], bbox.data_ptr(),
                                     L.current_stream_handle()), "link_coords_bbox")
    b = bbox.tolist()
    return tuple(b[:4]), tuple(b[4:])


class GridTooLarge(L.LinkAmdError):
    pass


class BlockIndex:
    """Block structure of `coords` for block edge `s` (built on the GPU, see module docstring)."""

    def __init__(self, coords: torch.Tensor, s: int, bounds: Optional[Tuple[Sequence[int], Sequence[int]]] = None,
                 want_idx64: bool = True):
        if coords.device.type != "cuda":
            raise L.LinkAmdError("BlockIndex needs GPU coords (HIP path; no CPU fallback)")
        if coords.dtype != torch.int32 or coords.ndim != 2 or coords.shape[1] != 4:
            raise ValueError(f"coords must be int32[N,4], got {coords.dtype} {tuple(coords.shape)}")
        if not isinstance(s, int) or s <= 0:
            raise ValueError(f"block edge s must be a positive int, got {s!r}")
        self.coords = coords.contiguous()
        self.s = s
        self.n = n = coords.shape[0]
        dev = coords.device
        # bounds handed in by the caller (spatial_shape) are trusted without a sync: a voxel outside them is
        # dropped and reported in hdr[STATUS]; consumers zero-fill their outputs in that case (rows_checked)
        self.rows_checked = bounds is None
        self._m: Optional[int] = None
        self._nbr: Dict[Tuple[int, bool], torch.Tensor] = {}
        if bounds is None and n > 0 and SPECULATE_GRID:
            # Nobody told us the bounds (voxel_to_aux through the reference's surface, utils.py:44-52): instead of measuring
            # them (one round trip), building the index, and reading M back (a second one), the index is built on the grid of
            # the LAST index built for this block edge -- frames of a stream share their extents -- while the bounding box is
            # taken in the same pass, and ONE 64-byte read answers both: the box lies inside that grid and no voxel was dropped
            # (then block ids, which are ranks among occupied cells in coordinate order, do not depend on the grid's extent) or
            # the index is built again on the measured bounds.  A grid more than twice the measured one is not kept either.
            spec = _SPEC_GRID.get((dev, s))
            if spec is not None:
                got = self._build(spec, want_idx64, with_bbox=True)
                box = (tuple(got[:4]), tuple(got[4:8]))
                ok = got[8 + L.HDR_STATUS] == 0 and all(a <= b for a, b in zip(spec[0], box[0])) and all(
                    a >= b for a, b in zip(spec[1], box[1]))
                if ok:
                    try:
                        ok = self.v <= 2 * max(L.grid_from_bounds(box[0], box[1], s).cells, 1)
                    except L.LinkAmdError:
                        ok = False
                if ok:
                    self.bounds = box                       # what other map builders read from cmaps: the measured box
                    self._m = int(got[8 + L.HDR_M])
                    return
                bounds = box
        if bounds is None:
            if n == 0:
                bounds = ((0, 0, 0, 0), (0, 0, 0, 0))
            else:
                bounds = coords_bounds(self.coords)
        bounds = (tuple(int(x) for x in bounds[0]), tuple(int(x) for x in bounds[1]))
        self._build(bounds, want_idx64, with_bbox=False)
        self.bounds = bounds
        if self.rows_checked and n > 0:
            q = int(s)                                      # remembered padded to whole blocks: equal extents, equal grid
            _SPEC_GRID[(dev, s)] = (tuple((v // q) * q for v in bounds[0][:3]) + (bounds[0][3],),
                                    tuple((v // q) * q + q - 1 for v in bounds[1][:3]) + (bounds[1][3],))

    def _build(self, bounds, want_idx64: bool, with_bbox: bool):
        """Allocate and launch the index build on the grid of `bounds`; with_bbox: the bounding-box kernel in the same pass,
        returns [bbox(8) | hdr(8)] read back in one transfer."""
        n, s, dev = self.n, self.s, self.coords.device
        try:
            self.grid = L.grid_from_bounds(bounds[0], bounds[1], s)
        except L.LinkAmdError as e:
            raise GridTooLarge(str(e))
        self.v = v = self.grid.cells
        if v > MAX_CELLS:
            raise GridTooLarge(f"dense block grid would need {v} cells (> {MAX_CELLS})")
        ws = _workspace(dev)
        cell_counts, scratch = ws.ensure(n, v)
        i32 = dict(dtype=torch.int32, device=dev)
        self.cell_blk = torch.empty(v, **i32)
        self.vox_blk = torch.empty(n, **i32)
        self.idx_query = torch.empty(n, dtype=torch.int64, device=dev) if want_idx64 else None
        self.perm = torch.empty(n, **i32)
        self.vox_sorted = torch.empty((max(n, 1), 4), **i32)
        self.pos_blk = torch.empty(max(n, 1), **i32)
        self.blk_start = torch.empty(n + 1, **i32)
        self.blk_coords = torch.empty((max(n, 1), 4), **i32)
        self.counts_buf = torch.empty(max(n, 1), **i32)
        both = None
        if with_bbox:
            init = _BBOX_INIT.get(dev)
            if init is None:
                imax, imin = 2 ** 31 - 1, -2 ** 31
                init = _BBOX_INIT[dev] = torch.tensor([imax] * 4 + [imin] * 4, dtype=torch.int32, device=dev)
            both = torch.empty(8 + L.HDR_WORDS, **i32)
            both[:8].copy_(init)
            self.hdr = both[8:]
            L.check(L.lib().link_coords_bbox(self.coords.data_ptr(), n, both.data_ptr(), L.current_stream_handle()),
                    "link_coords_bbox")
        else:
            self.hdr = torch.empty(L.HDR_WORDS, **i32)
        L.check(L.lib().link_index_build(
            self.coords.data_ptr(), n, ctypes.byref(self.grid), cell_counts.data_ptr(), scratch.data_ptr(),
            scratch.numel(), self.cell_blk.data_ptr(), self.vox_blk.data_ptr(),
            self.idx_query.data_ptr() if want_idx64 else None, self.perm.data_ptr(), self.vox_sorted.data_ptr(),
            self.pos_blk.data_ptr(),
            self.blk_start.data_ptr(), self.blk_coords.data_ptr(), self.counts_buf.data_ptr(),
            self.hdr.data_ptr(), L.current_stream_handle()), "link_index_build")
        return both.tolist() if with_bbox else None

    # -- lazily synchronised facts ------------------------------------------------------------------
    @property
    def M(self) -> int:
        """Number of blocks (one 32-byte D2H sync on first use; also validates the status word)."""
        if self._m is None:
            h = self.hdr.tolist()
            if h[L.HDR_STATUS] != 0:
                raise L.LinkAmdError("BlockIndex: voxels outside the supplied bounds "
                                     f"{self.bounds} (status={h[L.HDR_STATUS]})")
            self._m = int(h[L.HDR_M])
        return self._m

    @property
    def counts(self) -> torch.Tensor:
        return self.counts_buf[: self.M]

    @property
    def block_coords(self) -> torch.Tensor:
        return self.blk_coords[: self.M]

    def neighbor_map(self, r: int, transpose: bool = False) -> torch.Tensor:
        """int32[M, r^3] neighbour block ids in get_kernel_offsets(r) order, -1 = absent."""
        if r % 2 == 1:
            transpose = False   # symmetric offset set: the adjoint relation is the relation itself
        key = (int(r), bool(transpose))
        if key not in self._nbr:
            m = self.M
            nbr = torch.empty((m, r ** 3), dtype=torch.int32, device=self.coords.device)
            L.check(L.lib().link_neighbor_map(self.blk_coords.data_ptr(), self.cell_blk.data_ptr(),
                                              ctypes.byref(self.grid), self.hdr.data_ptr(), m, int(r), 1,
                                              1 if transpose else 0, nbr.data_ptr(),
                                              L.current_stream_handle()), "link_neighbor_map")
            self._nbr[key] = nbr
        return self._nbr[key]


def unique_cells(rows: torch.Tensor, bounds) -> Tuple[torch.Tensor, torch.Tensor]:
    """Sorted unique rows of int32 `rows`[n,4] inside `bounds` (rows outside are dropped), lexicographic over the four
    columns, through the dense cell grid (include/link_amd.h: link_index_cells).  Returns (rows buffer int32[n,4] of
    which the first M are valid, hdr): M = hdr[HDR_M] stays on the device until the caller reads it."""
    rows = rows.contiguous()
    n, dev = rows.shape[0], rows.device
    try:
        grid = L.grid_from_bounds(bounds[0], bounds[1], 1)
    except L.LinkAmdError as e:
        raise GridTooLarge(str(e))
    v = grid.cells
    if v > MAX_CELLS:
        raise GridTooLarge(f"dense grid would need {v} cells (> {MAX_CELLS})")
    cell_counts, scratch = _workspace(dev).ensure(n, v)
    i32 = dict(dtype=torch.int32, device=dev)
    cell_blk = torch.empty(v, **i32)
    blk_start = torch.empty(n + 1, **i32)
    blk_coords = torch.empty((max(n, 1), 4), **i32)
    counts = torch.empty(max(n, 1), **i32)
    hdr = torch.empty(L.HDR_WORDS, **i32)
    L.check(L.lib().link_index_cells(rows.data_ptr(), n, ctypes.byref(grid), cell_counts.data_ptr(), scratch.data_ptr(),
                                     scratch.numel(), cell_blk.data_ptr(), blk_start.data_ptr(), blk_coords.data_ptr(),
                                     counts.data_ptr(), hdr.data_ptr(), L.current_stream_handle()), "link_index_cells")
    return blk_coords, hdr


def foreign_neighbor_map(rows: torch.Tensor, r: int, transpose: bool = False, step: int = 1,
                         table_rows: Optional[torch.Tensor] = None, bounds=None) -> torch.Tensor:
    """Neighbour map for arbitrary rows int32[M,4] (not produced by a BlockIndex): dense cell table
    over the bounding box of `table_rows` (default: the rows themselves; first duplicate row wins), then
    the same lookup kernel; neighbour offsets are multiplied by `step`.  With `table_rows`, entry [i,k] is
    the index in table_rows of rows[i] + offset_k*step (the kernel map of a strided convolution)."""
    rows = rows.contiguous()
    m = rows.shape[0]
    dev = rows.device
    nbr = torch.empty((m, r ** 3), dtype=torch.int32, device=dev)
    if m == 0:
        return nbr
    src = rows if table_rows is None else table_rows.contiguous()
    if src.shape[0] == 0:
        return nbr.fill_(-1)
    lo, hi = bounds if bounds is not None else coords_bounds(src)      # bounds of `src`, when the caller has them
    try:
        grid = L.grid_from_bounds(lo, hi, 1)
    except L.LinkAmdError as e:
        raise GridTooLarge(str(e))
    if grid.cells > MAX_CELLS:
        raise GridTooLarge(f"dense block grid would need {grid.cells} cells")
    # the cell table lives in the (device, stream) workspace and is kept all zero between calls: built by m scattered
    # writes, undone by m scattered zeros -- not a memset over every cell of a sparse grid per call
    ws = _workspace(dev)
    table = ws.cell_table(grid.cells)
    st = L.current_stream_handle()
    try:
        L.check(L.lib().link_cell_table_build(src.data_ptr(), src.shape[0], ctypes.byref(grid), table.data_ptr(), None,
                                              st), "link_cell_table_build")
        L.check(L.lib().link_neighbor_map(rows.data_ptr(), table.data_ptr(), ctypes.byref(grid), None, m, int(r),
                                          int(step), 1 if transpose else 0, nbr.data_ptr(), st), "link_neighbor_map")
        L.check(L.lib().link_cell_table_clear(src.data_ptr(), src.shape[0], ctypes.byref(grid), table.data_ptr(), st),
                "link_cell_table_clear")
    except Exception:
        ws.drop_cell_table()          # the 'all zero between calls' invariant no longer holds for this table
        raise
    return nbr
