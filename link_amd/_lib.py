"""link_amd/_lib.py -- ctypes binding of liblink_amd.so (the C ABI declared in include/link_amd.h).

The product path is HIP-only: if the shared library is missing this module raises at import of the
first symbol -- there is no CPU or PyTorch fallback (the oracle under oracle/ is test infrastructure
and is never imported from here).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "liblink_amd.so")

LINK_OK, LINK_ERR_ARG, LINK_ERR_LAUNCH, LINK_ERR_WORKSPACE, LINK_BATCH_TIMEOUT = 0, -1, -2, -3, -4
HDR_M, HDR_STATUS, HDR_NVALID, HDR_STATUS_ACC, HDR_WORDS = 0, 1, 2, 3, 8
OP_COS, OP_SIN, OP_COSX = 0, 1, 2
ELK_LANE_CHANNEL, ELK_NO_PAIR, ELK_FUSED_GATHER, ELK_NO_DENSE_GRID, ELK_TILES = 1, 2, 4, 8, 16     # link_elk_desc_t::flags
ELK_LEAN_CS, ELK_LEAN_NO_CS, ELK_LEAN_PM, ELK_LEAN_NO_PM = 32, 64, 128, 256
IO_F32, IO_F16, IO_BF16 = 0, 1, 2
ABI_VERSION = 12
# LINK_AMD_DEBUG=1: read the device status word back after every core call (one 32-byte D2H sync per call) and
# raise when the index dropped a voxel -- the sync-free default trusts the caller's bounds
DEBUG = os.environ.get("LINK_AMD_DEBUG", "0") not in ("", "0")


class LinkGrid(Structure):
    """link_grid_t"""
    _fields_ = [("s", c_int32), ("lo", c_int32 * 4), ("dim", c_int32 * 4)]

    @property
    def cells(self) -> int:
        v = 1
        for d in self.dim:
            v *= int(d)
        return v

    def key(self):
        return (int(self.s), tuple(int(x) for x in self.lo), tuple(int(x) for x in self.dim))


class LinkElkDesc(Structure):
    """link_elk_desc_t"""
    _fields_ = [("op", c_int32), ("c", c_int32), ("cg", c_int32), ("r", c_int32),
                ("coord_div", c_float), ("eps", c_float), ("flags", c_int32)]


class LinkElkBuffers(Structure):
    """link_elk_buffers_t"""
    _fields_ = [(k, c_void_p) for k in ("feats", "coords", "w_pre", "pre_ln_w", "pre_ln_b", "w_pos", "alpha",
                                        "ln_w", "ln_b", "cell_counts", "scratch")] + \
               [("scratch_bytes", c_size_t)] + \
               [(k, c_void_p) for k in ("cell_blk", "vox_blk", "idx_query", "perm", "vox_sorted", "pos_blk", "blk_start",
                                        "blk_coords", "counts", "hdr", "fin", "S", "A", "out")] + \
               [("s_bytes", c_int64), ("io_dtype", c_int32), ("reserved", c_int32), ("cell_pair", c_void_p)]


class LinkDcGrid(Structure):
    """link_dc_grid_t (dense-cell path: padded grid + slot capacity)"""
    _fields_ = [("s", c_int32), ("lo", c_int32 * 4), ("dim", c_int32 * 4), ("pdim", c_int32 * 3),
                ("k", c_int32), ("vp", c_int64)]

    @property
    def interior_cells(self) -> int:
        v = 1
        for d in self.dim:
            v *= int(d)
        return v


class LinkDcTuning(Structure):
    """link_dc_tuning_t: launch geometry / kernel selection of ONE plan (all zero = defaults)"""
    _fields_ = [(k, c_int32) for k in ("k1_wgs", "k2_zsplit", "k1_lds_pad", "k2_lds_pad", "k1_form", "k2_form", "mode",
                                       "reserved0", "reserved")] + [("k1_dbg", c_void_p), ("k2_dbg", c_void_p)]


class LinkDcBuffers(Structure):
    """link_dc_buffers_t"""
    _fields_ = [(k, c_void_p) for k in ("feats", "coords", "w_pre", "pre_ln_w", "pre_ln_b", "w_pos", "alpha",
                                        "ln_w", "ln_b", "cnt", "slots", "sid", "vrec", "vcell", "cell_n", "hdr", "fin",
                                        "S", "A", "out")] + [("io_dtype", c_int32), ("tune", LinkDcTuning)]


class LinkLeanBuffers(Structure):
    """link_lean_buffers_t (lean form of R_core: three launches with the index rebuilt)"""
    _fields_ = [(k, c_void_p) for k in ("feats", "coords", "w_pre", "pre_ln_w", "pre_ln_b", "w_pos", "alpha", "ln_w", "ln_b",
                                        "cnt", "cnt_prev", "list", "rec2", "occ", "occ_prev", "ctrl", "ctrl_prev", "X", "S",
                                        "hdr", "out")] + [("seg_cap", c_int64), ("k", c_int32), ("io_dtype", c_int32), ("cnt_shift", c_int32), ("reserved", c_int32)]


LEAN_KMAX, LEAN_SEGS, LEAN_CHUNK = 352, 16, 32
BLOCK_DONE, BLOCK_MISS = 0, 1


class LinkBlockArgs(Structure):
    """link_block_args_t (section G: one host call per LinK block on a new coordinate set)"""
    _fields_ = [("buf", POINTER(LinkDcBuffers)), ("g", POINTER(LinkDcGrid)), ("desc", POINTER(LinkElkDesc)), ("n", c_int64),
                ("mean_max", c_int32), ("cell_max", c_int32), ("ts", c_int32), ("subm", c_int32),
                ("nbr", c_void_p), ("pair_arena", c_void_p),
                ("pair_arena_words", c_int64), ("contrib", c_void_p), ("contrib_rows", c_int64), ("w", c_void_p), ("ws", c_void_p),
                ("w_big", c_void_p), ("nl_w", c_void_p), ("nl_b", c_void_p), ("nl_eps", c_float), ("flags", c_int32),
                ("out", c_void_p), ("bbox", c_int32 * 8), ("stats", c_int32 * 4), ("verdict", c_int32), ("reserved", c_int32)]


# name -> (restype, argtypes); every symbol include/link_amd.h declares
SIGNATURES = {
    "link_abi_version": (c_int, []),
    "link_abi_struct_size": (c_int32, [c_int32]),
    "link_last_error": (c_char_p, []),
    "link_hash": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "link_kernel_hash": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "link_hash_query_workspace_bytes": (c_size_t, [c_int64]),
    "link_hash_query": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p,
                                c_size_t, c_void_p]),
    "link_count": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "link_ti_weights": (c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    "link_voxelize_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                      c_void_p]),
    "link_voxelize_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "link_devoxelize_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p,
                                        c_void_p]),
    "link_devoxelize_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64,
                                         c_void_p, c_void_p]),
    "link_grid_from_bounds": (c_int64, [POINTER(c_int32), POINTER(c_int32), c_int32, POINTER(LinkGrid)]),
    "link_coords_bbox": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "link_index_scratch_bytes": (c_size_t, [c_int64, c_int64]),
    "link_index_build": (c_int, [c_void_p, c_int64, POINTER(LinkGrid), c_void_p, c_void_p, c_size_t,
                                 c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p, c_void_p]),
    "link_index_build_first": (c_int, [c_void_p, c_int64, POINTER(LinkGrid), c_void_p, c_void_p, c_size_t] + [c_void_p] * 11),
    "link_block_gather": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(LinkGrid), c_void_p,
                                  POINTER(LinkElkDesc), c_int64, c_void_p, c_void_p]),
    "link_voxel_demod_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, POINTER(LinkElkDesc), c_int64, c_void_p, c_void_p]),
    "link_neighbor_map": (c_int, [c_void_p, c_void_p, POINTER(LinkGrid), c_void_p, c_int64, c_int32,
                                  c_int32, c_int32, c_void_p, c_void_p]),
    "link_cell_table_build": (c_int, [c_void_p, c_int64, POINTER(LinkGrid), c_void_p, c_void_p, c_void_p]),
    "link_conv_site_table": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "link_conv_gather_table": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p,
                                       c_void_p]),
    "link_cell_table_clear": (c_int, [c_void_p, c_int64, POINTER(LinkGrid), c_void_p, c_void_p]),
    "link_block_mean": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64,
                                c_void_p, c_void_p]),
    "link_aux_to_voxel_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                          c_int64, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "link_aux_to_voxel_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int64, c_int64, c_int64, c_int64, c_void_p, c_void_p,
                                           c_void_p]),
    "link_premix_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float,
                               c_void_p, c_void_p]),
    "link_modulate_block_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, POINTER(LinkElkDesc), c_int64, c_int64, c_void_p,
                                        c_void_p]),
    "link_gather_demod_ln": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, POINTER(LinkGrid), c_void_p,
                                     POINTER(LinkElkDesc), c_int64, c_int64, c_void_p, c_void_p]),
    "link_elk_core_forward": (c_int, [POINTER(LinkElkBuffers), POINTER(LinkGrid), POINTER(LinkElkDesc),
                                      c_int64, c_int64, c_int32, c_void_p]),
    "link_elk_tiles_table_bytes": (c_int64, [POINTER(LinkElkDesc), c_int64, c_int64]),
    "link_elk_premix_modsum_tiles": (c_int, [c_void_p] * 10 + [POINTER(LinkElkDesc), c_int64, c_int64, c_void_p, c_int64,
                                             c_void_p, c_void_p]),
    "link_elk_gather_demod_tiles": (c_int, [c_void_p] * 6 + [POINTER(LinkGrid)] + [c_void_p] * 5 +
                                    [POINTER(LinkElkDesc), c_int64, c_int64, c_void_p, c_void_p]),
    "link_elk_premix_modsum_tiles_io": (c_int, [c_void_p, c_int32] + [c_void_p] * 9 + [POINTER(LinkElkDesc), c_int64, c_int64, c_void_p,
                                                c_int64, c_void_p, c_void_p]),
    "link_elk_gather_demod_tiles_io": (c_int, [c_void_p] * 6 + [POINTER(LinkGrid)] + [c_void_p] * 5 +
                                       [POINTER(LinkElkDesc), c_int64, c_int64, c_void_p, c_int32, c_void_p]),
    "link_elk_mid_forward": (c_int, [c_void_p] * 6 + [POINTER(LinkGrid), c_void_p, c_void_p, c_void_p,
                                     POINTER(LinkElkDesc), c_void_p, c_void_p, c_int64, c_int64] + [c_void_p] * 5),
    "link_elk_out_ln_backward": (c_int, [c_void_p] * 9 + [POINTER(LinkElkDesc), c_int64, c_void_p, c_void_p,
                                         c_void_p]),
    "link_subm_conv_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                       c_void_p, c_void_p]),
    "link_aux_to_voxel_forward_grid": (c_int, [c_void_p] * 4 + [POINTER(LinkGrid), c_void_p, c_void_p, c_int64,
                                               c_int64, c_int32, c_int32] + [c_void_p] * 5),
    "link_aux_to_voxel_forward_scatter": (c_int, [c_void_p] * 4 + [POINTER(LinkGrid), c_void_p, c_void_p, c_void_p, c_int64,
                                                  c_int64, c_int32, c_int32] + [c_void_p] * 4),
    "link_conv_pairs_supported": (c_int, [c_int32, c_int32]),
    "link_conv_pairs_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "link_conv_pairs_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                    c_float, c_void_p, c_int32, c_void_p, c_void_p]),
    "link_index_cells": (c_int, [c_void_p, c_int64, POINTER(LinkGrid), c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_void_p, c_void_p]),
    "link_conv_out_candidate_count": (c_int32, [c_void_p, c_void_p]),
    "link_conv_out_candidates": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "link_conv_pairs_gemm_io": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "link_conv_pairs_sum_io": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                       c_float, c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "link_conv_centre_sum_io": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                        c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "link_conv_pairs_gemm_amp": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "link_conv_centre_sum_amp": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                         c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_int32, c_void_p]),
    "link_bn_partial_workgroups": (c_int32, [c_int64, c_int32]),
    "link_bn_forward_stats": (c_int, [c_void_p, c_int64, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "link_bn_backward_reduce": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p]),
    "link_bn_apply_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "link_bn_backward_reduce_relu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "link_bn_apply_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p]),
    "link_conv_pairs_gemm_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "link_pair_plan_count": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    "link_pair_plan_fill": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                    c_void_p]),
    "link_subm_conv_resident_amp": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                            c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_void_p]),
    "link_pair_plan_build": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int64] + [c_void_p] * 13),
    "link_pair_plan_layout": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "link_conv_pairs_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int64,
                                      c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "link_conv_centre_sum": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int32, c_int32,
                                     c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_void_p]),
    "link_subm_conv_wgrad_chunks": (c_int32, []),
    "link_subm_conv_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    "link_subm_conv_wgrad_split": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "link_subm_conv_wgrad_reduce": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "link_subm_conv_ln_add_relu": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                           c_void_p, c_void_p, c_float, c_void_p, c_int32, c_void_p, c_void_p]),
    "link_ln_add_relu_forward": (c_int, [c_void_p] * 4 + [c_int64, c_int32, c_float, c_void_p, c_void_p]),
    "link_ln_add_relu_backward": (c_int, [c_void_p] * 4 + [c_int64, c_int32, c_float] + [c_void_p] * 4),
    "link_sum_partials": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int64, c_void_p,
                                  c_void_p]),
    "link_premix_ln_backward": (c_int, [c_void_p] * 4 + [c_int64, c_int32, ctypes.c_float] + [c_void_p] * 4),
    "link_elk_mid_partial_rows": (c_int32, []),
    "link_dc_grid_from": (c_int64, [POINTER(LinkGrid), c_int32, POINTER(LinkDcGrid)]),
    "link_dc_premix_insert": (c_int, [c_void_p] * 5 + [c_int64, c_int32, c_float, POINTER(LinkDcGrid), c_int32] +
                              [c_void_p] * 7),
    "link_dc_modsum": (c_int, [c_void_p] * 6 + [POINTER(LinkElkDesc), POINTER(LinkDcGrid), c_int32, c_void_p,
                               c_void_p, c_void_p]),
    "link_dc_gather": (c_int, [c_void_p, c_void_p, POINTER(LinkElkDesc), POINTER(LinkDcGrid), c_void_p, c_void_p]),
    "link_elk_core_dense_forward": (c_int, [POINTER(LinkDcBuffers), POINTER(LinkDcGrid), POINTER(LinkElkDesc),
                                            c_int64, c_int32, c_void_p]),
    "link_elk_core_lean_forward": (c_int, [POINTER(LinkLeanBuffers), POINTER(LinkGrid), POINTER(LinkElkDesc), c_int64, c_int64,
                                           c_int32, c_void_p]),
    "link_dc_index_probe": (c_int, [POINTER(LinkDcBuffers), POINTER(LinkDcGrid), c_int64, c_void_p, c_void_p]),
    "link_dc_index_ids": (c_int, [c_void_p, c_int64, POINTER(LinkDcGrid)] + [c_void_p] * 5),
    "link_dc_index": (c_int, [c_void_p, c_int64, POINTER(LinkDcGrid)] + [c_void_p] * 5),
    "link_dc_premix_modsum": (c_int, [POINTER(LinkDcBuffers), POINTER(LinkDcGrid), POINTER(LinkElkDesc), c_int64,
                                      c_int32, c_void_p]),
    "link_dc_gather_demod": (c_int, [POINTER(LinkDcBuffers), POINTER(LinkDcGrid), POINTER(LinkElkDesc), c_int64, c_void_p]),
    "link_dc_demod": (c_int, [c_void_p] * 8 + [POINTER(LinkElkDesc), POINTER(LinkDcGrid), c_int64, c_void_p, c_int32,
                              c_void_p]),
    "link_block_ctx_create": (c_int, [POINTER(c_void_p)]),
    "link_block_ctx_destroy": (c_int, [c_void_p]),
    "link_pair_plan_arena": (c_int64, [c_int64, c_int32, c_int32, POINTER(c_int64)]),
    "link_dc_neighbor_map": (c_int, [c_void_p, c_int64, POINTER(LinkDcGrid), c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "link_elk_block_forward": (c_int, [c_void_p, POINTER(LinkBlockArgs), c_void_p]),
    "link_dc_batch_create": (c_int, [POINTER(c_void_p)]),
    "link_dc_batch_destroy": (c_int, [c_void_p]),
    "link_elk_core_dense_forward_batch": (c_int, [c_void_p, POINTER(LinkDcBuffers), POINTER(c_int64), c_int32, POINTER(LinkDcGrid),
                                                  POINTER(LinkElkDesc), c_void_p]),
    "link_dc_batch_submit": (c_int, [c_void_p, POINTER(LinkDcBuffers), POINTER(c_int64), c_int32, POINTER(LinkDcGrid),
                                     POINTER(LinkElkDesc), c_void_p, POINTER(c_int64)]),
    "link_dc_batch_join": (c_int, [c_void_p, c_int64, c_void_p]),
    "link_dc_batch_set_timing": (c_int, [c_void_p, c_int32]),
    "link_dc_batch_kernel_times": (c_int, [c_void_p, c_int64, POINTER(c_float)]),
    "link_dc_batch_status": (c_int, [c_void_p, POINTER(c_int32)]),
    "link_dc_batch_set_debug": (c_int, [c_void_p, c_void_p, c_void_p]),
    "link_dc_batch_probe_streams": (c_int, [c_void_p, c_void_p, POINTER(ctypes.c_double)]),
    "link_streams_share_queue": (c_int, [c_void_p, c_void_p, POINTER(ctypes.c_double)]),
    "link_elk_mid_backward": (c_int, [c_void_p] * 9 + [POINTER(LinkGrid), c_void_p, c_void_p, c_void_p,
                                      POINTER(LinkElkDesc), c_int64, c_int64] + [c_void_p] * 5),
}

_lib = None


class LinkAmdError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    """Load liblink_amd.so (once).  Fails loudly: no fallback exists."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise LinkAmdError(
                f"{SO_PATH} not found: the HIP extension is not built. Run `python link_amd/build.py` "
                "(or __graft_entry__.build()). link_amd has no CPU/PyTorch fallback.")
        handle = ctypes.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if handle.link_abi_version() != ABI_VERSION:
            raise LinkAmdError("liblink_amd.so ABI version mismatch; rebuild with link_amd/build.py")
        for which, cls in enumerate((LinkGrid, LinkElkDesc, LinkElkBuffers, LinkDcGrid, LinkDcTuning, LinkDcBuffers, LinkLeanBuffers, LinkBlockArgs)):
            if handle.link_abi_struct_size(which) != ctypes.sizeof(cls):
                raise LinkAmdError(f"liblink_amd.so: layout of {cls.__name__} differs from include/link_amd.h "
                                   f"({ctypes.sizeof(cls)} bytes here, {handle.link_abi_struct_size(which)} in the library)")
        _lib = handle
    return _lib


def current_stream_handle() -> int:
    """hipStream_t of torch's current stream on the current device.  torch.cuda.current_stream() builds a Python
    Stream object (~10 us); the raw query is what torch's own compiled code uses (~0.3 us) -- it matters on the
    module path, where a stage is a dozen FFI calls."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:                      # private API moved: fall back to the public one
        return torch.cuda.current_stream().cuda_stream


def check(rc: int, what: str) -> None:
    if rc != LINK_OK:
        msg = {LINK_ERR_ARG: "invalid argument", LINK_ERR_LAUNCH: "HIP launch failure",
               LINK_ERR_WORKSPACE: "workspace too small",
               LINK_BATCH_TIMEOUT: "a bounded wait inside a persistent batch kernel gave up"}.get(rc, f"error {rc}")
        detail = lib().link_last_error()
        raise LinkAmdError(f"{what}: {msg}" + (f" ({detail.decode()})" if detail else ""))


def grid_from_bounds(lo, hi, s: int) -> LinkGrid:
    """Host helper (no GPU needed): block grid covering voxel coords lo..hi (inclusive)."""
    g = LinkGrid()
    lo_a = (c_int32 * 4)(*[int(x) for x in lo])
    hi_a = (c_int32 * 4)(*[int(x) for x in hi])
    v = lib().link_grid_from_bounds(lo_a, hi_a, int(s), ctypes.byref(g))
    if v < 0:
        raise LinkAmdError(f"grid_from_bounds({list(lo)}, {list(hi)}, s={s}): bounds invalid or grid "
                           ">= 2^30 cells")
    return g


def dc_grid_from(grid: LinkGrid, k: int = 0):
    """Padded dense-cell grid of `grid` (slot capacity k, 0 = s^3), or None when the dense-cell path cannot
    address it (>= 2^30 cells or >= 2^31 slots)."""
    g = LinkDcGrid()
    v = lib().link_dc_grid_from(ctypes.byref(grid), int(k), ctypes.byref(g))
    return g if v > 0 else None
