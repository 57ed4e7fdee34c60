// link_amd/csrc/dense_fused.hip -- the fused kernels of the dense-cell layout (include/link_amd.h section E).
//
// Why fuse: counters on cfg2 (profiles/r02_*) show pre_mix bound by the f32 MFMA pipe with the VALU idle, and
// the per-cell modulate+sum bound by VALU issue with the MFMA pipe idle -- and `fin` making a 51 MB round trip
// between them.  k_dc_premix_modsum does both on the same 16-voxel tile: MFMA of one wave runs beside the
// sincos/modulate VALU work of the other wave on its SIMD, `fin` never leaves registers, and the per-cell
// sums are formed from an LDS image of the tile.
//
//   k_dc_index            coords -> cell, rank = cnt[cell]++, slots[cell][rank] = (x,y,z,id); vcell[i] = cell
//   k_dc_premix_modsum    a wave owns a range of cells: counts -> wave prefix scan -> flat, id-ordered voxel
//                         list in LDS -> tiles of 16 voxels: gather F rows, LayerNorm(F Wpre^T) on MFMA, theta /
//                         sincos / modulate in the MFMA layout (a lane holds 16 channels of ONE voxel: no lane
//                         is ever idle, whatever the cell sizes), X tile -> LDS -> per-cell sums -> S rows
//   k_dc_demod            per voxel pair, original order, persistent + software-pipelined: A[cell] row,
//                         de-modulate, LayerNorm, store
#define DC_IO 0
#define DC_IO_NS dcio_f32
#include "dense_fused_impl.h"

using namespace link;

namespace link {
// tile form of the fused pre_mix kernel (dense_tiles.hip)
int dc_tiles_modsum(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d, int64_t n, bool warm, hipStream_t st);
}

// the fp16 / bf16 instantiations live in their own translation units (dense_fused_f16.hip, dense_fused_bf16.hip)
#define DC_DECL_IO(NS)                                                                                                    \
  namespace NS {                                                                                                           \
  int run_premix_modsum(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t, bool,        \
                        hipStream_t);                                                                                      \
  int run_demod(const float *, const float *, const int32_t *, const int32_t *, const float *, const float *,             \
                const float *, const float *, const link_elk_desc_t &, const link_dc_grid_t &, int64_t, void *,            \
                hipStream_t);                                                                                              \
  int run_gather_demod(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t, hipStream_t); \
  int run_premix_modsum_sparse(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t, bool,  \
                               const int32_t *, hipStream_t);                                                              \
  int run_gather_demod_sparse(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t,         \
                              const int32_t *, hipStream_t);                                                               \
  int run_step3(const link_dc_buffers_t *, const link_dc_buffers_t *, const link_dc_buffers_t *, const link_dc_grid_t &,   \
                const link_elk_desc_t &, int64_t, int64_t, int64_t, int, hipStream_t);                                     \
  }
DC_DECL_IO(dcio_f16)
DC_DECL_IO(dcio_bf16)
#undef DC_DECL_IO

// ---------------------------------------------------------------------------------------------
// index: slot insert
// ---------------------------------------------------------------------------------------------
template <bool STATS>
#ifndef DC_INDEX_THREADS
#define DC_INDEX_THREADS 256     /* 128 / 512 / 1024: no difference (A/B on one box: 37.0-38.2 us/frame, index 10.0-10.5 us with events) */
#endif
__global__ void __launch_bounds__(DC_INDEX_THREADS) k_dc_index(const int4 *__restrict__ coords, int64_t n, link_dc_grid_t g,
                                                  uint32_t *__restrict__ cnt, int4 *__restrict__ slots,
                                                  int32_t *__restrict__ vcell, int32_t *__restrict__ hdr,
                                                  int32_t *__restrict__ stats) {
  int st_in = 0, st_first = 0, st_max = 0;
  dc_index_body<STATS>(coords, n, g, cnt, slots, vcell, hdr, (int)blockIdx.x, (int)gridDim.x, DC_INDEX_THREADS, st_in, st_first, st_max);
  if (STATS) dc_index_stats_flush(stats, st_in, st_first, st_max);
}

extern "C" int link_dc_index(const int32_t *coords, int64_t n, const link_dc_grid_t *g, uint32_t *cnt,
                             int32_t *slots, int32_t *vcell, int32_t *hdr, void *stream) {
  if (n < 0 || !g) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!coords || !cnt || !slots || !vcell || !hdr) return LINK_ERR_ARG;
  if (g->k < DC_INL || g->vp * (int64_t)g->k * 16 >= (1LL << 32) || n >= (1LL << 29)) return LINK_ERR_ARG;
  int64_t wgs = (n + DC_INDEX_THREADS - 1) / DC_INDEX_THREADS;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index<false>, dim3((unsigned)wgs), dim3(DC_INDEX_THREADS), 0, S(stream), reinterpret_cast<const int4 *>(coords), n,
                     *g, cnt, reinterpret_cast<int4 *>(slots), vcell, hdr, (int32_t *)nullptr);
  return check_launch("link_dc_index");
}

// ---------------------------------------------------------------------------------------------
// sparse-cell layout (dense_gather_sparse_impl.h): slot insert + first-voxel marks, one R_core step in three launches
// ---------------------------------------------------------------------------------------------
// The insert of k_dc_index, plus: occ[v] = the cell voxel v was the first of (insert rank 0), 0 otherwise -- the two fused
// kernels walk ranges of voxel ids and take the cells marked there -- and cell_n of the PREVIOUS frame's cells back to zero
// (its marks, occ_prev[0 .. n_prev)): a neighbour is present iff its published count is non-zero, so stale counts of cells
// this frame does not occupy must not survive; the cells it does occupy are rewritten by the pre_mix kernel afterwards.
__global__ void __launch_bounds__(DC_INDEX_THREADS) k_dc_index_sparse(const int4 *__restrict__ coords, int64_t n, link_dc_grid_t g,
                                                                      uint32_t *__restrict__ cnt, int4 *__restrict__ slots,
                                                                      int32_t *__restrict__ vcell, int32_t *__restrict__ hdr,
                                                                      int32_t *__restrict__ occ, const int32_t *__restrict__ occ_prev,
                                                                      int64_t n_prev, int32_t *__restrict__ cell_n) {
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_n = dc_rsrc(cell_n, (uint32_t)(g.vp * 4));
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[LINK_HDR_NVALID] = (int32_t)n;
  const int64_t top = n > n_prev ? n : n_prev;
  for (int64_t v = (int64_t)blockIdx.x * DC_INDEX_THREADS + threadIdx.x; v < top; v += (int64_t)gridDim.x * DC_INDEX_THREADS) {
    if (v < n_prev) {
      const int pp = occ_prev[v];
      st4i(r_n, pp ? (uint32_t)pp * 4u : DC_OOB, 0);
    }
    if (v >= n) continue;
    const int4 rc = coords[v];
    const unsigned ux = (unsigned)(floordiv(rc.x, g.s) - g.lo[0]), uy = (unsigned)(floordiv(rc.y, g.s) - g.lo[1]);
    const unsigned uz = (unsigned)(floordiv(rc.z, g.s) - g.lo[2]), ub = (unsigned)(rc.w - g.lo[3]);
    const bool inside = ux < (unsigned)g.dim[0] && uy < (unsigned)g.dim[1] && uz < (unsigned)g.dim[2] &&
                        ub < (unsigned)g.dim[3];
    if (!inside) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 1);
    const int pcell = inside ? dc_cell(g, (int)ux, (int)uy, (int)uz, (int)ub) : 0;
    const int rank = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_cnt, pcell ? (uint32_t)pcell * 4u : DC_OOB, 0, 0);
    const bool full = pcell != 0 && rank >= g.k;
    if (full) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 2);
    const bool keep = pcell != 0 && !full;
    st16i(r_slots, keep ? dc_slot(g, pcell, rank) * 16u : DC_OOB, make_int4(rc.x, rc.y, rc.z, (int)v));
    vcell[v] = keep ? pcell : 0;
    occ[v] = (keep && rank == 0) ? pcell : 0;
  }
}

static int dc_common_ok(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d, int64_t n);

extern "C" int link_elk_core_sparse_forward(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                            int64_t n, int32_t build_index, int32_t *occ, const int32_t *occ_prev,
                                            int64_t n_prev, void *stream) {
  if (dc_common_ok(b, g, d, n) != LINK_OK || n_prev < 0) return LINK_ERR_ARG;
  if (d->c != 16 && d->c != 32 && d->c != 64) return LINK_ERR_ARG;
  if ((d->r != 2 && d->r != 3) || g->k > 64) return LINK_ERR_ARG;       // a cell is one wave's serial work: small cells only
  if (d->op == LINK_OP_COSX && !b->fin) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!b->feats || !b->coords || !b->slots || !b->cnt || !b->cell_n || !b->vcell || !b->w_pre || !b->pre_ln_w || !b->pre_ln_b ||
      !b->w_pos || !b->ln_w || !b->ln_b || !b->S || !b->hdr || !b->out || !occ || (n_prev > 0 && !occ_prev))
    return LINK_ERR_ARG;
  if (n >= (1LL << 29) || n_prev >= (1LL << 29)) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (build_index) {
    const int64_t top = n > n_prev ? n : n_prev;
    int64_t wgs = (top + DC_INDEX_THREADS - 1) / DC_INDEX_THREADS;
    if (wgs > 4096) wgs = 4096;
    hipLaunchKernelGGL(k_dc_index_sparse, dim3((unsigned)wgs), dim3(DC_INDEX_THREADS), 0, st, reinterpret_cast<const int4 *>(b->coords), n,
                       *g, b->cnt, reinterpret_cast<int4 *>(b->slots), b->vcell, b->hdr, occ, occ_prev, n_prev, b->cell_n);
    int rc = check_launch("link_dc_index(sparse)");
    if (rc != LINK_OK) return rc;
  }
  int rc;
  const bool warm = build_index == 0;
  switch (b->io_dtype) {
    case 1: rc = dcio_f16::run_premix_modsum_sparse(b, *g, *d, n, warm, occ, st); break;
    case 2: rc = dcio_bf16::run_premix_modsum_sparse(b, *g, *d, n, warm, occ, st); break;
    default: rc = dcio_f32::run_premix_modsum_sparse(b, *g, *d, n, warm, occ, st); break;
  }
  if (rc != LINK_OK) return rc;
  switch (b->io_dtype) {
    case 1: return dcio_f16::run_gather_demod_sparse(b, *g, *d, n, occ, st);
    case 2: return dcio_bf16::run_gather_demod_sparse(b, *g, *d, n, occ, st);
    default: return dcio_f32::run_gather_demod_sparse(b, *g, *d, n, occ, st);
  }
}

// One step of the three-frame pipeline (dense_step3_impl.h): slot insert of the frame in b_insert, K1 of the frame in b_k1, K2 of
// the frame in b_k2, in one launch.  A null frame is an absent stage.
static int dc_step3_frame_ok(const link_dc_buffers_t *b, int64_t n, int io) {
  if (!b) return LINK_OK;
  if (n < 0 || n >= (1LL << 29) || b->io_dtype != io || b->alpha) return LINK_ERR_ARG;
  if (!b->feats || !b->coords || !b->slots || !b->cnt || !b->cell_n || !b->vcell || !b->w_pre || !b->pre_ln_w || !b->pre_ln_b ||
      !b->w_pos || !b->ln_w || !b->ln_b || !b->S || !b->hdr || !b->out)
    return LINK_ERR_ARG;
  return LINK_OK;
}
extern "C" int link_elk_core_dense_step3(const link_dc_buffers_t *b_insert, int64_t n_insert, const link_dc_buffers_t *b_k1,
                                         int64_t n_k1, const link_dc_buffers_t *b_k2, int64_t n_k2, const link_dc_grid_t *g,
                                         const link_elk_desc_t *d, int32_t insert_wgs, void *stream) {
  const link_dc_buffers_t *any = b_k2 ? b_k2 : (b_k1 ? b_k1 : b_insert);
  if (!any) return LINK_OK;
  if (dc_common_ok(any, g, d, 0) != LINK_OK || g->k > 352) return LINK_ERR_ARG;
  const int io = any->io_dtype;
  if (dc_step3_frame_ok(b_insert, n_insert, io) || dc_step3_frame_ok(b_k1, n_k1, io) || dc_step3_frame_ok(b_k2, n_k2, io))
    return LINK_ERR_ARG;
  const int64_t nmax = std::max(std::max(b_insert ? n_insert : 0, b_k1 ? n_k1 : 0), b_k2 ? n_k2 : 0);
  if (nmax * (int64_t)d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  if ((b_insert && (b_insert == b_k1 || b_insert == b_k2)) || (b_k1 && b_k1 == b_k2)) return LINK_ERR_ARG;   // three different frames
  switch (io) {
    case 1: return dcio_f16::run_step3(b_insert, b_k1, b_k2, *g, *d, n_insert, n_k1, n_k2, insert_wgs, S(stream));
    case 2: return dcio_bf16::run_step3(b_insert, b_k1, b_k2, *g, *d, n_insert, n_k1, n_k2, insert_wgs, S(stream));
    default: return dcio_f32::run_step3(b_insert, b_k1, b_k2, *g, *d, n_insert, n_k1, n_k2, insert_wgs, S(stream));
  }
}

namespace link {
// the insert of link_dc_index + occupancy statistics (behind link_dc_index_probe, dense.hip)
int dc_index_stats_run(const link_dc_buffers_t *b, const link_dc_grid_t *g, int64_t n, int32_t *stats, hipStream_t st) {
  if (!b->coords || !b->cnt || !b->slots || !b->vcell || !b->hdr) return LINK_ERR_ARG;
  if (g->k < DC_INL || g->vp * (int64_t)g->k * 16 >= (1LL << 32) || n >= (1LL << 29)) return LINK_ERR_ARG;
  int64_t wgs = (n + DC_INDEX_THREADS - 1) / DC_INDEX_THREADS;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index<true>, dim3((unsigned)wgs), dim3(DC_INDEX_THREADS), 0, st, reinterpret_cast<const int4 *>(b->coords), n, *g, b->cnt,
                     reinterpret_cast<int4 *>(b->slots), b->vcell, b->hdr, stats);
  return check_launch("link_dc_index_probe");
}
}  // namespace link

static int dc_common_ok(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d, int64_t n) {
  if (!b || !g || !d || n < 0) return LINK_ERR_ARG;
  if (b->io_dtype < 0 || b->io_dtype > 2) return LINK_ERR_ARG;
  if (d->op < 0 || d->op > 2 || d->cg <= 0 || d->c % d->cg != 0 || g->k < DC_INL) return LINK_ERR_ARG;
  const int parts = d->op == LINK_OP_COSX ? 3 : 2;
  if ((g->vp + 1) * (int64_t)parts * d->c * 4 >= (1LL << 32) || g->vp * (int64_t)g->k * 16 >= (1LL << 32)) return LINK_ERR_ARG;
  if (n * (int64_t)d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  return LINK_OK;
}

extern "C" int link_dc_premix_modsum(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                     int64_t n, int32_t warm, void *stream) {
  if (dc_common_ok(b, g, d, n) != LINK_OK) return LINK_ERR_ARG;
  if (d->c != 16 && d->c != 32 && d->c != 64) return LINK_ERR_ARG;
  if (g->k > 352) return LINK_ERR_ARG;                 // a cell's records must fit the wave's LDS list (LCAP)
  if (d->op == LINK_OP_COSX && !b->fin) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!b->feats || !b->slots || !b->cnt || !b->cell_n || !b->w_pre || !b->pre_ln_w || !b->pre_ln_b || !b->w_pos ||
      !b->S || !b->hdr)
    return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (b->tune.k1_form == 1) return dc_tiles_modsum(b, g, d, n, warm != 0, st);
  switch (b->io_dtype) {
    case 1: return dcio_f16::run_premix_modsum(b, *g, *d, n, warm != 0, st);
    case 2: return dcio_bf16::run_premix_modsum(b, *g, *d, n, warm != 0, st);
    default: return dcio_f32::run_premix_modsum(b, *g, *d, n, warm != 0, st);
  }
}

extern "C" int link_dc_demod(const float *A, const float *fin, const int32_t *coords, const int32_t *vcell,
                             const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                             const link_elk_desc_t *d, const link_dc_grid_t *g, int64_t n, void *out, int32_t io_dtype,
                             void *stream) {
  if (!d || !g || n < 0 || !dc_width_ok(d->c) || d->op < 0 || d->op > 2 || d->cg <= 0 || d->c % d->cg != 0) return LINK_ERR_ARG;
  if (io_dtype < 0 || io_dtype > 2) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!A || !coords || !vcell || !w_pos || !ln_w || !ln_b || !out || (d->op == LINK_OP_COSX && !fin)) return LINK_ERR_ARG;
  const int parts = d->op == LINK_OP_COSX ? 3 : 2;
  if ((g->vp + 1) * (int64_t)parts * d->c * 4 >= (1LL << 32) || n * (int64_t)d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (io_dtype) {
    case 1: return dcio_f16::run_demod(A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, *d, *g, n, out, st);
    case 2: return dcio_bf16::run_demod(A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, *d, *g, n, out, st);
    default: return dcio_f32::run_demod(A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, *d, *g, n, out, st);
  }
}

extern "C" int link_dc_gather_demod(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                    int64_t n, void *stream) {
  if (dc_common_ok(b, g, d, n) != LINK_OK || !dc_width_ok(d->c) || (d->r != 2 && d->r != 3)) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!b->S || !b->cell_n || !b->slots || !b->w_pos || !b->ln_w || !b->ln_b || !b->out || (d->op == LINK_OP_COSX && !b->fin))
    return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (d->c != 64) return LINK_ERR_ARG;               // the fused gather + de-modulate kernels are built for C = 64; other widths: link_dc_gather + link_dc_demod
  switch (b->io_dtype) {
    case 1: return dcio_f16::run_gather_demod(b, *g, *d, n, st);
    case 2: return dcio_bf16::run_gather_demod(b, *g, *d, n, st);
    default: return dcio_f32::run_gather_demod(b, *g, *d, n, st);
  }
}
