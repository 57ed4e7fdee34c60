// link_amd/csrc/dense_tiles.hip -- tile form of the fused pre_mix kernel (dense_tiles_impl.h), fp32 feature rows, and the
// slot insert that feeds it (voxel ids instead of records).
#define DC_IO 0
#define DC_IO_NS dcio_f32
#include "dense_tiles_impl.h"

using namespace link;

// coords -> cell, rank = cnt[cell]++, sid[cell][rank] = voxel id, vcell[id] = cell.  Every store is an unconditional
// buffer store whose offset is out of range when the lane has nothing to write (hardware drops it): no branch around
// a VMEM op, hence counted vmcnt waits (load -> atomic -> store is the whole kernel: latency only).
//
// STATS (first visit of a coordinate set, link_dc_index_probe): voxels inside the grid, occupied cells and the fullest
// cell's count also go to stats[0..2], one atomic per wave and pass -- the frame's occupancy without a second insert.
template <bool STATS>
__global__ void __launch_bounds__(256) k_dc_index_ids(const int4 *__restrict__ coords, int64_t n, link_dc_grid_t g,
                                                      uint32_t *__restrict__ cnt, uint32_t *__restrict__ sid,
                                                      int32_t *__restrict__ vcell, int32_t *__restrict__ hdr, uint32_t sid_bytes,
                                                      int32_t *__restrict__ stats) {
  const __amdgpu_buffer_rsrc_t r_sid = dc_rsrc(sid, sid_bytes);
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  int st_in = 0, st_first = 0, st_max = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[LINK_HDR_NVALID] = (int32_t)n;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
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
    st4i(r_sid, keep ? dc_sid_off(g, pcell, rank) * 4u : DC_OOB, (int)v);
    vcell[v] = keep ? pcell : 0;
    if (STATS) { st_in += pcell != 0; st_first += (pcell != 0 && rank == 0); st_max = max(st_max, pcell != 0 ? rank + 1 : 0); }
  }
  if (STATS) dc_index_stats_flush(stats, st_in, st_first, st_max);
}

static inline int64_t dc_sid_words(const link_dc_grid_t *g) { return g->vp * (int64_t)(g->k > DC_SID_INL ? g->k : DC_SID_INL); }

extern "C" int link_dc_index_ids(const int32_t *coords, int64_t n, const link_dc_grid_t *g, uint32_t *cnt, uint32_t *sid,
                                 int32_t *vcell, int32_t *hdr, void *stream) {
  if (n < 0 || !g) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!coords || !cnt || !sid || !vcell || !hdr) return LINK_ERR_ARG;
  if (g->k < 1 || dc_sid_words(g) * 4 >= (1LL << 32) || n > (1LL << DC_ID_BITS)) return LINK_ERR_ARG;
  int64_t wgs = (n + 255) / 256;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index_ids<false>, dim3((unsigned)wgs), dim3(256), 0, S(stream), reinterpret_cast<const int4 *>(coords), n,
                     *g, cnt, sid, vcell, hdr, (uint32_t)(dc_sid_words(g) * 4), (int32_t *)nullptr);
  return check_launch("link_dc_index_ids");
}

namespace link {
// the insert of link_dc_index_ids + occupancy statistics (behind link_dc_index_probe, dense.hip)
int dc_index_ids_stats(const link_dc_buffers_t *b, const link_dc_grid_t *g, int64_t n, int32_t *stats, hipStream_t st) {
  if (!b->coords || !b->cnt || !b->sid || !b->vcell || !b->hdr) return LINK_ERR_ARG;
  if (g->k < 1 || dc_sid_words(g) * 4 >= (1LL << 32) || n > (1LL << DC_ID_BITS)) return LINK_ERR_ARG;
  int64_t wgs = (n + 255) / 256;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index_ids<true>, dim3((unsigned)wgs), dim3(256), 0, st, reinterpret_cast<const int4 *>(b->coords), n, *g,
                     b->cnt, b->sid, b->vcell, b->hdr, (uint32_t)(dc_sid_words(g) * 4), stats);
  return check_launch("link_dc_index_probe");
}
}  // namespace link

namespace dcio_f16 { int run_tiles_modsum(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t, bool, hipStream_t); }
namespace dcio_bf16 { int run_tiles_modsum(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t, bool, hipStream_t); }

namespace link {
// tile form behind link_dc_premix_modsum (dense_fused.hip validated the arguments)
int dc_tiles_modsum(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d, int64_t n, bool warm, hipStream_t st) {
  if (!b->sid || !b->coords || n > (1LL << DC_ID_BITS) || dc_sid_words(g) * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  switch (b->io_dtype) {
    case 1: return dcio_f16::run_tiles_modsum(b, *g, *d, n, warm, st);
    case 2: return dcio_bf16::run_tiles_modsum(b, *g, *d, n, warm, st);
    default: return dcio_f32::run_tiles_modsum(b, *g, *d, n, warm, st);
  }
}
}  // namespace link
