// link_amd/csrc/elk_tiles.hip -- tile form of R_core on the general layout (elk_tiles_impl.h): C ABI; the kernels for fp32
// feature rows (fp16 / bf16 rows at the kernel boundary: elk_tiles_f16.hip, elk_tiles_bf16.hip).
#define DC_IO 0
#define DC_IO_NS elkt_f32
#include "elk_tiles_impl.h"
#include "elk_tiles_dispatch.h"

using namespace link;
using namespace elkt_f32;

#define ELKT_DECL(NS)                                                                                                              \
  namespace NS {                                                                                                                   \
  int run_premix(const void *, const int32_t *, const int32_t *, const int32_t *, const int32_t *, const float *, const float *,   \
                 const float *, const float *, const float *, const link_elk_desc_t &, int64_t, int64_t, float *, int64_t, float *, \
                 hipStream_t);                                                                                                     \
  int run_gather(const float *, const float *, const int32_t *, const int32_t *, const int32_t *, const int32_t *,                 \
                 const link_grid_t &, const int32_t *, const float *, const float *, const float *, const float *,                 \
                 const link_elk_desc_t &, int64_t, int64_t, void *, hipStream_t);                                                  \
  }
ELKT_DECL(elkt_f16)
ELKT_DECL(elkt_bf16)
#undef ELKT_DECL

#ifdef ELK_T_DBG
extern "C" int link_elk_tiles_debug_read(void *host_dst, int64_t bytes, int which) {      // profiling builds only
  const hipError_t e = which ? hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(elkt_f32::elk_g_dbg), (size_t)bytes)
                             : hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(elkt_f32::elk_t_dbg), (size_t)bytes);
  return e == hipSuccess ? LINK_OK : LINK_ERR_LAUNCH;
}
#endif

static bool tiles_desc_ok(const link_elk_desc_t *d) {
  return d && dc_width_ok(d->c) && (d->r == 2 || d->r == 3) && d->cg >= 1 && d->c % d->cg == 0 &&
         (d->op == LINK_OP_COS || d->op == LINK_OP_SIN || d->op == LINK_OP_COSX);
}

extern "C" int64_t link_elk_tiles_table_bytes(const link_elk_desc_t *desc, int64_t n, int64_t m_cap) {
  if (!tiles_desc_ok(desc) || n < 0 || m_cap < 0) return -1;
  const int rs = (desc->op == LINK_OP_COSX ? 3 : 2) * desc->c;
  // two partial rows per wave, at one 16-voxel tile per wave (the smallest span): monotone in n, so a table sized for a
  // plan's capacity serves every smaller frame
  return elk_t_part_off(m_cap, rs) + ((n > 0 ? n : 1) + 63) / 64 * 4 * 2 * (int64_t)rs * 4;
}

extern "C" int link_elk_premix_modsum_tiles_io(const void *feats, int32_t io_dtype, const int32_t *vox_sorted, const int32_t *pos_blk,
                                               const int32_t *blk_start, const int32_t *hdr, const float *w_pre,
                                               const float *pre_ln_w, const float *pre_ln_b, const float *w_pos,
                                               const float *alpha, const link_elk_desc_t *desc, int64_t n, int64_t m_cap,
                                               float *S_, int64_t s_bytes, float *fin, void *stream) {
  if (n < 0 || !tiles_desc_ok(desc) || io_dtype < LINK_IO_F32 || io_dtype > LINK_IO_BF16) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!feats || !vox_sorted || !pos_blk || !blk_start || !hdr || !w_pre || !pre_ln_w || !pre_ln_b || !w_pos || !S_) return LINK_ERR_ARG;
  if (desc->op == LINK_OP_COSX && !fin) return LINK_ERR_ARG;
  const int64_t need = link_elk_tiles_table_bytes(desc, n, m_cap);
  // 32-bit byte offsets into the table, the rows and the records
  if (m_cap < 1 || s_bytes < need || need >= (1LL << 32) || n * desc->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
#define LINK_T_CALL(NS) NS::run_premix(feats, vox_sorted, pos_blk, blk_start, hdr, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, *desc, n, m_cap, S_, need, fin, st)
  switch (io_dtype) {
    case LINK_IO_F16: return LINK_T_CALL(elkt_f16);
    case LINK_IO_BF16: return LINK_T_CALL(elkt_bf16);
    default: return LINK_T_CALL(elkt_f32);
  }
#undef LINK_T_CALL
}

extern "C" int link_elk_premix_modsum_tiles(const float *feats, const int32_t *vox_sorted, const int32_t *pos_blk,
                                            const int32_t *blk_start, const int32_t *hdr, const float *w_pre,
                                            const float *pre_ln_w, const float *pre_ln_b, const float *w_pos,
                                            const float *alpha, const link_elk_desc_t *desc, int64_t n, int64_t m_cap,
                                            float *S_, int64_t s_bytes, float *fin, void *stream) {
  return link_elk_premix_modsum_tiles_io(feats, LINK_IO_F32, vox_sorted, pos_blk, blk_start, hdr, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha,
                                         desc, n, m_cap, S_, s_bytes, fin, stream);
}

extern "C" int link_elk_gather_demod_tiles_io(const float *S_, const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk,
                                              const int32_t *blk_coords, const int32_t *cell_blk, const link_grid_t *grid,
                                              const int32_t *hdr, const float *w_pos, const float *alpha, const float *ln_w,
                                              const float *ln_b, const link_elk_desc_t *desc, int64_t n, int64_t m_cap,
                                              void *out, int32_t io_dtype, void *stream) {
  if (n < 0 || !tiles_desc_ok(desc) || !grid || io_dtype < LINK_IO_F32 || io_dtype > LINK_IO_BF16) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!S_ || !vox_sorted || !pos_blk || !blk_coords || !cell_blk || !hdr || !w_pos || !ln_w || !ln_b || !out || m_cap < 1) return LINK_ERR_ARG;
  if (desc->op == LINK_OP_COSX && !fin) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
#define LINK_G_CALL(NS) NS::run_gather(S_, fin, vox_sorted, pos_blk, blk_coords, cell_blk, *grid, hdr, w_pos, alpha, ln_w, ln_b, *desc, n, m_cap, out, st)
  switch (io_dtype) {
    case LINK_IO_F16: return LINK_G_CALL(elkt_f16);
    case LINK_IO_BF16: return LINK_G_CALL(elkt_bf16);
    default: return LINK_G_CALL(elkt_f32);
  }
#undef LINK_G_CALL
}

extern "C" int link_elk_gather_demod_tiles(const float *S_, const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk,
                                           const int32_t *blk_coords, const int32_t *cell_blk, const link_grid_t *grid,
                                           const int32_t *hdr, const float *w_pos, const float *alpha, const float *ln_w,
                                           const float *ln_b, const link_elk_desc_t *desc, int64_t n, int64_t m_cap,
                                           float *out, void *stream) {
  return link_elk_gather_demod_tiles_io(S_, fin, vox_sorted, pos_blk, blk_coords, cell_blk, grid, hdr, w_pos, alpha, ln_w, ln_b, desc, n,
                                        m_cap, out, LINK_IO_F32, stream);
}
