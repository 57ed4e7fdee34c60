// link_amd/csrc/elk_tiles_dispatch.h -- launch + template dispatch of the tile-form kernels (elk_tiles_impl.h) for the I/O
// type of the including translation unit (DC_IO / DC_IO_NS: elk_tiles.hip fp32, elk_tiles_f16.hip, elk_tiles_bf16.hip).
// Defines DC_IO_NS::run_premix / run_gather, which the C ABI in elk_tiles.hip calls after validating the arguments.
namespace DC_IO_NS {
using namespace link;

// Sorted positions per workgroup of k_elk_tiles: one 16-voxel tile per wave on small frames (a tile is ~4 us of one wave's
// dependent work, so a frame's waves should all be resident together), two from 32k voxels (fewer prologues: 20.4 against
// 24.1 us at 59k voxels, C = 64; 11.7 against 15.1 at 3k), more only beyond 4096 workgroups.
#ifdef ELK_T_SPAN
static int tiles_span(int64_t) { return ELK_T_SPAN; }
#else
static int tiles_span(int64_t n) {
  if (n <= 32768) return 64;
  const int64_t k = (n + 128 * 4096 - 1) / (128 * 4096);
  return 128 * (int)(k > 1 ? k : 1);
}
#endif

static int64_t tiles_wgs(int64_t n) { return (n + tiles_span(n) - 1) / tiles_span(n); }


template <int C, int OP, int NB>
static int launch_tiles(const void *feats, const int32_t *vox_sorted, const int32_t *pos_blk, const int32_t *blk_start,
                        const int32_t *hdr, const float *w_pre, const float *ln_w, const float *ln_b, const float *w_pos,
                        const float *alpha, const link_elk_desc_t &d, int64_t n, int64_t m_cap, float *S_, int64_t s_bytes,
                        float *fin, hipStream_t st) {
  using K = elk_t_cfg<C, OP>;
  const int lds = K::LDS_BYTES;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_elk_tiles<C, OP, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k_elk_tiles<C, OP, NB>), dim3((unsigned)tiles_wgs(n)), dim3(64 * K::NW), lds, st, feats,
                     reinterpret_cast<const int4 *>(vox_sorted), pos_blk, blk_start, hdr, w_pre, ln_w, ln_b, w_pos, alpha, d.cg,
                     d.coord_div, d.eps, n, m_cap, tiles_span(n), S_, (uint32_t)s_bytes, (uint32_t)elk_t_part_off(m_cap, K::P * C), fin);
  return check_launch("link_elk_premix_modsum_tiles");
}

template <int C, int OP, typename... A>
static int tiles_nb(const link_elk_desc_t &d, A... a) {
  constexpr int T = C / 16;
  int nb = (d.cg % 16 == 0) ? d.cg / 16 : T;
  if (nb > T) nb = T;
  if (T >= 2 && nb == T / 2) return launch_tiles<C, OP, (T >= 2 ? T / 2 : 1)>(a...);
  if (T >= 4 && nb == T / 4) return launch_tiles<C, OP, (T >= 4 ? T / 4 : 1)>(a...);
  return launch_tiles<C, OP, T>(a...);                  // any other grouping: every 16-channel block evaluates its own theta
}

template <int C, typename... A>
static int tiles_op(const link_elk_desc_t &d, A... a) {
  switch (d.op) {
    case LINK_OP_COS: return tiles_nb<C, LINK_OP_COS>(d, a...);
    case LINK_OP_SIN: return tiles_nb<C, LINK_OP_SIN>(d, a...);
    default: return tiles_nb<C, LINK_OP_COSX>(d, a...);
  }
}

int run_premix(const void *feats, const int32_t *vox_sorted, const int32_t *pos_blk, const int32_t *blk_start, const int32_t *hdr,
               const float *w_pre, const float *pre_ln_w, const float *pre_ln_b, const float *w_pos, const float *alpha,
               const link_elk_desc_t &d, int64_t n, int64_t m_cap, float *S_, int64_t need, float *fin, hipStream_t st) {
#define LINK_T_ARGS d, feats, vox_sorted, pos_blk, blk_start, hdr, w_pre, pre_ln_w, pre_ln_b, w_pos, alpha, d, n, m_cap, S_, need, fin, st
  switch (d.c) {
    case 16: return tiles_op<16>(LINK_T_ARGS);
    case 32: return tiles_op<32>(LINK_T_ARGS);
    case 64: return tiles_op<64>(LINK_T_ARGS);
    default: return tiles_op<128>(LINK_T_ARGS);
  }
#undef LINK_T_ARGS
}

template <int C, int OP, int R>
static int launch_gather(const float *S_, const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk,
                         const int32_t *blk_coords, const int32_t *cell_blk, const link_grid_t &g, const int32_t *hdr,
                         const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                         const link_elk_desc_t &d, int64_t n, int64_t m_cap, void *out, hipStream_t st) {
  using K = elk_g_cfg<C, OP, R>;
  // a multiple of 8: the kernel deals contiguous eighths of the tiles to the XCDs (workgroup w runs on XCD w % 8)
  const int64_t wgs = ((n + (int64_t)K::WP * K::NW - 1) / ((int64_t)K::WP * K::NW) + 7) & ~(int64_t)7;
  if (K::LDS_BYTES > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_elk_gather_tiles<C, OP, R>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              K::LDS_BYTES);
  hipLaunchKernelGGL((k_elk_gather_tiles<C, OP, R>), dim3((unsigned)wgs), dim3(64 * K::NW), K::LDS_BYTES, st, S_, fin,
                     reinterpret_cast<const int4 *>(vox_sorted), pos_blk, reinterpret_cast<const int4 *>(blk_coords), cell_blk, g, hdr,
                     w_pos, alpha, ln_w, ln_b, d.cg, d.coord_div, d.eps, m_cap, out);
  return check_launch("link_elk_gather_demod_tiles");
}

template <int C, typename... A>
static int gather_op_r(const link_elk_desc_t &d, A... a) {
#define LINK_G_CASE(OPV)                                                        \
  return d.r == 2 ? launch_gather<C, OPV, 2>(a...) : launch_gather<C, OPV, 3>(a...)
  switch (d.op) {
    case LINK_OP_COS: LINK_G_CASE(LINK_OP_COS);
    case LINK_OP_SIN: LINK_G_CASE(LINK_OP_SIN);
    default: LINK_G_CASE(LINK_OP_COSX);
  }
#undef LINK_G_CASE
}

int run_gather(const float *S_, const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk, const int32_t *blk_coords,
               const int32_t *cell_blk, const link_grid_t &grid, const int32_t *hdr, const float *w_pos, const float *alpha,
               const float *ln_w, const float *ln_b, const link_elk_desc_t &d, int64_t n, int64_t m_cap, void *out, hipStream_t st) {
#define LINK_G_ARGS d, S_, fin, vox_sorted, pos_blk, blk_coords, cell_blk, grid, hdr, w_pos, alpha, ln_w, ln_b, d, n, m_cap, out, st
  switch (d.c) {
    case 16: return gather_op_r<16>(LINK_G_ARGS);
    case 32: return gather_op_r<32>(LINK_G_ARGS);
    case 64: return gather_op_r<64>(LINK_G_ARGS);
    default: return gather_op_r<128>(LINK_G_ARGS);
  }
#undef LINK_G_ARGS
}

}  // namespace DC_IO_NS
