// link_amd/csrc/dense_k2_cfg.h -- LDS layout of the fused box sum + de-modulate kernels (K2 of the dense-cell layout, C = 64) and the
// inline-asm LDS accessors their bodies use.  Included INSIDE an IO namespace by dense_fused_impl.h (all K2 forms) and by
// dense_batch.hip (round 6: the persistent K2-role kernel runs dc_k2q_body of dense_gather_quad_impl.h over (frame, tile) items).
#pragma once

template <int OP, int R>
struct dc_k2_cfg {
  static constexpr int C = 64, LPR = 16, NG = 16;
  static constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  using G = dc_gather_cfg<C, P, R>;
  static constexpr int RB = P * C * 4;
  static constexpr int REC_OFF = G::PLANE_BYTES + G::CNT_BYTES;
  static constexpr int REC_BYTES = NG * DC_INL * 16;
  static constexpr int BUF_BYTES = REC_OFF + REC_BYTES;
  static constexpr int ABUF_OFF = 3 * BUF_BYTES;
  static constexpr int NCNT_OFF = ABUF_OFF + NG * RB;
  static constexpr int LDS_BYTES = NCNT_OFF + NG * 4;
  // producer / consumer form: two A images (ABUF_OFF, 2 * NG * RB), two count images, a ring of 4 record images
  // (plane-ring slots without the record image: 2 workgroups of 79 KB per CU)
  static constexpr int SPLIT_PLANE = G::NPC * 16;      // no padding pass: surplus DMA lanes re-load pass 0's pieces
  static constexpr int SPLIT_BUF_BYTES = SPLIT_PLANE + G::CNT_BYTES;
  static constexpr int SPLIT_ABUF_OFF = 3 * SPLIT_BUF_BYTES;
  static constexpr int SPLIT_NCNT_OFF = SPLIT_ABUF_OFF + 2 * NG * RB;
  static constexpr int SPLIT_REC_OFF = SPLIT_NCNT_OFF + 2 * NG * 4;
  static constexpr int SPLIT_LDS_BYTES = SPLIT_REC_OFF + 4 * REC_BYTES;
  static constexpr bool SPLIT_FITS = 2 * SPLIT_LDS_BYTES <= 160 * 1024;      // two workgroups per CU (not cos_x: 3-part rows)
  static constexpr int NI = G::PASSES + 2;            // DMA instructions per plane and wave
  static_assert(G::NG == NG && G::TX * G::TY == NG, "16 columns");
};

__device__ __forceinline__ void lds_rd2_b128(uint32_t a0, uint32_t a1, v4f_t &x0, v4f_t &x1) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(a0), "v"(a1) : "memory");
}
// six b128 reads in flight, ONE wait (the pair's two records and four A-row pieces: three round trips -> one)
__device__ __forceinline__ void lds_rd6_b128(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                             v4f_t &x0, v4f_t &x1, v4f_t &x2, v4f_t &x3, v4f_t &x4, v4f_t &x5) {
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
               "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5)
               : "memory");
}
__device__ __forceinline__ int lds_rd_b32(uint32_t a) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wr_b128(uint32_t a, float4 v) {
  const v4f_t x = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(x) : "memory");
}
__device__ __forceinline__ void lds_wr_b32(uint32_t a, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}
