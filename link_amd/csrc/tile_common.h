// link_amd/csrc/tile_common.h -- what the tile-form kernels share (dense_tiles_impl.h: dense-cell layout; elk_tiles_impl.h:
// general layout): the fp16 hi | lo image of the pre_mix weight in LDS, the matrix-core contraction of one 16-voxel tile,
// the segmented-scan step along a DPP row, the row-swap butterfly over the four lane groups of a voxel.  Compiled per
// feature I/O type like dense_io.h (DC_IO / DC_IO_NS).
#pragma once
#include "dense_io.h"

namespace DC_IO_NS {
using namespace link;

// LDS image of the parameters: W as fp16, row co = [hi(C) | lo(C) | pad] (conflict-free ds_read_b64), then the LayerNorm
// weight | bias, then the theta weights per channel (w0 | w1 | w2 | alpha)
template <int C>
struct dc_wimg {
  static constexpr int LDH = 2 * C + 8;
  static constexpr int WIMG_BYTES = C * LDH * 2;
  static constexpr int W_BYTES = WIMG_BYTES + (2 * C + 4 * C) * 4;
};

#define DC_ID_BITS 26
#define DC_ID_MASK ((1u << DC_ID_BITS) - 1u)

// one step of the segmented scan over a DPP row for four values: v += mask * v[lane - 2^k] (mask = 1 where that lane
// belongs to the same cell).  The s_nop covers the VALU-write -> DPP-read hazard, which hipcc cannot see inside asm.
#define DC_SCAN4(CTRL, m, a, b, c, d)                                                        \
  asm volatile("s_nop 1\n\t"                                                                  \
               "v_fmac_f32_dpp %0, %0, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
               "v_fmac_f32_dpp %1, %1, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
               "v_fmac_f32_dpp %2, %2, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t" \
               "v_fmac_f32_dpp %3, %3, %4 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1"     \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                           \
               : "v"(m))

// v + v[lane ^ 16] + v[lane ^ 32] + v[lane ^ 48]: the four lane groups of a voxel, through the gfx950 row swaps (VALU; a
// __shfl_xor is an LDS round trip each).  v_permlane32_swap exchanges the upper half of one operand with the lower half
// of the other, v_permlane16_swap the odd rows of one with the even rows of the other: with both operands = v, the two
// results hold {own, partner} in some order, so their sum is the butterfly step.
__device__ __forceinline__ float dc_sum_groups(float v) {
  // inline asm: hipcc (ROCm 7.2) returns the first result of __builtin_amdgcn_permlane{16,32}_swap for both elements;
  // the instruction itself rewrites BOTH of its operands.  s_nop: VALU write -> permlane read hazard
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a += b;
  b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

template <int CTRL>
__device__ __forceinline__ int dc_dpp_i(int v, int oob) {      // lanes without a source keep `oob`
  return __builtin_amdgcn_update_dpp(oob, v, CTRL, 0xF, 0xF, false);
}

// Stage the image with NT threads (NT >= 2C): all loads, ONE wait, then the writes.  Returns whether THIS thread saw a
// weight outside the fp16 split's range (the caller ORs over the workgroup with its barrier: then the fp32 instruction runs).
// EXACT (round 5, cos_x: elk_common.h LINK_COSX_EXACT): the image holds W itself in fp32, row co = C floats + 4 of padding -- the
// same bytes per row as the hi | lo image, so every LDS layout behind it is unchanged -- and dc_premix_tile<C, true> runs the exact
// fp32 matrix instruction from it (the "outside the fp16 range" path reads W from global memory per tile: fine for a rarity, not
// for the form every cos_x block takes).
template <int C, int NT, bool EXACT = false>
__device__ __forceinline__ bool dc_stage_weights(char *smem_raw, const float *__restrict__ w_pre, const float *__restrict__ ln_w,
                                                 const float *__restrict__ ln_b, const float *__restrict__ w_pos,
                                                 const float *__restrict__ alpha, int cg, int tid) {
  using WI = dc_wimg<C>;
  static_assert(NT >= 2 * C, "staging assumes one thread per LayerNorm parameter");
  float *ln_lds_ = reinterpret_cast<float *>(smem_raw + WI::WIMG_BYTES);
  float *pw_lds_ = ln_lds_ + 2 * C;
  bool w_big = false;
  {                                                    // stage W (fp16 hi | lo image) and the parameters: all loads, ONE wait, then the writes
    constexpr int NF4 = C * C / 4;
    constexpr int NVW = (NF4 + NT - 1) / NT;
    float4 wv[NVW];
#pragma unroll
    for (int i = 0; i < NVW; i++) {
      const int e = (i * NT + tid) * 4;
      wv[i] = *reinterpret_cast<const float4 *>(&w_pre[(NF4 % NT == 0 || e < C * C) ? e : 0]);
    }
#pragma unroll
    for (int i = 0; i < NVW; i++) {
      int e = (i * NT + tid) * 4;
      if (NF4 % NT != 0 && e >= C * C) e = 0;         // C = 16: surplus lanes rewrite piece 0 with its own value
      const int r = e / C, col = e - r * C;
      const float4 wq = (NF4 % NT == 0 || (i * NT + tid) * 4 < C * C) ? wv[i] : *reinterpret_cast<const float4 *>(&w_pre[0]);
      if constexpr (EXACT) {
        static_assert(WI::LDH * 2 == (C + 4) * 4, "fp32 rows of C + 4 floats fill the hi | lo image's rows exactly");
        *reinterpret_cast<float4 *>(reinterpret_cast<float *>(smem_raw) + r * (C + 4) + col) = wq;
      } else {
        uint2 hi, lo;
        dc_split4(wq, hi, lo);
        unsigned short *wh = reinterpret_cast<unsigned short *>(smem_raw);
        *reinterpret_cast<uint2 *>(&wh[r * WI::LDH + col]) = hi;
        *reinterpret_cast<uint2 *>(&wh[r * WI::LDH + C + col]) = lo;
        w_big |= !(fmaxf(fmaxf(fabsf(wq.x), fabsf(wq.y)), fmaxf(fabsf(wq.z), fabsf(wq.w))) < 32768.0f);
      }
    }
    if (tid < C) ln_lds_[tid] = ln_w[tid];
    else if (tid < 2 * C) ln_lds_[tid] = ln_b[tid - C];
    if (tid < C) {                                     // theta weights of channel tid (channel ch uses theta[ch % cg])
      const int tc = tid % cg;
      pw_lds_[tid] = w_pos[3 * tc + 0]; pw_lds_[C + tid] = w_pos[3 * tc + 1]; pw_lds_[2 * C + tid] = w_pos[3 * tc + 2];
      pw_lds_[3 * C + tid] = alpha ? alpha[tc] : 1.0f;
    }
  }
  return w_big;
}

// ---- pre_mix contraction of one tile: D[co][voxel] = sum_ci W[co][ci] x[voxel][ci] as an fp16 hi/lo split of both
// operands on v_mfma_f32_16x16x32_f16 (exact products, fp32 accumulation, the dropped lo*lo term is 2^-22 relative;
// fp16 rows have lo = 0); |x| or |w| >= 2^15 takes the fp32 instruction with W from global memory, wave-uniform ----
template <int C, bool EXACT = false>
__device__ __forceinline__ void dc_premix_tile(const unsigned short *wh, const float *__restrict__ w_pre, bool w_big, int li, int gq,
                                               const float4 (&ff)[C / 16], floatx4 (&cc)[C / 16]) {
  using WI = dc_wimg<C>;
  constexpr int T = C / 16;
#pragma unroll
  for (int tp = 0; tp < T; tp++) cc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
  if constexpr (EXACT) {                                // W in fp32 in LDS (dc_stage_weights<C, NT, true>): the exact matrix instruction
    const float *wf = reinterpret_cast<const float *>(wh);
#pragma unroll
    for (int tt = 0; tt < T; tt++) {
      float4 a[T];
#pragma unroll
      for (int tp = 0; tp < T; tp++) a[tp] = *reinterpret_cast<const float4 *>(&wf[(16 * tp + li) * (C + 4) + 16 * tt + 4 * gq]);
#pragma unroll
      for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].x, ff[tt].x, cc[tp], 0, 0, 0);
#pragma unroll
      for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].y, ff[tt].y, cc[tp], 0, 0, 0);
#pragma unroll
      for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].z, ff[tt].z, cc[tp], 0, 0, 0);
#pragma unroll
      for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].w, ff[tt].w, cc[tp], 0, 0, 0);
    }
    return;
  }
  uint2 bh[T], bl[T];
  float mx = 0.f;
#pragma unroll
  for (int tt = 0; tt < T; tt++) {
    dc_split4(ff[tt], bh[tt], bl[tt]);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(ff[tt].x), fabsf(ff[tt].y)), fmaxf(fabsf(ff[tt].z), fabsf(ff[tt].w))));
  }
  if (__builtin_expect(!(w_big || __any(!(mx < 32768.0f))), 1)) {
    if constexpr (T % 2 == 0) {
      // two output blocks x two input-block pairs at a time: 8 operand pieces (16 registers) in flight instead of 4T
#pragma unroll
      for (int tq = 0; tq < T; tq += 2)
#pragma unroll
        for (int tt = 0; tt < T; tt += 2) {
          uint2 ah[2][2], al[2][2];
#pragma unroll
          for (int h = 0; h < 2; h++)
#pragma unroll
            for (int u = 0; u < 2; u++) {
              ah[h][u] = *reinterpret_cast<const uint2 *>(&wh[(16 * (tq + u) + li) * WI::LDH + 16 * (tt + h) + 4 * gq]);
              al[h][u] = *reinterpret_cast<const uint2 *>(&wh[(16 * (tq + u) + li) * WI::LDH + C + 16 * (tt + h) + 4 * gq]);
            }
#pragma unroll
          for (int u = 0; u < 2; u++) cc[tq + u] = dc_mfma_f16x2(al[0][u], al[1][u], bh[tt], bh[tt + 1], cc[tq + u]);
          if constexpr (IO != 1) {
#pragma unroll
            for (int u = 0; u < 2; u++) cc[tq + u] = dc_mfma_f16x2(ah[0][u], ah[1][u], bl[tt], bl[tt + 1], cc[tq + u]);
          }
#pragma unroll
          for (int u = 0; u < 2; u++) cc[tq + u] = dc_mfma_f16x2(ah[0][u], ah[1][u], bh[tt], bh[tt + 1], cc[tq + u]);
        }
    } else {
#pragma unroll
      for (int tt = 0; tt < T; tt++) {
        uint2 ah[T], al[T];
#pragma unroll
        for (int tp = 0; tp < T; tp++) {
          ah[tp] = *reinterpret_cast<const uint2 *>(&wh[(16 * tp + li) * WI::LDH + 16 * tt + 4 * gq]);
          al[tp] = *reinterpret_cast<const uint2 *>(&wh[(16 * tp + li) * WI::LDH + C + 16 * tt + 4 * gq]);
        }
#pragma unroll
        for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16(al[tp], bh[tt], cc[tp]);
        if constexpr (IO != 1) {
#pragma unroll
          for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16(ah[tp], bl[tt], cc[tp]);
        }
#pragma unroll
        for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16(ah[tp], bh[tt], cc[tp]);
      }
    }
    return;
  }
#pragma unroll
  for (int tt = 0; tt < T; tt++) {
    float4 a[T];
#pragma unroll
    for (int tp = 0; tp < T; tp++) a[tp] = *reinterpret_cast<const float4 *>(&w_pre[(16 * tp + li) * C + 16 * tt + 4 * gq]);
#pragma unroll
    for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].x, ff[tt].x, cc[tp], 0, 0, 0);
#pragma unroll
    for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].y, ff[tt].y, cc[tp], 0, 0, 0);
#pragma unroll
    for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].z, ff[tt].z, cc[tp], 0, 0, 0);
#pragma unroll
    for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].w, ff[tt].w, cc[tp], 0, 0, 0);
  }
}

}  // namespace DC_IO_NS
