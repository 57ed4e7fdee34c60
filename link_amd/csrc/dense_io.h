// link_amd/csrc/dense_io.h -- feature-row I/O and matrix-core helpers of the fused dense-cell kernels, compiled once per
// feature I/O type: the including translation unit defines DC_IO (0 fp32, 1 fp16, 2 bf16) and DC_IO_NS.  Feature rows are
// read (feats) and written (out) in that type; everything in between -- the MFMA contraction, theta, the block table,
// LayerNorm statistics -- is fp32 (the reference's AMP contract: custom_fwd(cast_inputs=torch.half) on the voxelize /
// devoxelize ops, torchsparse/nn/functional/voxelize.py:13, devoxelize.py:54, with fp32 accumulation; SURVEY.md 8b "AMP").
#pragma once
#include <type_traits>

#include "dense_common.h"

namespace DC_IO_NS {
using namespace link;

constexpr int IO = DC_IO;
constexpr int IO_BYTES = IO == 0 ? 4 : 2;             // bytes per feature element at the kernel boundary

typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
typedef unsigned short us4_t __attribute__((ext_vector_type(4)));
typedef int v2i_t __attribute__((ext_vector_type(2)));

// four consecutive channels starting at element index e (a multiple of 4)
__device__ __forceinline__ float4 io_ld4(const void *base, int64_t e) {
  if constexpr (IO == 0) {
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + e);
  } else if constexpr (IO == 1) {
    const h4_t h = *reinterpret_cast<const h4_t *>(reinterpret_cast<const _Float16 *>(base) + e);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  } else {
    const us4_t u = *reinterpret_cast<const us4_t *>(reinterpret_cast<const unsigned short *>(base) + e);
    return make_float4(__uint_as_float((unsigned)u.x << 16), __uint_as_float((unsigned)u.y << 16),
                       __uint_as_float((unsigned)u.z << 16), __uint_as_float((unsigned)u.w << 16));
  }
}
// the same through a buffer descriptor: 32-bit byte offset `row_off` of the row + a compile-time element offset (an
// immediate of the instruction): one address register for all the pieces of a row
template <int ELEM>
__device__ __forceinline__ float4 io_ldb4(__amdgpu_buffer_rsrc_t r, uint32_t row_off) {
  if constexpr (IO == 0) {
    const v4i_t x = __builtin_amdgcn_raw_buffer_load_b128(r, row_off + (uint32_t)(ELEM * 4), 0, 0);
    return make_float4(__int_as_float(x.x), __int_as_float(x.y), __int_as_float(x.z), __int_as_float(x.w));
  } else {
    const v2i_t x = __builtin_amdgcn_raw_buffer_load_b64(r, row_off + (uint32_t)(ELEM * 2), 0, 0);
    if constexpr (IO == 1) {
      const h4_t h = __builtin_bit_cast(h4_t, x);
      return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
    } else {
      return make_float4(__uint_as_float((unsigned)x.x << 16), __uint_as_float((unsigned)x.x & 0xFFFF0000u),
                         __uint_as_float((unsigned)x.y << 16), __uint_as_float((unsigned)x.y & 0xFFFF0000u));
    }
  }
}
__device__ __forceinline__ unsigned bf16_rne(float f) {
  const unsigned u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (u >> 16) | 0x40u;      // NaN stays NaN
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// four channels as 16-bit elements of the boundary type (IO = 1, 2)
__device__ __forceinline__ v2i_t io_pack4(float4 v) {
  v2i_t x;
  if constexpr (IO == 1) {
    const h4_t h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    x = __builtin_bit_cast(v2i_t, h);
  } else {
    x.x = (int)(bf16_rne(v.x) | (bf16_rne(v.y) << 16));
    x.y = (int)(bf16_rne(v.z) | (bf16_rne(v.w) << 16));
  }
  return x;
}
// store four channels at BYTE offset of the fp32 layout / 4 * IO_BYTES, i.e. callers pass the element offset
__device__ __forceinline__ void io_st4(__amdgpu_buffer_rsrc_t r, uint32_t elem_off, bool valid, float4 v) {
  if constexpr (IO == 0) {
    st16(r, valid ? elem_off * 4u : DC_OOB, v);
  } else {
    __builtin_amdgcn_raw_buffer_store_b64(io_pack4(v), r, valid ? elem_off * 2u : DC_OOB, 0, DC_ST_AUX);
  }
}
// the same through a 64-bit element index into `base` (rows beyond the 4 GiB a descriptor spans)
__device__ __forceinline__ void io_st4_ptr(void *base, int64_t elem, float4 v) {
  if constexpr (IO == 0) *reinterpret_cast<float4 *>(reinterpret_cast<float *>(base) + elem) = v;
  else *reinterpret_cast<v2i_t *>(reinterpret_cast<char *>(base) + elem * 2) = io_pack4(v);
}

// ---------------------------------------------------------------------------------------------
// pre_mix + LayerNorm + modulate + per-cell sum
// ---------------------------------------------------------------------------------------------
// x = hi + lo, hi = fp16(x) (round to nearest), lo = fp16(x - hi): four values -> two packed operands
typedef _Float16 dc_h4v __attribute__((ext_vector_type(4)));
#ifndef DC_SPLIT_ASM
#define DC_SPLIT_ASM 1   /* round 5: the split as THREE instructions per pair of values -- v_cvt_pk_f16_f32 (both hi), then
                            v_fma_mixlo_f16 / v_fma_mixhi_f16 forming lo = fp16(x - hi) straight from the packed hi (the mix
                            instructions read an fp16 half as an fp32 operand: (-1) * hi + x is exact, ONE rounding to fp16 --
                            the same bits as fp16(fp32(x - hi)), since x - hi is exactly representable).  hipcc's own code for the
                            C++ below is 52 VALU instructions per 16 values (unpacking conversions, scalar subtractions, 8
                            conversions done twice); this is 24.  0 = the C++ form */
#endif
__device__ __forceinline__ void dc_split2(float x0, float x1, uint32_t &hi, uint32_t &lo) {
  asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
      "v_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(hi), "=&v"(lo)
      : "v"(x0), "v"(x1));
}
__device__ __forceinline__ void dc_split4(const float4 &v, uint2 &hi, uint2 &lo) {
#if DC_SPLIT_ASM
  dc_split2(v.x, v.y, hi.x, lo.x);
  dc_split2(v.z, v.w, hi.y, lo.y);
#else
  const dc_h4v h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  const dc_h4v l = {(_Float16)(v.x - (float)h.x), (_Float16)(v.y - (float)h.y), (_Float16)(v.z - (float)h.z), (_Float16)(v.w - (float)h.w)};
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
#endif
}
__device__ __forceinline__ floatx4 dc_mfma_f16(uint2 a, uint2 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(dc_h4v, a), __builtin_bit_cast(dc_h4v, b), c, 0, 0, 0);
}
// two 16-channel blocks in one instruction (gfx950: v_mfma_f32_16x16x32_f16, K = 32 in the passes of K = 16): the
// instruction's k = 8g + j is mapped to channel 4g + j of the first block for j < 4 and of the second for j >= 4 --
// the same for both operands, so any such permutation of k is correct
typedef _Float16 dc_h8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ floatx4 dc_mfma_f16x2(uint2 a0, uint2 a1, uint2 b0, uint2 b1, floatx4 c) {
  const uint4 a = make_uint4(a0.x, a0.y, a1.x, a1.y), b = make_uint4(b0.x, b0.y, b1.x, b1.y);
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(dc_h8v, a), __builtin_bit_cast(dc_h8v, b), c, 0, 0, 0);
}

}  // namespace DC_IO_NS
