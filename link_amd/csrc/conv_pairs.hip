// link_amd/csrc/conv_pairs.hip -- the sparse-frame form of the sparse convolution (row N1).
//
// conv.hip's output-stationary kernel walks all K offsets for every tile of 16 output voxels; that is the right
// shape when a voxel has most of its 27 neighbours.  On sparse frames (cfg2: 1.2 of 27 present, LiDAR sweeps:
// ~6) nearly every (tile, offset) step finds one or two of its 16 voxels with work, and the f32 MFMA issues at
// tile granularity -- 15/16 of the matrix pipe's time multiplies zeros.  The reference's gather-GEMM-scatter
// (convolution_cuda.cu:90-165) is dense by construction because it runs over the *pair list* of each offset.
// This file keeps that property without its 27 x 3 launches, its scatter atomics or its intermediate copies:
//
//   k_conv_pairs_gemm   rows = pairs (input row j -> contribution row p), grouped by kernel offset and padded
//                       to 128-row granules; a workgroup owns one granule, stages W_k once in LDS and its 8
//                       waves run one dense 16-pair MFMA tile each: contrib[p] = feats[pair_in[p]] . W_k
//                       (v_mfma_f32_16x16x4_f32, exact f32).
//   k_conv_pairs_sum    output-stationary and deterministic: voxel i adds the rows of its CSR list (ascending
//                       offset) in registers, then bias / LayerNorm + add + ReLU epilogue (the block's tail,
//                       linkunet.py:183) and ONE store.  No atomics anywhere.  (Strided / transposed maps.)
//   k_conv_centre_sum   submanifold maps: the centre offset's pairs are the identity, so its GEMM runs on the
//                       output tile itself and the same epilogue finishes the voxel in the MFMA accumulators;
//                       the pair list then holds only the other offsets.
//
// The pair lists are the kernel map (the reference's nbmaps/nbsizes, nn/functional/conv.py:109-122): built once
// per coordinate set on the host side (link_amd/elk.py::_pair_plan) and cached with it.
#include "common.h"

using namespace link;

typedef float floatx4 __attribute__((ext_vector_type(4)));

// Feature rows at the kernel boundary may be fp32, fp16 or bf16 (io = LINK_IO_*; the reference's AMP contract for its
// convolution: custom_fwd(cast_inputs=torch.half), nn/functional/conv.py:18, with fp32 accumulation).  Everything in
// between -- weights, contribution rows, statistics -- is fp32.  The switch is wave-uniform.
typedef _Float16 cp_h4 __attribute__((ext_vector_type(4)));
typedef unsigned short cp_us4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 cp_ld4(const void *base, int64_t e, int io) {
  if (io == LINK_IO_F32) return *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(base) + e);
  if (io == LINK_IO_F16) {
    const cp_h4 h = *reinterpret_cast<const cp_h4 *>(reinterpret_cast<const _Float16 *>(base) + e);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
  }
  const cp_us4 u = *reinterpret_cast<const cp_us4 *>(reinterpret_cast<const unsigned short *>(base) + e);
  return make_float4(__uint_as_float((unsigned)u.x << 16), __uint_as_float((unsigned)u.y << 16),
                     __uint_as_float((unsigned)u.z << 16), __uint_as_float((unsigned)u.w << 16));
}
__device__ __forceinline__ unsigned short cp_bf16(float f) {
  const unsigned u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void cp_st4(void *base, int64_t e, float4 v, int io) {
  if (io == LINK_IO_F32) { *reinterpret_cast<float4 *>(reinterpret_cast<float *>(base) + e) = v; return; }
  if (io == LINK_IO_F16) {
    const cp_h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    *reinterpret_cast<cp_h4 *>(reinterpret_cast<_Float16 *>(base) + e) = h;
    return;
  }
  const cp_us4 u = {cp_bf16(v.x), cp_bf16(v.y), cp_bf16(v.z), cp_bf16(v.w)};
  *reinterpret_cast<cp_us4 *>(reinterpret_cast<unsigned short *>(base) + e) = u;
}

template <int CI, int CO>
__global__ void __launch_bounds__(512) k_conv_pairs_gemm(const void *__restrict__ feats, int io,
                                                         const int32_t *__restrict__ pair_in,
                                                         const int32_t *__restrict__ wg_k,
                                                         const float *__restrict__ w, float *__restrict__ contrib) {
  constexpr int TI = CI / 16, TO = CO / 16;
  constexpr int LD = CO + 4;                 // row stride of W_k in LDS: 4*LD = 16 (mod 32) -> the two 16-lane
                                             // rows of a half-wave read disjoint banks
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int k = wg_k[blockIdx.x];
  if (k < 0) return;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  // the wave's 16 input rows first: their latency hides behind the W_k staging
  const int j0 = pair_in[row0 + li];
  float4 f0[TI];
  {
    const int64_t r0 = (int64_t)(j0 < 0 ? 0 : j0) * CI + 4 * g;
#pragma unroll
    for (int t = 0; t < TI; t++) f0[t] = cp_ld4(feats, r0 + 16 * t, io);
  }
  const float *wk = w + (int64_t)k * CI * CO;
  for (int e = tid * 4; e < CI * CO; e += 512 * 4) {
    const int r = e / CO, col = e - r * CO;
    *reinterpret_cast<float4 *>(&w_lds[r * LD + col]) = *reinterpret_cast<const float4 *>(&wk[e]);
  }
  __syncthreads();
  if (__all(j0 < 0)) return;                           // granule padding
  floatx4 a0[TO];
#pragma unroll
  for (int tp = 0; tp < TO; tp++) a0[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < TI; t++) {
    const float *wr = &w_lds[(16 * t + 4 * g) * LD + li];
#pragma unroll
    for (int tp = 0; tp < TO; tp++) {
      // A operand: W_k^T[co = 16tp + li][ci = 16t + 4g + j]
      a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * tp], f0[t].x, a0[tp], 0, 0, 0);
      a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[LD + 16 * tp], f0[t].y, a0[tp], 0, 0, 0);
      a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[2 * LD + 16 * tp], f0[t].z, a0[tp], 0, 0, 0);
      a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3 * LD + 16 * tp], f0[t].w, a0[tp], 0, 0, 0);
    }
  }
  // D[co = 16tp + 4g + r][pair li]: a lane holds 4 consecutive channels of its pair's row per output tile
  float *o0 = contrib + (row0 + li) * CO + 4 * g;
  if (j0 >= 0) {
#pragma unroll
    for (int tp = 0; tp < TO; tp++)
      *reinterpret_cast<float4 *>(o0 + 16 * tp) = make_float4(a0[tp][0], a0[tp][1], a0[tp][2], a0[tp][3]);
  }
}

// ---- AMP form: 16-bit rows x 16-bit weights on the f16 / bf16 matrix cores (fp32 accumulation) ----------------
// custom_fwd(cast_inputs=torch.half) (nn/functional/conv.py:18) rounds the kernel as well as the rows, so with half
// rows the products are exact in fp32 and v_mfma_f32_16x16x16_{f16,bf16} replaces four v_mfma_f32_16x16x4_f32 at
// 1/8 of their matrix-pipe time.  The weights arrive already rounded and transposed, wt[k][co][ci] (16-bit): a
// lane's A operand -- W_k^T[co = 16tp + li][ci = 16t + 4g .. +3] -- is then one 8-byte LDS read, and the B operand
// is the 8 bytes of the row it loaded (no conversion anywhere).  LDS row stride CI + 8 elements = 4 * odd dwords:
// the 32 lanes of a ds_read_b64 half hit 32 distinct bank pairs.
typedef _Float16 cp_f16x4 __attribute__((ext_vector_type(4)));
typedef short cp_s16x4 __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ floatx4 cp_mfma16(uint2 a, uint2 b, floatx4 c) {
  if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(cp_s16x4, a), __builtin_bit_cast(cp_s16x4, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(cp_f16x4, a), __builtin_bit_cast(cp_f16x4, b), c, 0, 0, 0);
}
template <int CI, int CO>
__device__ __forceinline__ void cp_stage_wt(unsigned short *w_lds, const unsigned short *wk, int tid, int nt = 512) {
  constexpr int LDH = CI + 8;
  for (int e = tid * 8; e < CI * CO; e += nt * 8) {
    const int r = e / CI, col = e - r * CI;
    *reinterpret_cast<uint4 *>(&w_lds[r * LDH + col]) = *reinterpret_cast<const uint4 *>(&wk[e]);
  }
}

// C16: the contribution rows are stored in the row type as well (what the reference's half torch.mm produces,
// convolution_cuda.cu:127-140); their sum in k_conv_centre_sum stays fp32.
template <int CI, int CO, bool BF, bool C16>
__global__ void __launch_bounds__(512) k_conv_pairs_gemm_h(const unsigned short *__restrict__ feats,
                                                           const int32_t *__restrict__ pair_in,
                                                           const int32_t *__restrict__ wg_k,
                                                           const unsigned short *__restrict__ wt,
                                                           void *__restrict__ contrib) {
  constexpr int TI = CI / 16, TO = CO / 16, LDH = CI + 8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short *w_lds = reinterpret_cast<unsigned short *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int k = wg_k[blockIdx.x];
  if (k < 0) return;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  const int j0 = pair_in[row0 + li];
  uint2 f0[TI];
  {
    const unsigned short *fr = feats + (int64_t)(j0 < 0 ? 0 : j0) * CI + 4 * g;
#pragma unroll
    for (int t = 0; t < TI; t++) f0[t] = *reinterpret_cast<const uint2 *>(fr + 16 * t);
  }
  cp_stage_wt<CI, CO>(w_lds, wt + (int64_t)k * CI * CO, tid);
  __syncthreads();
  if (__all(j0 < 0)) return;                           // granule padding
  floatx4 a0[TO];
#pragma unroll
  for (int tp = 0; tp < TO; tp++) a0[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < TI; t++) {
#pragma unroll
    for (int tp = 0; tp < TO; tp++) {
      const uint2 a = *reinterpret_cast<const uint2 *>(&w_lds[(16 * tp + li) * LDH + 16 * t + 4 * g]);
      a0[tp] = cp_mfma16<BF>(a, f0[t], a0[tp]);
    }
  }
  if (j0 >= 0) {
#pragma unroll
    for (int tp = 0; tp < TO; tp++)
      cp_st4(contrib, (row0 + li) * CO + 4 * g + 16 * tp, make_float4(a0[tp][0], a0[tp][1], a0[tp][2], a0[tp][3]),
             C16 ? (BF ? LINK_IO_BF16 : LINK_IO_F16) : LINK_IO_F32);
  }
}

// fp32 rows on the f16 matrix cores: x = hi + lo and w = hi + lo as fp16 pairs (22 mantissa bits), products
// hi*hi + hi*lo + lo*hi with fp32 accumulation -- exact products, the dropped lo*lo term is 2^-22 relative -- three
// 4-pass instructions per 16 input channels instead of four 8-pass ones (the fp32 form is bound by the matrix pipe from
// C = 64 up).  The weights arrive split and transposed, ws[k][co][hi(CI) | lo(CI)] (fp16, cached per parameter version by
// the host side), with a device flag that is non-zero when some |w| >= 2^15; that, or a row value outside the range,
// sends the wave through the fp32 instruction with W from global memory (never on normalised networks).
template <int CI, int CO>
__global__ void __launch_bounds__(512) k_conv_pairs_gemm_split(const float *__restrict__ feats,
                                                               const int32_t *__restrict__ pair_in,
                                                               const int32_t *__restrict__ wg_k,
                                                               const unsigned short *__restrict__ ws,
                                                               const float *__restrict__ w, const int32_t *__restrict__ w_big,
                                                               float *__restrict__ contrib) {
  constexpr int TI = CI / 16, TO = CO / 16, LDH = 2 * CI + 8;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short *w_lds = reinterpret_cast<unsigned short *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int k = wg_k[blockIdx.x];
  if (k < 0) return;
  const int64_t row0 = (int64_t)blockIdx.x * 128 + wave * 16;
  const int j0 = pair_in[row0 + li];
  const bool wbig = w_big[0] != 0;
  float4 f0[TI];
  {
    const float *fr = feats + (int64_t)(j0 < 0 ? 0 : j0) * CI + 4 * g;
#pragma unroll
    for (int t = 0; t < TI; t++) f0[t] = *reinterpret_cast<const float4 *>(fr + 16 * t);
  }
  const unsigned short *wk = ws + (int64_t)k * CO * 2 * CI;
  for (int e = tid * 8; e < CO * 2 * CI; e += 512 * 8) {
    const int r = e / (2 * CI), col = e - r * (2 * CI);
    *reinterpret_cast<uint4 *>(&w_lds[r * LDH + col]) = *reinterpret_cast<const uint4 *>(&wk[e]);
  }
  __syncthreads();
  if (__all(j0 < 0)) return;                           // granule padding
  floatx4 a0[TO];
#pragma unroll
  for (int tp = 0; tp < TO; tp++) a0[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
  uint2 bh[TI], bl[TI];
  float mx = 0.f;
#pragma unroll
  for (int t = 0; t < TI; t++) {
    const cp_h4 h = {(_Float16)f0[t].x, (_Float16)f0[t].y, (_Float16)f0[t].z, (_Float16)f0[t].w};
    const cp_h4 l = {(_Float16)(f0[t].x - (float)h.x), (_Float16)(f0[t].y - (float)h.y), (_Float16)(f0[t].z - (float)h.z),
                     (_Float16)(f0[t].w - (float)h.w)};
    bh[t] = __builtin_bit_cast(uint2, h);
    bl[t] = __builtin_bit_cast(uint2, l);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(f0[t].x), fabsf(f0[t].y)), fmaxf(fabsf(f0[t].z), fabsf(f0[t].w))));
  }
  if (__builtin_expect(!(wbig || __any(!(mx < 32768.0f))), 1)) {
#pragma unroll
    for (int t = 0; t < TI; t++) {
#pragma unroll
      for (int tp = 0; tp < TO; tp++) {
        const uint2 ah = *reinterpret_cast<const uint2 *>(&w_lds[(16 * tp + li) * LDH + 16 * t + 4 * g]);
        const uint2 al = *reinterpret_cast<const uint2 *>(&w_lds[(16 * tp + li) * LDH + CI + 16 * t + 4 * g]);
        a0[tp] = cp_mfma16<false>(al, bh[t], a0[tp]);
        a0[tp] = cp_mfma16<false>(ah, bl[t], a0[tp]);
        a0[tp] = cp_mfma16<false>(ah, bh[t], a0[tp]);
      }
    }
  } else {
    const float *wf = w + (int64_t)k * CI * CO;
#pragma unroll
    for (int t = 0; t < TI; t++) {
#pragma unroll
      for (int tp = 0; tp < TO; tp++) {
        const float *wr = wf + (16 * t + 4 * g) * CO + 16 * tp + li;
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[0], f0[t].x, a0[tp], 0, 0, 0);
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[CO], f0[t].y, a0[tp], 0, 0, 0);
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[2 * CO], f0[t].z, a0[tp], 0, 0, 0);
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3 * CO], f0[t].w, a0[tp], 0, 0, 0);
      }
    }
  }
  float *o0 = contrib + (row0 + li) * CO + 4 * g;
  if (j0 >= 0) {
#pragma unroll
    for (int tp = 0; tp < TO; tp++)
      *reinterpret_cast<float4 *>(o0 + 16 * tp) = make_float4(a0[tp][0], a0[tp][1], a0[tp][2], a0[tp][3]);
  }
}

// Submanifold tables: the centre offset's pairs are the identity, so its GEMM needs no gather and no
// contribution rows -- this kernel runs it per tile of 16 output voxels and finishes the voxel in the MFMA
// accumulator layout (a lane holds 4 channels x CO/16 tiles of ONE voxel): + the voxel's CSR rows of the other
// offsets (computed before by k_conv_pairs_gemm), + bias, LayerNorm as in-lane adds + 2 cross-lane steps,
// + addend, ReLU, one store.  On cfg2 (0.15 other neighbours per voxel) the whole convolution is this kernel
// plus a 17k-row GEMM.
template <int CI, int CO, bool TAIL, int MM>      // MM: 0 = fp32 weights w[k][ci][co]; 1 / 2 = the AMP form, f16 / bf16 weights wt[k][co][ci]; 3 / 4 = AMP with 16-bit contribution rows
__global__ void __launch_bounds__(512) k_conv_centre_sum(const void *__restrict__ feats, int io, const void *__restrict__ w,
                                                         int centre, const void *__restrict__ contrib,
                                                         uint32_t contrib_bytes, const int32_t *__restrict__ ext_start,
                                                         const int32_t *__restrict__ ext_list, int64_t n,
                                                         const float *__restrict__ bias, const float *__restrict__ ln_w,
                                                         const float *__restrict__ ln_b, float eps,
                                                         const void *__restrict__ addend, int relu,
                                                         void *__restrict__ out) {
  constexpr int TI = CI / 16, TO = CO / 16;
  constexpr int LD = CO + 4;
  constexpr bool BFW = MM == 2 || MM == 4, C16 = MM >= 3;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int nt = blockDim.x;                           // 512, or fewer waves per workgroup on small frames (more CUs pulling rows)
  const int64_t row0 = (int64_t)blockIdx.x * (nt / 4) + wave * 16;
  const int64_t v0 = row0 + li;
  const bool ok0 = v0 < n;
  const int64_t c0 = ok0 ? v0 : n - 1;
  float4 f0[MM ? 1 : TI];
  uint2 h0[MM ? TI : 1];
  if constexpr (MM != 0) {
    const unsigned short *fr = reinterpret_cast<const unsigned short *>(feats) + c0 * CI + 4 * g;
#pragma unroll
    for (int t = 0; t < TI; t++) h0[t] = *reinterpret_cast<const uint2 *>(fr + 16 * t);
  } else {
#pragma unroll
    for (int t = 0; t < TI; t++) f0[t] = cp_ld4(feats, c0 * CI + 16 * t + 4 * g, io);
  }
  const int s0 = ext_start[c0], e0 = ok0 ? ext_start[c0 + 1] : s0;
  if constexpr (MM != 0) {
    cp_stage_wt<CI, CO>(reinterpret_cast<unsigned short *>(smem_raw),
                        reinterpret_cast<const unsigned short *>(w) + (int64_t)centre * CI * CO, tid, nt);
  } else {
    const float *wk = reinterpret_cast<const float *>(w) + (int64_t)centre * CI * CO;
    for (int e = tid * 4; e < CI * CO; e += nt * 4) {
      const int r = e / CO, col = e - r * CO;
      *reinterpret_cast<float4 *>(&w_lds[r * LD + col]) = *reinterpret_cast<const float4 *>(&wk[e]);
    }
  }
  const int pf0 = s0 < e0 ? ext_list[s0] : -1, pf1 = s0 + 1 < e0 ? ext_list[s0 + 1] : -1;
  __syncthreads();
  if (row0 >= n) return;
  floatx4 a0[TO];
#pragma unroll
  for (int tp = 0; tp < TO; tp++) a0[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
  if constexpr (MM != 0) {
    constexpr int LDH = CI + 8;
    const unsigned short *wh = reinterpret_cast<const unsigned short *>(smem_raw);
#pragma unroll
    for (int t = 0; t < TI; t++) {
#pragma unroll
      for (int tp = 0; tp < TO; tp++) {
        const uint2 a = *reinterpret_cast<const uint2 *>(&wh[(16 * tp + li) * LDH + 16 * t + 4 * g]);
        a0[tp] = cp_mfma16<BFW>(a, h0[t], a0[tp]);
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < TI; t++) {
      const float *wr = &w_lds[(16 * t + 4 * g) * LD + li];
#pragma unroll
      for (int tp = 0; tp < TO; tp++) {
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[16 * tp], f0[t].x, a0[tp], 0, 0, 0);
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[LD + 16 * tp], f0[t].y, a0[tp], 0, 0, 0);
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[2 * LD + 16 * tp], f0[t].z, a0[tp], 0, 0, 0);
        a0[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[3 * LD + 16 * tp], f0[t].w, a0[tp], 0, 0, 0);
      }
    }
  }
  // the other offsets' rows, ascending kernel offset (fixed summation order), NQ per trip: the lists are
  // short but the trip count of a wave is the maximum over its 16 voxels, and a trip is two dependent round
  // trips.  Absent rows read through an out-of-range buffer offset (returns 0): no branch around the loads.
  if (__any(s0 < e0)) {
    #ifndef CP_NQ_SMALL
#define CP_NQ_SMALL 4
#define CP_NQ_LARGE 2
#endif
    constexpr int NQ = (TO <= 4) ? CP_NQ_SMALL : CP_NQ_LARGE;   // rows in flight per trip (register budget: NQ * TO dwordx4)
    const __amdgpu_buffer_rsrc_t r_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(contrib), 0, contrib_bytes, 0x00020000);
    int p[NQ];
    p[0] = pf0; p[1] = pf1;                            // the first trip's row ids were fetched before the MFMAs
#pragma unroll
    for (int j = 2; j < NQ; j++) p[j] = s0 + j < e0 ? ext_list[s0 + j] : -1;
    for (int q0 = s0; __any(q0 < e0); q0 += NQ) {
      floatx4 c[C16 ? 1 : NQ][C16 ? 1 : TO];
      uint2 h[C16 ? NQ : 1][C16 ? TO : 1];
#pragma unroll
      for (int j = 0; j < NQ; j++)
#pragma unroll
        for (int tp = 0; tp < TO; tp++) {
          if constexpr (C16) {
            const uint32_t off = p[j] >= 0 ? (uint32_t)p[j] * (uint32_t)(CO * 2) + (uint32_t)((16 * tp + 4 * g) * 2) : 0xFFFFFFF0u;
            h[j][tp] = __builtin_bit_cast(uint2, __builtin_amdgcn_raw_buffer_load_b64(r_c, off, 0, 0));
          } else {
            const uint32_t off = p[j] >= 0 ? (uint32_t)p[j] * (uint32_t)(CO * 4) + (uint32_t)((16 * tp + 4 * g) * 4) : 0xFFFFFFF0u;
            c[j][tp] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r_c, off, 0, 0));
          }
        }
#pragma unroll
      for (int j = 0; j < NQ; j++) p[j] = q0 + NQ + j < e0 ? ext_list[q0 + NQ + j] : -1;
#pragma unroll
      for (int tp = 0; tp < TO; tp++)
#pragma unroll
        for (int j = 0; j < NQ; j++) {
          if constexpr (C16) {
            float4 v;
            if constexpr (BFW) {
              v = make_float4(__uint_as_float(h[j][tp].x << 16), __uint_as_float(h[j][tp].x & 0xFFFF0000u),
                              __uint_as_float(h[j][tp].y << 16), __uint_as_float(h[j][tp].y & 0xFFFF0000u));
            } else {
              const cp_h4 q = __builtin_bit_cast(cp_h4, h[j][tp]);
              v = make_float4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
            }
            a0[tp][0] += v.x; a0[tp][1] += v.y; a0[tp][2] += v.z; a0[tp][3] += v.w;
          } else {
            a0[tp] += c[j][tp];
          }
        }
    }
  }
  if (bias) {
#pragma unroll
    for (int tp = 0; tp < TO; tp++) {
      const float4 b = *reinterpret_cast<const float4 *>(bias + 16 * tp + 4 * g);
      a0[tp][0] += b.x; a0[tp][1] += b.y; a0[tp][2] += b.z; a0[tp][3] += b.w;
    }
  }
  float mean = 0.f, rstd = 1.f;
  if (TAIL && !(relu & 2)) {                         // bit 1: ln_w / ln_b are a per-channel affine (folded BatchNorm)
    float sm = 0.f;
#pragma unroll
    for (int tp = 0; tp < TO; tp++) sm += (a0[tp][0] + a0[tp][1]) + (a0[tp][2] + a0[tp][3]);
    sm += __shfl_xor(sm, 16, 64);
    sm += __shfl_xor(sm, 32, 64);
    mean = sm * (1.0f / CO);
    float qv = 0.f;
#pragma unroll
    for (int tp = 0; tp < TO; tp++)
#pragma unroll
      for (int r = 0; r < 4; r++) { const float d = a0[tp][r] - mean; qv += d * d; }
    qv += __shfl_xor(qv, 16, 64);
    qv += __shfl_xor(qv, 32, 64);
    rstd = 1.0f / sqrtf(qv * (1.0f / CO) + eps);
  }
  if (!ok0) return;
#pragma unroll
  for (int tp = 0; tp < TO; tp++) {
    float4 o = make_float4(a0[tp][0], a0[tp][1], a0[tp][2], a0[tp][3]);
    if (TAIL) {
      const float4 lw = *reinterpret_cast<const float4 *>(ln_w + 16 * tp + 4 * g);
      const float4 lb = *reinterpret_cast<const float4 *>(ln_b + 16 * tp + 4 * g);
      o.x = (o.x - mean) * rstd * lw.x + lb.x; o.y = (o.y - mean) * rstd * lw.y + lb.y;
      o.z = (o.z - mean) * rstd * lw.z + lb.z; o.w = (o.w - mean) * rstd * lw.w + lb.w;
      if (addend) {
        const float4 ad = cp_ld4(addend, v0 * CO + 16 * tp + 4 * g, io);
        o.x += ad.x; o.y += ad.y; o.z += ad.z; o.w += ad.w;
      }
      if (relu & 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    }
    cp_st4(out, v0 * CO + 16 * tp + 4 * g, o, io);
  }
}

template <int LPR>
__device__ __forceinline__ float cp_grp_sum(float v) {
  if (LPR >= 2) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  if (LPR >= 4) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  if (LPR >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  if (LPR >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  if (LPR >= 32) v += __shfl_xor(v, 16, 64);
  return v;
}

// one LPR-lane group per output voxel, a lane owns 4 channels (C = 4 * LPR)
template <int LPR, bool TAIL>
__global__ void __launch_bounds__(256) k_conv_pairs_sum(const float *__restrict__ contrib,
                                                        const int32_t *__restrict__ ext_start,
                                                        const int32_t *__restrict__ ext_list, int64_t n,
                                                        int64_t n_direct, const float *__restrict__ bias,
                                                        const float *__restrict__ ln_w, const float *__restrict__ ln_b,
                                                        float eps, const void *__restrict__ addend, int relu,
                                                        void *__restrict__ out, int io) {
  constexpr int C = 4 * LPR, G = 64 / LPR;
  const int lane = threadIdx.x & 63, li = lane & (LPR - 1);
  const int64_t ngroups = (int64_t)gridDim.x * 4 * G;
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), gw = bv, gb = bv;
  if (bias) bv = *reinterpret_cast<const float4 *>(bias + 4 * li);
  if (TAIL) { gw = *reinterpret_cast<const float4 *>(ln_w + 4 * li); gb = *reinterpret_cast<const float4 *>(ln_b + 4 * li); }
  for (int64_t v = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR; v < n; v += ngroups) {
    const int s = ext_start[v], e = ext_start[v + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (v < n_direct) acc = *reinterpret_cast<const float4 *>(contrib + v * C + 4 * li);
    float4 ad = make_float4(0.f, 0.f, 0.f, 0.f);
    if (TAIL && addend) ad = cp_ld4(addend, v * C + 4 * li, io);
    int q = s;
    for (; q + 1 < e; q += 2) {                       // two rows in flight per trip
      const int p0 = ext_list[q], p1 = ext_list[q + 1];
      const float4 c0 = *reinterpret_cast<const float4 *>(contrib + (int64_t)p0 * C + 4 * li);
      const float4 c1 = *reinterpret_cast<const float4 *>(contrib + (int64_t)p1 * C + 4 * li);
      acc.x += c0.x; acc.y += c0.y; acc.z += c0.z; acc.w += c0.w;
      acc.x += c1.x; acc.y += c1.y; acc.z += c1.z; acc.w += c1.w;
    }
    if (q < e) {
      const float4 c0 = *reinterpret_cast<const float4 *>(contrib + (int64_t)ext_list[q] * C + 4 * li);
      acc.x += c0.x; acc.y += c0.y; acc.z += c0.z; acc.w += c0.w;
    }
    acc.x += bv.x; acc.y += bv.y; acc.z += bv.z; acc.w += bv.w;
    if (TAIL) {
      float mean = 0.f, rstd = 1.f;
      if (!(relu & 2)) {
        mean = cp_grp_sum<LPR>((acc.x + acc.y) + (acc.z + acc.w)) * (1.0f / C);
        const float ex = acc.x - mean, ey = acc.y - mean, ez = acc.z - mean, ew = acc.w - mean;
        rstd = 1.0f / sqrtf(cp_grp_sum<LPR>((ex * ex + ey * ey) + (ez * ez + ew * ew)) * (1.0f / C) + eps);
      }
      const float dx = acc.x - mean, dy = acc.y - mean, dz = acc.z - mean, dw = acc.w - mean;
      acc.x = dx * rstd * gw.x + gb.x + ad.x; acc.y = dy * rstd * gw.y + gb.y + ad.y;
      acc.z = dz * rstd * gw.z + gb.z + ad.z; acc.w = dw * rstd * gw.w + gb.w + ad.w;
      if (relu & 1) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
    }
    cp_st4(out, v * C + 4 * li, acc, io);
  }
}

template <int CI, int CO>
static int launch_pairs_gemm(const void *feats, int io, const int32_t *pair_in, const int32_t *wg_k, int64_t granules,
                             const float *w, float *contrib, hipStream_t st) {
  const size_t lds = (size_t)CI * (CO + 4) * sizeof(float);
  if (lds > 64 * 1024)      // per device and cheap: no process-wide "done" flag
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_pairs_gemm<CI, CO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_conv_pairs_gemm<CI, CO>), dim3((unsigned)granules), dim3(512), lds, st, feats, io, pair_in, wg_k, w, contrib);
  return check_launch("link_conv_pairs_gemm");
}

extern "C" int link_conv_pairs_supported(int32_t cin, int32_t cout) {
  const bool sq = cin == cout && (cin == 16 || cin == 32 || cin == 64 || cin == 128);
  const bool rect = (cin == 16 && cout == 32) || (cin == 32 && cout == 16) || (cin == 32 && cout == 64) ||
                    (cin == 64 && cout == 32) || (cin == 64 && cout == 128) || (cin == 128 && cout == 64) ||
                    (cin == 16 && cout == 64) || (cin == 64 && cout == 16);
  return (sq || rect) ? 1 : 0;
}

extern "C" int link_conv_pairs_gemm_io(const void *feats, int32_t io_dtype, const int32_t *pair_in, const int32_t *wg_k,
                                       int64_t rows_pad, const float *w, int32_t cin, int32_t cout, float *contrib, void *stream);
extern "C" int link_conv_pairs_gemm(const float *feats, const int32_t *pair_in, const int32_t *wg_k, int64_t rows_pad,
                                    const float *w, int32_t cin, int32_t cout, float *contrib, void *stream) {
  return link_conv_pairs_gemm_io(feats, LINK_IO_F32, pair_in, wg_k, rows_pad, w, cin, cout, contrib, stream);
}
extern "C" int link_conv_pairs_gemm_io(const void *feats, int32_t io_dtype, const int32_t *pair_in, const int32_t *wg_k,
                                       int64_t rows_pad, const float *w, int32_t cin, int32_t cout, float *contrib, void *stream) {
  if (io_dtype < 0 || io_dtype > 2) return LINK_ERR_ARG;
  if (rows_pad < 0 || (rows_pad & 127) || rows_pad >= (1LL << 31) || !link_conv_pairs_supported(cin, cout)) return LINK_ERR_ARG;
  if (rows_pad == 0) return LINK_OK;
  if (!feats || !pair_in || !wg_k || !w || !contrib) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  const int64_t gr = rows_pad / 128;
#define LINK_CP(I, O) if (cin == I && cout == O) return launch_pairs_gemm<I, O>(feats, (int)io_dtype, pair_in, wg_k, gr, w, contrib, st)
  LINK_CP(16, 16); LINK_CP(32, 32); LINK_CP(64, 64); LINK_CP(128, 128);
  LINK_CP(16, 32); LINK_CP(32, 16); LINK_CP(32, 64); LINK_CP(64, 32); LINK_CP(64, 128); LINK_CP(128, 64);
  LINK_CP(16, 64); LINK_CP(64, 16);
#undef LINK_CP
  return LINK_ERR_ARG;
}

extern "C" int link_conv_pairs_sum_io(const float *contrib, const int32_t *ext_start, const int32_t *ext_list, int64_t n,
                                      int64_t n_direct, int32_t cout, const float *bias, const float *ln_w, const float *ln_b,
                                      float eps, const void *addend, int32_t relu, void *out, int32_t io_dtype, void *stream);
extern "C" int link_conv_pairs_sum(const float *contrib, const int32_t *ext_start, const int32_t *ext_list, int64_t n,
                                   int64_t n_direct, int32_t cout, const float *bias, const float *ln_w,
                                   const float *ln_b, float eps, const float *addend, int32_t relu, float *out,
                                   void *stream) {
  return link_conv_pairs_sum_io(contrib, ext_start, ext_list, n, n_direct, cout, bias, ln_w, ln_b, eps, addend, relu, out,
                                LINK_IO_F32, stream);
}
extern "C" int link_conv_pairs_sum_io(const float *contrib, const int32_t *ext_start, const int32_t *ext_list, int64_t n,
                                      int64_t n_direct, int32_t cout, const float *bias, const float *ln_w, const float *ln_b,
                                      float eps, const void *addend, int32_t relu, void *out, int32_t io_dtype, void *stream) {
  if (io_dtype < 0 || io_dtype > 2) return LINK_ERR_ARG;
  if (n < 0 || n_direct < 0 || n_direct > n) return LINK_ERR_ARG;
  if (cout != 16 && cout != 32 && cout != 64 && cout != 128) return LINK_ERR_ARG;   // power-of-two lane groups
  if ((ln_w == nullptr) != (ln_b == nullptr)) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!contrib || !ext_start || !ext_list || !out) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  const bool tail = ln_w != nullptr;
  const int lpr = cout / 4, gpw = 64 / lpr;
  int64_t wgs = (n + 4 * gpw - 1) / (4 * gpw);
  if (wgs > 4096) wgs = 4096;
#define LINK_CS(LPRV)                                                                                                  \
  if (lpr == LPRV) {                                                                                                   \
    if (tail) hipLaunchKernelGGL((k_conv_pairs_sum<LPRV, true>), dim3((unsigned)wgs), dim3(256), 0, st, contrib, ext_start,  \
                                 ext_list, n, n_direct, bias, ln_w, ln_b, eps, addend, (int)relu, out, (int)io_dtype); \
    else hipLaunchKernelGGL((k_conv_pairs_sum<LPRV, false>), dim3((unsigned)wgs), dim3(256), 0, st, contrib, ext_start,      \
                            ext_list, n, n_direct, bias, ln_w, ln_b, eps, addend, (int)relu, out, (int)io_dtype);      \
    return check_launch("link_conv_pairs_sum");                                                                        \
  }
  LINK_CS(4) LINK_CS(8) LINK_CS(16) LINK_CS(32)
#undef LINK_CS
  return LINK_ERR_ARG;
}

template <int CI, int CO, int MM>
static int launch_centre_sum(const void *feats, int io, const void *w, int centre, const void *contrib, uint32_t cbytes, const int32_t *ext_start,
                             const int32_t *ext_list, int64_t n, const float *bias, const float *ln_w, const float *ln_b,
                             float eps, const void *addend, int relu, void *out, hipStream_t st) {
  const size_t lds = MM ? (size_t)CO * (CI + 8) * 2 : (size_t)CI * (CO + 4) * sizeof(float);
  // a workgroup of 8 waves finishes 128 voxels; frames too small to give every CU one of those run 4 or 2 waves per
  // workgroup instead (the kernel is a row gather: what counts is how many CUs pull rows, W_k staging is per workgroup)
#ifndef CP_NT_BIG
#define CP_NT_BIG 262144
#define CP_NT_MID 24576
#endif
  const unsigned nt = n > CP_NT_BIG ? 512u : (n > CP_NT_MID ? 256u : 128u);
  const unsigned wgs = (unsigned)((n + nt / 4 - 1) / (nt / 4));
  if (ln_w) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_centre_sum<CI, CO, true, MM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_conv_centre_sum<CI, CO, true, MM>), dim3(wgs), dim3(nt), lds, st, feats, io, w, centre, contrib, cbytes, ext_start,
                       ext_list, n, bias, ln_w, ln_b, eps, addend, relu, out);
  } else {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_centre_sum<CI, CO, false, MM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_conv_centre_sum<CI, CO, false, MM>), dim3(wgs), dim3(nt), lds, st, feats, io, w, centre, contrib, cbytes, ext_start,
                       ext_list, n, bias, ln_w, ln_b, eps, addend, relu, out);
  }
  return check_launch("link_conv_centre_sum");
}

extern "C" int link_conv_centre_sum_io(const void *feats, const float *w, int32_t centre, const float *contrib,
                                       int64_t contrib_rows, const int32_t *ext_start, const int32_t *ext_list, int64_t n,
                                       int32_t cin, int32_t cout, const float *bias, const float *ln_w, const float *ln_b, float eps,
                                       const void *addend, int32_t relu, void *out, int32_t io_dtype, void *stream);
extern "C" int link_conv_centre_sum(const float *feats, const float *w, int32_t centre, const float *contrib,
                                    int64_t contrib_rows, const int32_t *ext_start, const int32_t *ext_list, int64_t n, int32_t cin,
                                    int32_t cout, const float *bias, const float *ln_w, const float *ln_b, float eps,
                                    const float *addend, int32_t relu, float *out, void *stream) {
  return link_conv_centre_sum_io(feats, w, centre, contrib, contrib_rows, ext_start, ext_list, n, cin, cout, bias, ln_w, ln_b, eps,
                                 addend, relu, out, LINK_IO_F32, stream);
}
extern "C" int link_conv_centre_sum_io(const void *feats, const float *w, int32_t centre, const float *contrib,
                                       int64_t contrib_rows, const int32_t *ext_start, const int32_t *ext_list, int64_t n,
                                       int32_t cin, int32_t cout, const float *bias, const float *ln_w, const float *ln_b, float eps,
                                       const void *addend, int32_t relu, void *out, int32_t io_dtype, void *stream) {
  if (io_dtype < 0 || io_dtype > 2) return LINK_ERR_ARG;
  if (n < 0 || centre < 0 || !link_conv_pairs_supported(cin, cout) || (ln_w == nullptr) != (ln_b == nullptr)) return LINK_ERR_ARG;
  if (contrib_rows < 0 || contrib_rows * (int64_t)cout * 4 >= 0xFFFFFFF0LL) return LINK_ERR_ARG;   // 32-bit row offsets
  const uint32_t cbytes = (uint32_t)(contrib_rows * cout * 4);
  if (n == 0) return LINK_OK;
  if (!feats || !w || !ext_start || !out || (contrib_rows > 0 && (!contrib || !ext_list))) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
#define LINK_CC(I, O) if (cin == I && cout == O) return launch_centre_sum<I, O, 0>(feats, (int)io_dtype, w, centre, contrib, cbytes, ext_start, ext_list, n, bias, ln_w, ln_b, eps, addend, (int)relu, out, st)
  LINK_CC(16, 16); LINK_CC(32, 32); LINK_CC(64, 64); LINK_CC(128, 128);
  LINK_CC(16, 32); LINK_CC(32, 16); LINK_CC(32, 64); LINK_CC(64, 32); LINK_CC(64, 128); LINK_CC(128, 64);
  LINK_CC(16, 64); LINK_CC(64, 16);
#undef LINK_CC
  return LINK_ERR_ARG;
}

template <int CI, int CO>
static int launch_pairs_gemm_split(const float *feats, const int32_t *pair_in, const int32_t *wg_k, int64_t granules,
                                   const void *ws, const float *w, const int32_t *w_big, float *contrib, hipStream_t st) {
  const size_t lds = (size_t)CO * (2 * CI + 8) * 2;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_pairs_gemm_split<CI, CO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_conv_pairs_gemm_split<CI, CO>), dim3((unsigned)granules), dim3(512), lds, st, feats, pair_in, wg_k,
                     reinterpret_cast<const unsigned short *>(ws), w, w_big, contrib);
  return check_launch("link_conv_pairs_gemm_split");
}

extern "C" int link_conv_pairs_gemm_split(const float *feats, const int32_t *pair_in, const int32_t *wg_k, int64_t rows_pad,
                                          const void *ws, const float *w, const int32_t *w_big, int32_t cin, int32_t cout,
                                          float *contrib, void *stream) {
  if (rows_pad < 0 || (rows_pad & 127) || rows_pad >= (1LL << 31) || !link_conv_pairs_supported(cin, cout)) return LINK_ERR_ARG;
  if (rows_pad == 0) return LINK_OK;
  if (!feats || !pair_in || !wg_k || !ws || !w || !w_big || !contrib) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  const int64_t gr = rows_pad / 128;
#define LINK_CP(I, O) if (cin == I && cout == O) return launch_pairs_gemm_split<I, O>(feats, pair_in, wg_k, gr, ws, w, w_big, contrib, st)
  LINK_CP(16, 16); LINK_CP(32, 32); LINK_CP(64, 64); LINK_CP(128, 128);
  LINK_CP(16, 32); LINK_CP(32, 16); LINK_CP(32, 64); LINK_CP(64, 32); LINK_CP(64, 128); LINK_CP(128, 64);
  LINK_CP(16, 64); LINK_CP(64, 16);
#undef LINK_CP
  return LINK_ERR_ARG;
}

// AMP form (see cp_mfma16): 16-bit rows AND 16-bit weights wt[k][cout][cin] of the same type, fp32 accumulation.
template <int CI, int CO, bool BF, bool C16>
static int launch_pairs_gemm_h(const void *feats, const int32_t *pair_in, const int32_t *wg_k, int64_t granules,
                               const void *wt, void *contrib, hipStream_t st) {
  const size_t lds = (size_t)CO * (CI + 8) * 2;
  hipLaunchKernelGGL((k_conv_pairs_gemm_h<CI, CO, BF, C16>), dim3((unsigned)granules), dim3(512), lds, st,
                     reinterpret_cast<const unsigned short *>(feats), pair_in, wg_k, reinterpret_cast<const unsigned short *>(wt), contrib);
  return check_launch("link_conv_pairs_gemm_amp");
}

extern "C" int link_conv_pairs_gemm_amp(const void *feats, int32_t io_dtype, const int32_t *pair_in, const int32_t *wg_k,
                                        int64_t rows_pad, const void *wt, int32_t cin, int32_t cout, void *contrib,
                                        int32_t contrib_dtype, void *stream) {
  if (io_dtype != LINK_IO_F16 && io_dtype != LINK_IO_BF16) return LINK_ERR_ARG;
  if (contrib_dtype != LINK_IO_F32 && contrib_dtype != io_dtype) return LINK_ERR_ARG;
  const bool c16 = contrib_dtype != LINK_IO_F32;
  if (rows_pad < 0 || (rows_pad & 127) || rows_pad >= (1LL << 31) || !link_conv_pairs_supported(cin, cout)) return LINK_ERR_ARG;
  if (rows_pad == 0) return LINK_OK;
  if (!feats || !pair_in || !wg_k || !wt || !contrib) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  const int64_t gr = rows_pad / 128;
#define LINK_CP(I, O)                                                                                                   \
  if (cin == I && cout == O)                                                                                            \
    return io_dtype == LINK_IO_BF16                                                                                     \
               ? (c16 ? launch_pairs_gemm_h<I, O, true, true>(feats, pair_in, wg_k, gr, wt, contrib, st)                \
                      : launch_pairs_gemm_h<I, O, true, false>(feats, pair_in, wg_k, gr, wt, contrib, st))              \
               : (c16 ? launch_pairs_gemm_h<I, O, false, true>(feats, pair_in, wg_k, gr, wt, contrib, st)               \
                      : launch_pairs_gemm_h<I, O, false, false>(feats, pair_in, wg_k, gr, wt, contrib, st))
  LINK_CP(16, 16); LINK_CP(32, 32); LINK_CP(64, 64); LINK_CP(128, 128);
  LINK_CP(16, 32); LINK_CP(32, 16); LINK_CP(32, 64); LINK_CP(64, 32); LINK_CP(64, 128); LINK_CP(128, 64);
  LINK_CP(16, 64); LINK_CP(64, 16);
#undef LINK_CP
  return LINK_ERR_ARG;
}

extern "C" int link_conv_centre_sum_amp(const void *feats, const void *wt, int32_t centre, const void *contrib,
                                        int32_t contrib_dtype, int64_t contrib_rows, const int32_t *ext_start,
                                        const int32_t *ext_list, int64_t n, int32_t cin, int32_t cout, const float *bias,
                                        const float *ln_w, const float *ln_b, float eps, const void *addend, int32_t relu,
                                        void *out, int32_t io_dtype, void *stream) {
  if (io_dtype != LINK_IO_F16 && io_dtype != LINK_IO_BF16) return LINK_ERR_ARG;
  if (contrib_dtype != LINK_IO_F32 && contrib_dtype != io_dtype) return LINK_ERR_ARG;
  const bool c16 = contrib_dtype != LINK_IO_F32;
  if (n < 0 || centre < 0 || !link_conv_pairs_supported(cin, cout) || (ln_w == nullptr) != (ln_b == nullptr)) return LINK_ERR_ARG;
  if (contrib_rows < 0 || contrib_rows * (int64_t)cout * 4 >= 0xFFFFFFF0LL) return LINK_ERR_ARG;   // 32-bit row offsets
  const uint32_t cbytes = (uint32_t)(contrib_rows * cout * (c16 ? 2 : 4));
  if (n == 0) return LINK_OK;
  if (!feats || !wt || !ext_start || !out || (contrib_rows > 0 && (!contrib || !ext_list))) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
#define LINK_CA(I, O, M) launch_centre_sum<I, O, M>(feats, (int)io_dtype, wt, centre, contrib, cbytes, ext_start, ext_list, n, bias, ln_w, ln_b, eps, addend, (int)relu, out, st)
#define LINK_CC(I, O)                                                                                                   \
  if (cin == I && cout == O)                                                                                            \
    return io_dtype == LINK_IO_BF16 ? (c16 ? LINK_CA(I, O, 4) : LINK_CA(I, O, 2)) : (c16 ? LINK_CA(I, O, 3) : LINK_CA(I, O, 1))
  LINK_CC(16, 16); LINK_CC(32, 32); LINK_CC(64, 64); LINK_CC(128, 128);
  LINK_CC(16, 32); LINK_CC(32, 16); LINK_CC(32, 64); LINK_CC(64, 32); LINK_CC(64, 128); LINK_CC(128, 64);
  LINK_CC(16, 64); LINK_CC(64, 16);
#undef LINK_CC
  return LINK_ERR_ARG;
}

// ---------------------------------------------------------------------------------------------
// building the pair plan from a per-output neighbour table (what link_amd/elk.py::_PairPlan holds)
// ---------------------------------------------------------------------------------------------
// Two passes over the table with one host round trip between them (the host needs the per-offset pair counts
// to lay out the 128-row granules and to size the contribution buffer -- the same numbers the reference's
// nbsizes holds on the host, nn/functional/conv.py:114-116):
//   count pass     per workgroup (256 table rows): pairs of every offset, and rows whose centre entry is not the
//                  row itself (0 in total <=> submanifold table); row_info[i] = (#valid entries) | (centre valid) << 16
//   fill pass      contribution row p of every pair = first row of its offset's granules + pairs of that offset in
//                  earlier workgroups (the host's scan of the count pass) + in earlier waves / lanes (LDS, ballot):
//                  no atomics, placement is deterministic; pair_in[p] = input row, and the voxel's CSR list in
//                  ascending offset order.
// The workgroup stages its 256 table rows through LDS (coalesced reads; row stride kvol is odd for every
// kernel the networks use, so the per-lane walk is conflict-free).
template <bool FILL>
__global__ void __launch_bounds__(256) k_pair_plan(const int32_t *__restrict__ nbr, int64_t n, int kvol, int centre,
                                                   int skip_centre, const int32_t *__restrict__ base_k,
                                                   const int32_t *__restrict__ wg_base, const int32_t *__restrict__ ext_start,
                                                   int32_t *__restrict__ wg_counts, int32_t *__restrict__ row_info,
                                                   int32_t *__restrict__ pair_in, int32_t *__restrict__ pair_out,
                                                   int32_t *__restrict__ ext_list, const int32_t *__restrict__ wg_ext = nullptr,
                                                   int32_t *__restrict__ ext_start_out = nullptr) {
  extern __shared__ int32_t smem[];
  int32_t *tile = smem;                                // [256][kvol]
  int32_t *wcnt = smem + 256 * kvol;                   // [4][kvol + 1]: pairs per wave and offset (+ identity misses)
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int rows = (int)((n - row0 < 256) ? n - row0 : 256);
  const int tot = rows * kvol;
  const int32_t *src = nbr + row0 * kvol;
  for (int e = threadIdx.x; e < tot; e += 256) tile[e] = src[e];
  __syncthreads();
  const int r = threadIdx.x;
  const bool live = r < rows;
  const int32_t *mine = tile + (live ? r : 0) * kvol;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int nvalid = 0, cvalid = 0;
  // pass A: this wave's pair count per offset -> LDS (no global atomics: 27 counters shared by every wave of the
  // grid serialise in the L2 and made this kernel the longest of a cold frame)
  for (int k = 0; k < kvol; k++) {
    const int v = live ? mine[k] : -1;
    bool valid = v >= 0;
    if (k == centre) {
      cvalid = valid ? 1 : 0;
      const unsigned long long bad = __ballot(live && v != (int)(row0 + r));
      if (lane == 0) wcnt[wave * (kvol + 1) + kvol] = __popcll(bad);
      if (FILL && skip_centre) valid = false;
    }
    const unsigned long long m = __ballot(valid);
    if (lane == 0) wcnt[wave * (kvol + 1) + k] = __popcll(m);
    nvalid += valid ? 1 : 0;
  }
  __syncthreads();
  if (!FILL) {
    if (live) row_info[row0 + r] = nvalid | (cvalid << 16);
    if (threadIdx.x <= kvol) {                         // per-workgroup totals; the host sums / scans them
      const int k = threadIdx.x;
      wg_counts[(int64_t)blockIdx.x * (kvol + 1) + k] =
          wcnt[k] + wcnt[(kvol + 1) + k] + wcnt[2 * (kvol + 1) + k] + wcnt[3 * (kvol + 1) + k];
    }
    return;
  }
  // pass B: contribution row = first row of the offset's granules + pairs of earlier workgroups + earlier waves of
  // this workgroup + earlier lanes of this wave: deterministic placement, voxel ascending inside an offset
  int q;
  if (wg_ext) {
    // the CSR start of every row without a separate prefix-sum pass: rows of earlier workgroups (wg_ext, from the layout
    // kernel) + the rows before this one in the workgroup (wave prefix + the earlier waves' totals through LDS)
    int incl = live ? nvalid : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    __shared__ int s_wave_tot[4];
    if (lane == 63) s_wave_tot[wave] = incl;
    __syncthreads();
    int before_w = 0;
    for (int w = 0; w < wave; w++) before_w += s_wave_tot[w];
    q = wg_ext[blockIdx.x] + before_w + incl - (live ? nvalid : 0);
    if (live) ext_start_out[row0 + r] = q;
  } else {
    q = live ? ext_start[row0 + r] : 0;
  }
  for (int k = 0; k < kvol; k++) {
    if (skip_centre && k == centre) continue;
    const int v = live ? mine[k] : -1;
    const bool valid = v >= 0;
    const unsigned long long m = __ballot(valid);
    if (valid) {
      int before = 0;
      for (int w = 0; w < wave; w++) before += wcnt[w * (kvol + 1) + k];
      const int p = base_k[k] + wg_base[(int64_t)blockIdx.x * kvol + k] + before + __popcll(m & lt);
      pair_in[p] = v;
      pair_out[p] = (int)(row0 + r);
      ext_list[q++] = p;
    }
  }
}

extern "C" int link_pair_plan_count(const int32_t *nbr, int64_t n, int32_t kvol, int32_t *wg_counts, int32_t *row_info,
                                    void *stream) {
  if (n < 0 || kvol <= 0 || kvol > 64) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!nbr || !wg_counts || !row_info) return LINK_ERR_ARG;
  const unsigned wgs = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_pair_plan<false>, dim3(wgs), dim3(256), (size_t)(256 * kvol + 4 * (kvol + 1)) * 4, S(stream), nbr, n,
                     (int)kvol, (int)(kvol / 2), 0, (const int32_t *)nullptr, (const int32_t *)nullptr, (const int32_t *)nullptr,
                     wg_counts, row_info, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
  return check_launch("link_pair_plan_count");
}

extern "C" int link_pair_plan_fill(const int32_t *nbr, int64_t n, int32_t kvol, int32_t skip_centre, const int32_t *base_k,
                                   const int32_t *wg_base, const int32_t *ext_start, int32_t *pair_in, int32_t *pair_out,
                                   int32_t *ext_list, void *stream) {
  if (n < 0 || kvol <= 0 || kvol > 64) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!nbr || !base_k || !wg_base || !ext_start || !pair_in || !pair_out || !ext_list) return LINK_ERR_ARG;
  const unsigned wgs = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(k_pair_plan<true>, dim3(wgs), dim3(256), (size_t)(256 * kvol + 4 * (kvol + 1)) * 4, S(stream), nbr, n,
                     (int)kvol, (int)(kvol / 2), (int)skip_centre, base_k, wg_base, ext_start, (int32_t *)nullptr,
                     (int32_t *)nullptr, pair_in, pair_out, ext_list);
  return check_launch("link_pair_plan_fill");
}

// Layout of a pair plan ON THE DEVICE (what the host otherwise computes between link_pair_plan_count and
// link_pair_plan_fill after reading the counts back), two launches (round 5):
//   k_pair_plan_colscan   a wave per offset column: exclusive scan of the per-workgroup counts down the column (-> wg_base) and
//                         the column's total (left in base_k[k]; the centre-miss column's in hdr[3])
//   k_pair_plan_finish    one workgroup: granule-aligned row ranges per offset (one lane), rows of earlier workgroups over all
//                         offsets (wg_ext), -1 tails, the offset of every granule up to the caller's capacity (-1 behind the
//                         last one: the GEMM kernels return there)
// Until round 5 this was ONE workgroup doing everything: 27 us on a 100k-voxel table (391 count workgroups) -- 46-58 us next to
// other kernels -- of which 35 us were the column scans (tools/layout_bench.hip: a single workgroup is bound by the latency
// of its own instruction stream, one wave per SIMD; staging the table through LDS, DPP scans and batched loads each moved it by
// a few us only).  A wave per column on 28 CUs does the scans in ~3 us.
#ifdef PP_DBG
__device__ unsigned long long g_pp_dbg[8];
#define PP_T(i) do { __syncthreads(); if (threadIdx.x == 0) g_pp_dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PP_T(i) do { } while (0)
#endif
__global__ void __launch_bounds__(64) k_pair_plan_colscan(const int32_t *__restrict__ wg_counts, int nwg, int kvol,
                                                          int32_t *__restrict__ wg_base, int32_t *__restrict__ base_k,
                                                          int32_t *__restrict__ hdr) {
  const int k = (int)blockIdx.x, lane = (int)threadIdx.x;          // column 0 .. kvol (kvol = rows whose centre entry is not the row)
  const int kc = kvol + 1;
  int running = 0;
  for (int c0 = 0; c0 < nwg; c0 += 512) {
    int vv[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {                        // eight chunks of 64 rows requested together
      const int w = c0 + 64 * j + lane;
      vv[j] = w < nwg ? wg_counts[(int64_t)w * kc + k] : 0;
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (c0 + 64 * j >= nwg) break;                     // uniform
      const int w = c0 + 64 * j + lane;
      int incl = vv[j];
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
      const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47);
      const int gq = lane >> 4;
      incl += gq == 0 ? 0 : (gq == 1 ? t0 : (gq == 2 ? t0 + t1 : t0 + t1 + t2));
      if (w < nwg && k < kvol) wg_base[(int64_t)w * kvol + k] = running + incl - vv[j];
      running += __builtin_amdgcn_readlane(incl, 63);
    }
  }
  if (lane == 0) {
    if (k < kvol) base_k[k] = running; else hdr[3] = running;
  }
}

__global__ void __launch_bounds__(256) k_pair_plan_finish(int nwg, int kvol, int centre, int skip_centre, int64_t gran_cap,
                                                          int32_t *__restrict__ base_k, const int32_t *__restrict__ wg_base,
                                                          int32_t *__restrict__ gran_start, int32_t *__restrict__ wg_k,
                                                          int32_t *__restrict__ hdr, int32_t *__restrict__ wg_ext,
                                                          int32_t *__restrict__ pair_in, int32_t *__restrict__ pair_out,
                                                          int32_t *__restrict__ ext_total) {
  __shared__ int s_tot[65], s_gs[66], s_base[65];
  const int nthr = (int)blockDim.x;
  PP_T(0);
  if ((int)threadIdx.x < kvol) s_tot[threadIdx.x] = base_k[threadIdx.x];      // the column totals k_pair_plan_colscan left there
  if (wg_ext) {
    // rows of earlier workgroups over all offsets (the fill kernel's CSR starts): a thread per count workgroup, its kvol
    // values requested together
    for (int w = threadIdx.x; w < nwg; w += nthr) {
      int acc = 0;
#pragma unroll 32
      for (int k = 0; k < kvol; k++) {
        const int v = wg_base[(int64_t)w * kvol + k];
        acc += (skip_centre && k == centre) ? 0 : v;
      }
      wg_ext[w] = acc;
    }
  }
  __syncthreads();
  PP_T(1);
  if (threadIdx.x == 0) {
    int64_t rows = 0, gran = 0, pairs = 0;
    for (int k = 0; k < kvol; k++) {
      const int cnt = (skip_centre && k == centre) ? 0 : s_tot[k];
      base_k[k] = s_base[k] = (int32_t)rows;
      gran_start[k] = s_gs[k] = (int32_t)gran;
      const int64_t gk = (cnt + 127) / 128;
      rows += gk * 128;
      gran += gk;
      pairs += cnt;
    }
    gran_start[kvol] = s_gs[kvol] = (int32_t)gran;
    hdr[0] = (int32_t)pairs;
    hdr[1] = (int32_t)rows;
    hdr[2] = (int32_t)gran;                            // (hdr[3], rows whose centre neighbour is not the row itself: k_pair_plan_colscan)
    hdr[4] = gran > gran_cap ? 1 : 0;                  // capacity exceeded: cannot happen with the bound of the C ABI comment
    hdr[5] = hdr[6] = hdr[7] = 0;
    if (ext_total) *ext_total = (int32_t)pairs;       // ext_start[n]
  }
  __syncthreads();
  PP_T(2);
  if (pair_in) {
    // -1 in the unused tail of every offset's last granule (the only padding a GEMM workgroup ever reads: granules behind
    // the last one return on wg_k)
    for (int k = 0; k < kvol; k++) {
      const int cnt = (skip_centre && k == centre) ? 0 : s_tot[k];
      const int end = ((cnt + 127) / 128) * 128;
      for (int r = cnt + threadIdx.x; r < end; r += nthr) {
        pair_in[s_base[k] + r] = -1;
        pair_out[s_base[k] + r] = -1;
      }
    }
  }
  PP_T(3);
  const int total = s_gs[kvol];
  // the granules in use (a few hundred) one by one; the capacity behind them (tens of thousands of -1) in 16-byte stores
  int64_t g_vec = gran_cap;                              // first granule of the vector part
  if ((reinterpret_cast<uintptr_t>(wg_k) & 15) == 0) g_vec = ((int64_t)total + 3) & ~(int64_t)3;
  if (g_vec > gran_cap) g_vec = gran_cap;
  for (int64_t g = threadIdx.x; g < g_vec; g += nthr) {
    int k = -1;
    if (g < total) {
      k = 0;
      while (k + 1 < kvol && s_gs[k + 1] <= g) k++;
    }
    wg_k[g] = k;
  }
  const int64_t nvec = (gran_cap - g_vec) >> 2;
  for (int64_t q = threadIdx.x; q < nvec; q += nthr) reinterpret_cast<int4 *>(wg_k + g_vec)[q] = make_int4(-1, -1, -1, -1);
  for (int64_t g = g_vec + (nvec << 2) + threadIdx.x; g < gran_cap; g += nthr) wg_k[g] = -1;
  PP_T(4);
}

static int pair_plan_layout_run(const int32_t *wg_counts, int64_t nwg, int32_t kvol, int32_t skip_centre, int64_t gran_cap, int32_t *base_k,
                                int32_t *wg_base, int32_t *gran_start, int32_t *wg_k, int32_t *hdr, int32_t *wg_ext, int32_t *pair_in,
                                int32_t *pair_out, int32_t *ext_total, hipStream_t st, const char *what) {
  hipLaunchKernelGGL(k_pair_plan_colscan, dim3((unsigned)(kvol + 1)), dim3(64), 0, st, wg_counts, (int)nwg, (int)kvol, wg_base, base_k, hdr);
  int rc = check_launch(what);
  if (rc != LINK_OK) return rc;
  hipLaunchKernelGGL(k_pair_plan_finish, dim3(1), dim3(256), 0, st, (int)nwg, (int)kvol, (int)(kvol / 2), (int)skip_centre, gran_cap, base_k,
                     (const int32_t *)wg_base, gran_start, wg_k, hdr, wg_ext, pair_in, pair_out, ext_total);
  return check_launch(what);
}

extern "C" int link_pair_plan_layout(const int32_t *wg_counts, int64_t n, int32_t kvol, int32_t skip_centre, int64_t gran_cap,
                                     int32_t *base_k, int32_t *wg_base, int32_t *gran_start, int32_t *wg_k, int32_t *hdr,
                                     void *stream) {
  if (n < 0 || kvol <= 0 || kvol > 64 || gran_cap < 0) return LINK_ERR_ARG;
  if (!wg_counts || !base_k || !wg_base || !gran_start || !hdr || (gran_cap > 0 && !wg_k)) return LINK_ERR_ARG;
  const int64_t nwg = (n + 255) / 256;
  if (nwg >= (1LL << 31)) return LINK_ERR_ARG;
  return pair_plan_layout_run(wg_counts, nwg, kvol, skip_centre, gran_cap, base_k, wg_base, gran_start, wg_k, hdr, nullptr, nullptr, nullptr,
                              nullptr, S(stream), "link_pair_plan_layout");
}

// ---------------------------------------------------------------------------------------------
// output sites of a site-creating (regular) sparse convolution: candidate rows
// ---------------------------------------------------------------------------------------------
// A site o exists when some active input i and tap a satisfy o * s = i + p - a (per axis).  Per axis an input
// offers few candidates: kernel 3 / stride 2 -> floor((i+p)/2) and, when i+p is even, one less; otherwise one
// per tap whose numerator divides.  Thread (input, combination) writes its candidate row (b, z, y, x) -- or a
// row of -1 when a factor is invalid or outside the output shape; duplicates and the -1 rows are what the
// dense-grid block index (link_index_build with block edge 1) is built to drop, and its sorted unique rows are
// the output sites.
struct conv_geom { int k[3], s[3], p[3], oshape[3], slots[3]; };

__global__ void __launch_bounds__(256) k_conv_out_candidates(const int4 *__restrict__ ind, int64_t n, conv_geom g, int ncomb,
                                                             int4 *__restrict__ cand) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * ncomb) return;
  const int64_t i = t / ncomb;
  int c = (int)(t - i * ncomb);
  const int4 r = ind[i];                               // (b, z, y, x)
  const int pos[3] = {r.y, r.z, r.w};
  int o[3];
  bool ok = true;
#pragma unroll
  for (int d = 2; d >= 0; d--) {
    const int j = c % g.slots[d];
    c /= g.slots[d];
    const int v = pos[d] + g.p[d];
    if (g.k[d] == 3 && g.s[d] == 2) {
      o[d] = (v >> 1) - j;
      ok &= (j == 0) || ((v & 1) == 0);
    } else {
      const int num = v - j;                           // tap j
      o[d] = num / g.s[d];
      ok &= num >= 0 && num % g.s[d] == 0;
    }
    ok &= o[d] >= 0 && o[d] < g.oshape[d];
  }
  cand[t] = ok ? make_int4(r.x, o[0], o[1], o[2]) : make_int4(-1, -1, -1, -1);
}

extern "C" int link_conv_out_candidates(const int32_t *indices, int64_t n, const int32_t *kernel, const int32_t *stride,
                                        const int32_t *padding, const int32_t *out_shape, int32_t *cand, void *stream) {
  if (n < 0 || !kernel || !stride || !padding || !out_shape) return LINK_ERR_ARG;
  conv_geom g;
  int ncomb = 1;
  for (int d = 0; d < 3; d++) {
    g.k[d] = kernel[d]; g.s[d] = stride[d]; g.p[d] = padding[d]; g.oshape[d] = out_shape[d];
    if ((g.k[d] != 1 && g.k[d] != 3) || (g.s[d] != 1 && g.s[d] != 2) || g.p[d] < 0 || g.oshape[d] <= 0) return LINK_ERR_ARG;
    g.slots[d] = (g.k[d] == 3 && g.s[d] == 2) ? 2 : g.k[d];
    ncomb *= g.slots[d];
  }
  if (n == 0) return LINK_OK;
  if (!indices || !cand || n * ncomb >= (1LL << 31)) return LINK_ERR_ARG;
  const int64_t total = n * ncomb;
  hipLaunchKernelGGL(k_conv_out_candidates, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const int4 *>(indices), n, g, ncomb, reinterpret_cast<int4 *>(cand));
  return check_launch("link_conv_out_candidates");
}

// count -> layout -> fill in one call (what link_amd's device-laid-out plans run): the fill pass computes the CSR starts
// itself and the layout pass writes the padding tails, so no prefix-sum pass and no -1 fill of the capacity-sized lists.
extern "C" int link_pair_plan_build(const int32_t *nbr, int64_t n, int32_t kvol, int32_t skip_centre, int64_t gran_cap,
                                    int32_t *wg_counts, int32_t *row_info, int32_t *base_k, int32_t *wg_base, int32_t *gran_start,
                                    int32_t *wg_ext, int32_t *wg_k, int32_t *hdr, int32_t *ext_start, int32_t *pair_in,
                                    int32_t *pair_out, int32_t *ext_list, void *stream) {
  if (n <= 0 || kvol <= 0 || kvol > 64 || gran_cap <= 0) return LINK_ERR_ARG;
  if (!nbr || !wg_counts || !row_info || !base_k || !wg_base || !gran_start || !wg_ext || !wg_k || !hdr || !ext_start || !pair_in ||
      !pair_out || !ext_list)
    return LINK_ERR_ARG;
  const int64_t nwg = (n + 255) / 256;
  if (nwg >= (1LL << 31)) return LINK_ERR_ARG;
  // pair_in / pair_out hold gran_cap granules of 128 rows: every (voxel, offset) may be a pair and every offset's last granule
  // partly filled -- a smaller capacity would be written past (the header's overflow flag is raised only afterwards)
  if (gran_cap < (n * (int64_t)(kvol - (skip_centre ? 1 : 0)) + 127 * (int64_t)kvol + 127) / 128) return LINK_ERR_ARG;
  int rc = link_pair_plan_count(nbr, n, kvol, wg_counts, row_info, stream);
  if (rc != LINK_OK) return rc;
  rc = pair_plan_layout_run(wg_counts, nwg, kvol, skip_centre, gran_cap, base_k, wg_base, gran_start, wg_k, hdr, wg_ext, pair_in, pair_out,
                            ext_start + n, S(stream), "link_pair_plan_build");
  if (rc != LINK_OK) return rc;
  hipLaunchKernelGGL(k_pair_plan<true>, dim3((unsigned)nwg), dim3(256), (size_t)(256 * kvol + 4 * (kvol + 1)) * 4, S(stream), nbr, n,
                     (int)kvol, (int)(kvol / 2), (int)skip_centre, base_k, wg_base, (const int32_t *)nullptr, (int32_t *)nullptr,
                     (int32_t *)nullptr, pair_in, pair_out, ext_list, wg_ext, ext_start);
  return check_launch("link_pair_plan_build");
}

namespace link {
// link_pair_plan_build behind a count pass somebody else has done (the block driver's neighbour-map kernel counts on the tile
// it holds: dense_fused.hip, k_dc_neighbor_map<true>): layout + fill
int pair_plan_build_counted(const int32_t *nbr, int64_t n, int32_t kvol, int32_t skip_centre, int64_t gran_cap, const int32_t *wg_counts,
                            int32_t *base_k, int32_t *wg_base, int32_t *gran_start, int32_t *wg_ext, int32_t *wg_k, int32_t *hdr,
                            int32_t *ext_start, int32_t *pair_in, int32_t *pair_out, int32_t *ext_list, hipStream_t st) {
  if (n <= 0 || kvol <= 0 || kvol > 64 || gran_cap <= 0) return LINK_ERR_ARG;
  const int64_t nwg = (n + 255) / 256;
  if (gran_cap < (n * (int64_t)(kvol - (skip_centre ? 1 : 0)) + 127 * (int64_t)kvol + 127) / 128) return LINK_ERR_ARG;
  int rc = pair_plan_layout_run(wg_counts, nwg, kvol, skip_centre, gran_cap, base_k, wg_base, gran_start, wg_k, hdr, wg_ext, pair_in, pair_out,
                                ext_start + n, st, "link_elk_block_forward (pair plan)");
  if (rc != LINK_OK) return rc;
  hipLaunchKernelGGL(k_pair_plan<true>, dim3((unsigned)nwg), dim3(256), (size_t)(256 * kvol + 4 * (kvol + 1)) * 4, st, nbr, n, (int)kvol,
                     (int)(kvol / 2), (int)skip_centre, base_k, wg_base, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
                     pair_in, pair_out, ext_list, wg_ext, ext_start);
  return check_launch("link_elk_block_forward (pair plan)");
}
}  // namespace link

// ---------------------------------------------------------------------------------------------
// gather table of a site-creating convolution, straight from the (b, z, y, x) rows
// ---------------------------------------------------------------------------------------------
// site table: cell (b, z, y, x) of the input shape -> input row + 1 (0 = no site); the smallest row wins a duplicate.
// CLEAR: undo it (the table lives in the caller's workspace and stays all zero between uses).
template <bool CLEAR>
__global__ void __launch_bounds__(256) k_conv_site_table(const int4 *__restrict__ ind, int64_t n, int Z, int Y, int X, int B,
                                                         unsigned int *__restrict__ table) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int4 r = ind[i];                               // (b, z, y, x)
  if ((unsigned)r.x >= (unsigned)B || (unsigned)r.y >= (unsigned)Z || (unsigned)r.z >= (unsigned)Y || (unsigned)r.w >= (unsigned)X) return;
  const int64_t cell = (((int64_t)r.x * Z + r.y) * Y + r.z) * X + r.w;
  if (CLEAR) { table[cell] = 0u; return; }
  const unsigned v = (unsigned)i + 1u;
  unsigned old = atomicCAS(&table[cell], 0u, v);
  while (old != 0u && old > v) {
    const unsigned prev = atomicCAS(&table[cell], old, v);
    if (prev == old) break;
    old = prev;
  }
}

// table[j, t] = input row at out[j] * stride - padding + tap_t (per axis; taps (a, b, c) row-major over the kernel), -1 absent
__global__ void __launch_bounds__(256) k_conv_gather_table(const int4 *__restrict__ out_ind, int64_t m, conv_geom g, int Z, int Y,
                                                           int X, int B, const unsigned int *__restrict__ site,
                                                           int32_t *__restrict__ table) {
  const int ntap = g.k[0] * g.k[1] * g.k[2];
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= m * ntap) return;
  const int64_t j = t / ntap;
  int tap = (int)(t - j * ntap);
  const int c = tap % g.k[2]; tap /= g.k[2];
  const int b = tap % g.k[1];
  const int a = tap / g.k[1];
  const int4 o = out_ind[j];                           // (b, z, y, x)
  const int z = o.y * g.s[0] - g.p[0] + a, y = o.z * g.s[1] - g.p[1] + b, x = o.w * g.s[2] - g.p[2] + c;
  int32_t v = -1;
  if ((unsigned)o.x < (unsigned)B && (unsigned)z < (unsigned)Z && (unsigned)y < (unsigned)Y && (unsigned)x < (unsigned)X)
    v = (int32_t)site[(((int64_t)o.x * Z + z) * Y + y) * X + x] - 1;
  table[t] = v;
}

extern "C" int link_conv_site_table(const int32_t *indices, int64_t n, const int32_t *in_shape, int32_t batch, int32_t *table,
                                    int32_t clear, void *stream) {
  if (n < 0 || !in_shape || batch <= 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!indices || !table || in_shape[0] <= 0 || in_shape[1] <= 0 || in_shape[2] <= 0) return LINK_ERR_ARG;
  if ((int64_t)batch * in_shape[0] * in_shape[1] * in_shape[2] >= (1LL << 31)) return LINK_ERR_ARG;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (clear)
    hipLaunchKernelGGL(k_conv_site_table<true>, grid, dim3(256), 0, S(stream), reinterpret_cast<const int4 *>(indices), n,
                       in_shape[0], in_shape[1], in_shape[2], batch, reinterpret_cast<unsigned int *>(table));
  else
    hipLaunchKernelGGL(k_conv_site_table<false>, grid, dim3(256), 0, S(stream), reinterpret_cast<const int4 *>(indices), n,
                       in_shape[0], in_shape[1], in_shape[2], batch, reinterpret_cast<unsigned int *>(table));
  return check_launch("link_conv_site_table");
}

extern "C" int link_conv_gather_table(const int32_t *out_indices, int64_t m, const int32_t *kernel, const int32_t *stride,
                                      const int32_t *padding, const int32_t *in_shape, int32_t batch, const int32_t *site_table,
                                      int32_t *table, void *stream) {
  if (m < 0 || !kernel || !stride || !padding || !in_shape || batch <= 0) return LINK_ERR_ARG;
  if (m == 0) return LINK_OK;
  if (!out_indices || !site_table || !table) return LINK_ERR_ARG;
  conv_geom g = {};
  for (int a = 0; a < 3; a++) {
    if (kernel[a] < 1 || kernel[a] > 3 || stride[a] < 1) return LINK_ERR_ARG;
    g.k[a] = kernel[a]; g.s[a] = stride[a]; g.p[a] = padding[a];
  }
  if ((int64_t)batch * in_shape[0] * in_shape[1] * in_shape[2] >= (1LL << 31)) return LINK_ERR_ARG;
  const int ntap = g.k[0] * g.k[1] * g.k[2];
  hipLaunchKernelGGL(k_conv_gather_table, dim3((unsigned)((m * ntap + 255) / 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const int4 *>(out_indices), m, g, in_shape[0], in_shape[1], in_shape[2], batch,
                     reinterpret_cast<const unsigned int *>(site_table), table);
  return check_launch("link_conv_gather_table");
}

extern "C" int32_t link_conv_out_candidate_count(const int32_t *kernel, const int32_t *stride) {
  int ncomb = 1;
  for (int d = 0; d < 3; d++) ncomb *= (kernel[d] == 3 && stride[d] == 2) ? 2 : kernel[d];
  return ncomb;
}

// ---------------------------------------------------------------------------------------------
// weight gradient over the pair list: g_w[k] = sum over the pairs of offset k of feats[in]^T . g_out[out]
// ---------------------------------------------------------------------------------------------
// (the weight half of convolution_backward_cuda, convolution_cuda.cu:167-278: per offset gather + cuBLAS mm(in^T,
// grad)).  A workgroup walks WGRAD_RUN consecutive 128-pair granules: both row sets of a granule are staged in LDS
// (padding pairs as zero rows), wave w accumulates the 16-row strip ci in [16w, 16w+16) of the [CI x CO] product with
// the pairs as the MFMA k-dimension, and a strip goes to a partial slot whenever the offset changes or the walk ends
// -- the per-offset sums are formed afterwards in slot order by k_conv_pairs_wgrad_sum: fixed order, no atomics, any
// width pair the forward kernels take.
constexpr int WGRAD_RUN = 4;                           // granules one workgroup folds into a partial (same offset only)

template <int CI, int CO>
__global__ void __launch_bounds__(64 * (CI / 16)) k_conv_pairs_wgrad(const float *__restrict__ feats, const float *__restrict__ gout,
                                                                     const int32_t *__restrict__ pair_in,
                                                                     const int32_t *__restrict__ pair_out,
                                                                     const int32_t *__restrict__ wg_k, int64_t granules,
                                                                     int64_t plan_granules, int centre, int64_t n_rows,
                                                                     float *__restrict__ partial) {
  // granules [plan_granules, granules): the identity pairs (i -> i) of a submanifold table's centre offset
  constexpr int NW = CI / 16, NT = 64 * NW, TO = CO / 16;
  constexpr int LDF = CI + 16, LDG = CO + 16;          // row strides: 16 (mod 32) banks apart for the 4 pairs of a k-step
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *fl = reinterpret_cast<float *>(smem_raw);     // [128][LDF]
  float *gl = fl + 128 * LDF;                          // [128][LDG]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g4 = lane >> 4;
  floatx4 acc[TO];
#pragma unroll
  for (int tt = 0; tt < TO; tt++) acc[tt] = (floatx4){0.f, 0.f, 0.f, 0.f};
  const float *fa = fl + g4 * LDF + 16 * wave + li;    // A[ci = 16 wave + li][pair = 4 q + g4]
  const float *gb = gl + g4 * LDG + li;                // B[pair = 4 q + g4][co = 16 tt + li]
  auto flush = [&](int64_t slot) {                     // D[ci = 16 wave + 4 g4 + r][co = 16 tt + li]
    float *dst = partial + slot * CI * CO + (16 * wave + 4 * g4) * CO + li;
#pragma unroll
    for (int tt = 0; tt < TO; tt++) {
#pragma unroll
      for (int r = 0; r < 4; r++) dst[r * CO + 16 * tt] = acc[tt][r];
      acc[tt] = (floatx4){0.f, 0.f, 0.f, 0.f};
    }
  };
  const int64_t g0 = (int64_t)blockIdx.x * WGRAD_RUN;
  int64_t run = g0;                                    // first granule of the run being accumulated = its partial slot
  int cur = g0 < plan_granules ? wg_k[g0] : centre;
  for (int64_t gi = g0; gi < g0 + WGRAD_RUN && gi < granules; gi++) {
    const bool ident = gi >= plan_granules;
    const int k = ident ? centre : wg_k[gi];
    if (k != cur || gi == plan_granules) { if (gi != g0) flush(run); run = gi; cur = k; }
    const int64_t row0 = gi * 128;
    const int64_t id0 = (gi - plan_granules) * 128;
    __syncthreads();                                   // the previous granule's operands are consumed
    for (int e = tid; e < 128 * (CI / 4); e += NT) {
      const int pr = e / (CI / 4), c4 = e - pr * (CI / 4);
      const int j = ident ? (id0 + pr < n_rows ? (int)(id0 + pr) : -1) : pair_in[row0 + pr];
      const float4 v = j >= 0 ? *reinterpret_cast<const float4 *>(feats + (int64_t)j * CI + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(&fl[pr * LDF + 4 * c4]) = v;
    }
    for (int e = tid; e < 128 * (CO / 4); e += NT) {
      const int pr = e / (CO / 4), c4 = e - pr * (CO / 4);
      const int j = ident ? (id0 + pr < n_rows ? (int)(id0 + pr) : -1) : pair_out[row0 + pr];
      const float4 v = j >= 0 ? *reinterpret_cast<const float4 *>(gout + (int64_t)j * CO + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4 *>(&gl[pr * LDG + 4 * c4]) = v;
    }
    __syncthreads();
#pragma unroll 4
    for (int q = 0; q < 32; q++) {
      const float a = fa[4 * q * LDF];
#pragma unroll
      for (int tt = 0; tt < TO; tt++) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, gb[4 * q * LDG + 16 * tt], acc[tt], 0, 0, 0);
    }
  }
  flush(run);
}

// g_w[k][e] = sum of the partial slots of offset k in granule order: a slot exists at the offset's first granule and at
// every later multiple of WGRAD_RUN (where a workgroup started a run); gran_start i32[kvol + 1]
__global__ void __launch_bounds__(256) k_conv_pairs_wgrad_sum(const float *__restrict__ partial, const int32_t *__restrict__ gran_start,
                                                              int kvol, int elems, int id_k, int id_first, int id_end,
                                                              float *__restrict__ gw) {
  const int k = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= elems) return;
  float s = 0.f;
  const int first = gran_start[k], end = gran_start[k + 1];
  for (int gi = first; gi < end; gi = (gi == first) ? (first / WGRAD_RUN + 1) * WGRAD_RUN : gi + WGRAD_RUN)
    s += partial[(int64_t)gi * elems + e];
  if (k == id_k)                                       // the identity granules of the centre offset (their own slot range)
    for (int gi = id_first; gi < id_end; gi = (gi == id_first) ? (id_first / WGRAD_RUN + 1) * WGRAD_RUN : gi + WGRAD_RUN)
      s += partial[(int64_t)gi * elems + e];
  gw[(int64_t)k * elems + e] = s;
}

template <int CI, int CO>
static int launch_pairs_wgrad(const float *feats, const float *gout, const int32_t *pair_in, const int32_t *pair_out,
                              const int32_t *wg_k, int64_t granules, int64_t plan_granules, int centre, int64_t n_rows,
                              float *partial, hipStream_t st) {
  const size_t lds = (size_t)128 * ((CI + 16) + (CO + 16)) * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_conv_pairs_wgrad<CI, CO>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_conv_pairs_wgrad<CI, CO>), dim3((unsigned)((granules + WGRAD_RUN - 1) / WGRAD_RUN)), dim3(64 * (CI / 16)), lds,
                     st, feats, gout, pair_in, pair_out, wg_k, granules, plan_granules, centre, n_rows, partial);
  return check_launch("link_conv_pairs_wgrad");
}

extern "C" int link_conv_pairs_wgrad(const float *feats, const float *gout, const int32_t *pair_in, const int32_t *pair_out,
                                     const int32_t *wg_k, const int32_t *gran_start, int64_t rows_pad, int32_t kvol,
                                     int64_t n_direct, int32_t cin, int32_t cout, float *partial, float *gw, void *stream) {
  if (rows_pad < 0 || (rows_pad & 127) || kvol <= 0 || n_direct < 0 || !link_conv_pairs_supported(cin, cout)) return LINK_ERR_ARG;
  if (!gw || !gran_start) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  const int64_t pg = rows_pad / 128;
  const int64_t gr = pg + (n_direct + 127) / 128;      // + identity granules of the centre offset (submanifold tables)
  int rc = LINK_OK;
  if (gr > 0) {
    if (!feats || !gout || !partial || (pg > 0 && (!pair_in || !pair_out || !wg_k))) return LINK_ERR_ARG;
    rc = LINK_ERR_ARG;
#define LINK_CW(I, O) \
  if (cin == I && cout == O) rc = launch_pairs_wgrad<I, O>(feats, gout, pair_in, pair_out, wg_k, gr, pg, (int)(kvol / 2), n_direct, partial, st)
    LINK_CW(16, 16); LINK_CW(32, 32); LINK_CW(64, 64); LINK_CW(128, 128);
    LINK_CW(16, 32); LINK_CW(32, 16); LINK_CW(32, 64); LINK_CW(64, 32); LINK_CW(64, 128); LINK_CW(128, 64);
    LINK_CW(16, 64); LINK_CW(64, 16);
#undef LINK_CW
    if (rc != LINK_OK) return rc;
  }
  const int elems = cin * cout;
  hipLaunchKernelGGL(k_conv_pairs_wgrad_sum, dim3((unsigned)((elems + 255) / 256), (unsigned)kvol), dim3(256), 0, st, partial, gran_start,
                     (int)kvol, elems, n_direct > 0 ? (int)(kvol / 2) : -1, (int)pg, (int)gr, gw);
  return check_launch("link_conv_pairs_wgrad_sum");
}
