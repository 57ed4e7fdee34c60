// link_amd/csrc/conv.hip -- row N1 of SURVEY.md section 8f: the stride-1 submanifold sparse convolution
// that ELKBlock.local_mix runs (spnn.Conv3d(inc, inc, kernel_size=3), linkunet.py:109,125), for gfx950.
//
// The reference (torchsparse-u/torchsparse/backend/convolution/convolution_cuda.cu:53-165) loops over the
// 27 kernel offsets and, per offset, gathers the participating input rows into a buffer, calls cuBLAS,
// and scatter-adds the result: 27 x (gather kernel + GEMM + scatter kernel), every intermediate
// through HBM.  Here it is ONE output-stationary kernel: a wave owns NT tiles of 16 output voxels and
// keeps their [16 x C] accumulators in registers across the whole offset loop; per offset the
// workgroup stages W_k (transposed, padded) in LDS once for all its 4*NT tiles, each lane gathers the
// neighbour row of its voxel straight into the MFMA B-operand layout (one dwordx4 per 16-channel
// tile), and v_mfma_f32_16x16x4_f32 accumulates D[co][voxel] += W_k^T[co][ci] * F[nbr][ci] (exact f32).
// The offset loop is software-pipelined: W_next travels global -> registers -> the other LDS buffer
// while W_k feeds the MFMAs (one barrier per offset), and the wave's 27 x 16 x NT neighbour ids sit in LDS.
// Tiles in which no voxel has the offset's neighbour skip the MFMAs, and an offset no tile of the
// workgroup needs skips the W_k staging too; `order` (optional) lists the voxels in a spatially coherent
// order so that the 16 voxels of a tile share their present/absent pattern.  Output is written once.
//
// out[v, co] = sum_k sum_ci feats[nbr[v, k], ci] * w[k, ci, co]          (nbr < 0: absent)
#include "common.h"

using namespace link;

typedef float floatx4 __attribute__((ext_vector_type(4)));

// optional fused epilogue (row N2 of SURVEY.md section 8f; linkunet.py:183 / ts_elk.py:228):
// out = relu(addend + LayerNorm(conv) * ln_w + ln_b); ln_w == NULL -> plain convolution output
struct conv_epilogue {
  const float *ln_w, *ln_b, *addend;
  float eps;
  int relu;
};

// Store phase shared by the matrix-core kernels: the accumulator layout holds 16 channels of ONE voxel per lane group
// pass (lane (li, g): voxel li, channels 16 tp + 4 g .. + 3), so LayerNorm statistics are in-lane adds + 2 cross-lane steps.
// row N2: out = relu(addend + LayerNorm(conv) * ln_w + ln_b); bit 1 of ep.relu: ln_w / ln_b are a plain per-channel affine
// (folded BatchNorm, scn.py:486-489), no statistics; ln_w == NULL: the plain convolution output.
template <int CO>
__device__ __forceinline__ void conv_finish(const floatx4 (&acc)[CO / 16], int64_t v, int g, const conv_epilogue &ep,
                                            float *__restrict__ out) {
  constexpr int TO = CO / 16;
  if (ep.ln_w) {
    float mean = 0.f, rstd = 1.f;
    if (!(ep.relu & 2)) {
      float sm = 0.f;
#pragma unroll
      for (int tp = 0; tp < TO; tp++) sm += (acc[tp][0] + acc[tp][1]) + (acc[tp][2] + acc[tp][3]);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      mean = sm * (1.0f / CO);
      float q = 0.f;
#pragma unroll
      for (int tp = 0; tp < TO; tp++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float d = acc[tp][r] - mean;
          q += d * d;
        }
      q += __shfl_xor(q, 16, 64);
      q += __shfl_xor(q, 32, 64);
      rstd = 1.0f / sqrtf(q * (1.0f / CO) + ep.eps);
    }
    if (v >= 0) {
#pragma unroll
      for (int tp = 0; tp < TO; tp++) {
        const int ch = 16 * tp + 4 * g;
        const float4 lw = *reinterpret_cast<const float4 *>(&ep.ln_w[ch]);
        const float4 lb = *reinterpret_cast<const float4 *>(&ep.ln_b[ch]);
        float4 o;
        o.x = (acc[tp][0] - mean) * rstd * lw.x + lb.x;
        o.y = (acc[tp][1] - mean) * rstd * lw.y + lb.y;
        o.z = (acc[tp][2] - mean) * rstd * lw.z + lb.z;
        o.w = (acc[tp][3] - mean) * rstd * lw.w + lb.w;
        if (ep.addend) {
          const float4 a4 = *reinterpret_cast<const float4 *>(&ep.addend[v * CO + ch]);
          o.x = a4.x + o.x; o.y = a4.y + o.y; o.z = a4.z + o.z; o.w = a4.w + o.w;
        }
        if (ep.relu & 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4 *>(&out[v * CO + ch]) = o;
      }
    }
  } else if (v >= 0) {
#pragma unroll
    for (int tp = 0; tp < TO; tp++)
      *reinterpret_cast<float4 *>(&out[v * CO + 16 * tp + 4 * g]) = make_float4(acc[tp][0], acc[tp][1], acc[tp][2], acc[tp][3]);
  }
}

template <int CI, int CO, int NT, bool DEEP>
__global__ void __launch_bounds__(256) k_subm_conv_mfma(const float *__restrict__ feats,
                                                        const int32_t *__restrict__ nbr,
                                                        const float *__restrict__ w,
                                                        const int32_t *__restrict__ order, int64_t n,
                                                        int kvol, float *__restrict__ out, conv_epilogue ep) {
  constexpr int TI = CI / 16, TO = CO / 16;         // 16-channel tiles of the input (MFMA k-steps) / output rows
  constexpr int LDW = CI + 4;
  constexpr int WREG = (CI * CO + 1023) / 1024;     // float4 of W_k per thread (last one guarded)
  constexpr int KMAX = 27;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *wt_lds = reinterpret_cast<float *>(smem_raw);          // 2 x W_k^T [co][ci], row stride LDW
  int32_t *nb_lds = reinterpret_cast<int32_t *>(wt_lds + 2 * CO * LDW);   // [4 waves][KMAX][NT*16] neighbour ids
  __shared__ uint32_t wg_mask;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  int32_t *my_nb = nb_lds + wave * (KMAX * NT * 16);
  const int64_t tiles = (n + 15) / 16;
  const int64_t tiles_per_pass = (int64_t)gridDim.x * 4 * NT;
  for (int64_t base = 0; base < tiles; base += tiles_per_pass) {
    const int64_t tile0 = base + ((int64_t)blockIdx.x * 4 + wave) * NT;
    floatx4 acc[NT][TO];
    int64_t vox[NT];                                // this lane's output voxel of each tile (-1: past the end)
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int64_t q = (tile0 + j) * 16 + li;
      vox[j] = (q < n) ? (order ? (int64_t)order[q] : q) : -1;
#pragma unroll
      for (int tp = 0; tp < TO; tp++) acc[j][tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
    }
    if (tid == 0) wg_mask = 0u;
    // the wave's neighbour ids -> LDS in one round trip (the 4 quarter-waves split the offsets)
#pragma unroll
    for (int j = 0; j < NT; j++)
      for (int k = g; k < kvol; k += 4)
        my_nb[(k * NT + j) * 16 + li] = (vox[j] >= 0) ? nbr[vox[j] * kvol + k] : -1;
    __syncthreads();                                // ids visible; wg_mask zeroed; previous pass done with LDS
    uint32_t wmask = 0u;                            // offsets for which some voxel of this wave has a neighbour
    for (int k = 0; k < kvol; k++) {
      bool need = false;
#pragma unroll
      for (int j = 0; j < NT; j++) need |= my_nb[(k * NT + j) * 16 + li] >= 0;
      if (__any(need)) wmask |= 1u << k;
    }
    if (lane == 0 && wmask) atomicOr(&wg_mask, wmask);
    __syncthreads();
    uint32_t m = wg_mask;                           // offsets the workgroup computes; all others skipped outright
    if (m == 0u) {                                  // (uniform) nothing to do: rows of zeros
      __syncthreads();
    } else if constexpr (DEEP) {
      // Small frames (a handful of tiles per wave): a step's MFMAs are over in a few hundred cycles, so the offset loop
      // is a chain of load latencies.  Everything is requested two steps ahead: W of offsets k+1 / k+2 travel in two
      // register sets (one is written to the idle LDS buffer per step), and the neighbour rows of offset k+1 are
      // gathered while offset k is multiplied.  NT == 1.
      static_assert(!DEEP || NT == 1, "deep pipeline: one tile per wave");
      auto next_off = [&]() { const int kk = m ? (__ffs(m) - 1) : -1; m &= m - 1; return kk; };
      auto ld_w = [&](int kk, float4 (&wr)[WREG]) {
        const float *wk = w + (int64_t)(kk < 0 ? 0 : kk) * CI * CO;
#pragma unroll
        for (int q = 0; q < WREG; q++)
          if ((q * 256 + tid) * 4 < CI * CO) wr[q] = *reinterpret_cast<const float4 *>(&wk[(q * 256 + tid) * 4]);
      };
      auto st_w = [&](int b, const float4 (&wr)[WREG]) {
        float *dst = wt_lds + b * (CO * LDW);
#pragma unroll
        for (int q = 0; q < WREG; q++) {
          const int e = (q * 256 + tid) * 4, ci = e / CO, co = e - ci * CO;
          if (e < CI * CO) {
            dst[(co + 0) * LDW + ci] = wr[q].x; dst[(co + 1) * LDW + ci] = wr[q].y;
            dst[(co + 2) * LDW + ci] = wr[q].z; dst[(co + 3) * LDW + ci] = wr[q].w;
          }
        }
      };
      auto ld_f = [&](int kk, int &id, float4 (&ff)[TI]) {
        id = kk >= 0 ? my_nb[(kk * NT) * 16 + li] : -1;
        const int64_t row = id >= 0 ? id : 0;
#pragma unroll
        for (int tt = 0; tt < TI; tt++) ff[tt] = *reinterpret_cast<const float4 *>(&feats[row * CI + 16 * tt + 4 * g]);
      };
      auto mma = [&](int b, int id, const float4 (&ff)[TI]) {
        const bool has = id >= 0;
        if (!__any(has)) return;                      // wave-uniform
        const float *wt = wt_lds + b * (CO * LDW);
#pragma unroll
        for (int tt = 0; tt < TI; tt++) {
          const float fx = has ? ff[tt].x : 0.f, fy = has ? ff[tt].y : 0.f, fz = has ? ff[tt].z : 0.f, fw = has ? ff[tt].w : 0.f;
#pragma unroll
          for (int tp = 0; tp < TO; tp++) {
            const float4 a = *reinterpret_cast<const float4 *>(&wt[(16 * tp + li) * LDW + 16 * tt + 4 * g]);
            acc[0][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, fx, acc[0][tp], 0, 0, 0);
            acc[0][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, fy, acc[0][tp], 0, 0, 0);
            acc[0][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, fz, acc[0][tp], 0, 0, 0);
            acc[0][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, fw, acc[0][tp], 0, 0, 0);
          }
        }
      };
      float4 wA[WREG], wB[WREG], fA[TI], fB[TI];
      int idA = -1, idB = -1;
      int k0 = next_off(), k1 = next_off(), k2 = next_off();
      ld_w(k0, wA);
      ld_f(k0, idA, fA);
      if (k1 >= 0) ld_w(k1, wB);
      st_w(0, wA);                                    // W_k0 -> buffer 0
      if (k2 >= 0) ld_w(k2, wA);                      // set A is free again: W_k2
      __syncthreads();
      // invariant at the top of a step on (k0, buffer b): W_k0 in LDS buffer b; W_k1 in flight / in set X; W_k2 in
      // flight in set Y; rows of k0 in the current row set
      int b = 0;
      while (true) {
        // -- even step: rows in fA, W_k1 in wB, W_k2 in wA
        if (k1 >= 0) ld_f(k1, idB, fB);
        mma(b, idA, fA);
        if (k1 < 0) break;
        st_w(b ^ 1, wB);
        { const int k3 = next_off(); if (k3 >= 0) ld_w(k3, wB); k0 = k1; k1 = k2; k2 = k3; }
        __syncthreads();
        b ^= 1;
        // -- odd step: rows in fB, W_k1 in wA, W_k2 in wB
        if (k1 >= 0) ld_f(k1, idA, fA);
        mma(b, idB, fB);
        if (k1 < 0) break;
        st_w(b ^ 1, wA);
        { const int k3 = next_off(); if (k3 >= 0) ld_w(k3, wA); k0 = k1; k1 = k2; k2 = k3; }
        __syncthreads();
        b ^= 1;
      }
      __syncthreads();                               // LDS free for the next pass
    } else {
      // software pipeline over the needed offsets: W_next travels global -> registers while W_k is used
      float4 wreg[WREG];
      int k = __ffs(m) - 1;
      m &= m - 1;
      {
        const float *wk = w + (int64_t)k * CI * CO;
#pragma unroll
        for (int q = 0; q < WREG; q++)
          if ((q * 256 + tid) * 4 < CI * CO) wreg[q] = *reinterpret_cast<const float4 *>(&wk[(q * 256 + tid) * 4]);
      }
      int buf = 0;
      {
        float *dst = wt_lds;
#pragma unroll
        for (int q = 0; q < WREG; q++) {
          const int e = (q * 256 + tid) * 4, ci = e / CO, co = e - ci * CO;
          if (e < CI * CO) {
            dst[(co + 0) * LDW + ci] = wreg[q].x; dst[(co + 1) * LDW + ci] = wreg[q].y;
            dst[(co + 2) * LDW + ci] = wreg[q].z; dst[(co + 3) * LDW + ci] = wreg[q].w;
          }
        }
      }
      __syncthreads();
      while (true) {
        const int kn = m ? (__ffs(m) - 1) : -1;      // next needed offset (uniform)
        m &= m - 1;
        if (kn >= 0) {
          const float *wk = w + (int64_t)kn * CI * CO;
#pragma unroll
          for (int q = 0; q < WREG; q++)
          if ((q * 256 + tid) * 4 < CI * CO) wreg[q] = *reinterpret_cast<const float4 *>(&wk[(q * 256 + tid) * 4]);
        }
        if ((wmask >> k) & 1u) {                     // wave-uniform: this wave has work at offset k
          const float *wt = wt_lds + buf * (CO * LDW);
          int id[NT];
#pragma unroll
          for (int j = 0; j < NT; j++) id[j] = my_nb[(k * NT + j) * 16 + li];
          float4 f[NT][TI];
#pragma unroll
          for (int j = 0; j < NT; j++) {            // all gathers of the step back to back
            const int64_t row = (id[j] >= 0) ? id[j] : 0;
#pragma unroll
            for (int t = 0; t < TI; t++) f[j][t] = *reinterpret_cast<const float4 *>(&feats[row * CI + 16 * t + 4 * g]);
          }
#pragma unroll
          for (int j = 0; j < NT; j++) {
            const bool has = id[j] >= 0;
            if (!__any(has)) continue;              // wave-uniform: no voxel of the tile has neighbour k
#pragma unroll
            for (int t = 0; t < TI; t++) {
              const float fx = has ? f[j][t].x : 0.f, fy = has ? f[j][t].y : 0.f;
              const float fz = has ? f[j][t].z : 0.f, fw = has ? f[j][t].w : 0.f;
#pragma unroll
              for (int tp = 0; tp < TO; tp++) {
                const float4 a = *reinterpret_cast<const float4 *>(&wt[(16 * tp + li) * LDW + 16 * t + 4 * g]);
                acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, fx, acc[j][tp], 0, 0, 0);
                acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, fy, acc[j][tp], 0, 0, 0);
                acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, fz, acc[j][tp], 0, 0, 0);
                acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, fw, acc[j][tp], 0, 0, 0);
              }
            }
          }
        }
        if (kn < 0) break;
        {                                            // W_next -> the other LDS buffer (nobody reads it now)
          float *dst = wt_lds + (buf ^ 1) * (CO * LDW);
#pragma unroll
          for (int q = 0; q < WREG; q++) {
            const int e = (q * 256 + tid) * 4, ci = e / CO, co = e - ci * CO;
            if (e < CI * CO) {
              dst[(co + 0) * LDW + ci] = wreg[q].x; dst[(co + 1) * LDW + ci] = wreg[q].y;
              dst[(co + 2) * LDW + ci] = wreg[q].z; dst[(co + 3) * LDW + ci] = wreg[q].w;
            }
          }
        }
        __syncthreads();                             // ONE barrier per offset
        buf ^= 1;
        k = kn;
      }
      __syncthreads();                               // LDS free for the next pass
    }
#pragma unroll
    for (int j = 0; j < NT; j++) conv_finish<CO>(acc[j], vox[j], g, ep, out);
  }
}

// ---------------------------------------------------------------------------------------------
// Narrow layers (C = 16 / 32: the first two stages of the detection backbone, scn.py:467-470, the stem of the
// segmentation encoders): ALL 27 W_k fit in LDS at once (27 x 16 x 16 floats = 27 KB, 27 x 32 x 32 = 108 KB), so the
// workgroup stages them ONCE and its waves walk tiles of 16 output voxels with no barrier and no weight traffic in the
// offset loop -- the kernel above re-stages W_k and synchronises per offset, which at these widths is all it does
// (4 MFMAs between two barriers), and the pair-list form moves every (voxel, offset) contribution through HBM twice
// (105 MB out + 105 MB back for 150 k voxels at C = 16: at the HBM roofline for bytes this form never moves).  Per tile:
// the 27 x 16 neighbour ids -> LDS in one round trip, the offsets some voxel of the tile has (7 ballots), then per such
// offset one gathered row piece per lane and TI x TO x 4 v_mfma_f32_16x16x4_f32 (exact f32, A operand from the resident
// image), the rows of 8 offsets requested back to back, two accumulator sets.  Same store phase (conv_finish).
// ---------------------------------------------------------------------------------------------
#ifdef CONV_RES_DBG        /* profiling builds only: per-wave s_memtime phases of the resident-weights kernel */
__device__ unsigned long long conv_res_dbg[8 * 16384];
extern "C" int link_conv_resident_debug_read(void *host_dst, int64_t bytes) {
  return hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(conv_res_dbg), (size_t)bytes) == hipSuccess ? LINK_OK : LINK_ERR_LAUNCH;
}
#define CONV_TICK(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define CONV_TICK(v)
#endif

template <int CI, int CO>
struct conv_res_cfg {
  static constexpr int TI = CI / 16, TO = CO / 16, KV = 27;
  static constexpr int LDW = CI + 4;                   // row stride of W_k^T [co][ci]: 16 (mod 32) banks between the half-wave's two rows
  static constexpr int W_FLOATS = KV * CO * LDW;
  // waves sharing one image: the tile loop is a chain of round trips (ids, batches of rows, addend), so the CU wants 4-6
  // waves per SIMD; an image beyond 48 KB admits one workgroup per CU: 16 waves there, 8 (x 3 workgroups) below
  static constexpr int NW = (W_FLOATS * 4 > 48 * 1024) ? 16 : 8;
  static constexpr int LDS_BYTES = W_FLOATS * 4 + NW * (KV + 1) * 16 * 4;     // + per wave the tile's 27 x 16 neighbour ids (+ a spare row)
};

typedef _Float16 conv_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 conv_h8 __attribute__((ext_vector_type(8)));
// x = hi + lo, hi = fp16(x), lo = fp16(x - hi): four values -> two packed operands
__device__ __forceinline__ void conv_split4(const float4 &v, uint2 &hi, uint2 &lo) {
  const conv_h4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  const conv_h4 l = {(_Float16)(v.x - (float)h.x), (_Float16)(v.y - (float)h.y), (_Float16)(v.z - (float)h.z), (_Float16)(v.w - (float)h.w)};
  hi = __builtin_bit_cast(uint2, h);
  lo = __builtin_bit_cast(uint2, l);
}

// SPLIT: the products run on the f16 matrix cores with both operands split into fp16 hi + lo (hi*hi + hi*lo + lo*hi, fp32
// accumulation; the dropped lo*lo term is 2^-22 relative) -- what the pair-list form's inference kernel does
// (k_conv_pairs_gemm_split): a 16-channel k-step is ONE v_mfma_f32_16x16x16_f16 of 16 cycles instead of four
// v_mfma_f32_16x16x4_f32 of 32, two k-steps one v_mfma_f32_16x16x32_f16.  The resident image holds [hi(CI) | lo(CI)] halves
// per output channel (the same 4 CI + 16 bytes per row as the fp32 image).  A row or a weight beyond the fp16 range
// (|x| >= 2^15) sends the tile to the exact instruction with W from global memory (wave-uniform, never on sane data).
template <int CI, int CO, bool SPLIT>
__global__ void __launch_bounds__(64 * (conv_res_cfg<CI, CO>::NW)) k_subm_conv_resident(
    const float *__restrict__ feats, const int32_t *__restrict__ nbr, const float *__restrict__ w,
    const int32_t *__restrict__ order, int64_t n, int kvol, float *__restrict__ out, conv_epilogue ep) {
  using K = conv_res_cfg<CI, CO>;
  constexpr int TI = K::TI, TO = K::TO, LDW = K::LDW, NT = 64 * K::NW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *wt_lds = reinterpret_cast<float *>(smem_raw);                  // [k][co][ci], row stride LDW
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  int32_t *my_nb = reinterpret_cast<int32_t *>(wt_lds + K::W_FLOATS) + wave * ((K::KV + 1) * 16);
  CONV_TICK(tc0);
#ifdef CONV_RES_DBG
  unsigned long long tc_ids = 0, tc_ld = 0, tc_mma = 0, tc_fin = 0, tc_tiles = 0;
#endif
  // W [k][ci][co] -> LDS [k][co][ci]: 16-byte loads along co, four scattered writes each (fp32, or fp16 hi and lo)
  constexpr int LDH = 2 * CI + 8;                      // halves per row of the split image: [hi(CI) | lo(CI) | pad]
  bool w_big = false;
  for (int e = tid * 4; e < kvol * CI * CO; e += NT * 4) {
    const float4 v = *reinterpret_cast<const float4 *>(&w[e]);
    const int k = e / (CI * CO), r = e - k * (CI * CO), ci = r / CO, co = r - ci * CO;
    if constexpr (SPLIT) {
      _Float16 *dst = reinterpret_cast<_Float16 *>(smem_raw) + (k * CO + co) * LDH + ci;
      const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        const _Float16 h = (_Float16)vv[q4];
        dst[q4 * LDH] = h;
        dst[q4 * LDH + CI] = (_Float16)(vv[q4] - (float)h);
        w_big |= !(fabsf(vv[q4]) < 32768.0f);
      }
    } else {
      float *dst = wt_lds + (k * CO + co) * LDW + ci;
      dst[0] = v.x; dst[LDW] = v.y; dst[2 * LDW] = v.z; dst[3 * LDW] = v.w;
    }
  }
  if constexpr (SPLIT) w_big = __syncthreads_or(w_big) != 0;
  else __syncthreads();
  CONV_TICK(tc1);
  const int64_t tiles = (n + 15) / 16;
  // (contiguous eighths of the tiles per XCD were tried: 37.2 against 36.3 us at C = 16 -- the gathers are not what bounds it)
  const int64_t waves_total = (int64_t)gridDim.x * K::NW;
  for (int64_t tile = (int64_t)blockIdx.x * K::NW + wave; tile < tiles; tile += waves_total) {
    const int64_t q = tile * 16 + li;
    const int64_t v = (q < n) ? (order ? (int64_t)order[q] : q) : -1;
    CONV_TICK(ta);
    // neighbour ids of the tile: lane group g loads the offsets k = g (mod 4) -- unconditional loads on clamped indices, so
    // the seven requests are in flight together (a conditional load is a branch and a full wait each); which offsets
    // does the tile have at all?
    const int64_t vc = v >= 0 ? v : 0;
    int idr[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const int k = 4 * j + g;
      idr[j] = nbr[vc * kvol + (k < kvol ? k : 0)];
    }
    uint32_t mask = 0u;
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const int k = 4 * j + g;
      const int id = (k < kvol && v >= 0) ? idr[j] : -1;
      my_nb[k * 16 + li] = id;                         // k = 27 (j = 6, g = 3) lands in a spare slot
      const unsigned long long bal = __ballot(id >= 0);
#pragma unroll
      for (int gg = 0; gg < 4; gg++)
        if ((bal >> (16 * gg)) & 0xFFFFull) mask |= 1u << (4 * j + gg);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CONV_TICK(tb);
#ifdef CONV_RES_DBG
    tc_ids += tb - ta;
#endif
    // two accumulator sets, alternating by offset: consecutive MFMAs do not wait for each other's result
    floatx4 acc[TO], acc2[TO];
#pragma unroll
    for (int tp = 0; tp < TO; tp++) acc[tp] = acc2[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
    auto next_off = [&]() { const int kk = mask ? (__ffs(mask) - 1) : -1; mask &= mask - 1; return kk; };
    auto mma = [&](int kk, bool has, const float4 (&ff)[TI], floatx4 (&ac)[TO], bool exact) {
      if constexpr (SPLIT) {
        if (!exact) {
          const _Float16 *wh = reinterpret_cast<const _Float16 *>(smem_raw) + (size_t)kk * (CO * LDH);
          uint2 bh[TI], bl[TI];
#pragma unroll
          for (int tt = 0; tt < TI; tt++) {
            conv_split4(ff[tt], bh[tt], bl[tt]);
            bh[tt].x = has ? bh[tt].x : 0u; bh[tt].y = has ? bh[tt].y : 0u;
            bl[tt].x = has ? bl[tt].x : 0u; bl[tt].y = has ? bl[tt].y : 0u;
          }
#pragma unroll
          for (int tp = 0; tp < TO; tp++) {
            const _Float16 *row = wh + (16 * tp + li) * LDH + 4 * g;
            if constexpr (TI == 2) {
              // two 16-channel k-steps in one instruction: its k = 8 g + j is channel 4 g + j of the first block (j < 4) and
              // of the second (j >= 4), for both operands alike
              const uint2 ah0 = *reinterpret_cast<const uint2 *>(row), ah1 = *reinterpret_cast<const uint2 *>(row + 16);
              const uint2 al0 = *reinterpret_cast<const uint2 *>(row + CI), al1 = *reinterpret_cast<const uint2 *>(row + CI + 16);
              const uint4 Ah = make_uint4(ah0.x, ah0.y, ah1.x, ah1.y), Al = make_uint4(al0.x, al0.y, al1.x, al1.y);
              const uint4 Bh = make_uint4(bh[0].x, bh[0].y, bh[1].x, bh[1].y), Bl = make_uint4(bl[0].x, bl[0].y, bl[1].x, bl[1].y);
              ac[tp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(conv_h8, Al), __builtin_bit_cast(conv_h8, Bh), ac[tp], 0, 0, 0);
              ac[tp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(conv_h8, Ah), __builtin_bit_cast(conv_h8, Bl), ac[tp], 0, 0, 0);
              ac[tp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(conv_h8, Ah), __builtin_bit_cast(conv_h8, Bh), ac[tp], 0, 0, 0);
            } else {
#pragma unroll
              for (int tt = 0; tt < TI; tt++) {
                const uint2 ah = *reinterpret_cast<const uint2 *>(row + 16 * tt), al = *reinterpret_cast<const uint2 *>(row + CI + 16 * tt);
                ac[tp] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(conv_h4, al), __builtin_bit_cast(conv_h4, bh[tt]), ac[tp], 0, 0, 0);
                ac[tp] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(conv_h4, ah), __builtin_bit_cast(conv_h4, bl[tt]), ac[tp], 0, 0, 0);
                ac[tp] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(conv_h4, ah), __builtin_bit_cast(conv_h4, bh[tt]), ac[tp], 0, 0, 0);
              }
            }
          }
          return;
        }
      }
      // exact f32: A operand from the resident fp32 image, or (SPLIT kernels' out-of-range tiles) from global memory
#pragma unroll
      for (int tt = 0; tt < TI; tt++) {
        const float fx = has ? ff[tt].x : 0.f, fy = has ? ff[tt].y : 0.f, fz = has ? ff[tt].z : 0.f, fw = has ? ff[tt].w : 0.f;
#pragma unroll
        for (int tp = 0; tp < TO; tp++) {
          float4 a;
          if constexpr (SPLIT) {
            const float *wk = w + (size_t)kk * (CI * CO) + (size_t)(16 * tt + 4 * g) * CO + 16 * tp + li;
            a = make_float4(wk[0], wk[CO], wk[2 * CO], wk[3 * CO]);
          } else {
            a = *reinterpret_cast<const float4 *>(&wt_lds[(size_t)kk * (CO * LDW) + (16 * tp + li) * LDW + 16 * tt + 4 * g]);
          }
          ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, fx, ac[tp], 0, 0, 0);
          ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, fy, ac[tp], 0, 0, 0);
          ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, fz, ac[tp], 0, 0, 0);
          ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, fw, ac[tp], 0, 0, 0);
        }
      }
    };
    // batches of BATCH offsets, straight-line: all ids from LDS, all rows requested, then the products (an offset slot
    // beyond the tile's last one multiplies zeros: at most BATCH - 1 per tile)
    constexpr int BATCH = 8 / TI;
    auto load_batch = [&](int (&ks)[BATCH], int (&ids)[BATCH], float4 (&fs)[BATCH][TI]) {
      const bool any = mask != 0u;
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        const int kk = next_off();
        ks[i] = kk < 0 ? 0 : kk;
        const int id = my_nb[ks[i] * 16 + li];
        ids[i] = kk < 0 ? -1 : id;
      }
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        const int64_t row = ids[i] >= 0 ? ids[i] : 0;
#pragma unroll
        for (int tt = 0; tt < TI; tt++) fs[i][tt] = *reinterpret_cast<const float4 *>(&feats[row * CI + 16 * tt + 4 * g]);
      }
      __builtin_amdgcn_sched_barrier(0);               // the batch's requests go out before anything that follows
      return any;
    };
    auto products = [&](const int (&ks)[BATCH], const int (&ids)[BATCH], const float4 (&fs)[BATCH][TI]) {
      bool exact = !SPLIT;
      if constexpr (SPLIT) {
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < BATCH; i++)
#pragma unroll
          for (int tt = 0; tt < TI; tt++)
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf(fs[i][tt].x), fabsf(fs[i][tt].y)), fmaxf(fabsf(fs[i][tt].z), fabsf(fs[i][tt].w))));
        exact = w_big || __any(!(mx < 32768.0f));
      }
#pragma unroll
      for (int i = 0; i < BATCH; i++) mma(ks[i], ids[i] >= 0, fs[i], (i & 1) ? acc2 : acc, exact);
    };
    // (two row sets with the next batch in flight during the products were tried: 161 registers halve the waves per SIMD
    // and the kernel got slower, 36 -> 49 us at C = 16)
    int ksA[BATCH], idsA[BATCH];
    float4 fA[BATCH][TI];
    CONV_TICK(tl0);
    while (load_batch(ksA, idsA, fA)) products(ksA, idsA, fA);
#ifdef CONV_RES_DBG
    asm volatile("s_nop 0" ::"v"(acc[0][0]), "v"(acc2[0][0]));
    CONV_TICK(tl2);
    tc_mma += tl2 - tl0;
#endif
#pragma unroll
    for (int tp = 0; tp < TO; tp++) acc[tp] += acc2[tp];
    CONV_TICK(tf0);
    conv_finish<CO>(acc, v, g, ep, out);
    __builtin_amdgcn_wave_barrier();                   // the id list is rewritten by the next tile
#ifdef CONV_RES_DBG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CONV_TICK(tf1);
    tc_fin += tf1 - tf0;
    tc_tiles++;
#endif
  }
#ifdef CONV_RES_DBG
  if (lane == 0 && blockIdx.x * K::NW + wave < 16384) {
    unsigned long long *d = conv_res_dbg + (size_t)(blockIdx.x * K::NW + wave) * 8;
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    d[0] = tc1 - tc0; d[1] = tc_ids; d[2] = tc_ld; d[3] = tc_mma; d[4] = tc_fin; d[5] = tc_tiles; d[6] = tc0; d[7] = te - tc0;
  }
#endif
}

template <int CI, int CO>
static int launch_conv_resident(const float *feats, const int32_t *nbr, const float *w, const int32_t *order, int64_t n,
                                int kvol, float *out, const conv_epilogue &ep, bool split, hipStream_t st) {
  using K = conv_res_cfg<CI, CO>;
  const int64_t tiles = (n + 15) / 16;
  // persistent workgroups: as many as stay resident (LDS-bound), each wave strides over the tiles -- the image is staged
  // once per workgroup, not once per tile
  const int per_cu = (160 * 1024) / K::LDS_BYTES;
  int64_t wgs = (tiles + K::NW - 1) / K::NW;
  if (wgs > 256 * per_cu) wgs = 256 * per_cu;
  if (split) {
    if (K::LDS_BYTES > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_subm_conv_resident<CI, CO, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    hipLaunchKernelGGL((k_subm_conv_resident<CI, CO, true>), dim3((unsigned)wgs), dim3(64 * K::NW), K::LDS_BYTES, st, feats, nbr, w,
                       order, n, kvol, out, ep);
  } else {
    if (K::LDS_BYTES > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_subm_conv_resident<CI, CO, false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    hipLaunchKernelGGL((k_subm_conv_resident<CI, CO, false>), dim3((unsigned)wgs), dim3(64 * K::NW), K::LDS_BYTES, st, feats, nbr, w,
                       order, n, kvol, out, ep);
  }
  return check_launch("link_subm_conv_forward");
}

// ---------------------------------------------------------------------------------------------
// AMP form of the resident-weights kernel: rows AND weights are 16-bit (fp16 or bf16: the reference's
// custom_fwd(cast_inputs=torch.half), nn/functional/conv.py:18), products on the 16-bit matrix cores with fp32 accumulation
// over ALL offsets (the pair-list AMP form rounds each offset's product to the row type first, as the reference's half mm
// does: this one is closer to the fp32 result), statistics / affine in fp32, 16-bit out.  wt = [kvol][cout][cin] in the row
// type (what link_conv_pairs_gemm_amp takes): the LDS image is a padded copy, a gathered row piece is 8 bytes and goes to
// the matrix core as it is -- no conversion in the offset loop.
// ---------------------------------------------------------------------------------------------
struct conv_epilogue_amp {
  const float *ln_w, *ln_b;
  const unsigned short *addend;
  float eps;
  int relu;
};
typedef short conv_s4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float conv_h2f(unsigned short v, bool bf) {
  if (bf) return __uint_as_float((unsigned)v << 16);
  return (float)__builtin_bit_cast(_Float16, v);
}
__device__ __forceinline__ unsigned short conv_f2h(float f, bool bf) {
  if (bf) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
  }
  return __builtin_bit_cast(unsigned short, (_Float16)f);
}

template <int CI, int CO>
struct conv_amp_cfg {
  static constexpr int TI = CI / 16, TO = CO / 16, KV = 27, NW = 8;
  static constexpr int LDH = CI + 8;                   // halves per row of the image
  static constexpr int W_BYTES = KV * CO * LDH * 2;
  static constexpr int LDS_BYTES = W_BYTES + NW * (KV + 1) * 16 * 4;
};

template <int CI, int CO, bool BF>
__global__ void __launch_bounds__(512) k_subm_conv_resident_amp(
    const unsigned short *__restrict__ feats, const int32_t *__restrict__ nbr, const unsigned short *__restrict__ wt,
    const int32_t *__restrict__ order, int64_t n, int kvol, unsigned short *__restrict__ out, conv_epilogue_amp ep) {
  using K = conv_amp_cfg<CI, CO>;
  constexpr int TI = K::TI, TO = K::TO, LDH = K::LDH;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  unsigned short *wh = reinterpret_cast<unsigned short *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, g = lane >> 4;
  int32_t *my_nb = reinterpret_cast<int32_t *>(smem_raw + K::W_BYTES) + wave * ((K::KV + 1) * 16);
  for (int e = tid * 4; e < kvol * CO * CI; e += 512 * 4) {              // [k][co][ci] -> rows padded to LDH
    const uint2 v = *reinterpret_cast<const uint2 *>(&wt[e]);
    const int rowi = e / CI, ci = e - rowi * CI;
    *reinterpret_cast<uint2 *>(&wh[rowi * LDH + ci]) = v;
  }
  __syncthreads();
  const int64_t tiles = (n + 15) / 16;
  const int64_t waves_total = (int64_t)gridDim.x * K::NW;
  for (int64_t tile = (int64_t)blockIdx.x * K::NW + wave; tile < tiles; tile += waves_total) {
    const int64_t q = tile * 16 + li;
    const int64_t v = (q < n) ? (order ? (int64_t)order[q] : q) : -1;
    const int64_t vc = v >= 0 ? v : 0;
    int idr[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const int k = 4 * j + g;
      idr[j] = nbr[vc * kvol + (k < kvol ? k : 0)];
    }
    uint32_t mask = 0u;
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const int k = 4 * j + g;
      const int id = (k < kvol && v >= 0) ? idr[j] : -1;
      my_nb[k * 16 + li] = id;
      const unsigned long long bal = __ballot(id >= 0);
#pragma unroll
      for (int gg = 0; gg < 4; gg++)
        if ((bal >> (16 * gg)) & 0xFFFFull) mask |= 1u << (4 * j + gg);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    floatx4 acc[TO], acc2[TO];
#pragma unroll
    for (int tp = 0; tp < TO; tp++) acc[tp] = acc2[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
    auto next_off = [&]() { const int kk = mask ? (__ffs(mask) - 1) : -1; mask &= mask - 1; return kk; };
    constexpr int BATCH = 16 / TI;
    while (mask) {
      int ks[BATCH], ids[BATCH];
      uint2 fs[BATCH][TI];
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        const int kk = next_off();
        ks[i] = kk < 0 ? 0 : kk;
        const int id = my_nb[ks[i] * 16 + li];
        ids[i] = kk < 0 ? -1 : id;
      }
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        const int64_t row = ids[i] >= 0 ? ids[i] : 0;
#pragma unroll
        for (int tt = 0; tt < TI; tt++) fs[i][tt] = *reinterpret_cast<const uint2 *>(&feats[row * CI + 16 * tt + 4 * g]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        const bool has = ids[i] >= 0;
        const unsigned short *wk = wh + (size_t)ks[i] * (CO * LDH);
#pragma unroll
        for (int tt = 0; tt < TI; tt++) {
          uint2 b = fs[i][tt];
          b.x = has ? b.x : 0u; b.y = has ? b.y : 0u;
#pragma unroll
          for (int tp = 0; tp < TO; tp++) {
            const uint2 a = *reinterpret_cast<const uint2 *>(&wk[(16 * tp + li) * LDH + 16 * tt + 4 * g]);
            floatx4 &ac = (i & 1) ? acc2[tp] : acc[tp];
            if constexpr (BF) ac = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(conv_s4, a), __builtin_bit_cast(conv_s4, b), ac, 0, 0, 0);
            else ac = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(conv_h4, a), __builtin_bit_cast(conv_h4, b), ac, 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int tp = 0; tp < TO; tp++) acc[tp] += acc2[tp];
    // store phase: as conv_finish, 16-bit addend / out
    float mean = 0.f, rstd = 1.f;
    if (ep.ln_w && !(ep.relu & 2)) {
      float sm = 0.f;
#pragma unroll
      for (int tp = 0; tp < TO; tp++) sm += (acc[tp][0] + acc[tp][1]) + (acc[tp][2] + acc[tp][3]);
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      mean = sm * (1.0f / CO);
      float qv = 0.f;
#pragma unroll
      for (int tp = 0; tp < TO; tp++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float d = acc[tp][r] - mean;
          qv += d * d;
        }
      qv += __shfl_xor(qv, 16, 64);
      qv += __shfl_xor(qv, 32, 64);
      rstd = 1.0f / sqrtf(qv * (1.0f / CO) + ep.eps);
    }
    if (v >= 0) {
#pragma unroll
      for (int tp = 0; tp < TO; tp++) {
        const int ch = 16 * tp + 4 * g;
        float o[4] = {acc[tp][0], acc[tp][1], acc[tp][2], acc[tp][3]};
        if (ep.ln_w) {
          const float4 lw = *reinterpret_cast<const float4 *>(&ep.ln_w[ch]);
          const float4 lb = *reinterpret_cast<const float4 *>(&ep.ln_b[ch]);
          o[0] = (o[0] - mean) * rstd * lw.x + lb.x; o[1] = (o[1] - mean) * rstd * lw.y + lb.y;
          o[2] = (o[2] - mean) * rstd * lw.z + lb.z; o[3] = (o[3] - mean) * rstd * lw.w + lb.w;
        }
        if (ep.addend) {
          const uint2 a4 = *reinterpret_cast<const uint2 *>(&ep.addend[v * CO + ch]);
          o[0] += conv_h2f((unsigned short)(a4.x & 0xFFFFu), BF); o[1] += conv_h2f((unsigned short)(a4.x >> 16), BF);
          o[2] += conv_h2f((unsigned short)(a4.y & 0xFFFFu), BF); o[3] += conv_h2f((unsigned short)(a4.y >> 16), BF);
        }
        if (ep.relu & 1) { o[0] = fmaxf(o[0], 0.f); o[1] = fmaxf(o[1], 0.f); o[2] = fmaxf(o[2], 0.f); o[3] = fmaxf(o[3], 0.f); }
        uint2 pk;
        pk.x = (unsigned)conv_f2h(o[0], BF) | ((unsigned)conv_f2h(o[1], BF) << 16);
        pk.y = (unsigned)conv_f2h(o[2], BF) | ((unsigned)conv_f2h(o[3], BF) << 16);
        *reinterpret_cast<uint2 *>(&out[v * CO + ch]) = pk;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int CI, int CO>
static int launch_conv_resident_amp(const void *feats, int io, const int32_t *nbr, const void *wt, const int32_t *order, int64_t n,
                                    int kvol, void *out, const conv_epilogue_amp &ep, hipStream_t st) {
  using K = conv_amp_cfg<CI, CO>;
  const int64_t tiles = (n + 15) / 16;
  const int per_cu = (160 * 1024) / K::LDS_BYTES > 4 ? 4 : (160 * 1024) / K::LDS_BYTES;
  int64_t wgs = (tiles + K::NW - 1) / K::NW;
  if (wgs > 256 * per_cu) wgs = 256 * per_cu;
  const unsigned short *f = static_cast<const unsigned short *>(feats), *w16 = static_cast<const unsigned short *>(wt);
  unsigned short *o = static_cast<unsigned short *>(out);
  if (io == LINK_IO_BF16) {
    if (K::LDS_BYTES > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_subm_conv_resident_amp<CI, CO, true>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    hipLaunchKernelGGL((k_subm_conv_resident_amp<CI, CO, true>), dim3((unsigned)wgs), dim3(512), K::LDS_BYTES, st, f, nbr, w16, order, n, kvol, o, ep);
  } else {
    if (K::LDS_BYTES > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_subm_conv_resident_amp<CI, CO, false>), hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
    hipLaunchKernelGGL((k_subm_conv_resident_amp<CI, CO, false>), dim3((unsigned)wgs), dim3(512), K::LDS_BYTES, st, f, nbr, w16, order, n, kvol, o, ep);
  }
  return check_launch("link_subm_conv_resident_amp");
}

extern "C" int link_subm_conv_resident_amp(const void *feats, int32_t io_dtype, const int32_t *nbr, const void *wt,
                                           const int32_t *order, int64_t n, int32_t cin, int32_t cout, int32_t kvol,
                                           const float *ln_w, const float *ln_b, float eps, const void *addend, int32_t relu,
                                           void *out, void *stream) {
  if (io_dtype != LINK_IO_F16 && io_dtype != LINK_IO_BF16) return LINK_ERR_ARG;
  if (n < 0 || kvol <= 0 || kvol > 27 || (ln_w == nullptr) != (ln_b == nullptr)) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!feats || !nbr || !wt || !out) return LINK_ERR_ARG;
  const conv_epilogue_amp ep = {ln_w, ln_b, static_cast<const unsigned short *>(addend), eps, relu & 3};
  hipStream_t st = S(stream);
  if (cin == 16 && cout == 16) return launch_conv_resident_amp<16, 16>(feats, io_dtype, nbr, wt, order, n, kvol, out, ep, st);
  if (cin == 32 && cout == 32) return launch_conv_resident_amp<32, 32>(feats, io_dtype, nbr, wt, order, n, kvol, out, ep, st);
  if (cin == 16 && cout == 32) return launch_conv_resident_amp<16, 32>(feats, io_dtype, nbr, wt, order, n, kvol, out, ep, st);
  if (cin == 32 && cout == 16) return launch_conv_resident_amp<32, 16>(feats, io_dtype, nbr, wt, order, n, kvol, out, ep, st);
  return LINK_ERR_ARG;
}

// any Cin / Cout <= 256: one wave per output voxel, lanes = output channels (no MFMA; small widths)
template <int CPL>
__global__ void __launch_bounds__(256) k_subm_conv_generic(const float *__restrict__ feats,
                                                           const int32_t *__restrict__ nbr,
                                                           const float *__restrict__ w, int64_t n, int cin,
                                                           int cout, int kvol, float *__restrict__ out,
                                                           conv_epilogue ep) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (v >= n) return;
  float acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) acc[q] = 0.f;
  for (int k = 0; k < kvol; k++) {
    const int id = nbr[v * kvol + k];
    if (id < 0) continue;                            // wave-uniform
    const float *row = feats + (int64_t)id * cin;
    const float *wk = w + (int64_t)k * cin * cout;
    for (int ci = 0; ci < cin; ci++) {
      const float fv = row[ci];
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        const int co = lane + 64 * q;
        if (co < cout) acc[q] = fmaf(fv, wk[(int64_t)ci * cout + co], acc[q]);
      }
    }
  }
  if (ep.ln_w) {
    float mean = 0.f, rstd = 1.f;
    if (!(ep.relu & 2)) {
      float sm = 0.f;
#pragma unroll
      for (int q = 0; q < CPL; q++) sm += (lane + 64 * q < cout) ? acc[q] : 0.f;
      mean = wave_sum(sm) / cout;
      float qq = 0.f;
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        const float d = (lane + 64 * q < cout) ? acc[q] - mean : 0.f;
        qq += d * d;
      }
      rstd = 1.0f / sqrtf(wave_sum(qq) / cout + ep.eps);
    }
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      const int co = lane + 64 * q;
      if (co < cout) {
        float o = (acc[q] - mean) * rstd * ep.ln_w[co] + ep.ln_b[co];
        if (ep.addend) o = ep.addend[v * cout + co] + o;
        if (ep.relu & 1) o = fmaxf(o, 0.f);
        out[v * cout + co] = o;
      }
    }
    return;
  }
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    const int co = lane + 64 * q;
    if (co < cout) out[v * cout + co] = acc[q];
  }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient  g_w[k][ci][co] = sum_v feats[nbr[v,k]][ci] * g_out[v][co]   (C <= 64, C % 4 == 0)
//
// The reduction runs over voxels, so voxels are the MFMA k-dimension: a step takes 4 (input row,
// output row) pairs, lane (li, g) loads ONE float4 of pair g's input row (channels 4li..4li+3) and one
// of its grad row, and the 16 products {r, q} of v_mfma_f32_16x16x4_f32 give
// D_{r,q}[i][j] = sum_pairs F[4i + r] * G[4j + q] -- all 64 x 64 entries from two 16-byte loads per
// lane.  Workgroup (k, chunk): its 4 waves walk the chunk's voxels 64 at a time, read the offset's
// column of the transposed neighbour table (coalesced), compact the present (in, out) pairs with a
// ballot into a per-wave LDS queue (exact sparsity skipping), and consume them 16 at a time; the
// waves' accumulators are summed through LDS in a fixed order and written as one [C x C] partial per
// (chunk, k): deterministic, no atomics.
// ---------------------------------------------------------------------------------------------
// Balance (round 5): on a submanifold table the centre offset's column is DENSE (every voxel pairs with itself) while the
// other 26 hold a fifth of that on LiDAR frames -- with one workgroup per (offset, chunk) the centre's workgroups set the
// kernel's time (284 us on the 113k-voxel stem of the cfg3 encoder, a quarter of whose training step was this kernel).  The
// centre column is therefore cut into `chunks + centre_extra` pieces: workgroups [kvol * chunks, kvol * chunks + centre_extra)
// take the extra pieces; partial slot = workgroup number either way.
__global__ void __launch_bounds__(256, 4) k_subm_conv_wgrad(const float *__restrict__ feats,
                                                         const float *__restrict__ gout,
                                                         const int32_t *__restrict__ nbr_t, int64_t n, int c,
                                                         int kvol, int chunks, int centre_extra, float *__restrict__ partial) {
  __shared__ int2 queue[4][96];                     // per wave: pending (in row, out row) pairs
  __shared__ float red[4][4][4][64];                // [wave][q][e][lane] of ONE r at a time (16 KB: the cross-wave sum runs in four
                                                    // passes -- with the whole 64 KB image two workgroups filled a CU's LDS, i.e. two
                                                    // waves per SIMD to hide the row gathers of a kernel that lives on them; round 5)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int centre = kvol >> 1;
  const bool extra_wg = (int)blockIdx.x >= kvol * chunks;
  const int k = extra_wg ? centre : (int)blockIdx.x % kvol;
  const int chunk = extra_wg ? chunks + ((int)blockIdx.x - kvol * chunks) : (int)blockIdx.x / kvol;
  const int pieces = (k == centre && centre_extra > 0) ? chunks + centre_extra : chunks;
  const bool act = 4 * li < c;
  const int cofs = act ? 4 * li : 0;
  floatx4 acc[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int q = 0; q < 4; q++) acc[r][q] = (floatx4){0.f, 0.f, 0.f, 0.f};
  // voxel range of this wave: the chunk's range cut in 4, in units of 64 voxels
  const int64_t units = (n + 63) / 64;
  const int64_t per_chunk = (units + pieces - 1) / pieces;
  const int64_t u0 = (int64_t)chunk * per_chunk, u1 = (u0 + per_chunk < units) ? u0 + per_chunk : units;
  const int32_t *col = nbr_t + (int64_t)k * n;
  int2 *my_q = queue[wave];
  int qn = 0;                                       // pairs waiting in the queue (wave-uniform)
  auto consume = [&](int base, int cnt) {           // cnt <= 16 pairs starting at queue[base]: 4 MFMA steps
    float4 fa[4], ga[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int qi = 4 * s + g;
      const bool ok = qi < cnt;
      const int2 pr = my_q[base + (ok ? qi : 0)];
      fa[s] = *reinterpret_cast<const float4 *>(&feats[(int64_t)pr.x * c + cofs]);
      ga[s] = *reinterpret_cast<const float4 *>(&gout[(int64_t)pr.y * c + cofs]);
      if (!ok || !act) fa[s] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (!act) ga[s] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
      if (4 * s >= cnt) break;                      // wave-uniform
      const float fv[4] = {fa[s].x, fa[s].y, fa[s].z, fa[s].w}, gv[4] = {ga[s].x, ga[s].y, ga[s].z, ga[s].w};
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[r][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fv[r], gv[q], acc[r][q], 0, 0, 0);
    }
  };
  for (int64_t u = u0 + wave; u < u1; u += 4) {
    const int64_t v = u * 64 + lane;
    const int id = (v < n) ? col[v] : -1;
    const uint64_t m = __ballot(id >= 0);
    if (id >= 0) my_q[qn + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(id, (int)v);
    qn += __popcll(m);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int done = 0;
    while (qn - done >= 16) { consume(done, 16); done += 16; }
    if (done) {                                     // move the < 16 leftovers to the front
      const int left = qn - done;
      int2 keep = make_int2(0, 0);
      if (lane < left) keep = my_q[done + lane];
      __builtin_amdgcn_wave_barrier();
      if (lane < left) my_q[lane] = keep;
      qn = left;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (qn > 0) consume(0, qn);                       // tail (qn < 16)
  // fixed-order sum of the 4 waves, then one [C x C] partial per (chunk, k)
  float *dst = partial + (int64_t)blockIdx.x * c * c;             // slot = workgroup number (= chunk * kvol + k below kvol * chunks)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    if (r) __syncthreads();                          // the previous pass has been read
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[wave][q][e][lane] = acc[r][q][e];
    __syncthreads();
    for (int idx = tid; idx < 4 * 4 * 64; idx += 256) {
      const int l = idx & 63, e = (idx >> 6) & 3, q = idx >> 8;
      const int ci = 4 * (4 * (l >> 4) + e) + r, co = 4 * (l & 15) + q;   // D[i = 4g+e][j = li] of product (r, q)
      if (ci < c && co < c)
        dst[ci * c + co] = (red[0][q][e][l] + red[1][q][e][l]) + (red[2][q][e][l] + red[3][q][e][l]);
    }
  }
}

extern "C" int32_t link_subm_conv_wgrad_chunks(void) { return 16; }

extern "C" int link_subm_conv_wgrad(const float *feats, const float *gout, const int32_t *nbr_t, int64_t n,
                                    int32_t c, int32_t kvol, float *partial, void *stream) {
  if (n < 0 || c <= 0 || c > 64 || (c & 3) != 0 || kvol <= 0) return LINK_ERR_ARG;
  if (!partial) return LINK_ERR_ARG;
  if (n > 0 && (!feats || !gout || !nbr_t)) return LINK_ERR_ARG;
  const int chunks = link_subm_conv_wgrad_chunks();
  hipLaunchKernelGGL(k_subm_conv_wgrad, dim3((unsigned)(kvol * chunks)), dim3(256), 0, S(stream), feats, gout, nbr_t, n,
                     (int)c, (int)kvol, chunks, 0, partial);
  return check_launch("link_subm_conv_wgrad");
}

// Round 5: the caller picks the split.  partial = f32[(kvol * chunks + centre_extra)][c][c]; g_w[k] = sum over chunk of slot
// [chunk * kvol + k], plus, for k = kvol / 2, the slots [kvol * chunks, kvol * chunks + centre_extra).  centre_extra > 0 only
// for tables whose centre column is dense (odd kernel over one coordinate set).
extern "C" int link_subm_conv_wgrad_split(const float *feats, const float *gout, const int32_t *nbr_t, int64_t n, int32_t c,
                                          int32_t kvol, int32_t chunks, int32_t centre_extra, float *partial, void *stream) {
  if (n < 0 || c <= 0 || c > 64 || (c & 3) != 0 || kvol <= 0 || chunks < 1 || chunks > 1024 || centre_extra < 0 || centre_extra > 4096)
    return LINK_ERR_ARG;
  if (!partial) return LINK_ERR_ARG;
  if (n > 0 && (!feats || !gout || !nbr_t)) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_subm_conv_wgrad, dim3((unsigned)(kvol * chunks + centre_extra)), dim3(256), 0, S(stream), feats, gout, nbr_t, n,
                     (int)c, (int)kvol, (int)chunks, (int)centre_extra, partial);
  return check_launch("link_subm_conv_wgrad_split");
}

// g_w from the partial slots of link_subm_conv_wgrad_split in ONE launch (round 6): g_w[k] = sum over chunk of slot [chunk * kvol + k]
// (+, for k = kvol / 2, the centre_extra slots behind them), in a fixed association -- deterministic.  Was three torch launches per convolution
// and backward call (two reductions and an add: 885 + 885 of the ~900 launches of a cfg3 training step were these).
__global__ void __launch_bounds__(256) k_subm_wgrad_reduce(const float *__restrict__ part, int kvol, int chunks, int extra, int cc,
                                                           float *__restrict__ gw) {
  // eight adjacent lanes share one float4 of one offset's c x c matrix: lane `sub` adds slots sub, sub + 8, ... (the centre offset's
  // extra slots behind its chunk slots), then a fixed xor tree over the eight -- the same association in every run.  (One thread per
  // float4 walking all slots -- up to 64 + 192 of them for the centre of a big layer -- took 20 us per call, 59 at worst.)
  const int e = (int)(blockIdx.x * 256 + threadIdx.x);
  const int o = e >> 3, sub = e & 7;
  const int per = cc >> 2;
  const bool live = o < kvol * per;
  const int k = live ? o / per : 0, q = live ? o - k * per : 0;
  const int nslot = chunks + ((extra > 0 && k == kvol / 2) ? extra : 0);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = sub; s0 < nslot; s0 += 32) {            // four slots of this lane in flight
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      int sl = s0 + 8 * u;
      sl = sl < nslot ? sl : nslot - 1;
      const int64_t row = sl < chunks ? (int64_t)sl * kvol + k : (int64_t)chunks * kvol + (sl - chunks);
      v[u] = *reinterpret_cast<const float4 *>(&part[row * cc + 4 * q]);
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (s0 + 8 * u < nslot) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
  }
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) {
    acc.x += __shfl_xor(acc.x, m, 64); acc.y += __shfl_xor(acc.y, m, 64); acc.z += __shfl_xor(acc.z, m, 64); acc.w += __shfl_xor(acc.w, m, 64);
  }
  if (live && sub == 0) *reinterpret_cast<float4 *>(&gw[(int64_t)k * cc + 4 * q]) = acc;
}
extern "C" int link_subm_conv_wgrad_reduce(const float *partial, int32_t c, int32_t kvol, int32_t chunks, int32_t centre_extra, float *gw,
                                           void *stream) {
  if (!partial || !gw || c <= 0 || c > 64 || (c & 3) != 0 || kvol <= 0 || chunks < 1 || chunks > 1024 || centre_extra < 0 || centre_extra > 4096)
    return LINK_ERR_ARG;
  const int cc = c * c, total = kvol * (cc >> 2) * 8;
  hipLaunchKernelGGL(k_subm_wgrad_reduce, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(stream), partial, (int)kvol, (int)chunks,
                     (int)centre_extra, cc, gw);
  return check_launch("link_subm_conv_wgrad_reduce");
}

static constexpr int g_conv_wgs = 1024;  // cap (sweep recorded in DESIGN.md 4b)
static constexpr int g_conv_nt = 0;      // tiles per wave: by size
static constexpr int g_conv_deep = 1;    // two-steps-ahead pipeline of the table kernel on small frames
#ifndef CONV_RESIDENT_MIN
#define CONV_RESIDENT_MIN 1              /* voxels from which the resident-weights kernel takes the narrow layers */
#endif

template <int CI, int CO, int NT>
static int launch_conv_mfma_nt(const float *feats, const int32_t *nbr, const float *w, const int32_t *order,
                               int64_t n, int kvol, float *out, const conv_epilogue &ep, hipStream_t st) {
  const size_t lds = ((size_t)2 * CO * (CI + 4) + (size_t)4 * 27 * NT * 16) * sizeof(float);
  if (lds > 64 * 1024) {
    // per device and cheap (a host-side table write): no process-wide once-flag, which a second GPU or a
    // device reset would never pass again
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_subm_conv_mfma<CI, CO, NT, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  const int64_t tiles = (n + 15) / 16;
  int64_t wgs = (tiles + 4 * NT - 1) / (4 * NT);
  if (wgs > g_conv_wgs) wgs = g_conv_wgs;
  if (NT == 1 && g_conv_deep && tiles <= 1024) {      // <= 16k voxels: a wave or two per SIMD, the latency-bound regime (larger frames lose occupancy to the extra register sets)
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_subm_conv_mfma<CI, CO, 1, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_subm_conv_mfma<CI, CO, 1, true>), dim3((unsigned)wgs), dim3(256), lds, st, feats, nbr, w, order, n, kvol, out, ep);
    return check_launch("link_subm_conv_forward");
  }
  hipLaunchKernelGGL((k_subm_conv_mfma<CI, CO, NT, false>), dim3((unsigned)wgs), dim3(256), lds, st, feats, nbr, w, order, n, kvol, out, ep);
  return check_launch("link_subm_conv_forward");
}

// tiles per wave: as few as still fills the chip (more workgroups = more latency hiding; fewer = less
// W_k staging traffic); the widest layers are capped at 2 by the 160 KB of LDS
template <int CI, int CO>
static int launch_conv_mfma(const float *feats, const int32_t *nbr, const float *w, const int32_t *order,
                            int64_t n, int kvol, float *out, const conv_epilogue &ep, hipStream_t st) {
  const int64_t tiles = (n + 15) / 16;
  constexpr bool nt4_fits = ((size_t)2 * CO * (CI + 4) + (size_t)4 * 27 * 4 * 16) * sizeof(float) <= 160 * 1024 && CO <= 112;
  if (g_conv_nt == 4 && nt4_fits) return launch_conv_mfma_nt<CI, CO, 4>(feats, nbr, w, order, n, kvol, out, ep, st);
  if (g_conv_nt == 2 || g_conv_nt == 4) return launch_conv_mfma_nt<CI, CO, 2>(feats, nbr, w, order, n, kvol, out, ep, st);
  if (g_conv_nt == 1) return launch_conv_mfma_nt<CI, CO, 1>(feats, nbr, w, order, n, kvol, out, ep, st);
  if (tiles > (int64_t)g_conv_wgs * 4 * 2 && nt4_fits) return launch_conv_mfma_nt<CI, CO, 4>(feats, nbr, w, order, n, kvol, out, ep, st);
  if (tiles > (int64_t)g_conv_wgs * 4) return launch_conv_mfma_nt<CI, CO, 2>(feats, nbr, w, order, n, kvol, out, ep, st);
  return launch_conv_mfma_nt<CI, CO, 1>(feats, nbr, w, order, n, kvol, out, ep, st);
}

static int subm_conv_impl(const float *feats, const int32_t *nbr, const float *w, const int32_t *order,
                          int64_t n, int32_t cin, int32_t cout, int32_t kvol, float *out,
                          const conv_epilogue &ep, void *stream);

extern "C" int link_subm_conv_forward(const float *feats, const int32_t *nbr, const float *w,
                                      const int32_t *order, int64_t n, int32_t cin, int32_t cout,
                                      int32_t kvol, float *out, void *stream) {
  const conv_epilogue ep = {nullptr, nullptr, nullptr, 0.f, 0};
  return subm_conv_impl(feats, nbr, w, order, n, cin, cout, kvol, out, ep, stream);
}

extern "C" int link_subm_conv_ln_add_relu(const float *feats, const int32_t *nbr, const float *w,
                                          const int32_t *order, int64_t n, int32_t cin, int32_t cout,
                                          int32_t kvol, const float *ln_w, const float *ln_b, float eps,
                                          const float *addend, int32_t relu, float *out, void *stream) {
  if (!ln_w || !ln_b) return LINK_ERR_ARG;
  const conv_epilogue ep = {ln_w, ln_b, addend, eps, relu & 3};
  return subm_conv_impl(feats, nbr, w, order, n, cin, cout, kvol, out, ep, stream);
}

static int subm_conv_impl(const float *feats, const int32_t *nbr, const float *w, const int32_t *order,
                          int64_t n, int32_t cin, int32_t cout, int32_t kvol, float *out,
                          const conv_epilogue &ep, void *stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || cin > 256 || cout > 256 || kvol <= 0) return LINK_ERR_ARG;
  const bool mfma_ok = (cin & 15) == 0 && (cout & 15) == 0 && cin <= 128 && cout <= 128 && kvol <= 27;
  if (n == 0) return LINK_OK;
  if (!feats || !nbr || !w || !out) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (mfma_ok && kvol <= 27 && n >= CONV_RESIDENT_MIN) {          // narrow layers: every W_k resident in LDS
    // the fused inference forms (epilogue set: link_subm_conv_ln_add_relu) take the fp16-split products, like the pair-list
    // form's inference kernel; the plain forward (what autograd records) stays on the exact f32 instruction
    const bool split = ep.ln_w != nullptr;
    if (cin == 16 && cout == 16) return launch_conv_resident<16, 16>(feats, nbr, w, order, n, kvol, out, ep, split, st);
    if (cin == 32 && cout == 32) return launch_conv_resident<32, 32>(feats, nbr, w, order, n, kvol, out, ep, split, st);
    if (cin == 16 && cout == 32) return launch_conv_resident<16, 32>(feats, nbr, w, order, n, kvol, out, ep, split, st);
    if (cin == 32 && cout == 16) return launch_conv_resident<32, 16>(feats, nbr, w, order, n, kvol, out, ep, split, st);
  }
  if (mfma_ok) {
#define LINK_CONV(I, O) if (cin == I && cout == O) return launch_conv_mfma<I, O>(feats, nbr, w, order, n, kvol, out, ep, st)
    LINK_CONV(16, 16); LINK_CONV(32, 32); LINK_CONV(48, 48); LINK_CONV(64, 64); LINK_CONV(80, 80); LINK_CONV(96, 96);
    LINK_CONV(112, 112); LINK_CONV(128, 128);
    // the channel changes of the reference encoders (cs = [32,32,64,128,256,256,128,96,96] x cr, linkunet.py:262)
    LINK_CONV(16, 32); LINK_CONV(32, 16); LINK_CONV(32, 64); LINK_CONV(64, 32); LINK_CONV(64, 128); LINK_CONV(128, 64);
    LINK_CONV(96, 128); LINK_CONV(128, 96); LINK_CONV(64, 96); LINK_CONV(96, 64); LINK_CONV(32, 96); LINK_CONV(96, 32);
    LINK_CONV(16, 64); LINK_CONV(64, 16); LINK_CONV(32, 128); LINK_CONV(128, 32);
#undef LINK_CONV
  }
  dim3 grid(blocks_for(n * 64, 256)), block(256);
  const int cpl = (cout + 63) / 64;
  if (cpl == 1) hipLaunchKernelGGL(k_subm_conv_generic<1>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out, ep);
  else if (cpl == 2) hipLaunchKernelGGL(k_subm_conv_generic<2>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out, ep);
  else if (cpl == 3) hipLaunchKernelGGL(k_subm_conv_generic<3>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out, ep);
  else hipLaunchKernelGGL(k_subm_conv_generic<4>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out, ep);
  return check_launch("link_subm_conv_forward");
}
