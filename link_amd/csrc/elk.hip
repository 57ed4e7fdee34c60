// link_amd/csrc/elk.hip -- section C of include/link_amd.h: the fused R_core of ELKBlock.forward
// (segmentation/core/models/semantic_kitti/linkunet.py:124-185; detection/det3d/models/utils/
// ts_elk.py:144-230) for gfx950.
//
//   k_premix_ln_tlp      fin = LayerNorm(F @ Wpre^T): the only GEMM-shaped step -> f32 MFMA
//                        (v_mfma_f32_16x16x4_f32, exact f32), W staged once per workgroup in LDS
//                        (row stride padded by 4 dwords: conflict-free ds_read_b128), D = W * F^T so
//                        that a lane ends up holding 16t+4g+r channels of ONE voxel: the LayerNorm
//                        reduction is 16 in-lane adds + 2 cross-lane steps, and stores are 16 B/lane.
//   k_modulate_sum_g     theta / sincos / modulate / per-block pre-aggregation: 16-lane groups, each
//                        walking a run of consecutive blocks, sums in registers in ascending voxel id,
//                        ONE non-atomic row write per block.
//   k_block_gather_g     r^3 neighbour-block sum as r column sums with a z-sliding register ring,
//                        neighbour ids of 8 blocks resolved into LDS in one round trip -> A table.
//   k_voxel_demod_ln_g   loop-free per-voxel(-pair) de-modulate + LayerNorm + store.
//   k_gather_demod_ln*   fused gather+demod forms (lane=channel generic fallback for any C / r <= 5,
//                        and the group form used when no A scratch is supplied).
//
// S layout: m_cap rows [part0 C | part1 C | (part2 C)] fp32 (row = P*C floats: 512 B at C=64, i.e.
// exactly 4 aligned 128-B lines -- the gather is bound by L2->L1 line requests), followed by the
// (m_cap+1)-th all-zero row (row id m_cap: what an absent neighbour points at, so the gather needs no
// validity selects) and then the m_cap+1 per-block counts (fp32) at S + (m_cap+1)*P*C.
#include <limits.h>

#include "elk_common.h"

using namespace link;

static int g_premix_wgs_fwd();
static bool g_wt_fwd();

// ---------------------------------------------------------------------------------------------
// pre_mix + LayerNorm (MFMA path, C % 16 == 0, C <= 128)
// ---------------------------------------------------------------------------------------------
// TLP variant: W staged once per workgroup in LDS (padded rows: conflict-free ds_read_b128), no
// software pipelining at all -- a wave is load -> 64 MFMA -> LayerNorm -> store per tile, and 5-6
// resident waves per SIMD overlap each other's phases (register rotation / predication defeat the
// compiler's counted waits on this target, occupancy does not).
template <int C>
__global__ void __launch_bounds__(256) k_premix_ln_tlp(const float *__restrict__ feats,
                                                       const float *__restrict__ w_pre,
                                                       const float *__restrict__ ln_w,
                                                       const float *__restrict__ ln_b, int64_t n,
                                                       float eps, float *__restrict__ fin, bool wt) {
  constexpr int T = C / 16;
  constexpr int LDW = C + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  for (int e = tid * 4; e < C * C; e += 256 * 4) {
    int r = e / C, col = e - r * C;
    *reinterpret_cast<float4 *>(&w_lds[r * LDW + col]) = *reinterpret_cast<const float4 *>(&w_pre[e]);
  }
  __syncthreads();
  const int64_t tiles = (n + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t v = tile * 16 + li;
    const bool ok = v < n;
    const int64_t vl = ok ? v : n - 1;
    float4 f[T];
#pragma unroll
    for (int t = 0; t < T; t++) f[t] = *reinterpret_cast<const float4 *>(&feats[vl * C + 16 * t + 4 * g]);
    floatx4 acc[T];
#pragma unroll
    for (int tp = 0; tp < T; tp++) acc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; t++) {
#pragma unroll
      for (int tp = 0; tp < T; tp++) {
        float4 a = *reinterpret_cast<const float4 *>(&w_lds[(16 * tp + li) * LDW + 16 * t + 4 * g]);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f[t].x, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f[t].y, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f[t].z, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f[t].w, acc[tp], 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int tp = 0; tp < T; tp++) s += (acc[tp][0] + acc[tp][1]) + (acc[tp][2] + acc[tp][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int tp = 0; tp < T; tp++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float d = acc[tp][r] - mean;
        q += d * d;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
    if (ok) {
#pragma unroll
      for (int tp = 0; tp < T; tp++) {
        const float4 lw = *reinterpret_cast<const float4 *>(&ln_w[16 * tp + 4 * g]);   // L1-resident
        const float4 lb = *reinterpret_cast<const float4 *>(&ln_b[16 * tp + 4 * g]);
        float4 o;
        o.x = (acc[tp][0] - mean) * rstd * lw.x + lb.x;
        o.y = (acc[tp][1] - mean) * rstd * lw.y + lb.y;
        o.z = (acc[tp][2] - mean) * rstd * lw.z + lb.z;
        o.w = (acc[tp][3] - mean) * rstd * lw.w + lb.w;
        store_out(fin, v * C + 16 * tp + 4 * g, o, wt);
      }
    }
  }
}

template <int C>
static int launch_premix_tlp(const float *feats, const float *w_pre, const float *ln_w, const float *ln_b,
                             int64_t n, float eps, float *fin, hipStream_t st) {
  size_t lds = (size_t)C * (C + 4) * sizeof(float);
  int64_t tiles = (n + 15) / 16;
  int64_t wgs = (tiles + 3) / 4;
  if (wgs > g_premix_wgs_fwd()) wgs = g_premix_wgs_fwd();
  if (lds > 64 * 1024) {   // beyond the default dynamic-LDS limit: opt in once per kernel (160 KB per CU on gfx950)
    // per device and cheap (a host-side table write): no process-wide once-flag, which a second GPU or a
    // device reset would never pass again
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_premix_ln_tlp<C>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(k_premix_ln_tlp<C>, dim3((unsigned)wgs), dim3(256), lds, st, feats, w_pre, ln_w, ln_b, n,
                     eps, fin, g_wt_fwd() && wt_ok(n, C));
  return check_launch("link_premix_ln");
}

// generic fallback (any C <= 256): one wave per voxel, lanes = output channels, W read through L1/L2
template <int CPL>
__global__ void __launch_bounds__(256) k_premix_ln_generic(const float *__restrict__ feats,
                                                           const float *__restrict__ w_pre,
                                                           const float *__restrict__ ln_w,
                                                           const float *__restrict__ ln_b, int64_t n,
                                                           int c, float eps, float *__restrict__ fin) {
  int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (v >= n) return;
  float acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) acc[q] = 0.f;
  const float *row = feats + v * c;
  for (int k = 0; k < c; k++) {
    float fk = row[k];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      int ch = lane + 64 * q;
      if (ch < c) acc[q] = fmaf(fk, w_pre[(int64_t)ch * c + k], acc[q]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < CPL; q++) s += (lane + 64 * q < c) ? acc[q] : 0.f;
  s = wave_sum(s);
  float mean = s / c;
  float qq = 0.f;
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    float d = (lane + 64 * q < c) ? acc[q] - mean : 0.f;
    qq += d * d;
  }
  qq = wave_sum(qq);
  float rstd = 1.0f / sqrtf(qq / c + eps);
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    int ch = lane + 64 * q;
    if (ch < c) fin[v * c + ch] = (acc[q] - mean) * rstd * ln_w[ch] + ln_b[ch];
  }
}

extern "C" int link_premix_ln(const float *feats, const float *w_pre, const float *ln_w,
                              const float *ln_b, int64_t n, int32_t c, float eps, float *fin,
                              void *stream) {
  if (n < 0 || c <= 0 || c > 256) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!feats || !w_pre || !ln_w || !ln_b || !fin) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (c) {
    case 16: return launch_premix_tlp<16>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 32: return launch_premix_tlp<32>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 48: return launch_premix_tlp<48>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 64: return launch_premix_tlp<64>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 80: return launch_premix_tlp<80>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 96: return launch_premix_tlp<96>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 112: return launch_premix_tlp<112>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 128: return launch_premix_tlp<128>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    default: break;
  }
  dim3 grid(blocks_for(n * 64, 256)), block(256);
  int cpl = (c + 63) / 64;
  if (cpl == 1) hipLaunchKernelGGL(k_premix_ln_generic<1>, grid, block, 0, st, feats, w_pre, ln_w, ln_b, n, (int)c, eps, fin);
  else if (cpl == 2) hipLaunchKernelGGL(k_premix_ln_generic<2>, grid, block, 0, st, feats, w_pre, ln_w, ln_b, n, (int)c, eps, fin);
  else if (cpl == 3) hipLaunchKernelGGL(k_premix_ln_generic<3>, grid, block, 0, st, feats, w_pre, ln_w, ln_b, n, (int)c, eps, fin);
  else hipLaunchKernelGGL(k_premix_ln_generic<4>, grid, block, 0, st, feats, w_pre, ln_w, ln_b, n, (int)c, eps, fin);
  return check_launch("link_premix_ln");
}


// Work partition shared by the two block kernels: the sorted block range [0,M) is cut into 8
// contiguous slabs, one per XCD (workgroup w runs on XCD w % 8 -- observed dispatch, used for L2
// affinity only), and each slab into equal chunks of consecutive blocks, one chunk per wave.
// Consecutive sorted blocks are z-neighbours in the grid, so a chunk is a run along z and a slab is a
// range of x-planes: the rows a slab gathers (its own + one halo plane each side) fit its XCD's L2.
__device__ __forceinline__ void wave_chunk(int m, int &b0, int &b1) {
  const int xcd = blockIdx.x & 7;
  const int waves_per_xcd = (gridDim.x >> 3) * (blockDim.x >> 6);
  const int wq = (blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int lo = (int)(((long long)m * xcd) >> 3), hi = (int)(((long long)m * (xcd + 1)) >> 3);
  const int ch = (hi - lo + waves_per_xcd - 1) / waves_per_xcd;
  b0 = lo + wq * ch;
  b1 = b0 + ch;
  if (b1 > hi) b1 = hi;
}

// ---------------------------------------------------------------------------------------------
// modulate + per-block pre-aggregation
// ---------------------------------------------------------------------------------------------
template <int CPL, int OP>
__global__ void __launch_bounds__(256) k_modulate_sum(const float *__restrict__ fin,
                                                      const int4 *__restrict__ vox_sorted,
                                                      const float *__restrict__ w_pos,
                                                      const float *__restrict__ alpha,
                                                      const int32_t *__restrict__ blk_start,
                                                      const int32_t *__restrict__ hdr, int c, int cg,
                                                      float coord_div, float *__restrict__ S,
                                                      int64_t m_cap) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  const int lane = threadIdx.x & 63;
  int b0, b1;
  wave_chunk(hdr[LINK_HDR_M], b0, b1);
  const int rs = P * c;
  float *__restrict__ Scnt = S + (m_cap + 1) * rs;
  if (blockIdx.x == 0 && threadIdx.x < 64) {       // the all-zero row absent neighbours point at
    float *zrow = S + m_cap * rs;
    for (int k = threadIdx.x; k < rs; k += 64) zrow[k] = 0.f;
    if (threadIdx.x == 0) Scnt[m_cap] = 0.f;
  }
  if (b0 >= b1) return;
  float w0[CPL], w1[CPL], w2[CPL], al[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    int ch = lane + 64 * q;
    int tc = (ch < c) ? ch % cg : 0;
    w0[q] = w_pos[3 * tc + 0]; w1[q] = w_pos[3 * tc + 1]; w2[q] = w_pos[3 * tc + 2];
    al[q] = alpha ? alpha[tc] : 1.0f;
  }
  const float inv_div = 1.0f;  (void)inv_div;
  for (int bb = b0; bb < b1; bb += 63) {          // up to 63 blocks per pass (64 segment bounds)
    const int nb = (b1 - bb < 63) ? (b1 - bb) : 63;
    const int bs_l = blk_start[bb + ((lane <= nb) ? lane : nb)];
    const int p_lo = __shfl(bs_l, 0, 64), p_hi = __shfl(bs_l, nb, 64);
    int j = 0;                                     // current block within the pass
    int seg_end = __shfl(bs_l, 1, 64);
    float a0[CPL], a1[CPL], a2[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) a0[q] = a1[q] = a2[q] = 0.f;
    for (int p0 = p_lo; p0 < p_hi; p0 += 64) {
      const int pl = p0 + lane;
      int4 vs = (pl < p_hi) ? vox_sorted[pl] : make_int4(0, 0, 0, 0);
      const int lim = (p_hi - p0 < 64) ? (p_hi - p0) : 64;
      for (int t = 0; t < lim; t++) {
        const int p = p0 + t;
        while (p >= seg_end) {                     // flush finished block(s)
          float *row = S + (int64_t)(bb + j) * rs;
#pragma unroll
          for (int q = 0; q < CPL; q++) {
            int ch = lane + 64 * q;
            if (ch < c) {
              row[ch] = a0[q]; row[c + ch] = a1[q];
              if (OP == LINK_OP_COSX) row[2 * c + ch] = a2[q];
            }
            a0[q] = a1[q] = a2[q] = 0.f;
          }
          const int seg_beg = __shfl(bs_l, j, 64);          // all lanes participate in the shuffle
          if (lane == 0) Scnt[bb + j] = (float)(seg_end - seg_beg);
          j++;
          seg_end = __shfl(bs_l, j + 1, 64);
        }
        const int i = __shfl(vs.w, t, 64);
        float x = (float)__shfl(vs.x, t, 64), y = (float)__shfl(vs.y, t, 64), z = (float)__shfl(vs.z, t, 64);
        if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
        for (int q = 0; q < CPL; q++) {
          int ch = lane + 64 * q;
          if (ch < c) {
            float f = fin[(int64_t)i * c + ch];
            float th = theta_of(x, y, z, w0[q], w1[q], w2[q], al[q]);
            float sn, cs;
            sincos_fast(th, sn, cs);
            if (OP == LINK_OP_SIN) { a0[q] += f * sn; a1[q] += f * cs; }
            else { a0[q] += f * cs; a1[q] += f * sn; }
            if (OP == LINK_OP_COSX) a2[q] += link_mul_rn(f, th);      // a rounded product, as the reference sums it (common.h)
          }
        }
      }
    }
    // flush the remaining block(s) of the pass
    while (j < nb) {
      float *row = S + (int64_t)(bb + j) * rs;
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        int ch = lane + 64 * q;
        if (ch < c) {
          row[ch] = a0[q]; row[c + ch] = a1[q];
          if (OP == LINK_OP_COSX) row[2 * c + ch] = a2[q];
        }
        a0[q] = a1[q] = a2[q] = 0.f;
      }
      const int seg_len = __shfl(bs_l, j + 1, 64) - __shfl(bs_l, j, 64);
      if (lane == 0) Scnt[bb + j] = (float)seg_len;
      j++;
    }
  }
}

// launch geometry (fixed: chosen from the sweeps recorded in DESIGN.md 5a; nothing here is mutable process state -- the
// alternative kernel paths are selected per call through link_elk_desc_t::flags)
static constexpr int g_modsum_wgs = 1024;   // flat 512..2048 in the sweep, smaller grids co-run better
static constexpr int g_gather_wgs = 1024;   // 4 waves/SIMD resident at ~100 VGPRs -> one resident round
static constexpr int g_premix_wgs = 1024;
static constexpr int g_coop_threshold = 4;  // mean voxels/block above which a wave's groups cooperate per block
static constexpr int g_bgather_wgs = 512;
static constexpr int g_wt = 15;             // write-through (sc1) output stores in all four streaming kernels

template <int CPL>
static void launch_modsum(int op, hipStream_t st, const float *fin, const int4 *vox, const float *w_pos,
                          const float *alpha, const int32_t *blk_start, const int32_t *hdr, int c, int cg,
                          float div, float *S, int64_t m_cap) {
  dim3 grid(g_modsum_wgs), block(256);
  if (op == LINK_OP_COS)
    hipLaunchKernelGGL((k_modulate_sum<CPL, LINK_OP_COS>), grid, block, 0, st, fin, vox, w_pos, alpha, blk_start, hdr, c, cg, div, S, m_cap);
  else if (op == LINK_OP_SIN)
    hipLaunchKernelGGL((k_modulate_sum<CPL, LINK_OP_SIN>), grid, block, 0, st, fin, vox, w_pos, alpha, blk_start, hdr, c, cg, div, S, m_cap);
  else
    hipLaunchKernelGGL((k_modulate_sum<CPL, LINK_OP_COSX>), grid, block, 0, st, fin, vox, w_pos, alpha, blk_start, hdr, c, cg, div, S, m_cap);
}

static bool modsum_group_path(const link_elk_desc_t *d, hipStream_t st, const float *fin, const int4 *vox,
                              const float *w_pos, const float *alpha, const int32_t *blk_start,
                              const int32_t *hdr, float *S_, int64_t m_cap, int op = -1,
                              const float *row_den = nullptr);
static bool gather_group_path(const link_elk_desc_t *d, hipStream_t st, const float *S_, const float *fin,
                              const int4 *vox, const float *w_pos, const float *alpha, const float *ln_w,
                              const float *ln_b, const int32_t *blk_start, const int4 *blk_coords,
                              const int32_t *cell_blk, const link_grid_t &g, const int32_t *hdr, float *out,
                              int64_t m_cap);
static int check_desc(const link_elk_desc_t *d) {
  if (!d) return LINK_ERR_ARG;
  if (d->op < 0 || d->op > 2 || d->c <= 0 || d->c > 256 || d->cg <= 0 || d->cg > d->c) return LINK_ERR_ARG;
  if (d->r <= 0 || d->r > 7 || d->coord_div == 0.f) return LINK_ERR_ARG;
  return LINK_OK;
}

extern "C" int link_modulate_block_sum(const float *fin, const int32_t *vox_sorted, const float *w_pos,
                                       const float *alpha, const int32_t *blk_start, const int32_t *hdr,
                                       const link_elk_desc_t *desc, int64_t n, int64_t m_cap, float *S_,
                                       void *stream) {
  if (check_desc(desc) != LINK_OK || n < 0 || m_cap < 0) return LINK_ERR_ARG;
  if (n == 0 || m_cap == 0) return LINK_OK;
  if (!fin || !vox_sorted || !w_pos || !blk_start || !hdr || !S_) return LINK_ERR_ARG;
  const int4 *v4 = reinterpret_cast<const int4 *>(vox_sorted);
  int cpl = (desc->c + 63) / 64;
  hipStream_t st = S(stream);
  if (modsum_group_path(desc, st, fin, v4, w_pos, alpha, blk_start, hdr, S_, m_cap)) return check_launch("link_modulate_block_sum");
  switch (cpl) {
    case 1: launch_modsum<1>(desc->op, st, fin, v4, w_pos, alpha, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_, m_cap); break;
    case 2: launch_modsum<2>(desc->op, st, fin, v4, w_pos, alpha, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_, m_cap); break;
    case 3: launch_modsum<3>(desc->op, st, fin, v4, w_pos, alpha, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_, m_cap); break;
    default: launch_modsum<4>(desc->op, st, fin, v4, w_pos, alpha, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_, m_cap); break;
  }
  return check_launch("link_modulate_block_sum");
}

// ---------------------------------------------------------------------------------------------
// neighbour-block sum + normalise + de-modulate + LayerNorm
//
// The r^3 neighbourhood sum is evaluated as r "column sums" col(z') = sum over the r^2 (dx,dy)
// neighbours at height z' (in get_kernel_offsets order within the plane), out = sum_z' col(z').
// A wave walks a run of consecutive sorted blocks = increasing z in one (x,y) column, so when it
// steps from z to z+1 it keeps r-1 of the r column sums in registers and gathers only r^2 new rows
// instead of r^3 (3x fewer row reads at r=3).  Summation order is fixed (deterministic), though not
// the k-ascending order of the reference kernel (difference ~1e-7 rel, far inside the 1e-4 gate).
// ---------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void plane_offset(int t, int &ox, int &oy) {
  constexpr int LO = -((R + 1) / 2) + 1;      // nn/utils/kernel.py:21
  if ((R & 1) != 0) { ox = LO + (t % R); oy = LO + (t / R); }   // odd volume: x fastest
  else { oy = LO + (t % R); ox = LO + (t / R); }                 // even volume: y faster than x
}

template <int CPL, int OP, int R>
__global__ void __launch_bounds__(256) k_gather_demod_ln(
    const float *__restrict__ S, const float *__restrict__ fin, const int4 *__restrict__ vox_sorted,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, const int32_t *__restrict__ blk_start,
    const int4 *__restrict__ blk_coords, const int32_t *__restrict__ cell_blk, link_grid_t g,
    const int32_t *__restrict__ hdr, int c, int cg, float coord_div, float eps, float *__restrict__ out,
    int64_t m_cap) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int ZLO = -((R + 1) / 2) + 1;
  constexpr int SUB = (R3 <= 27) ? 16 : 8;       // blocks per pass: their neighbour ids live in LDS
  __shared__ int32_t s_nb[4][SUB * R3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int b0, b1;
  wave_chunk(hdr[LINK_HDR_M], b0, b1);
  if (b0 >= b1) return;
  const int rs = P * c;
  const float *__restrict__ Scnt = S + (m_cap + 1) * rs;
  float w0[CPL], w1[CPL], w2[CPL], al[CPL], gw[CPL], gb[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    int ch = lane + 64 * q;
    int tc = (ch < c) ? ch % cg : 0;
    w0[q] = w_pos[3 * tc + 0]; w1[q] = w_pos[3 * tc + 1]; w2[q] = w_pos[3 * tc + 2];
    al[q] = alpha ? alpha[tc] : 1.0f;
    gw[q] = (ch < c) ? ln_w[ch] : 0.f;
    gb[q] = (ch < c) ? ln_b[ch] : 0.f;
  }
  // column-sum ring: slot d holds col(zc + ZLO + d) for the current centre (px,py,pw,zc)
  float col[R][P][CPL], cden[R];
  int px = 0, py = 0, pw = 0, pz = 0;
  bool have = false;
  int32_t *my_nb = s_nb[wave];

  for (int bb = b0; bb < b1; bb += SUB) {
    const int nbk = (b1 - bb < SUB) ? (b1 - bb) : SUB;
    // ---- pass prologue (ONE memory round trip for the whole pass): block coordinates, segment
    //      bounds, and all nbk*R^3 neighbour ids (cell table lookups, lane-parallel) -> LDS
    const int4 bc_l = blk_coords[bb + ((lane < nbk) ? lane : nbk - 1)];
    const int bs_l = blk_start[bb + ((lane <= nbk) ? lane : nbk)];
    for (int e = lane; e < nbk * R3; e += 64) {
      const int j = e / R3, k = e - j * R3;
      const int d = k / R2, t = k - d * R2;
      int ox, oy;
      plane_offset<R>(t, ox, oy);
      const int bx = __shfl(bc_l.x, j, 64), by = __shfl(bc_l.y, j, 64);
      const int bz = __shfl(bc_l.z, j, 64), bw = __shfl(bc_l.w, j, 64);
      const int32_t cell = cell_of(g, bx + ox, by + oy, bz + ZLO + d, bw);
      my_nb[e] = (cell >= 0) ? cell_blk[cell] - 1 : -1;
    }
    const int p_lo = __shfl(bs_l, 0, 64), p_hi = __shfl(bs_l, nbk, 64);
    int vs_base = p_lo;
    int4 vs = (p_lo + lane < p_hi) ? vox_sorted[p_lo + lane] : make_int4(0, 0, 0, 0);

    for (int j = 0; j < nbk; j++) {
      const int bx = __shfl(bc_l.x, j, 64), by = __shfl(bc_l.y, j, 64);
      const int bz = __shfl(bc_l.z, j, 64), bw = __shfl(bc_l.w, j, 64);
      const int st = __shfl(bs_l, j, 64), en = __shfl(bs_l, j + 1, 64);
      // how many ring slots survive the move to this block
      int keep = 0;
      if (have && bx == px && by == py && bw == pw) {
        int dz = bz - pz;
        if (dz > 0 && dz < R) keep = R - dz;
      }
      const int shift = R - keep;
#pragma unroll
      for (int d = 0; d < R; d++) {               // shift the surviving slots down (static indices)
#pragma unroll
        for (int sft = 1; sft < R; sft++) {
          if (shift == sft && d + sft < R) {
#pragma unroll
            for (int pp = 0; pp < P; pp++)
#pragma unroll
              for (int q = 0; q < CPL; q++) col[d][pp][q] = col[d + sft][pp][q];
            cden[d] = cden[d + sft];
          }
        }
      }
#pragma unroll
      for (int d = 0; d < R; d++) {
        if (d >= keep) {                          // wave-uniform: gather plane z + ZLO + d
          float v[R2][P][CPL], vd[R2];
          int32_t nbv[R2];
#pragma unroll
          for (int t = 0; t < R2; t++) nbv[t] = my_nb[(j * R + d) * R2 + t];   // LDS broadcast reads
#pragma unroll
          for (int t = 0; t < R2; t++) {          // issue all row loads of the plane back to back
            const float *row = S + (int64_t)((nbv[t] >= 0) ? nbv[t] : 0) * rs;
            vd[t] = Scnt[(nbv[t] >= 0) ? nbv[t] : 0];
#pragma unroll
            for (int pp = 0; pp < P; pp++)
#pragma unroll
              for (int q = 0; q < CPL; q++) {
                int ch = lane + 64 * q;
                v[t][pp][q] = row[pp * c + ((ch < c) ? ch : 0)];
              }
          }
          float acc[P][CPL], den = 0.f;
#pragma unroll
          for (int pp = 0; pp < P; pp++)
#pragma unroll
            for (int q = 0; q < CPL; q++) acc[pp][q] = 0.f;
#pragma unroll
          for (int t = 0; t < R2; t++) {          // then sum them in plane order
            if (nbv[t] >= 0) {
              den += vd[t];
#pragma unroll
              for (int pp = 0; pp < P; pp++)
#pragma unroll
                for (int q = 0; q < CPL; q++) acc[pp][q] += v[t][pp][q];
            }
          }
#pragma unroll
          for (int pp = 0; pp < P; pp++)
#pragma unroll
            for (int q = 0; q < CPL; q++) col[d][pp][q] = acc[pp][q];
          cden[d] = den;
        }
      }
      have = true; px = bx; py = by; pw = bw; pz = bz;
      // out = sum of the R column sums, normalised by the summed count (utils.py:80)
      float A[P][CPL], den = cden[0];
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int q = 0; q < CPL; q++) A[pp][q] = col[0][pp][q];
#pragma unroll
      for (int d = 1; d < R; d++) {
        den += cden[d];
#pragma unroll
        for (int pp = 0; pp < P; pp++)
#pragma unroll
          for (int q = 0; q < CPL; q++) A[pp][q] += col[d][pp][q];
      }
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int q = 0; q < CPL; q++) A[pp][q] = A[pp][q] / den;

      // voxels of the block (records come from the prefetched 64-wide window)
      for (int p = st; p < en; p++) {
        if (p - vs_base >= 64) {                  // refill the window (wave-uniform)
          vs_base = p;
          vs = (p + lane < p_hi) ? vox_sorted[p + lane] : make_int4(0, 0, 0, 0);
        }
        const int t = p - vs_base;
        const int i = __shfl(vs.w, t, 64);
        float x = (float)__shfl(vs.x, t, 64), y = (float)__shfl(vs.y, t, 64), z = (float)__shfl(vs.z, t, 64);
        if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
        float nv[CPL];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; q++) {
          int ch = lane + 64 * q;
          nv[q] = 0.f;
          if (ch < c) {
            float th = theta_of(x, y, z, w0[q], w1[q], w2[q], al[q]);
            float sn, cs;
            sincos_fast(th, sn, cs);
            float vv;
            if (OP == LINK_OP_SIN) vv = __fsub_rn(__fmul_rn(A[0][q], cs), __fmul_rn(A[1][q], sn));   // linkunet.py:148
            else vv = __fadd_rn(__fmul_rn(A[0][q], cs), __fmul_rn(A[1][q], sn));                      // :162
            if (OP == LINK_OP_COSX) {
              float f = fin[(int64_t)i * c + ch];
              vv = __fadd_rn(vv, __fsub_rn(A[P - 1][q], link_mul_rn(f, th)));                           // :176
            }
            nv[q] = vv;
            s += vv;
          }
        }
        s = wave_sum(s);
        const float mean = s / c;
        float qq = 0.f;
#pragma unroll
        for (int q = 0; q < CPL; q++) {
          float dd = (lane + 64 * q < c) ? nv[q] - mean : 0.f;
          qq += dd * dd;
        }
        qq = wave_sum(qq);
        const float rstd = 1.0f / sqrtf(qq / c + eps);
#pragma unroll
        for (int q = 0; q < CPL; q++) {
          int ch = lane + 64 * q;
          if (ch < c) out[(int64_t)i * c + ch] = (nv[q] - mean) * rstd * gw[q] + gb[q];
        }
      }
    }
  }
}

template <int CPL, int OP>
static void launch_gdl_r(int r, hipStream_t st, int64_t m_cap, const float *S_, const float *fin, const int4 *vox,
                         const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                         const int32_t *blk_start, const int4 *blk_coords, const int32_t *cell_blk,
                         const link_grid_t &g, const int32_t *hdr, const link_elk_desc_t &d, float *out) {
  dim3 grid(g_gather_wgs), block(256);
#define LINK_GDL(RR)                                                                                   \
  hipLaunchKernelGGL((k_gather_demod_ln<CPL, OP, RR>), grid, block, 0, st, S_, fin, vox, w_pos, alpha, \
                     ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d.c, d.cg, d.coord_div, d.eps, out, m_cap)
  switch (r) {
    case 1: LINK_GDL(1); break;
    case 2: LINK_GDL(2); break;
    case 3: LINK_GDL(3); break;
    case 4: LINK_GDL(4); break;
    default: LINK_GDL(5); break;
  }
#undef LINK_GDL
}

template <int CPL>
static void launch_gdl(int op, int r, hipStream_t st, int64_t m_cap, const float *S_, const float *fin, const int4 *vox,
                       const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                       const int32_t *blk_start, const int4 *blk_coords, const int32_t *cell_blk,
                       const link_grid_t &g, const int32_t *hdr, const link_elk_desc_t &d, float *out) {
  if (op == LINK_OP_COS)
    launch_gdl_r<CPL, LINK_OP_COS>(r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d, out);
  else if (op == LINK_OP_SIN)
    launch_gdl_r<CPL, LINK_OP_SIN>(r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d, out);
  else
    launch_gdl_r<CPL, LINK_OP_COSX>(r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d, out);
}

extern "C" int link_gather_demod_ln(const float *S_, const float *fin, const int32_t *vox_sorted,
                                    const float *w_pos, const float *alpha, const float *ln_w,
                                    const float *ln_b, const int32_t *blk_start,
                                    const int32_t *blk_coords, const int32_t *cell_blk,
                                    const link_grid_t *grid, const int32_t *hdr,
                                    const link_elk_desc_t *desc, int64_t n, int64_t m_cap, float *out,
                                    void *stream) {
  if (check_desc(desc) != LINK_OK || !grid || n < 0 || m_cap < 0 || desc->r > 5) return LINK_ERR_ARG;
  if (n == 0 || m_cap == 0) return LINK_OK;
  if (!S_ || !vox_sorted || !w_pos || !ln_w || !ln_b || !blk_start || !blk_coords || !cell_blk || !hdr || !out)
    return LINK_ERR_ARG;
  if (desc->op == LINK_OP_COSX && !fin) return LINK_ERR_ARG;
  const int4 *v4 = reinterpret_cast<const int4 *>(vox_sorted);
  const int4 *b4 = reinterpret_cast<const int4 *>(blk_coords);
  hipStream_t st = S(stream);
  int cpl = (desc->c + 63) / 64;
  if (gather_group_path(desc, st, S_, fin, v4, w_pos, alpha, ln_w, ln_b, blk_start, b4, cell_blk, *grid, hdr, out, m_cap))
    return check_launch("link_gather_demod_ln");
  switch (cpl) {
    case 1: launch_gdl<1>(desc->op, desc->r, st, m_cap, S_, fin, v4, w_pos, alpha, ln_w, ln_b, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
    case 2: launch_gdl<2>(desc->op, desc->r, st, m_cap, S_, fin, v4, w_pos, alpha, ln_w, ln_b, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
    case 3: launch_gdl<3>(desc->op, desc->r, st, m_cap, S_, fin, v4, w_pos, alpha, ln_w, ln_b, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
    default: launch_gdl<4>(desc->op, desc->r, st, m_cap, S_, fin, v4, w_pos, alpha, ln_w, ln_b, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
  }
  return check_launch("link_gather_demod_ln");
}

static int g_premix_wgs_fwd() { return g_premix_wgs; }
static bool g_wt_fwd() { return (g_wt & 1) != 0; }

// =============================================================================================
// Sub-wave ("group") kernels: the fast path for C % 4 == 0.
//
// A feature row of C floats is owned by LPR = pow2ceil(C/4) adjacent lanes, 4 consecutive channels
// (one 16-byte access) per lane, so a 256-byte row of C=64 is ONE 16-lane dwordx4 access instead of a
// 64-lane dword access, and a 64-lane wave carries 64/LPR independent work streams ("groups"), each
// walking its own run of consecutive blocks.  Versus lane=channel this cuts vector-memory
// instructions 4x, quadruples the loads in flight per wave (the kernels are latency/issue bound, not
// HBM bound, at LiDAR sizes), needs no wave-wide broadcasts, and keeps LayerNorm reductions inside
// a 16-lane DPP row.  Groups of one wave may diverge (different segment lengths); nothing is shared
// between them except the instruction stream.
// =============================================================================================

// chunk of consecutive sorted blocks owned by this GROUP (XCD slab -> equal chunks per group)
template <int LPR>
__device__ __forceinline__ void group_chunk(int m, int &b0, int &b1, int nwg = 0) {
  constexpr int G = 64 / LPR;
  const int xcd = blockIdx.x & 7;
  if (nwg == 0) nwg = gridDim.x;                  // workgroups taking part (a launch may be wider: block gather)
  const int groups_per_xcd = (nwg >> 3) * (blockDim.x >> 6) * G;
  const int gq = ((blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + ((threadIdx.x & 63) / LPR);
  const int lo = (int)(((long long)m * xcd) >> 3), hi = (int)(((long long)m * (xcd + 1)) >> 3);
  const int ch = (hi - lo + groups_per_xcd - 1) / groups_per_xcd;
  b0 = lo + gq * ch;
  b1 = b0 + ch;
  if (b1 > hi) b1 = hi;
  if (b0 > hi) b0 = hi;
}



// PAIR (channel j and j + C/2 share theta, i.e. groups == 2 and the row fills its lanes exactly):
// a step handles TWO consecutive voxels A,B; the low half of the group evaluates sincos(theta_A), the
// high half sincos(theta_B), and the halves swap results with one DPP op per value -- every sincos
// evaluated once instead of twice.
template <int LPR, int OP, bool PAIR>
__device__ __forceinline__ void modulate_sum_per_group(const float *__restrict__ fin,
                                                       const int4 *__restrict__ vox_sorted,
                                                       const float *__restrict__ w_pos,
                                                       const float *__restrict__ alpha,
                                                       const int32_t *__restrict__ blk_start,
                                                       const int32_t *__restrict__ hdr, int c, int cg,
                                                       float coord_div, float *__restrict__ S,
                                                       int64_t m_cap, bool wt,
                                                       const float *__restrict__ row_den) {
  constexpr int P = op_parts<OP>::value;
  constexpr int STEP = PAIR ? 2 : 1;
  const int li = (threadIdx.x & 63) & (LPR - 1);
  const int ch0 = 4 * li;
  const bool act = ch0 < c;                       // lanes past C (non power-of-two widths) idle
  const bool hi = PAIR && (li >= LPR / 2);
  int b0, b1;
  group_chunk<LPR>(hdr[LINK_HDR_M], b0, b1);
  const int rs = P * c;
  float *__restrict__ Scnt = S + (m_cap + 1) * rs;
  if (blockIdx.x == 0 && threadIdx.x < LPR) {      // the all-zero row absent neighbours point at
    float *zrow = S + m_cap * rs;
    if (act)
      for (int pp = 0; pp < P; pp++) *reinterpret_cast<float4 *>(&zrow[pp * c + ch0]) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (li == 0) Scnt[m_cap] = 0.f;
  }
  if (b0 >= b1) return;
  float w0[4], w1[4], w2[4], al[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int tc = act ? (ch0 + e) % cg : 0;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
  }
  const int p_end = blk_start[b1];
  int p = blk_start[b0];
  int b = b0;
  int seg_beg = p, seg_end = blk_start[b0 + 1];
  float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
  const int p_last = p_end - 1;
  const int cofs = act ? ch0 : 0;
  // unconditional (clamped) loads so that the compiler can keep them in flight with counted waits:
  // records two steps ahead, feature rows one step ahead
  auto ld_rec = [&](int q) { return vox_sorted[(q <= p_last) ? q : p_last]; };
  auto ld_row = [&](int i) { return *reinterpret_cast<const float4 *>(&fin[(int64_t)i * c + cofs]); };
  auto flush = [&]() {                            // block finished: one row write, no atomics
    float *row = S + (int64_t)b * rs;
    const float k = row_den ? 1.0f / row_den[b] : 1.0f;    // backward: rows pre-divided by the forward's denominator
    if (act) {
      store_out(S, (int64_t)b * rs + ch0, make_float4(a0[0] * k, a0[1] * k, a0[2] * k, a0[3] * k), wt);
      store_out(S, (int64_t)b * rs + c + ch0, make_float4(a1[0] * k, a1[1] * k, a1[2] * k, a1[3] * k), wt);
      if (P == 3) store_out(S, (int64_t)b * rs + 2 * c + ch0, make_float4(a2[0] * k, a2[1] * k, a2[2] * k, a2[3] * k), wt);
    }
    if (li == 0) Scnt[b] = (float)(seg_end - seg_beg);
#pragma unroll
    for (int e = 0; e < 4; e++) a0[e] = a1[e] = a2[e] = 0.f;
    b++;
    seg_beg = seg_end;
    seg_end = blk_start[b + 1];
  };
  int4 rcA = ld_rec(p), rcB = ld_rec(p + STEP - 1);
  int4 rnA = ld_rec(p + STEP), rnB = ld_rec(p + 2 * STEP - 1);
  float4 fcA = ld_row(rcA.w), fcB = ld_row(rcB.w);
  for (; p < p_end; p += STEP) {
    const int4 r2A = ld_rec(p + 2 * STEP), r2B = ld_rec(p + 3 * STEP - 1);
    const float4 fnA = ld_row(rnA.w), fnB = ld_row(rnB.w);
    const bool hasB = PAIR && (p + 1 < p_end);
    const int4 own = (hi && hasB) ? rcB : rcA;
    float x = (float)own.x, y = (float)own.y, z = (float)own.z;
    if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
    float snA[4], csA[4], snB[4], csB[4], thA[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
      float sn, cs;
      sincos_fast(th, sn, cs);
      thA[e] = th;
      if (PAIR) {
        const float so = partner<LPR>(sn), co = partner<LPR>(cs);
        const bool swapped = hi && hasB;          // this lane evaluated voxel B
        snA[e] = swapped ? so : sn; csA[e] = swapped ? co : cs;
        snB[e] = hi ? sn : so;      csB[e] = hi ? cs : co;
      } else {
        snA[e] = sn; csA[e] = cs; snB[e] = sn; csB[e] = cs;
      }
    }
    if (p == seg_end) flush();
    {
      const float fv[4] = {fcA.x, fcA.y, fcA.z, fcA.w};
#pragma unroll
      for (int e = 0; e < 4; e++) mod_accum<OP>(a0[e], a1[e], a2[e], fv[e], snA[e], csA[e], thA[e]);
    }
    if (hasB) {                                   // PAIR forms have no third part (never cos_x)
      if (p + 1 == seg_end) flush();
      const float fv[4] = {fcB.x, fcB.y, fcB.z, fcB.w};
#pragma unroll
      for (int e = 0; e < 4; e++) mod_accum<OP>(a0[e], a1[e], a2[e], fv[e], snB[e], csB[e], 0.f);
    }
    rcA = rnA; rcB = rnB; rnA = r2A; rnB = r2B; fcA = fnA; fcB = fnB;
  }
  {                                               // last block of the chunk
    float *row = S + (int64_t)b * rs;
    const float k = row_den ? 1.0f / row_den[b] : 1.0f;
    if (act) {
      store_out(S, (int64_t)b * rs + ch0, make_float4(a0[0] * k, a0[1] * k, a0[2] * k, a0[3] * k), wt);
      store_out(S, (int64_t)b * rs + c + ch0, make_float4(a1[0] * k, a1[1] * k, a1[2] * k, a1[3] * k), wt);
      if (P == 3) store_out(S, (int64_t)b * rs + 2 * c + ch0, make_float4(a2[0] * k, a2[1] * k, a2[2] * k, a2[3] * k), wt);
    }
    if (li == 0) Scnt[b] = (float)(seg_end - seg_beg);
  }
}

// Large-block mode: the G = 64/LPR groups of a wave COOPERATE on one block -- group q takes voxel
// steps q, q+G, q+2G, ... of the block's segment (ascending), and the G partial sums are combined with
// a fixed shuffle tree ((g0+g1)+(g2+g3)), so the result is deterministic.  Used when blocks hold many
// voxels (LiDAR surfaces: N/M of 5..300), where one group per block would serialise long segments
// while most groups idle; chosen on the device from N/M (the host never learns M).
template <int LPR, int OP, bool PAIR>
__device__ __forceinline__ void modulate_sum_cooperative(const float *__restrict__ fin,
                                                         const int4 *__restrict__ vox_sorted,
                                                         const float *__restrict__ w_pos,
                                                         const float *__restrict__ alpha,
                                                         const int32_t *__restrict__ blk_start,
                                                         const int32_t *__restrict__ hdr, int c, int cg,
                                                         float coord_div, float *__restrict__ S,
                                                         int64_t m_cap, bool wt,
                                                         const float *__restrict__ row_den) {
  constexpr int P = op_parts<OP>::value;
  constexpr int G = 64 / LPR;
  constexpr int STEP = PAIR ? 2 : 1;
  const int lane = threadIdx.x & 63;
  const int li = lane & (LPR - 1), q = lane / LPR;
  const int ch0 = 4 * li;
  const bool act = ch0 < c;
  const bool hi = PAIR && (li >= LPR / 2);
  const int cofs = act ? ch0 : 0;
  int b0, b1;
  wave_chunk(hdr[LINK_HDR_M], b0, b1);
  const int rs = P * c;
  float *__restrict__ Scnt = S + (m_cap + 1) * rs;
  if (blockIdx.x == 0 && threadIdx.x < LPR) {
    float *zrow = S + m_cap * rs;
    if (act)
      for (int pp = 0; pp < P; pp++) *reinterpret_cast<float4 *>(&zrow[pp * c + ch0]) = make_float4(0.f, 0.f, 0.f, 0.f);
    if (li == 0) Scnt[m_cap] = 0.f;
  }
  if (b0 >= b1) return;
  float w0[4], w1[4], w2[4], al[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int tc = act ? (ch0 + e) % cg : 0;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
  }
  int st = blk_start[b0];
  for (int b = b0; b < b1; b++) {
    const int en = blk_start[b + 1];
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    const int p_last = en - 1;
    for (int p = st + q * STEP; p < en; p += G * STEP) {        // this group's steps of the segment
      const bool hasB = PAIR && (p + 1 < en);
      const int4 rcA = vox_sorted[p], rcB = vox_sorted[(p + 1 <= p_last) ? p + 1 : p_last];
      const float4 fA = *reinterpret_cast<const float4 *>(&fin[(int64_t)rcA.w * c + cofs]);
      const float4 fB = *reinterpret_cast<const float4 *>(&fin[(int64_t)rcB.w * c + cofs]);
      const int4 own = (hi && hasB) ? rcB : rcA;
      float x = (float)own.x, y = (float)own.y, z = (float)own.z;
      if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
      const float fvA[4] = {fA.x, fA.y, fA.z, fA.w}, fvB[4] = {fB.x, fB.y, fB.z, fB.w};
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
        float sn, cs;
        sincos_fast(th, sn, cs);
        float snA = sn, csA = cs, snB = sn, csB = cs;
        if (PAIR) {
          const float so = partner<LPR>(sn), co = partner<LPR>(cs);
          const bool swapped = hi && hasB;
          snA = swapped ? so : sn; csA = swapped ? co : cs;
          snB = hi ? sn : so;      csB = hi ? cs : co;
        }
        mod_accum<OP>(a0[e], a1[e], a2[e], fvA[e], snA, csA, th);
        if (hasB) mod_accum<OP>(a0[e], a1[e], a2[e], fvB[e], snB, csB, 0.f);
      }
    }
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {             // fixed combine tree over the G groups
#pragma unroll
      for (int e = 0; e < 4; e++) {
        a0[e] += __shfl_xor(a0[e], o, 64);
        a1[e] += __shfl_xor(a1[e], o, 64);
        if (P == 3) a2[e] += __shfl_xor(a2[e], o, 64);
      }
    }
    if (q == 0) {
      float *row = S + (int64_t)b * rs;
      const float k = row_den ? 1.0f / row_den[b] : 1.0f;
      if (act) {
        *reinterpret_cast<float4 *>(&row[ch0]) = make_float4(a0[0] * k, a0[1] * k, a0[2] * k, a0[3] * k);
        *reinterpret_cast<float4 *>(&row[c + ch0]) = make_float4(a1[0] * k, a1[1] * k, a1[2] * k, a1[3] * k);
        if (P == 3) *reinterpret_cast<float4 *>(&row[2 * c + ch0]) = make_float4(a2[0] * k, a2[1] * k, a2[2] * k, a2[3] * k);
      }
      if (li == 0) Scnt[b] = (float)(en - st);
    }
    st = en;
  }
}

template <int LPR, int OP, bool PAIR>
__global__ void __launch_bounds__(256) k_modulate_sum_g(const float *__restrict__ fin,
                                                        const int4 *__restrict__ vox_sorted,
                                                        const float *__restrict__ w_pos,
                                                        const float *__restrict__ alpha,
                                                        const int32_t *__restrict__ blk_start,
                                                        const int32_t *__restrict__ hdr, int c, int cg,
                                                        float coord_div, float *__restrict__ S,
                                                        int64_t m_cap, int coop_threshold, bool wt,
                                                        const float *__restrict__ row_den) {
  // mean voxels per block decides the mode (grid-uniform, read from the device-side header)
  const int m = hdr[LINK_HDR_M], nv = hdr[LINK_HDR_NVALID];
  if (LPR < 64 && (int64_t)nv > (int64_t)coop_threshold * (m > 0 ? m : 1))
    modulate_sum_cooperative<LPR, OP, PAIR>(fin, vox_sorted, w_pos, alpha, blk_start, hdr, c, cg, coord_div, S, m_cap, wt, row_den);
  else
    modulate_sum_per_group<LPR, OP, PAIR>(fin, vox_sorted, w_pos, alpha, blk_start, hdr, c, cg, coord_div, S, m_cap, wt, row_den);
}

template <int LPR, int OP, int R, bool PAIR>
__global__ void __launch_bounds__(256) k_gather_demod_ln_g(
    const float *__restrict__ S, const float *__restrict__ fin, const int4 *__restrict__ vox_sorted,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, const int32_t *__restrict__ blk_start,
    const int4 *__restrict__ blk_coords, const int32_t *__restrict__ cell_blk, link_grid_t g,
    const int32_t *__restrict__ hdr, int c, int cg, float coord_div, float eps, float *__restrict__ out,
    int64_t m_cap) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  constexpr int G = 64 / LPR;
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int ZLO = -((R + 1) / 2) + 1;
  constexpr int SUB = 8;                           // blocks per pass whose neighbour ids sit in LDS
  constexpr int STEP = PAIR ? 2 : 1;
  __shared__ int32_t s_nb[4 * G][SUB * R3];
  __shared__ int32_t s_bs[4 * G][SUB + 1];
  __shared__ int4 s_bc[4 * G][SUB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (LPR - 1), grp = wave * G + lane / LPR;
  const int ch0 = 4 * li;
  const bool act = ch0 < c;
  const bool hi = PAIR && (li >= LPR / 2);
  const int cofs = act ? ch0 : 0;
  int b0, b1;
  group_chunk<LPR>(hdr[LINK_HDR_M], b0, b1);
  const bool live = b0 < b1;                       // dead groups still take part in wave-level votes
  const int rs = P * c;
  const float *__restrict__ Scnt = S + (m_cap + 1) * rs;
  float w0[4], w1[4], w2[4], al[4], gw[4], gb[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int ch = act ? ch0 + e : 0;
    int tc = ch % cg;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
    gw[e] = ln_w[ch]; gb[e] = ln_b[ch];
  }
  float col[R][P][4], cden[R];
#pragma unroll
  for (int d = 0; d < R; d++) {
    cden[d] = 0.f;
#pragma unroll
    for (int pp = 0; pp < P; pp++)
#pragma unroll
      for (int e = 0; e < 4; e++) col[d][pp][e] = 0.f;
  }
  int px = 0, py = 0, pw = 0, pz = 0;
  bool have = false;
  int32_t *my_nb = s_nb[grp];
  int32_t *my_bs = s_bs[grp];
  int4 *my_bc = s_bc[grp];
  const float inv_c = 1.0f / (float)c;
  const int nblk = live ? b1 - b0 : 0;
  const int npass = (nblk + SUB - 1) / SUB;
  int wave_pass = npass;                           // wave-uniform trip count (max over its groups)
#pragma unroll
  for (int o = 32; o >= LPR; o >>= 1) wave_pass = max(wave_pass, __shfl_xor(wave_pass, o, 64));

  for (int ps = 0; ps < wave_pass; ps++) {
    const int bb = b0 + ps * SUB;
    int nbk = live ? (b1 - bb) : 0;
    nbk = nbk < 0 ? 0 : (nbk > SUB ? SUB : nbk);
    // ---- pass prologue: ONE round trip for block coords, segment bounds and all neighbour ids
    for (int e = li; e < nbk; e += LPR) my_bc[e] = blk_coords[bb + e];
    for (int e = li; e <= nbk && nbk > 0; e += LPR) my_bs[e] = blk_start[bb + e];
    for (int e = li; e < nbk * R3; e += LPR) {
      const int j = e / R3, k = e - j * R3;
      const int d = k / R2, t = k - d * R2;
      int ox, oy;
      plane_offset<R>(t, ox, oy);
      const int4 bc = blk_coords[bb + j];           // L1/L2 hit (same lines as above)
      const int32_t cell = cell_of(g, bc.x + ox, bc.y + oy, bc.z + ZLO + d, bc.w);
      my_nb[e] = (cell >= 0) ? cell_blk[cell] - 1 : -1;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int j = 0; j < SUB; j++) {
      const bool on = j < nbk;                      // this group still has a block in this pass
      if (!__any(on)) break;
      const int jj = on ? j : 0;
      const int4 bc = on ? my_bc[jj] : make_int4(0, 0, 0, 0);
      const int st = on ? my_bs[jj] : 0, en = on ? my_bs[jj + 1] : 0;
      int keep = 0;
      if (on && have && bc.x == px && bc.y == py && bc.w == pw) {
        int dz = bc.z - pz;
        if (dz > 0 && dz < R) keep = R - dz;
      }
      const int shift = R - keep;
#pragma unroll
      for (int d = 0; d < R; d++) {
#pragma unroll
        for (int sft = 1; sft < R; sft++) {
          const bool mv = on && shift == sft && d + sft < R;
#pragma unroll
          for (int pp = 0; pp < P; pp++)
#pragma unroll
            for (int e = 0; e < 4; e++) col[d][pp][e] = mv ? col[(d + sft < R) ? d + sft : d][pp][e] : col[d][pp][e];
          cden[d] = mv ? cden[(d + sft < R) ? d + sft : d] : cden[d];
        }
      }
      // ---- missing planes, ONE plane body per iteration; each group works on ITS next missing plane
      int dcur = on ? keep : R;
      while (__any(dcur < R)) {
        const bool doit = dcur < R;
        const int d = doit ? dcur : R - 1;
        float acc[P][4], den = 0.f;
#pragma unroll
        for (int pp = 0; pp < P; pp++)
#pragma unroll
          for (int e = 0; e < 4; e++) acc[pp][e] = 0.f;
        float4 v[R2][P];
        float vd[R2];
        int32_t nbv[R2];
#pragma unroll
        for (int t = 0; t < R2; t++) nbv[t] = doit ? my_nb[(jj * R + d) * R2 + t] : -1;
#pragma unroll
        for (int t = 0; t < R2; t++) {              // all row loads of the plane issued back to back
          const float *row = S + (int64_t)((nbv[t] >= 0) ? nbv[t] : 0) * rs;
          vd[t] = Scnt[(nbv[t] >= 0) ? nbv[t] : 0];
#pragma unroll
          for (int pp = 0; pp < P; pp++) v[t][pp] = *reinterpret_cast<const float4 *>(&row[pp * c + cofs]);
        }
#pragma unroll
        for (int t = 0; t < R2; t++) {
          const bool okt = nbv[t] >= 0;
          den += okt ? vd[t] : 0.f;
#pragma unroll
          for (int pp = 0; pp < P; pp++) {
            acc[pp][0] += okt ? v[t][pp].x : 0.f; acc[pp][1] += okt ? v[t][pp].y : 0.f;
            acc[pp][2] += okt ? v[t][pp].z : 0.f; acc[pp][3] += okt ? v[t][pp].w : 0.f;
          }
        }
#pragma unroll
        for (int dd = 0; dd < R; dd++) {
          const bool put = doit && dd == d;
#pragma unroll
          for (int pp = 0; pp < P; pp++)
#pragma unroll
            for (int e = 0; e < 4; e++) col[dd][pp][e] = put ? acc[pp][e] : col[dd][pp][e];
          cden[dd] = put ? den : cden[dd];
        }
        dcur++;
      }
      if (on) { have = true; px = bc.x; py = bc.y; pw = bc.w; pz = bc.z; }
      float A[P][4], den = cden[0];
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int e = 0; e < 4; e++) A[pp][e] = col[0][pp][e];
#pragma unroll
      for (int d = 1; d < R; d++) {
        den += cden[d];
#pragma unroll
        for (int pp = 0; pp < P; pp++)
#pragma unroll
          for (int e = 0; e < 4; e++) A[pp][e] += col[d][pp][e];
      }
      const float rden = on ? den : 1.0f;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int e = 0; e < 4; e++) A[pp][e] = A[pp][e] / rden;          // utils.py:80

      // ---- voxels of the block, STEP at a time
      const int p_last = (en > st) ? en - 1 : st;
      auto ld_rec = [&](int q) { return vox_sorted[(q <= p_last) ? q : p_last]; };
      int4 rcA = ld_rec(st), rcB = ld_rec(st + STEP - 1);
      for (int p = st; __any(p < en); p += STEP) {
        const bool vok = p < en;
        const int4 rnA = ld_rec(p + STEP), rnB = ld_rec(p + 2 * STEP - 1);
        const bool hasB = PAIR && (p + 1 < en);
        const int4 own = (hi && hasB) ? rcB : rcA;
        float x = (float)own.x, y = (float)own.y, z = (float)own.z;
        if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
        float4 f4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (OP == LINK_OP_COSX) f4 = *reinterpret_cast<const float4 *>(&fin[(int64_t)rcA.w * c + cofs]);
        const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
        float nvA[4], nvB[4];
        float sA = 0.f, sB = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
          float sn, cs;
          sincos_fast(th, sn, cs);
          float snA_ = sn, csA_ = cs, snB_ = sn, csB_ = cs;
          if (PAIR) {
            const float so = partner<LPR>(sn), co = partner<LPR>(cs);
            const bool swapped = hi && hasB;
            snA_ = swapped ? so : sn; csA_ = swapped ? co : cs;
            snB_ = hi ? sn : so;      csB_ = hi ? cs : co;
          }
          float va, vb;
          if (OP == LINK_OP_SIN) {                                            // linkunet.py:148
            va = __fsub_rn(__fmul_rn(A[0][e], csA_), __fmul_rn(A[1][e], snA_));
            vb = __fsub_rn(__fmul_rn(A[0][e], csB_), __fmul_rn(A[1][e], snB_));
          } else {                                                            // :162
            va = __fadd_rn(__fmul_rn(A[0][e], csA_), __fmul_rn(A[1][e], snA_));
            vb = __fadd_rn(__fmul_rn(A[0][e], csB_), __fmul_rn(A[1][e], snB_));
          }
          if (OP == LINK_OP_COSX) va = __fadd_rn(va, __fsub_rn(A[P - 1][e], link_mul_rn(fv[e], th)));   // :176
          nvA[e] = act ? va : 0.f; nvB[e] = act ? vb : 0.f;
          sA += nvA[e]; sB += nvB[e];
        }
        sA = grp_sum<LPR>(sA);
        if (PAIR) sB = grp_sum<LPR>(sB);
        const float meanA = sA * inv_c, meanB = sB * inv_c;
        float qA = 0.f, qB = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          float dA = act ? nvA[e] - meanA : 0.f, dB = act ? nvB[e] - meanB : 0.f;
          qA += dA * dA; qB += dB * dB;
        }
        qA = grp_sum<LPR>(qA);
        if (PAIR) qB = grp_sum<LPR>(qB);
        const float rsA = 1.0f / sqrtf(qA * inv_c + eps), rsB = 1.0f / sqrtf(qB * inv_c + eps);
        if (act && vok) {
          float4 o;
          o.x = (nvA[0] - meanA) * rsA * gw[0] + gb[0];
          o.y = (nvA[1] - meanA) * rsA * gw[1] + gb[1];
          o.z = (nvA[2] - meanA) * rsA * gw[2] + gb[2];
          o.w = (nvA[3] - meanA) * rsA * gw[3] + gb[3];
          *reinterpret_cast<float4 *>(&out[(int64_t)rcA.w * c + ch0]) = o;
        }
        if (PAIR && act && hasB) {
          float4 o;
          o.x = (nvB[0] - meanB) * rsB * gw[0] + gb[0];
          o.y = (nvB[1] - meanB) * rsB * gw[1] + gb[1];
          o.z = (nvB[2] - meanB) * rsB * gw[2] + gb[2];
          o.w = (nvB[3] - meanB) * rsB * gw[3] + gb[3];
          *reinterpret_cast<float4 *>(&out[(int64_t)rcB.w * c + ch0]) = o;
        }
        rcA = rnA; rcB = rnB;
      }
    }
  }
}

// Regime switch of the block gather: the dense-grid kernel walks CELLS (tiles of the block grid), which pays
// off when most cells are occupied (cfg2: 85 %) and is hopeless when almost none are (LiDAR block grids:
// ~1 %).  Both kernels read the same device-side header, so the choice needs no host sync.
constexpr int DENSE_RATIO = 3;          // dense regime: V <= DENSE_RATIO * M
__device__ __forceinline__ bool dense_regime(const link_grid_t &g, int m) {
  const long long v = (long long)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  return v <= (long long)DENSE_RATIO * m;
}

// ---------------------------------------------------------------------------------------------
// Dense-grid block gather (r = 3).  A wave owns a tile of G = 64/LPR adjacent y-columns x TZ cells along z
// at fixed (x, batch); its G groups march z in LOCKSTEP.  Per z-step group q loads the three x-neighbour
// rows of ITS column (the two edge groups also those of the halo column next to the tile), adds them (x-sum),
// swaps x-sums with the neighbouring groups through the lane crossbar (y-sum), and keeps the last three
// yx-planes in a register ring whose slot is the (compile-time) step number -- no selects, no per-group
// height bookkeeping.  A tile of G x TZ cells fetches 3 (G+2)(TZ+2) rows: 6.75 per cell at G = TZ = 4
// against ~17 per block for the column-walking kernel, and the neighbour ids come from ONE round trip
// (cell arithmetic + cell_blk) instead of two.  Same summation order on every run: deterministic.
// ---------------------------------------------------------------------------------------------
// Optional epilogue of the block gather: the finished row of block `b` goes straight to the rows of the block's voxels
// (perm[blk_start[b] .. blk_start[b + 1]) = their ids) instead of -- or besides -- the [M, P*C] table: what aux_to_voxel
// needs (utils.py:84, `new_feat[idx]`), without the table's round trip through memory and the row-gather launch.
struct bg_scatter_t {
  const int32_t *blk_start;
  const int32_t *perm;
  float *out;                                          // nullptr: no scatter
};
template <int P>
__device__ __forceinline__ void bg_scatter_row(const bg_scatter_t &sc, uint32_t b, int rs, int c, int ch0, const float4 (&row)[P]) {
  const int p0 = sc.blk_start[b], p1 = sc.blk_start[b + 1];
  for (int p = p0; p < p1; p++) {
    float *dst = sc.out + (int64_t)sc.perm[p] * rs + ch0;
#pragma unroll
    for (int pp = 0; pp < P; pp++) *reinterpret_cast<float4 *>(dst + pp * c) = row[pp];
  }
}

template <int LPR, int P, int TZ>
__device__ __forceinline__ bool block_gather_dense_body(const float *__restrict__ S,
                                                            const int32_t *__restrict__ cell_blk, link_grid_t g,
                                                            const int32_t *__restrict__ hdr, int c, int64_t m_cap,
                                                            float *__restrict__ A_tab, bool wt, int flags,
                                                            float *__restrict__ den_out, int tiles_y, int tiles_z,
                                                            const bg_scatter_t &sc) {
  constexpr int G = 64 / LPR;
  constexpr int NY = G + 2, NZ = TZ + 2, NCELL = 3 * NY * NZ;
  static_assert(G >= 2, "needs at least two columns per wave");
  __shared__ uint32_t s_id[4][NCELL];
  const int m_now = hdr[LINK_HDR_M];                // regime check below; its load overlaps the id loads
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (LPR - 1), q = lane / LPR;
  const int ch0 = 4 * li;
  const bool act = ch0 < c;
  const int cofs = act ? ch0 : 0;
  // wave -> tile; x is the slowest tile coordinate and the workgroup range is cut into 8 XCD slabs
  const long long tiles = (long long)g.dim[0] * g.dim[3] * tiles_y * tiles_z;
  const long long wtiles = (tiles + 3) / 4;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const long long lo = (wtiles * xcd) >> 3, hi = (wtiles * (xcd + 1)) >> 3;
  const long long wgt = lo + slot;
  if (wgt >= hi) return dense_regime(g, m_now);
  const long long tile = wgt * 4 + wave;
  if (tile >= tiles) return dense_regime(g, m_now);
  const int zs = (int)(tile % tiles_z);
  long long t2 = tile / tiles_z;
  const int yq = (int)(t2 % tiles_y);
  t2 /= tiles_y;
  const int bb = g.lo[3] + (int)(t2 % g.dim[3]);
  const int bx = g.lo[0] + (int)(t2 / g.dim[3]);
  const int by0 = g.lo[1] + yq * G, bz0 = g.lo[2] + zs * TZ;
  const int rs = P * c;
  const uint32_t row_bytes = (uint32_t)rs * 4u;
  const uint32_t zero_row = (uint32_t)m_cap;
  uint32_t *ids = s_id[wave];
  bool any_out = false;
  for (int e = lane; e < NCELL; e += 64) {          // ids of the tile's input cells: one round trip
    const int ix = e / (NY * NZ), rem = e - ix * (NY * NZ);
    const int iy = rem / NZ, iz = rem - iy * NZ;
    const int32_t cell = cell_of(g, bx + ix - 1, by0 + iy - 1, bz0 + iz - 1, bb);
    const int32_t id = (cell >= 0) ? cell_blk[cell] - 1 : -1;
    ids[e] = (id >= 0) ? (uint32_t)id : zero_row;
    any_out |= (id >= 0) && ix == 1 && iy >= 1 && iy <= G && iz >= 1 && iz <= TZ;
  }
  if (!dense_regime(g, m_now)) return false;        // sparse block grid: the column-walking form takes over
  if (!__any(any_out)) return true;                 // no block in this tile
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const char *__restrict__ Sb = reinterpret_cast<const char *>(S) + cofs * 4;
  const char *__restrict__ Cb = reinterpret_cast<const char *>(S + (m_cap + 1) * rs);
  const int own = q + 1;                            // this group's column inside the (G+2)-wide input window
  const int halo = (q == 0) ? 0 : ((q == G - 1) ? G + 1 : -1);
  float ring[3][P][4], rden[3];
#pragma unroll
  for (int t = 0; t < NZ; t++) {                    // fully unrolled: the ring slot t % 3 is a constant
    uint32_t r[3], h[3];
#pragma unroll
    for (int ix = 0; ix < 3; ix++) {
      r[ix] = ids[(ix * NY + own) * NZ + t];
      h[ix] = (halo >= 0) ? ids[(ix * NY + (halo >= 0 ? halo : 0)) * NZ + t] : zero_row;
    }
    float4 v[3][P], w[3][P];
    float vd[3], wd[3];
#pragma unroll
    for (int ix = 0; ix < 3; ix++) {                // 3 own rows, all loads back to back
      vd[ix] = *reinterpret_cast<const float *>(Cb + r[ix] * 4u);
#pragma unroll
      for (int pp = 0; pp < P; pp++)
        v[ix][pp] = *reinterpret_cast<const float4 *>(Sb + r[ix] * row_bytes + (uint32_t)(pp * c) * 4u);
    }
#pragma unroll
    for (int ix = 0; ix < 3; ix++) {                // halo rows: only the two edge groups issue them (exec-masked)
      wd[ix] = 0.f;
#pragma unroll
      for (int pp = 0; pp < P; pp++) w[ix][pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (halo >= 0) {
#pragma unroll
      for (int ix = 0; ix < 3; ix++) {
        wd[ix] = *reinterpret_cast<const float *>(Cb + h[ix] * 4u);
#pragma unroll
        for (int pp = 0; pp < P; pp++)
          w[ix][pp] = *reinterpret_cast<const float4 *>(Sb + h[ix] * row_bytes + (uint32_t)(pp * c) * 4u);
      }
    }
    float cx[P][4], hx[P][4];
    const float cd = (vd[0] + vd[1]) + vd[2], hd = (wd[0] + wd[1]) + wd[2];
#pragma unroll
    for (int pp = 0; pp < P; pp++) {
      cx[pp][0] = (v[0][pp].x + v[1][pp].x) + v[2][pp].x; cx[pp][1] = (v[0][pp].y + v[1][pp].y) + v[2][pp].y;
      cx[pp][2] = (v[0][pp].z + v[1][pp].z) + v[2][pp].z; cx[pp][3] = (v[0][pp].w + v[1][pp].w) + v[2][pp].w;
      hx[pp][0] = (w[0][pp].x + w[1][pp].x) + w[2][pp].x; hx[pp][1] = (w[0][pp].y + w[1][pp].y) + w[2][pp].y;
      hx[pp][2] = (w[0][pp].z + w[1][pp].z) + w[2][pp].z; hx[pp][3] = (w[0][pp].w + w[1][pp].w) + w[2][pp].w;
    }
    // y-sum: x-sums of the left and right neighbour columns (the edge groups use their halo column)
    {
      const float ld = __shfl_up(cd, LPR, 64), rd = __shfl_down(cd, LPR, 64);
      rden[t % 3] = ((q == 0 ? hd : ld) + cd) + (q == G - 1 ? hd : rd);
    }
#pragma unroll
    for (int pp = 0; pp < P; pp++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float l = __shfl_up(cx[pp][e], LPR, 64), rr = __shfl_down(cx[pp][e], LPR, 64);
        ring[t % 3][pp][e] = ((q == 0 ? hx[pp][e] : l) + cx[pp][e]) + (q == G - 1 ? hx[pp][e] : rr);
      }
    if (t >= 2) {                                   // planes z-1, z, z+1 of output cell iz = t-1 are in the ring
      const uint32_t oid = ids[(1 * NY + own) * NZ + (t - 1)];
      if (oid != zero_row) {
        const float den = (rden[0] + rden[1]) + rden[2];
        const float k = (flags & 2) ? 1.0f : 1.0f / den;
        if (den_out && li == 0) den_out[oid] = den;
        if (act) {
          float4 row[P];
#pragma unroll
          for (int pp = 0; pp < P; pp++)
            row[pp] = make_float4(((ring[0][pp][0] + ring[1][pp][0]) + ring[2][pp][0]) * k,
                                  ((ring[0][pp][1] + ring[1][pp][1]) + ring[2][pp][1]) * k,
                                  ((ring[0][pp][2] + ring[1][pp][2]) + ring[2][pp][2]) * k,
                                  ((ring[0][pp][3] + ring[1][pp][3]) + ring[2][pp][3]) * k);
          if (A_tab) {
#pragma unroll
            for (int pp = 0; pp < P; pp++) store_out(A_tab, (int64_t)oid * rs + pp * c + ch0, row[pp], wt);
          }
          if (sc.out) bg_scatter_row<P>(sc, oid, rs, c, ch0, row);
        }
      }
    }
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Split form of the gather: (a) block level -- A[m] = (sum of the r^3 neighbour rows of S) / count,
// written once per block as a [M, P*C] table; (b) voxel level -- a loop-free streaming kernel, one
// group per voxel (pair), that reads its block's A row (consecutive voxels share it: L1/L2 hits),
// de-modulates, LayerNorms and stores.  (b) has no divergence and thousands of independent waves,
// which is what hides memory latency on this chip; (a) carries only uniform block-level work.
// ---------------------------------------------------------------------------------------------
template <int LPR, int P, int R, bool DENSE>
__global__ void __launch_bounds__(256) k_block_gather_g(const float *__restrict__ S,
                                                        const int4 *__restrict__ blk_coords,
                                                        const int32_t *__restrict__ cell_blk, link_grid_t g,
                                                        const int32_t *__restrict__ hdr, int c,
                                                        int64_t m_cap, float *__restrict__ A_tab, bool wt,
                                                        int flags, float *__restrict__ den_out, int sparse_wgs,
                                                        int tiles_y, int tiles_z, bg_scatter_t sc) {
  // flags bit2: the launch is wide enough for the dense-grid form, which takes the frame when the block grid
  // is at least 1/DENSE_RATIO occupied (device-side decision: the host never learns M)
  if constexpr (DENSE && R == 3 && LPR <= 32) {
    if ((flags & 4) &&
        block_gather_dense_body<LPR, P, 4>(S, cell_blk, g, hdr, c, m_cap, A_tab, wt, flags, den_out, tiles_y, tiles_z, sc))
      return;
  }
  if ((int)blockIdx.x >= sparse_wgs) return;        // column-walking form: the first sparse_wgs workgroups
  // flags bit0: TRANSPOSED neighbourhood (offsets negated: the blocks whose region contains this one --
  // what the backward pass gathers; identical for odd r); bit1: plain sum, no division by the count.
  constexpr int G = 64 / LPR;
  constexpr int R2 = R * R, R3 = R2 * R;
  constexpr int ZLO_F = -((R + 1) / 2) + 1;
  const bool tr = (flags & 1) != 0;
  const int ZLO = tr ? -(ZLO_F + R - 1) : ZLO_F;
  const int sgn = tr ? -1 : 1;
  constexpr int SUB = 8;
  __shared__ uint32_t s_nb[4 * G][SUB * R3];       // BYTE offsets of the neighbour rows (zero row if absent)
  __shared__ int4 s_bc[4 * G][SUB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (LPR - 1), grp = wave * G + lane / LPR;
  const int ch0 = 4 * li;
  const bool act = ch0 < c;
  const int cofs = act ? ch0 : 0;
  int b0, b1;
  group_chunk<LPR>(hdr[LINK_HDR_M], b0, b1, sparse_wgs);
  const bool live = b0 < b1;
  const int rs = P * c;
  const uint32_t row_bytes = (uint32_t)rs * 4u;
  const char *__restrict__ Sb = reinterpret_cast<const char *>(S) + cofs * 4;        // lane's column
  const char *__restrict__ Cb = reinterpret_cast<const char *>(S + (m_cap + 1) * rs); // counts
  const uint32_t zero_row = (uint32_t)m_cap;
  float col[R][P][4], cden[R];
  int ph[R];                                       // height held by ring slot (slot = height mod R)
#pragma unroll
  for (int d = 0; d < R; d++) {
    cden[d] = 0.f; ph[d] = INT_MIN;
#pragma unroll
    for (int pp = 0; pp < P; pp++)
#pragma unroll
      for (int e = 0; e < 4; e++) col[d][pp][e] = 0.f;
  }
  int px = INT_MIN, py = 0, pw = 0;
  uint32_t *my_nb = s_nb[grp];
  int4 *my_bc = s_bc[grp];
  const int nblk = live ? b1 - b0 : 0;
  int wave_pass = (nblk + SUB - 1) / SUB;
#pragma unroll
  for (int o = 32; o >= LPR; o >>= 1) wave_pass = max(wave_pass, __shfl_xor(wave_pass, o, 64));

  for (int ps = 0; ps < wave_pass; ps++) {
    const int bb = b0 + ps * SUB;
    int nbk = live ? (b1 - bb) : 0;
    nbk = nbk < 0 ? 0 : (nbk > SUB ? SUB : nbk);
    for (int e = li; e < nbk; e += LPR) my_bc[e] = blk_coords[bb + e];
    for (int e = li; e < nbk * R3; e += LPR) {
      const int j = e / R3, k = e - j * R3;
      const int d = k / R2, t = k - d * R2;
      int ox, oy;
      plane_offset<R>(t, ox, oy);
      const int4 bc = blk_coords[bb + j];
      const int32_t cell = cell_of(g, bc.x + sgn * ox, bc.y + sgn * oy, bc.z + ZLO + d, bc.w);
      const int32_t id = (cell >= 0) ? cell_blk[cell] - 1 : -1;
      my_nb[e] = (id >= 0) ? (uint32_t)id : zero_row;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    for (int j = 0; j < SUB; j++) {
      const bool on = j < nbk;
      if (!__any(on)) break;
      const int jj = on ? j : 0;
      const int4 bc = on ? my_bc[jj] : make_int4(0, 0, 0, 0);
      const bool same_col = on && bc.x == px && bc.y == py && bc.w == pw;
      // plane d of this block is height h = z + ZLO + d and lives in ring slot (h mod R)
      int need_mask = 0;
#pragma unroll
      for (int d = 0; d < R; d++) {
        const int h = bc.z + ZLO + d;
        const int slot = ((h % R) + R) % R;
        bool valid = false;
#pragma unroll
        for (int sl = 0; sl < R; sl++) valid |= (sl == slot) && same_col && (ph[sl] == h);
        if (on && !valid) need_mask |= 1 << d;
      }
      while (__any(need_mask != 0)) {              // ONE plane body per iteration, per-group plane
        const bool doit = need_mask != 0;
        const int d = doit ? (__ffs(need_mask) - 1) : 0;
        const int h = bc.z + ZLO + d;
        const int slot = ((h % R) + R) % R;
        float4 v[R2][P];
        float vd[R2];
#pragma unroll
        for (int t = 0; t < R2; t++) {             // all row loads of the plane back to back, 32-bit offsets
          const uint32_t rid = doit ? my_nb[(jj * R + d) * R2 + t] : zero_row;   // idle groups: zero row
          vd[t] = *reinterpret_cast<const float *>(Cb + rid * 4u);
#pragma unroll
          for (int pp = 0; pp < P; pp++)
            v[t][pp] = *reinterpret_cast<const float4 *>(Sb + rid * row_bytes + (uint32_t)(pp * c) * 4u);
        }
        float acc[P][4], den = vd[0];
#pragma unroll
        for (int pp = 0; pp < P; pp++) { acc[pp][0] = v[0][pp].x; acc[pp][1] = v[0][pp].y; acc[pp][2] = v[0][pp].z; acc[pp][3] = v[0][pp].w; }
#pragma unroll
        for (int t = 1; t < R2; t++) {             // absent neighbours read the zero row: no selects
          den += vd[t];
#pragma unroll
          for (int pp = 0; pp < P; pp++) {
            acc[pp][0] += v[t][pp].x; acc[pp][1] += v[t][pp].y; acc[pp][2] += v[t][pp].z; acc[pp][3] += v[t][pp].w;
          }
        }
#pragma unroll
        for (int sl = 0; sl < R; sl++) {
          const bool put = doit && sl == slot;
#pragma unroll
          for (int pp = 0; pp < P; pp++)
#pragma unroll
            for (int e = 0; e < 4; e++) col[sl][pp][e] = put ? acc[pp][e] : col[sl][pp][e];
          cden[sl] = put ? den : cden[sl];
          ph[sl] = put ? h : ph[sl];
        }
        need_mask &= need_mask - 1;
      }
      if (on) { px = bc.x; py = bc.y; pw = bc.w; }
      // all R slots now hold this block's planes: sum them in slot order (deterministic: the order
      // is a function of z mod R only)
      float den = cden[0];
      float Av[P][4];
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int e = 0; e < 4; e++) Av[pp][e] = col[0][pp][e];
#pragma unroll
      for (int sl = 1; sl < R; sl++) {
        den += cden[sl];
#pragma unroll
        for (int pp = 0; pp < P; pp++)
#pragma unroll
          for (int e = 0; e < 4; e++) Av[pp][e] += col[sl][pp][e];
      }
      const float rden = (flags & 2) ? 1.0f : 1.0f / den;   // den is an exact small integer; x*(1/d) vs x/d: <= 1 ulp
      if (den_out && on && li == 0) den_out[bb + jj] = den;
      if (on && act) {
        float4 row[P];
#pragma unroll
        for (int pp = 0; pp < P; pp++) row[pp] = make_float4(Av[pp][0] * rden, Av[pp][1] * rden, Av[pp][2] * rden, Av[pp][3] * rden);   // utils.py:80
        if (A_tab) {
#pragma unroll
          for (int pp = 0; pp < P; pp++) store_out(A_tab, (int64_t)(bb + jj) * rs + pp * c + ch0, row[pp], wt);
        }
        if (sc.out) bg_scatter_row<P>(sc, (uint32_t)(bb + jj), rs, c, ch0, row);
      }
    }
  }
}

template <int LPR, int OP, bool PAIR>
__global__ void __launch_bounds__(256) k_voxel_demod_ln_g(
    const float *__restrict__ A_tab, const float *__restrict__ fin, const int4 *__restrict__ vox_sorted,
    const int32_t *__restrict__ pos_blk, const float *__restrict__ w_pos, const float *__restrict__ alpha,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, const int32_t *__restrict__ hdr, int c,
    int cg, float coord_div, float eps, float *__restrict__ out, bool wt) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  constexpr int G = 64 / LPR;
  constexpr int STEP = PAIR ? 2 : 1;
  const int lane = threadIdx.x & 63;
  const int li = lane & (LPR - 1);
  const int ch0 = 4 * li;
  const bool act = ch0 < c;
  const bool hi = PAIR && (li >= LPR / 2);
  const int cofs = act ? ch0 : 0;
  const int n = hdr[LINK_HDR_NVALID];
  const int64_t gid = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + lane / LPR;
  const int64_t p64 = gid * STEP;
  const bool vok = p64 < n;
  const int p = vok ? (int)p64 : (n > 0 ? n - 1 : 0);
  const bool hasB = PAIR && (p64 + 1 < n);
  const int pB = hasB ? p + 1 : p;
  const int ra = P * c;
  // all loads of the step issued up front
  const int4 rcA = vox_sorted[p], rcB = vox_sorted[pB];
  const int bA = pos_blk[p], bB = pos_blk[pB];
  float4 aA[P], aB[P];
#pragma unroll
  for (int pp = 0; pp < P; pp++) {
    aA[pp] = *reinterpret_cast<const float4 *>(&A_tab[(int64_t)bA * ra + pp * c + cofs]);
    aB[pp] = *reinterpret_cast<const float4 *>(&A_tab[(int64_t)bB * ra + pp * c + cofs]);
  }
  float4 f4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (OP == LINK_OP_COSX) f4 = *reinterpret_cast<const float4 *>(&fin[(int64_t)rcA.w * c + cofs]);
  float w0[4], w1[4], w2[4], al[4], gw[4], gb[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int ch = act ? ch0 + e : 0;
    int tc = ch % cg;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
    gw[e] = ln_w ? ln_w[ch] : 1.0f; gb[e] = ln_w ? ln_b[ch] : 0.0f;
  }
  const int4 own = (hi && hasB) ? rcB : rcA;
  float x = (float)own.x, y = (float)own.y, z = (float)own.z;
  if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
  const float inv_c = 1.0f / (float)c;
  const float A0a[4] = {aA[0].x, aA[0].y, aA[0].z, aA[0].w}, A1a[4] = {aA[1].x, aA[1].y, aA[1].z, aA[1].w};
  const float A0b[4] = {aB[0].x, aB[0].y, aB[0].z, aB[0].w}, A1b[4] = {aB[1].x, aB[1].y, aB[1].z, aB[1].w};
  const float A2a[4] = {aA[P - 1].x, aA[P - 1].y, aA[P - 1].z, aA[P - 1].w};
  const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
  float nvA[4], nvB[4], sA = 0.f, sB = 0.f;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
    float sn, cs;
    sincos_fast(th, sn, cs);
    float snA_ = sn, csA_ = cs, snB_ = sn, csB_ = cs;
    if (PAIR) {
      const float so = partner<LPR>(sn), co = partner<LPR>(cs);
      const bool swapped = hi && hasB;
      snA_ = swapped ? so : sn; csA_ = swapped ? co : cs;
      snB_ = hi ? sn : so;      csB_ = hi ? cs : co;
    }
    float va, vb;
    if (OP == LINK_OP_SIN) {                                                 // linkunet.py:148
      va = __fsub_rn(__fmul_rn(A0a[e], csA_), __fmul_rn(A1a[e], snA_));
      vb = __fsub_rn(__fmul_rn(A0b[e], csB_), __fmul_rn(A1b[e], snB_));
    } else {                                                                 // :162
      va = __fadd_rn(__fmul_rn(A0a[e], csA_), __fmul_rn(A1a[e], snA_));
      vb = __fadd_rn(__fmul_rn(A0b[e], csB_), __fmul_rn(A1b[e], snB_));
    }
    if (OP == LINK_OP_COSX) va = __fadd_rn(va, __fsub_rn(A2a[e], link_mul_rn(fv[e], th)));          // :176
    nvA[e] = act ? va : 0.f; nvB[e] = act ? vb : 0.f;
    sA += nvA[e]; sB += nvB[e];
  }
  if (!ln_w) {                                   // training forward: the raw de-modulated rows (no LayerNorm)
    if (act && vok) store_out(out, (int64_t)rcA.w * c + ch0, make_float4(nvA[0], nvA[1], nvA[2], nvA[3]), wt);
    if (PAIR && act && hasB) store_out(out, (int64_t)rcB.w * c + ch0, make_float4(nvB[0], nvB[1], nvB[2], nvB[3]), wt);
    return;
  }
  sA = grp_sum<LPR>(sA);
  if (PAIR) sB = grp_sum<LPR>(sB);
  const float meanA = sA * inv_c, meanB = sB * inv_c;
  float qA = 0.f, qB = 0.f;
#pragma unroll
  for (int e = 0; e < 4; e++) {
    float dA = act ? nvA[e] - meanA : 0.f, dB = act ? nvB[e] - meanB : 0.f;
    qA += dA * dA; qB += dB * dB;
  }
  qA = grp_sum<LPR>(qA);
  if (PAIR) qB = grp_sum<LPR>(qB);
  const float rsA = 1.0f / sqrtf(qA * inv_c + eps), rsB = 1.0f / sqrtf(qB * inv_c + eps);
  if (act && vok) {
    float4 o;
    o.x = (nvA[0] - meanA) * rsA * gw[0] + gb[0];
    o.y = (nvA[1] - meanA) * rsA * gw[1] + gb[1];
    o.z = (nvA[2] - meanA) * rsA * gw[2] + gb[2];
    o.w = (nvA[3] - meanA) * rsA * gw[3] + gb[3];
    store_out(out, (int64_t)rcA.w * c + ch0, o, wt);
  }
  if (PAIR && act && hasB) {
    float4 o;
    o.x = (nvB[0] - meanB) * rsB * gw[0] + gb[0];
    o.y = (nvB[1] - meanB) * rsB * gw[1] + gb[1];
    o.z = (nvB[2] - meanB) * rsB * gw[2] + gb[2];
    o.w = (nvB[3] - meanB) * rsB * gw[3] + gb[3];
    store_out(out, (int64_t)rcB.w * c + ch0, o, wt);
  }
}

static inline int lanes_per_row(int c) {
  int need = (c + 3) / 4, l = 1;
  while (l < need) l <<= 1;
  return l;
}

template <int LPR>
static void launch_modsum_g(int op, hipStream_t st, const float *fin, const int4 *vox, const float *w_pos,
                            const float *alpha, const int32_t *blk_start, const int32_t *hdr, int c, int cg,
                            float div, float *S, int64_t m_cap, const float *row_den = nullptr, int flags = 0) {
  dim3 grid(g_modsum_wgs), block(256);
  const bool two_part = op == LINK_OP_COS || op == LINK_OP_SIN || op == LINK_OPI_SIN_BWD;
  const bool pair = !(flags & LINK_ELK_NO_PAIR) && LPR >= 2 && c == 2 * cg && c == 4 * LPR && two_part;
#define LINK_MS(OPP, PP)                                                                                         \
  hipLaunchKernelGGL((k_modulate_sum_g<LPR, OPP, PP>), grid, block, 0, st, fin, vox, w_pos, alpha, blk_start, hdr, \
                     c, cg, div, S, m_cap, g_coop_threshold, (g_wt & 2) != 0 && wt_ok(m_cap + 1, 3 * (int64_t)c), row_den)
  switch (op) {
    case LINK_OP_COS: if (pair) LINK_MS(LINK_OP_COS, true); else LINK_MS(LINK_OP_COS, false); break;
    case LINK_OP_SIN: if (pair) LINK_MS(LINK_OP_SIN, true); else LINK_MS(LINK_OP_SIN, false); break;
    case LINK_OPI_SIN_BWD: LINK_MS(LINK_OPI_SIN_BWD, false); break;
    case LINK_OPI_COSX_BWD: LINK_MS(LINK_OPI_COSX_BWD, false); break;
    default: LINK_MS(LINK_OP_COSX, false); break;
  }
#undef LINK_MS
}

template <int LPR, int OP>
static void launch_gdl_g_r(int r, hipStream_t st, int64_t m_cap, const float *S_, const float *fin, const int4 *vox,
                           const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                           const int32_t *blk_start, const int4 *blk_coords, const int32_t *cell_blk,
                           const link_grid_t &g, const int32_t *hdr, const link_elk_desc_t &d, float *out) {
  dim3 grid(g_gather_wgs), block(256);
  const bool pair = !(d.flags & LINK_ELK_NO_PAIR) && LPR >= 2 && d.c == 2 * d.cg && d.c == 4 * LPR && OP != LINK_OP_COSX;
#define LINK_GDLG(RR, PP)                                                                                     \
  hipLaunchKernelGGL((k_gather_demod_ln_g<LPR, OP, RR, PP>), grid, block, 0, st, S_, fin, vox, w_pos, alpha,  \
                     ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d.c, d.cg, d.coord_div, d.eps, out, m_cap)
  if (pair && OP != LINK_OP_COSX) {
    switch (r) {
      case 1: LINK_GDLG(1, (OP != LINK_OP_COSX)); break;
      case 2: LINK_GDLG(2, (OP != LINK_OP_COSX)); break;
      default: LINK_GDLG(3, (OP != LINK_OP_COSX)); break;
    }
  } else {
    switch (r) {
      case 1: LINK_GDLG(1, false); break;
      case 2: LINK_GDLG(2, false); break;
      default: LINK_GDLG(3, false); break;
    }
  }
#undef LINK_GDLG
}

template <int LPR>
static void launch_gdl_g(int op, int r, hipStream_t st, int64_t m_cap, const float *S_, const float *fin, const int4 *vox,
                         const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                         const int32_t *blk_start, const int4 *blk_coords, const int32_t *cell_blk,
                         const link_grid_t &g, const int32_t *hdr, const link_elk_desc_t &d, float *out) {
  if (op == LINK_OP_COS)
    launch_gdl_g_r<LPR, LINK_OP_COS>(r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d, out);
  else if (op == LINK_OP_SIN)
    launch_gdl_g_r<LPR, LINK_OP_SIN>(r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d, out);
  else
    launch_gdl_g_r<LPR, LINK_OP_COSX>(r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, d, out);
}

template <int LPR, int P>
static void launch_block_gather(int r, hipStream_t st, const float *S_, const int4 *blk_coords,
                                const int32_t *cell_blk, const link_grid_t &g, const int32_t *hdr, int c,
                                int64_t m_cap, float *A, int flags, float *den_out, bool allow_dense,
                                const bg_scatter_t &sc = bg_scatter_t{nullptr, nullptr, nullptr}) {
  const bool wt = (g_wt & 4) != 0 && wt_ok(m_cap + 1, 3 * (int64_t)c);
  unsigned wgs = (unsigned)g_bgather_wgs;
  int tiles_y = 0, tiles_z = 0;
  if (LPR <= 32 && r == 3 && allow_dense) {         // widen the launch for the dense-grid form (same kernel)
    constexpr int G = 64 / LPR, TZ = 4;
    tiles_y = (g.dim[1] + G - 1) / G;
    tiles_z = (g.dim[2] + TZ - 1) / TZ;
    const long long cells = (long long)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
    const long long tiles = (long long)g.dim[0] * g.dim[3] * tiles_y * tiles_z;
    const long long wtiles = (tiles + 3) / 4;
    // M <= m_cap: a grid with more than DENSE_RATIO * m_cap cells can never be in the dense regime, and the
    // column-walking-only instantiation (fewer registers, 4 waves/SIMD) is launched instead
    if (wtiles < (1LL << 22) && cells <= (long long)DENSE_RATIO * m_cap) {
      const unsigned dw = (unsigned)(((wtiles + 7) / 8 + 1) * 8);
      if (dw > wgs) wgs = dw;
      flags |= 4;
    }
  }
  dim3 grid(wgs), block(256);
#define LINK_BGK(RR, DD)                                                                                          \
  hipLaunchKernelGGL((k_block_gather_g<LPR, P, RR, DD>), grid, block, 0, st, S_, blk_coords, cell_blk, g, hdr, c, \
                     m_cap, A, wt, flags, den_out, g_bgather_wgs, tiles_y, tiles_z, sc)
  switch (r) {
    case 1: LINK_BGK(1, false); break;
    case 2: LINK_BGK(2, false); break;
    default: if (flags & 4) LINK_BGK(3, true); else LINK_BGK(3, false); break;
  }
#undef LINK_BGK
}

static int block_gather_impl(const float *S_, const int32_t *blk_coords, const int32_t *cell_blk,
                             const link_grid_t *grid, const int32_t *hdr, const link_elk_desc_t *desc,
                             int64_t m_cap, float *A, int flags, float *den_out, void *stream,
                             const bg_scatter_t &sc = bg_scatter_t{nullptr, nullptr, nullptr}) {
  if (check_desc(desc) != LINK_OK || !grid || m_cap < 0 || desc->r > 3 || (desc->c & 3) != 0) return LINK_ERR_ARG;
  if (m_cap == 0) return LINK_OK;
  if (!S_ || !blk_coords || !cell_blk || !hdr || (!A && !sc.out)) return LINK_ERR_ARG;
  if (sc.out && (!sc.blk_start || !sc.perm)) return LINK_ERR_ARG;
  if ((m_cap + 1) * (int64_t)(desc->c * 3 + 1) * 4 >= (1LL << 32)) return LINK_ERR_ARG;   // 32-bit row offsets
  const int4 *b4 = reinterpret_cast<const int4 *>(blk_coords);
  hipStream_t st = S(stream);
  const bool p3 = desc->op == LINK_OP_COSX;
  const bool dense_ok = !(desc->flags & LINK_ELK_NO_DENSE_GRID);
#define LINK_BG(L)                                                                                                   \
  if (p3) launch_block_gather<L, 3>(desc->r, st, S_, b4, cell_blk, *grid, hdr, desc->c, m_cap, A, flags, den_out, dense_ok, sc); \
  else launch_block_gather<L, 2>(desc->r, st, S_, b4, cell_blk, *grid, hdr, desc->c, m_cap, A, flags, den_out, dense_ok, sc)
  switch (lanes_per_row(desc->c)) {
    case 1: case 2: case 4: LINK_BG(4); break;
    case 8: LINK_BG(8); break;
    case 16: LINK_BG(16); break;
    case 32: LINK_BG(32); break;
    default: LINK_BG(64); break;
  }
#undef LINK_BG
  return check_launch("link_block_gather");
}

extern "C" int link_block_gather(const float *S_, const int32_t *blk_coords, const int32_t *cell_blk,
                                 const link_grid_t *grid, const int32_t *hdr, const link_elk_desc_t *desc,
                                 int64_t m_cap, float *A, void *stream) {
  return block_gather_impl(S_, blk_coords, cell_blk, grid, hdr, desc, m_cap, A, 0, nullptr, stream);
}

template <int LPR>
static void launch_voxel_demod(const link_elk_desc_t &d, int64_t n, hipStream_t st, const float *A,
                               const float *fin, const int4 *vox, const int32_t *pos_blk, const float *w_pos,
                               const float *alpha, const float *ln_w, const float *ln_b, const int32_t *hdr,
                               float *out) {
  constexpr int G = 64 / LPR;
  const bool pair = !(d.flags & LINK_ELK_NO_PAIR) && LPR >= 2 && d.c == 2 * d.cg && d.c == 4 * LPR && d.op != LINK_OP_COSX;
  const int64_t groups = pair ? (n + 1) / 2 : n;
  const int64_t wgs = (groups + 4 * G - 1) / (4 * G);
  dim3 grid((unsigned)wgs), block(256);
#define LINK_VD(OPP, PP)                                                                                  \
  hipLaunchKernelGGL((k_voxel_demod_ln_g<LPR, OPP, PP>), grid, block, 0, st, A, fin, vox, pos_blk, w_pos,  \
                     alpha, ln_w, ln_b, hdr, d.c, d.cg, d.coord_div, d.eps, out, (g_wt & 8) != 0 && wt_ok(n, d.c))
  if (d.op == LINK_OP_COS) { if (pair) LINK_VD(LINK_OP_COS, true); else LINK_VD(LINK_OP_COS, false); }
  else if (d.op == LINK_OP_SIN) { if (pair) LINK_VD(LINK_OP_SIN, true); else LINK_VD(LINK_OP_SIN, false); }
  else LINK_VD(LINK_OP_COSX, false);
#undef LINK_VD
}

static int voxel_demod_impl(const float *A, const float *fin, const int32_t *vox_sorted,
                            const int32_t *pos_blk, const float *w_pos, const float *alpha,
                            const float *ln_w, const float *ln_b, const int32_t *hdr,
                            const link_elk_desc_t *desc, int64_t n, float *out, void *stream);

extern "C" int link_voxel_demod_ln(const float *A, const float *fin, const int32_t *vox_sorted,
                                   const int32_t *pos_blk, const float *w_pos, const float *alpha,
                                   const float *ln_w, const float *ln_b, const int32_t *hdr,
                                   const link_elk_desc_t *desc, int64_t n, float *out, void *stream) {
  if (!ln_w || !ln_b) return LINK_ERR_ARG;
  return voxel_demod_impl(A, fin, vox_sorted, pos_blk, w_pos, alpha, ln_w, ln_b, hdr, desc, n, out, stream);
}

// ln_w == NULL: no LayerNorm (the training forward, whose LayerNorm is differentiated by the host)
static int voxel_demod_impl(const float *A, const float *fin, const int32_t *vox_sorted,
                            const int32_t *pos_blk, const float *w_pos, const float *alpha,
                            const float *ln_w, const float *ln_b, const int32_t *hdr,
                            const link_elk_desc_t *desc, int64_t n, float *out, void *stream) {
  if (check_desc(desc) != LINK_OK || n < 0 || (desc->c & 3) != 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!A || !vox_sorted || !pos_blk || !w_pos || !hdr || !out) return LINK_ERR_ARG;
  if (desc->op == LINK_OP_COSX && !fin) return LINK_ERR_ARG;
  const int4 *v4 = reinterpret_cast<const int4 *>(vox_sorted);
  hipStream_t st = S(stream);
  switch (lanes_per_row(desc->c)) {
    case 1: case 2: case 4: launch_voxel_demod<4>(*desc, n, st, A, fin, v4, pos_blk, w_pos, alpha, ln_w, ln_b, hdr, out); break;
    case 8: launch_voxel_demod<8>(*desc, n, st, A, fin, v4, pos_blk, w_pos, alpha, ln_w, ln_b, hdr, out); break;
    case 16: launch_voxel_demod<16>(*desc, n, st, A, fin, v4, pos_blk, w_pos, alpha, ln_w, ln_b, hdr, out); break;
    case 32: launch_voxel_demod<32>(*desc, n, st, A, fin, v4, pos_blk, w_pos, alpha, ln_w, ln_b, hdr, out); break;
    default: launch_voxel_demod<64>(*desc, n, st, A, fin, v4, pos_blk, w_pos, alpha, ln_w, ln_b, hdr, out); break;
  }
  return check_launch("link_voxel_demod_ln");
}

static bool modsum_group_path(const link_elk_desc_t *d, hipStream_t st, const float *fin, const int4 *vox,
                              const float *w_pos, const float *alpha, const int32_t *blk_start,
                              const int32_t *hdr, float *S_, int64_t m_cap, int op, const float *row_den) {
  if ((d->flags & LINK_ELK_LANE_CHANNEL) || (d->c & 3) != 0) return false;
  if (op < 0) op = d->op;
#define LINK_MSG(L) launch_modsum_g<L>(op, st, fin, vox, w_pos, alpha, blk_start, hdr, d->c, d->cg, d->coord_div, S_, m_cap, row_den, d->flags)
  switch (lanes_per_row(d->c)) {
    case 1: case 2: case 4: LINK_MSG(4); break;
    case 8: LINK_MSG(8); break;
    case 16: LINK_MSG(16); break;
    case 32: LINK_MSG(32); break;
    default: LINK_MSG(64); break;
  }
#undef LINK_MSG
  return true;
}

static bool gather_group_path(const link_elk_desc_t *d, hipStream_t st, const float *S_, const float *fin,
                              const int4 *vox, const float *w_pos, const float *alpha, const float *ln_w,
                              const float *ln_b, const int32_t *blk_start, const int4 *blk_coords,
                              const int32_t *cell_blk, const link_grid_t &g, const int32_t *hdr, float *out,
                              int64_t m_cap) {
  if ((d->flags & LINK_ELK_LANE_CHANNEL) || (d->c & 3) != 0 || d->r > 3) return false;
  switch (lanes_per_row(d->c)) {
    case 1: case 2: case 4: launch_gdl_g<4>(d->op, d->r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, *d, out); break;
    case 8: launch_gdl_g<8>(d->op, d->r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, *d, out); break;
    case 16: launch_gdl_g<16>(d->op, d->r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, *d, out); break;
    case 32: launch_gdl_g<32>(d->op, d->r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, *d, out); break;
    default: launch_gdl_g<64>(d->op, d->r, st, m_cap, S_, fin, vox, w_pos, alpha, ln_w, ln_b, blk_start, blk_coords, cell_blk, g, hdr, *d, out); break;
  }
  return true;
}

// ---------------------------------------------------------------------------------------------
// Training form of the middle of R_core:  new = demodulate(aggregate(modulate(fin, theta)), theta)
// (linkunet.py:151-176 between pre_mix and self.norm), forward and backward.  The two LayerNorms and
// the pre_mix Linear stay with the host framework's autograd (library GEMM + its LayerNorm).
//
// Backward, per part (v = A[block(i)], X = modulated features, g = grad(new)):
//   gA[m]   = (1/den[m]) * sum_{i in m} g_i * d(new)/d(v)          -> the forward's modulate+block-sum kernel
//             with the backward factors ([cos, sin] | [cos, -sin] | [cos, sin, 1]) and a row scale
//   gS[n]   = sum_{m : n in region(m)} gA[m]                       -> the forward's block gather, transposed
//             neighbourhood, no normalisation
//   g_fin_i = gS[block(i)] . d(X)/d(fin)  (+ the -theta*g term of cos_x), and the theta gradient folded
//             into per-workgroup partial sums of d/d(alpha) and d/d(pos_weight)  -> k_voxel_bwd_g
// ---------------------------------------------------------------------------------------------
template <int LPR, int OP>
__global__ void __launch_bounds__(256) k_voxel_bwd_g(
    const float *__restrict__ gS, const float *__restrict__ A_tab, const float *__restrict__ fin,
    const float *__restrict__ g_new, const int4 *__restrict__ vox_sorted, const int32_t *__restrict__ pos_blk,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, const int32_t *__restrict__ hdr, int c,
    int cg, float coord_div, float *__restrict__ g_fin, float *__restrict__ partials) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  constexpr int G = 64 / LPR;
  __shared__ float red[4][16][LPR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (LPR - 1);
  const int ch0 = 4 * li;
  const bool act = ch0 < c;
  const int cofs = act ? ch0 : 0;
  const int n = hdr[LINK_HDR_NVALID];
  const int ra = P * c;
  float w0[4], w1[4], w2[4], al[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int tc = act ? (ch0 + e) % cg : 0;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
  }
  float acc[4][4];                                 // [d alpha | d w.x | d w.y | d w.z][channel of the lane]
#pragma unroll
  for (int q = 0; q < 4; q++)
#pragma unroll
    for (int e = 0; e < 4; e++) acc[q][e] = 0.f;
  const int64_t ngroups = (int64_t)gridDim.x * 4 * G;
  for (int64_t p = ((int64_t)blockIdx.x * 4 + wave) * G + lane / LPR; p < n; p += ngroups) {
    const int4 rc = vox_sorted[p];
    const int b = pos_blk[p];
    float4 gx4[P], av4[P];
#pragma unroll
    for (int pp = 0; pp < P; pp++) {
      gx4[pp] = *reinterpret_cast<const float4 *>(&gS[(int64_t)b * ra + pp * c + cofs]);
      av4[pp] = *reinterpret_cast<const float4 *>(&A_tab[(int64_t)b * ra + pp * c + cofs]);
    }
    const float4 f4 = *reinterpret_cast<const float4 *>(&fin[(int64_t)rc.w * c + cofs]);
    const float4 g4 = *reinterpret_cast<const float4 *>(&g_new[(int64_t)rc.w * c + cofs]);
    float x = (float)rc.x, y = (float)rc.y, z = (float)rc.z;
    if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
    const float fv[4] = {f4.x, f4.y, f4.z, f4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
    const float gx0[4] = {gx4[0].x, gx4[0].y, gx4[0].z, gx4[0].w}, gx1[4] = {gx4[1].x, gx4[1].y, gx4[1].z, gx4[1].w};
    const float gx2[4] = {gx4[P - 1].x, gx4[P - 1].y, gx4[P - 1].z, gx4[P - 1].w};
    const float v0[4] = {av4[0].x, av4[0].y, av4[0].z, av4[0].w}, v1[4] = {av4[1].x, av4[1].y, av4[1].z, av4[1].w};
    float gf[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float t = fmaf(z, w2[e], fmaf(y, w1[e], x * w0[e]));
      const float th = t * al[e];
      float sn, cs;
      sincos_fast(th, sn, cs);
      float g_cs, g_sn, gth;
      if (OP == LINK_OP_SIN) {                     // new = v0 cos - v1 sin ; X = [f sin, f cos]
        gf[e] = gx0[e] * sn + gx1[e] * cs;
        g_cs = gv[e] * v0[e] + gx1[e] * fv[e];
        g_sn = gx0[e] * fv[e] - gv[e] * v1[e];
      } else {                                     // new = v0 cos + v1 sin (+ v2 - f theta) ; X = [f cos, f sin, (f theta)]
        gf[e] = gx0[e] * cs + gx1[e] * sn;
        g_cs = gv[e] * v0[e] + gx0[e] * fv[e];
        g_sn = gv[e] * v1[e] + gx1[e] * fv[e];
      }
      gth = cs * g_sn - sn * g_cs;
      if (OP == LINK_OP_COSX) {
        const float d = gx2[e] - gv[e];
        gf[e] = fmaf(d, th, gf[e]);
        gth = fmaf(d, fv[e], gth);
      }
      if (act) {
        acc[0][e] = fmaf(gth, t, acc[0][e]);
        const float ga = gth * al[e];
        acc[1][e] = fmaf(ga, x, acc[1][e]);
        acc[2][e] = fmaf(ga, y, acc[2][e]);
        acc[3][e] = fmaf(ga, z, acc[3][e]);
      }
    }
    if (act) *reinterpret_cast<float4 *>(&g_fin[(int64_t)rc.w * c + ch0]) = make_float4(gf[0], gf[1], gf[2], gf[3]);
  }
  // fixed reduction tree: groups of the wave, then the 4 waves through LDS -> one partial row per workgroup
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) acc[q][e] += __shfl_xor(acc[q][e], o, 64);
  if (lane < LPR)
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[wave][q * 4 + e][li] = acc[q][e];
  __syncthreads();
  if (wave == 0 && lane < LPR && act) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float4 o;
      o.x = (red[0][q * 4 + 0][li] + red[1][q * 4 + 0][li]) + (red[2][q * 4 + 0][li] + red[3][q * 4 + 0][li]);
      o.y = (red[0][q * 4 + 1][li] + red[1][q * 4 + 1][li]) + (red[2][q * 4 + 1][li] + red[3][q * 4 + 1][li]);
      o.z = (red[0][q * 4 + 2][li] + red[1][q * 4 + 2][li]) + (red[2][q * 4 + 2][li] + red[3][q * 4 + 2][li]);
      o.w = (red[0][q * 4 + 3][li] + red[1][q * 4 + 3][li]) + (red[2][q * 4 + 3][li] + red[3][q * 4 + 3][li]);
      *reinterpret_cast<float4 *>(&partials[((int64_t)blockIdx.x * 4 + q) * c + ch0]) = o;
    }
  }
}

template <int LPR>
static void launch_voxel_bwd(const link_elk_desc_t &d, int wgs, hipStream_t st, const float *gS, const float *A,
                             const float *fin, const float *g_new, const int4 *vox, const int32_t *pos_blk,
                             const float *w_pos, const float *alpha, const int32_t *hdr, float *g_fin,
                             float *partials) {
  dim3 grid(wgs), block(256);
#define LINK_VB(OPP)                                                                                          \
  hipLaunchKernelGGL((k_voxel_bwd_g<LPR, OPP>), grid, block, 0, st, gS, A, fin, g_new, vox, pos_blk, w_pos,    \
                     alpha, hdr, d.c, d.cg, d.coord_div, g_fin, partials)
  if (d.op == LINK_OP_COS) LINK_VB(LINK_OP_COS);
  else if (d.op == LINK_OP_SIN) LINK_VB(LINK_OP_SIN);
  else LINK_VB(LINK_OP_COSX);
#undef LINK_VB
}

// Backward of self.norm fused with the recomputation of its input: new_i is rebuilt from the saved A
// row of the voxel's block exactly as the forward's voxel kernel builds it (same operation order), its
// LayerNorm statistics are recomputed, and g_new = rstd * (gy*w - mean(gy*w) - xhat * mean(gy*w*xhat)).
// Per-workgroup partial sums of d/d(norm.weight) = sum gy*xhat and d/d(norm.bias) = sum gy.
template <int LPR, int OP>
__global__ void __launch_bounds__(256) k_out_ln_bwd_g(
    const float *__restrict__ g_out, const float *__restrict__ A_tab, const float *__restrict__ fin,
    const int4 *__restrict__ vox_sorted, const int32_t *__restrict__ pos_blk, const float *__restrict__ w_pos,
    const float *__restrict__ alpha, const float *__restrict__ ln_w, const int32_t *__restrict__ hdr, int c,
    int cg, float coord_div, float eps, float *__restrict__ g_new, float *__restrict__ partials) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  constexpr int G = 64 / LPR;
  __shared__ float red[4][8][LPR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (LPR - 1);
  const int ch0 = 4 * li;
  const bool act = ch0 < c;
  const int cofs = act ? ch0 : 0;
  const int n = hdr[LINK_HDR_NVALID];
  const int ra = P * c;
  const float inv_c = 1.0f / (float)c;
  float w0[4], w1[4], w2[4], al[4], gw[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    int ch = act ? ch0 + e : 0;
    int tc = ch % cg;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
    gw[e] = ln_w[ch];
  }
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t ngroups = (int64_t)gridDim.x * 4 * G;
  for (int64_t p = ((int64_t)blockIdx.x * 4 + wave) * G + lane / LPR; p < n; p += ngroups) {
    const int4 rc = vox_sorted[p];
    const int b = pos_blk[p];
    float4 av4[P];
#pragma unroll
    for (int pp = 0; pp < P; pp++) av4[pp] = *reinterpret_cast<const float4 *>(&A_tab[(int64_t)b * ra + pp * c + cofs]);
    float4 f4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (OP == LINK_OP_COSX) f4 = *reinterpret_cast<const float4 *>(&fin[(int64_t)rc.w * c + cofs]);
    const float4 g4 = *reinterpret_cast<const float4 *>(&g_out[(int64_t)rc.w * c + cofs]);
    float x = (float)rc.x, y = (float)rc.y, z = (float)rc.z;
    if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
    const float v0[4] = {av4[0].x, av4[0].y, av4[0].z, av4[0].w}, v1[4] = {av4[1].x, av4[1].y, av4[1].z, av4[1].w};
    const float v2[4] = {av4[P - 1].x, av4[P - 1].y, av4[P - 1].z, av4[P - 1].w};
    const float fv[4] = {f4.x, f4.y, f4.z, f4.w}, gy[4] = {g4.x, g4.y, g4.z, g4.w};
    float nv[4], sm = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
      float sn, cs;
      sincos_fast(th, sn, cs);
      float va;
      if (OP == LINK_OP_SIN) va = __fsub_rn(__fmul_rn(v0[e], cs), __fmul_rn(v1[e], sn));
      else va = __fadd_rn(__fmul_rn(v0[e], cs), __fmul_rn(v1[e], sn));
      if (OP == LINK_OP_COSX) va = __fadd_rn(va, __fsub_rn(v2[e], link_mul_rn(fv[e], th)));
      nv[e] = act ? va : 0.f;
      sm += nv[e];
    }
    const float mean = grp_sum<LPR>(sm) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) { const float d = act ? nv[e] - mean : 0.f; q += d * d; }
    const float rstd = 1.0f / sqrtf(grp_sum<LPR>(q) * inv_c + eps);
    float xh[4], gx[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      xh[e] = act ? (nv[e] - mean) * rstd : 0.f;
      gx[e] = act ? gy[e] * gw[e] : 0.f;
      s1 += gx[e];
      s2 = fmaf(gx[e], xh[e], s2);
      if (act) { aw[e] = fmaf(gy[e], xh[e], aw[e]); ab[e] += gy[e]; }
    }
    const float m1 = grp_sum<LPR>(s1) * inv_c, m2 = grp_sum<LPR>(s2) * inv_c;
    if (act) {
      float4 o;
      o.x = rstd * (gx[0] - m1 - xh[0] * m2);
      o.y = rstd * (gx[1] - m1 - xh[1] * m2);
      o.z = rstd * (gx[2] - m1 - xh[2] * m2);
      o.w = rstd * (gx[3] - m1 - xh[3] * m2);
      *reinterpret_cast<float4 *>(&g_new[(int64_t)rc.w * c + ch0]) = o;
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 4; e++) { aw[e] += __shfl_xor(aw[e], o, 64); ab[e] += __shfl_xor(ab[e], o, 64); }
  if (lane < LPR)
#pragma unroll
    for (int e = 0; e < 4; e++) { red[wave][e][li] = aw[e]; red[wave][4 + e][li] = ab[e]; }
  __syncthreads();
  if (wave == 0 && lane < LPR && act) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      float4 o;
      o.x = (red[0][q * 4 + 0][li] + red[1][q * 4 + 0][li]) + (red[2][q * 4 + 0][li] + red[3][q * 4 + 0][li]);
      o.y = (red[0][q * 4 + 1][li] + red[1][q * 4 + 1][li]) + (red[2][q * 4 + 1][li] + red[3][q * 4 + 1][li]);
      o.z = (red[0][q * 4 + 2][li] + red[1][q * 4 + 2][li]) + (red[2][q * 4 + 2][li] + red[3][q * 4 + 2][li]);
      o.w = (red[0][q * 4 + 3][li] + red[1][q * 4 + 3][li]) + (red[2][q * 4 + 3][li] + red[3][q * 4 + 3][li]);
      *reinterpret_cast<float4 *>(&partials[((int64_t)blockIdx.x * 2 + q) * c + ch0]) = o;
    }
  }
}

template <int LPR>
static void launch_out_ln_bwd(const link_elk_desc_t &d, int wgs, hipStream_t st, const float *g_out, const float *A,
                              const float *fin, const int4 *vox, const int32_t *pos_blk, const float *w_pos,
                              const float *alpha, const float *ln_w, const int32_t *hdr, float *g_new,
                              float *partials) {
  dim3 grid(wgs), block(256);
#define LINK_OB(OPP)                                                                                         \
  hipLaunchKernelGGL((k_out_ln_bwd_g<LPR, OPP>), grid, block, 0, st, g_out, A, fin, vox, pos_blk, w_pos,      \
                     alpha, ln_w, hdr, d.c, d.cg, d.coord_div, d.eps, g_new, partials)
  if (d.op == LINK_OP_COS) LINK_OB(LINK_OP_COS);
  else if (d.op == LINK_OP_SIN) LINK_OB(LINK_OP_SIN);
  else LINK_OB(LINK_OP_COSX);
#undef LINK_OB
}

// Backward of pre_mix = LayerNorm(F @ Wpre^T): the pre-LayerNorm activations are RECOMPUTED with the
// forward's MFMA schedule (so the statistics are bit-identical to the forward's), the LayerNorm backward
// is applied in the accumulator layout (lane = 4 channels x 4 tiles of one voxel), and
// g_F = g_pre @ Wpre runs as a second MFMA pass whose B operand IS that accumulator layout and whose A
// operand is a transposed copy of W in LDS.  g_pre is also stored (the weight gradient
// g_pre^T @ F is a plain GEMM left to the library), and per-workgroup partial sums of
// d/d(pre_mix.1.weight) = sum g_fin*xhat and d/d(pre_mix.1.bias) = sum g_fin are written.
template <int C>
__global__ void __launch_bounds__(256) k_premix_ln_bwd(const float *__restrict__ feats,
                                                       const float *__restrict__ w_pre,
                                                       const float *__restrict__ ln_w,
                                                       const float *__restrict__ g_fin, int64_t n, float eps,
                                                       float *__restrict__ g_pre, float *__restrict__ g_feats,
                                                       float *__restrict__ partials) {
  constexpr int T = C / 16;
  constexpr int LDW = C + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);          // W   [j][k]
  float *wt_lds = w_lds + C * LDW;                             // W^T [k][j]
  float *red = wt_lds + C * LDW;                               // [4 waves][2][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  for (int e = tid * 4; e < C * C; e += 256 * 4) {
    int r = e / C, col = e - r * C;
    const float4 w4 = *reinterpret_cast<const float4 *>(&w_pre[e]);
    *reinterpret_cast<float4 *>(&w_lds[r * LDW + col]) = w4;
    wt_lds[(col + 0) * LDW + r] = w4.x; wt_lds[(col + 1) * LDW + r] = w4.y;
    wt_lds[(col + 2) * LDW + r] = w4.z; wt_lds[(col + 3) * LDW + r] = w4.w;
  }
  __syncthreads();
  float lw[T][4];
#pragma unroll
  for (int tp = 0; tp < T; tp++) {
    const float4 l4 = *reinterpret_cast<const float4 *>(&ln_w[16 * tp + 4 * g]);
    lw[tp][0] = l4.x; lw[tp][1] = l4.y; lw[tp][2] = l4.z; lw[tp][3] = l4.w;
  }
  float pw[T][4], pb[T][4];
#pragma unroll
  for (int tp = 0; tp < T; tp++)
#pragma unroll
    for (int r = 0; r < 4; r++) pw[tp][r] = pb[tp][r] = 0.f;
  const int64_t tiles = (n + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t v = tile * 16 + li;
    const bool ok = v < n;
    const int64_t vl = ok ? v : n - 1;
    float4 f[T], gf4[T];
#pragma unroll
    for (int t = 0; t < T; t++) f[t] = *reinterpret_cast<const float4 *>(&feats[vl * C + 16 * t + 4 * g]);
#pragma unroll
    for (int t = 0; t < T; t++) gf4[t] = *reinterpret_cast<const float4 *>(&g_fin[vl * C + 16 * t + 4 * g]);
    floatx4 acc[T];
#pragma unroll
    for (int tp = 0; tp < T; tp++) acc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; t++) {
#pragma unroll
      for (int tp = 0; tp < T; tp++) {
        float4 a = *reinterpret_cast<const float4 *>(&w_lds[(16 * tp + li) * LDW + 16 * t + 4 * g]);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f[t].x, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f[t].y, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f[t].z, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f[t].w, acc[tp], 0, 0, 0);
      }
    }
    float s = 0.f;
#pragma unroll
    for (int tp = 0; tp < T; tp++) s += (acc[tp][0] + acc[tp][1]) + (acc[tp][2] + acc[tp][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int tp = 0; tp < T; tp++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float d = acc[tp][r] - mean;
        q += d * d;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int tp = 0; tp < T; tp++) {
      const float gv[4] = {gf4[tp].x, gf4[tp].y, gf4[tp].z, gf4[tp].w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const float xh = (acc[tp][r] - mean) * rstd;
        const float gx = gv[r] * lw[tp][r];
        s1 += gx;
        s2 = fmaf(gx, xh, s2);
        if (ok) { pw[tp][r] = fmaf(gv[r], xh, pw[tp][r]); pb[tp][r] += gv[r]; }
        acc[tp][r] = xh;                            // keep xhat; gx is recomputed below (saves 16 VGPRs)
      }
    }
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
#pragma unroll
    for (int tp = 0; tp < T; tp++) {
      const float gv[4] = {gf4[tp].x, gf4[tp].y, gf4[tp].z, gf4[tp].w};
#pragma unroll
      for (int r = 0; r < 4; r++) acc[tp][r] = rstd * (gv[r] * lw[tp][r] - m1 - acc[tp][r] * m2);   // g_pre
      if (ok)
        *reinterpret_cast<float4 *>(&g_pre[v * C + 16 * tp + 4 * g]) = make_float4(acc[tp][0], acc[tp][1], acc[tp][2], acc[tp][3]);
    }
    // g_F[v][k] = sum_j g_pre[v][j] W[j][k]:  D2[k][v] = sum_j W^T[k][j] g_pre[v][j]
    floatx4 acc2[T];
#pragma unroll
    for (int tp = 0; tp < T; tp++) acc2[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; t++) {
#pragma unroll
      for (int tp = 0; tp < T; tp++) {
        float4 a = *reinterpret_cast<const float4 *>(&wt_lds[(16 * tp + li) * LDW + 16 * t + 4 * g]);
        acc2[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, acc[t][0], acc2[tp], 0, 0, 0);
        acc2[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, acc[t][1], acc2[tp], 0, 0, 0);
        acc2[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, acc[t][2], acc2[tp], 0, 0, 0);
        acc2[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, acc[t][3], acc2[tp], 0, 0, 0);
      }
    }
    if (ok) {
#pragma unroll
      for (int tp = 0; tp < T; tp++)
        *reinterpret_cast<float4 *>(&g_feats[v * C + 16 * tp + 4 * g]) = make_float4(acc2[tp][0], acc2[tp][1], acc2[tp][2], acc2[tp][3]);
    }
  }
  // LayerNorm parameter gradients: sum over the 16 voxel lanes of each quarter-wave, then over waves
#pragma unroll
  for (int tp = 0; tp < T; tp++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      pw[tp][r] = grp_sum<16>(pw[tp][r]);
      pb[tp][r] = grp_sum<16>(pb[tp][r]);
    }
  if (li == 0) {
#pragma unroll
    for (int tp = 0; tp < T; tp++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        red[(wave * 2 + 0) * C + 16 * tp + 4 * g + r] = pw[tp][r];
        red[(wave * 2 + 1) * C + 16 * tp + 4 * g + r] = pb[tp][r];
      }
  }
  __syncthreads();
  for (int e = tid; e < 2 * C; e += 256) {
    const int qq = e / C, ch = e - qq * C;
    partials[((int64_t)blockIdx.x * 2 + qq) * C + ch] =
        (red[(0 * 2 + qq) * C + ch] + red[(1 * 2 + qq) * C + ch]) + (red[(2 * 2 + qq) * C + ch] + red[(3 * 2 + qq) * C + ch]);
  }
}

template <int C>
static int launch_premix_bwd(const float *feats, const float *w_pre, const float *ln_w, const float *g_fin,
                             int64_t n, float eps, float *g_pre, float *g_feats, float *partials, int wgs,
                             hipStream_t st) {
  size_t lds = ((size_t)2 * C * (C + 4) + 8 * C) * sizeof(float);
  if (lds > 64 * 1024) {
    // per device and cheap (a host-side table write): no process-wide once-flag, which a second GPU or a
    // device reset would never pass again
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_premix_ln_bwd<C>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  hipLaunchKernelGGL(k_premix_ln_bwd<C>, dim3((unsigned)wgs), dim3(256), lds, st, feats, w_pre, ln_w, g_fin, n, eps,
                     g_pre, g_feats, partials);
  return check_launch("link_premix_ln_backward");
}

extern "C" int32_t link_elk_mid_partial_rows(void) { return 1024; }

extern "C" int link_premix_ln_backward(const float *feats, const float *w_pre, const float *ln_w,
                                       const float *g_fin, int64_t n, int32_t c, float eps, float *g_pre,
                                       float *g_feats, float *partials, void *stream) {
  if (n < 0 || c <= 0 || (c & 15) != 0 || c > 128) return LINK_ERR_ARG;     // MFMA path only; callers fall back
  if (n == 0) return LINK_OK;
  if (!feats || !w_pre || !ln_w || !g_fin || !g_pre || !g_feats || !partials) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  const int wgs = link_elk_mid_partial_rows();
  switch (c) {
    case 16: return launch_premix_bwd<16>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
    case 32: return launch_premix_bwd<32>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
    case 48: return launch_premix_bwd<48>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
    case 64: return launch_premix_bwd<64>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
    case 80: return launch_premix_bwd<80>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
    case 96: return launch_premix_bwd<96>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
    case 112: return launch_premix_bwd<112>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
    default: return launch_premix_bwd<128>(feats, w_pre, ln_w, g_fin, n, eps, g_pre, g_feats, partials, wgs, st);
  }
}

extern "C" int link_elk_out_ln_backward(const float *g_out, const float *A, const float *fin,
                                        const int32_t *vox_sorted, const int32_t *pos_blk, const float *w_pos,
                                        const float *alpha, const float *ln_w, const int32_t *hdr,
                                        const link_elk_desc_t *desc, int64_t n, float *g_new, float *partials,
                                        void *stream) {
  if (check_desc(desc) != LINK_OK || n < 0 || (desc->c & 3) != 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!g_out || !A || !vox_sorted || !pos_blk || !w_pos || !ln_w || !hdr || !g_new || !partials) return LINK_ERR_ARG;
  if (desc->op == LINK_OP_COSX && !fin) return LINK_ERR_ARG;
  const int4 *v4 = reinterpret_cast<const int4 *>(vox_sorted);
  hipStream_t st = S(stream);
  const int wgs = link_elk_mid_partial_rows();
  switch (lanes_per_row(desc->c)) {
    case 1: case 2: case 4: launch_out_ln_bwd<4>(*desc, wgs, st, g_out, A, fin, v4, pos_blk, w_pos, alpha, ln_w, hdr, g_new, partials); break;
    case 8: launch_out_ln_bwd<8>(*desc, wgs, st, g_out, A, fin, v4, pos_blk, w_pos, alpha, ln_w, hdr, g_new, partials); break;
    case 16: launch_out_ln_bwd<16>(*desc, wgs, st, g_out, A, fin, v4, pos_blk, w_pos, alpha, ln_w, hdr, g_new, partials); break;
    case 32: launch_out_ln_bwd<32>(*desc, wgs, st, g_out, A, fin, v4, pos_blk, w_pos, alpha, ln_w, hdr, g_new, partials); break;
    default: launch_out_ln_bwd<64>(*desc, wgs, st, g_out, A, fin, v4, pos_blk, w_pos, alpha, ln_w, hdr, g_new, partials); break;
  }
  return check_launch("link_elk_out_ln_backward");
}

// ---------------------------------------------------------------------------------------------
// The block's tail for training (linkunet.py:183 / ts_elk.py:228): y = relu(addend + LayerNorm(x)),
// forward and backward, one 16-byte-per-lane group per row (persistent grid).  Inference fuses this
// tail into the convolution kernel instead (conv.hip, row N2).
// ---------------------------------------------------------------------------------------------
template <int LPR>
__global__ void __launch_bounds__(256) k_ln_add_relu_fwd_g(const float *__restrict__ x,
                                                           const float *__restrict__ addend,
                                                           const float *__restrict__ ln_w,
                                                           const float *__restrict__ ln_b, int64_t n, int c,
                                                           float eps, float *__restrict__ y) {
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (LPR - 1), ch0 = 4 * li;
  const bool act = ch0 < c;
  const int cofs = act ? ch0 : 0;
  const float inv_c = 1.0f / (float)c;
  const float4 w4 = *reinterpret_cast<const float4 *>(&ln_w[cofs]), b4 = *reinterpret_cast<const float4 *>(&ln_b[cofs]);
  const int64_t ngroups = (int64_t)gridDim.x * 4 * G;
  for (int64_t i = ((int64_t)blockIdx.x * 4 + wave) * G + lane / LPR; i < n; i += ngroups) {
    float4 v = *reinterpret_cast<const float4 *>(&x[i * c + cofs]);
    const float4 a = *reinterpret_cast<const float4 *>(&addend[i * c + cofs]);
    if (!act) v = make_float4(0.f, 0.f, 0.f, 0.f);
    const float mean = grp_sum<LPR>((v.x + v.y) + (v.z + v.w)) * inv_c;
    const float dx = act ? v.x - mean : 0.f, dy = act ? v.y - mean : 0.f, dz = act ? v.z - mean : 0.f, dw = act ? v.w - mean : 0.f;
    const float rstd = 1.0f / sqrtf(grp_sum<LPR>((dx * dx + dy * dy) + (dz * dz + dw * dw)) * inv_c + eps);
    if (act) {
      float4 o;
      o.x = fmaxf(a.x + (dx * rstd * w4.x + b4.x), 0.f);
      o.y = fmaxf(a.y + (dy * rstd * w4.y + b4.y), 0.f);
      o.z = fmaxf(a.z + (dz * rstd * w4.z + b4.z), 0.f);
      o.w = fmaxf(a.w + (dw * rstd * w4.w + b4.w), 0.f);
      *reinterpret_cast<float4 *>(&y[i * c + ch0]) = o;
    }
  }
}

template <int LPR>
__global__ void __launch_bounds__(256) k_ln_add_relu_bwd_g(const float *__restrict__ g_y,
                                                           const float *__restrict__ y,
                                                           const float *__restrict__ x,
                                                           const float *__restrict__ ln_w, int64_t n, int c,
                                                           float eps, float *__restrict__ g_addend,
                                                           float *__restrict__ g_x, float *__restrict__ partials) {
  constexpr int G = 64 / LPR;
  __shared__ float red[4][8][LPR];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & (LPR - 1), ch0 = 4 * li;
  const bool act = ch0 < c;
  const int cofs = act ? ch0 : 0;
  const float inv_c = 1.0f / (float)c;
  const float4 w4 = *reinterpret_cast<const float4 *>(&ln_w[cofs]);
  const float gw[4] = {w4.x, w4.y, w4.z, w4.w};
  float aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t ngroups = (int64_t)gridDim.x * 4 * G;
  for (int64_t i = ((int64_t)blockIdx.x * 4 + wave) * G + lane / LPR; i < n; i += ngroups) {
    const float4 v4 = *reinterpret_cast<const float4 *>(&x[i * c + cofs]);
    const float4 y4 = *reinterpret_cast<const float4 *>(&y[i * c + cofs]);
    const float4 g4 = *reinterpret_cast<const float4 *>(&g_y[i * c + cofs]);
    const float xv[4] = {v4.x, v4.y, v4.z, v4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w}, gv[4] = {g4.x, g4.y, g4.z, g4.w};
    float sm = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) sm += act ? xv[e] : 0.f;
    const float mean = grp_sum<LPR>(sm) * inv_c;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) { const float d = act ? xv[e] - mean : 0.f; q += d * d; }
    const float rstd = 1.0f / sqrtf(grp_sum<LPR>(q) * inv_c + eps);
    float g[4], xh[4], gx[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      g[e] = (act && yv[e] > 0.f) ? gv[e] : 0.f;              // ReLU mask from the saved output
      xh[e] = act ? (xv[e] - mean) * rstd : 0.f;
      gx[e] = g[e] * gw[e];
      s1 += gx[e];
      s2 = fmaf(gx[e], xh[e], s2);
      aw[e] = fmaf(g[e], xh[e], aw[e]);
      ab[e] += g[e];
    }
    const float m1 = grp_sum<LPR>(s1) * inv_c, m2 = grp_sum<LPR>(s2) * inv_c;
    if (act) {
      *reinterpret_cast<float4 *>(&g_addend[i * c + ch0]) = make_float4(g[0], g[1], g[2], g[3]);
      *reinterpret_cast<float4 *>(&g_x[i * c + ch0]) =
          make_float4(rstd * (gx[0] - m1 - xh[0] * m2), rstd * (gx[1] - m1 - xh[1] * m2),
                      rstd * (gx[2] - m1 - xh[2] * m2), rstd * (gx[3] - m1 - xh[3] * m2));
    }
  }
#pragma unroll
  for (int o = LPR; o < 64; o <<= 1)
#pragma unroll
    for (int e = 0; e < 4; e++) { aw[e] += __shfl_xor(aw[e], o, 64); ab[e] += __shfl_xor(ab[e], o, 64); }
  if (lane < LPR)
#pragma unroll
    for (int e = 0; e < 4; e++) { red[wave][e][li] = aw[e]; red[wave][4 + e][li] = ab[e]; }
  __syncthreads();
  if (wave == 0 && lane < LPR && act) {
#pragma unroll
    for (int q = 0; q < 2; q++) {
      float4 o;
      o.x = (red[0][q * 4 + 0][li] + red[1][q * 4 + 0][li]) + (red[2][q * 4 + 0][li] + red[3][q * 4 + 0][li]);
      o.y = (red[0][q * 4 + 1][li] + red[1][q * 4 + 1][li]) + (red[2][q * 4 + 1][li] + red[3][q * 4 + 1][li]);
      o.z = (red[0][q * 4 + 2][li] + red[1][q * 4 + 2][li]) + (red[2][q * 4 + 2][li] + red[3][q * 4 + 2][li]);
      o.w = (red[0][q * 4 + 3][li] + red[1][q * 4 + 3][li]) + (red[2][q * 4 + 3][li] + red[3][q * 4 + 3][li]);
      *reinterpret_cast<float4 *>(&partials[((int64_t)blockIdx.x * 2 + q) * c + ch0]) = o;
    }
  }
}

extern "C" int32_t link_elk_mid_partial_rows(void);

extern "C" int link_ln_add_relu_forward(const float *x, const float *addend, const float *ln_w,
                                        const float *ln_b, int64_t n, int32_t c, float eps, float *y,
                                        void *stream) {
  if (n < 0 || c <= 0 || (c & 3) != 0 || c > 256) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!x || !addend || !ln_w || !ln_b || !y) return LINK_ERR_ARG;
  dim3 grid(1024), block(256);
  hipStream_t st = S(stream);
  switch (lanes_per_row(c)) {
    case 1: case 2: case 4: hipLaunchKernelGGL(k_ln_add_relu_fwd_g<4>, grid, block, 0, st, x, addend, ln_w, ln_b, n, (int)c, eps, y); break;
    case 8: hipLaunchKernelGGL(k_ln_add_relu_fwd_g<8>, grid, block, 0, st, x, addend, ln_w, ln_b, n, (int)c, eps, y); break;
    case 16: hipLaunchKernelGGL(k_ln_add_relu_fwd_g<16>, grid, block, 0, st, x, addend, ln_w, ln_b, n, (int)c, eps, y); break;
    case 32: hipLaunchKernelGGL(k_ln_add_relu_fwd_g<32>, grid, block, 0, st, x, addend, ln_w, ln_b, n, (int)c, eps, y); break;
    default: hipLaunchKernelGGL(k_ln_add_relu_fwd_g<64>, grid, block, 0, st, x, addend, ln_w, ln_b, n, (int)c, eps, y); break;
  }
  return check_launch("link_ln_add_relu_forward");
}

extern "C" int link_ln_add_relu_backward(const float *g_y, const float *y, const float *x, const float *ln_w,
                                         int64_t n, int32_t c, float eps, float *g_addend, float *g_x,
                                         float *partials, void *stream) {
  if (n < 0 || c <= 0 || (c & 3) != 0 || c > 256) return LINK_ERR_ARG;
  if (!partials) return LINK_ERR_ARG;
  if (n > 0 && (!g_y || !y || !x || !ln_w || !g_addend || !g_x)) return LINK_ERR_ARG;
  dim3 grid(link_elk_mid_partial_rows()), block(256);
  hipStream_t st = S(stream);
  switch (lanes_per_row(c)) {
    case 1: case 2: case 4: hipLaunchKernelGGL(k_ln_add_relu_bwd_g<4>, grid, block, 0, st, g_y, y, x, ln_w, n, (int)c, eps, g_addend, g_x, partials); break;
    case 8: hipLaunchKernelGGL(k_ln_add_relu_bwd_g<8>, grid, block, 0, st, g_y, y, x, ln_w, n, (int)c, eps, g_addend, g_x, partials); break;
    case 16: hipLaunchKernelGGL(k_ln_add_relu_bwd_g<16>, grid, block, 0, st, g_y, y, x, ln_w, n, (int)c, eps, g_addend, g_x, partials); break;
    case 32: hipLaunchKernelGGL(k_ln_add_relu_bwd_g<32>, grid, block, 0, st, g_y, y, x, ln_w, n, (int)c, eps, g_addend, g_x, partials); break;
    default: hipLaunchKernelGGL(k_ln_add_relu_bwd_g<64>, grid, block, 0, st, g_y, y, x, ln_w, n, (int)c, eps, g_addend, g_x, partials); break;
  }
  return check_launch("link_ln_add_relu_backward");
}

// Column sums of up to three per-workgroup partial arrays [rows, cols_k] in one launch, fixed order
// (row lanes ascending, then a fixed LDS tree): the deterministic tail of every parameter gradient.
__global__ void __launch_bounds__(256) k_sum_partials(const float *__restrict__ p0, int c0,
                                                      const float *__restrict__ p1, int c1,
                                                      const float *__restrict__ p2, int c2, int64_t rows,
                                                      float *__restrict__ out) {
  __shared__ float red[32][8];
  const int col = blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
  const int total = c0 + c1 + c2;
  const float *src = nullptr;
  int stride = 0, off = 0;
  if (col < c0) { src = p0; stride = c0; off = col; }
  else if (col < c0 + c1) { src = p1; stride = c1; off = col - c0; }
  else if (col < total) { src = p2; stride = c2; off = col - c0 - c1; }
  float acc = 0.f;
  if (src)
    for (int64_t r = rl; r < rows; r += 32) acc += src[r * stride + off];
  red[rl][threadIdx.x & 7] = acc;
  __syncthreads();
  for (int h = 16; h >= 1; h >>= 1) {
    if (rl < h) red[rl][threadIdx.x & 7] += red[rl + h][threadIdx.x & 7];
    __syncthreads();
  }
  if (rl == 0 && col < total) out[col] = red[0][threadIdx.x & 7];
}

extern "C" int link_sum_partials(const float *p0, int32_t cols0, const float *p1, int32_t cols1,
                                 const float *p2, int32_t cols2, int64_t rows, float *out, void *stream) {
  if (cols0 < 0 || cols1 < 0 || cols2 < 0 || rows < 0 || !out) return LINK_ERR_ARG;
  if ((cols0 && !p0) || (cols1 && !p1) || (cols2 && !p2)) return LINK_ERR_ARG;
  const int total = cols0 + cols1 + cols2;
  if (total == 0) return LINK_OK;
  hipLaunchKernelGGL(k_sum_partials, dim3((total + 7) / 8), dim3(256), 0, S(stream), p0, (int)cols0, p1, (int)cols1,
                     p2, (int)cols2, rows, out);
  return check_launch("link_sum_partials");
}

static int train_args_ok(const link_elk_desc_t *desc, const link_grid_t *grid, int64_t n, int64_t m_cap) {
  if (check_desc(desc) != LINK_OK || !grid || n < 0 || m_cap < 0) return LINK_ERR_ARG;
  if ((desc->c & 3) != 0 || desc->r > 3) return LINK_ERR_ARG;      // group kernels only; callers fall back
  return LINK_OK;
}

extern "C" int link_elk_mid_forward(const float *fin, const int32_t *vox_sorted, const int32_t *pos_blk,
                                    const int32_t *blk_start, const int32_t *blk_coords,
                                    const int32_t *cell_blk, const link_grid_t *grid, const int32_t *hdr,
                                    const float *w_pos, const float *alpha, const link_elk_desc_t *desc,
                                    const float *ln_w, const float *ln_b, int64_t n, int64_t m_cap,
                                    float *S_, float *A, float *den, float *out, void *stream) {
  int rc = train_args_ok(desc, grid, n, m_cap);
  if (rc != LINK_OK) return rc;
  if (n == 0 || m_cap == 0) return LINK_OK;
  if (!fin || !vox_sorted || !pos_blk || !blk_start || !blk_coords || !cell_blk || !hdr || !w_pos || !S_ || !A ||
      !den || !out)
    return LINK_ERR_ARG;
  const int4 *v4 = reinterpret_cast<const int4 *>(vox_sorted);
  if (!modsum_group_path(desc, S(stream), fin, v4, w_pos, alpha, blk_start, hdr, S_, m_cap)) return LINK_ERR_ARG;
  rc = check_launch("link_elk_mid_forward");
  if (rc != LINK_OK) return rc;
  rc = block_gather_impl(S_, blk_coords, cell_blk, grid, hdr, desc, m_cap, A, 0, den, stream);
  if (rc != LINK_OK) return rc;
  if ((ln_w == nullptr) != (ln_b == nullptr)) return LINK_ERR_ARG;
  return voxel_demod_impl(A, fin, vox_sorted, pos_blk, w_pos, alpha, ln_w, ln_b, hdr, desc, n, out, stream);
}

extern "C" int link_elk_mid_backward(const float *g_out, const float *fin, const float *A, const float *den,
                                     const int32_t *vox_sorted, const int32_t *pos_blk,
                                     const int32_t *blk_start, const int32_t *blk_coords,
                                     const int32_t *cell_blk, const link_grid_t *grid, const int32_t *hdr,
                                     const float *w_pos, const float *alpha, const link_elk_desc_t *desc,
                                     int64_t n, int64_t m_cap, float *S_, float *gS, float *g_fin,
                                     float *partials, void *stream) {
  int rc = train_args_ok(desc, grid, n, m_cap);
  if (rc != LINK_OK) return rc;
  if (n == 0 || m_cap == 0) return LINK_OK;
  if (!g_out || !fin || !A || !den || !vox_sorted || !pos_blk || !blk_start || !blk_coords || !cell_blk || !hdr ||
      !w_pos || !S_ || !gS || !g_fin || !partials)
    return LINK_ERR_ARG;
  const int4 *v4 = reinterpret_cast<const int4 *>(vox_sorted);
  hipStream_t st = S(stream);
  const int bop = desc->op == LINK_OP_COS ? LINK_OP_COS : (desc->op == LINK_OP_SIN ? LINK_OPI_SIN_BWD : LINK_OPI_COSX_BWD);
  if (!modsum_group_path(desc, st, g_out, v4, w_pos, alpha, blk_start, hdr, S_, m_cap, bop, den)) return LINK_ERR_ARG;
  rc = check_launch("link_elk_mid_backward");
  if (rc != LINK_OK) return rc;
  rc = block_gather_impl(S_, blk_coords, cell_blk, grid, hdr, desc, m_cap, gS, 1 | 2, nullptr, stream);
  if (rc != LINK_OK) return rc;
  const int wgs = link_elk_mid_partial_rows();
  switch (lanes_per_row(desc->c)) {
    case 1: case 2: case 4: launch_voxel_bwd<4>(*desc, wgs, st, gS, A, fin, g_out, v4, pos_blk, w_pos, alpha, hdr, g_fin, partials); break;
    case 8: launch_voxel_bwd<8>(*desc, wgs, st, gS, A, fin, g_out, v4, pos_blk, w_pos, alpha, hdr, g_fin, partials); break;
    case 16: launch_voxel_bwd<16>(*desc, wgs, st, gS, A, fin, g_out, v4, pos_blk, w_pos, alpha, hdr, g_fin, partials); break;
    case 32: launch_voxel_bwd<32>(*desc, wgs, st, gS, A, fin, g_out, v4, pos_blk, w_pos, alpha, hdr, g_fin, partials); break;
    default: launch_voxel_bwd<64>(*desc, wgs, st, gS, A, fin, g_out, v4, pos_blk, w_pos, alpha, hdr, g_fin, partials); break;
  }
  return check_launch("link_elk_mid_backward");
}

// ---------------------------------------------------------------------------------------------
// aux_to_voxel on the block-gather kernel (the drop-in surface's R_agg):
// utils.py:75-82 multiplies the block MEANS by their counts again and sums r^3 neighbours; here the
// means are turned into the S layout (sum rows + zero row + count column) in one pass, the fused
// path's k_block_gather_g produces new_feat = sum/count-sum (and the denominators the backward needs),
// and a float4 row gather expands blocks to voxels.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_means_to_table(const float *__restrict__ mean,
                                                        const int32_t *__restrict__ counts, int64_t m, int w,
                                                        float *__restrict__ S) {
  const int w4 = w >> 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // one float4 of one row
  if (e < (m + 1) * w4) {
    const int64_t row = e / w4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                          // row m: the all-zero row
    if (row < m) {
      const float cnt = (float)counts[row];
      v = *reinterpret_cast<const float4 *>(&mean[e * 4]);
      v.x *= cnt; v.y *= cnt; v.z *= cnt; v.w *= cnt;
    }
    *reinterpret_cast<float4 *>(&S[e * 4]) = v;
  }
  if (e <= m) S[(m + 1) * (int64_t)w + e] = (e < m) ? (float)counts[e] : 0.f;
}

__global__ void __launch_bounds__(256) k_row_gather4(const float *__restrict__ tab,
                                                     const int64_t *__restrict__ idx, int64_t n, int w,
                                                     float *__restrict__ out) {
  const int w4 = w >> 2;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n * w4) return;
  const int64_t i = e / w4;
  const int j = (int)(e - i * w4);
  *reinterpret_cast<float4 *>(&out[e * 4]) = *reinterpret_cast<const float4 *>(&tab[idx[i] * w + 4 * j]);
}

extern "C" int link_aux_to_voxel_forward_grid(const float *small_f, const int32_t *counts,
                                              const int32_t *blk_coords, const int32_t *cell_blk,
                                              const link_grid_t *grid, const int32_t *hdr, const int64_t *idx,
                                              int64_t n, int64_t m, int32_t w, int32_t r, float *S_,
                                              float *new_feat, float *denom, float *out, void *stream) {
  if (n < 0 || m < 0 || w <= 0 || r <= 0 || r > 3 || !grid) return LINK_ERR_ARG;
  int parts = 0;
  if (w % 8 == 0 && w / 2 <= 256) parts = 2;
  else if (w % 12 == 0 && w / 3 <= 256) parts = 3;
  if (!parts) return LINK_ERR_ARG;                   // callers fall back to link_aux_to_voxel_forward
  if (m == 0) return LINK_OK;
  if (!small_f || !counts || !blk_coords || !cell_blk || !hdr || !S_ || !new_feat || !denom) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  hipLaunchKernelGGL(k_means_to_table, dim3(blocks_for((m + 1) * (w / 4), 256)), dim3(256), 0, st, small_f, counts,
                     m, (int)w, S_);
  link_elk_desc_t d = {parts == 3 ? LINK_OP_COSX : LINK_OP_COS, w / parts, w / parts, r, 1.0f, 1e-6f};
  int rc = block_gather_impl(S_, blk_coords, cell_blk, grid, hdr, &d, m, new_feat, 0, denom, stream);
  if (rc != LINK_OK) return rc;
  if (n > 0) {
    if (!idx || !out) return LINK_ERR_ARG;
    hipLaunchKernelGGL(k_row_gather4, dim3(blocks_for(n * (w / 4), 256)), dim3(256), 0, st, new_feat, idx, n, (int)w, out);
  }
  return check_launch("link_aux_to_voxel_forward_grid");
}

// The same with the rows scattered by the gather kernel itself: out[perm[p]] = row of block b for p in
// [blk_start[b], blk_start[b + 1]) -- no [M, W] table of neighbour means, no row-gather launch (two launches in all).
extern "C" int link_aux_to_voxel_forward_scatter(const float *small_f, const int32_t *counts, const int32_t *blk_coords,
                                                 const int32_t *cell_blk, const link_grid_t *grid, const int32_t *hdr,
                                                 const int32_t *blk_start, const int32_t *perm, int64_t n, int64_t m, int32_t w,
                                                 int32_t r, float *S_, float *denom, float *out, void *stream) {
  if (n < 0 || m < 0 || w <= 0 || r <= 0 || r > 3 || !grid) return LINK_ERR_ARG;
  int parts = 0;
  if (w % 8 == 0 && w / 2 <= 256) parts = 2;
  else if (w % 12 == 0 && w / 3 <= 256) parts = 3;
  if (!parts) return LINK_ERR_ARG;
  if (m == 0 || n == 0) return LINK_OK;
  if (!small_f || !counts || !blk_coords || !cell_blk || !hdr || !blk_start || !perm || !S_ || !denom || !out) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  hipLaunchKernelGGL(k_means_to_table, dim3(blocks_for((m + 1) * (w / 4), 256)), dim3(256), 0, st, small_f, counts, m, (int)w, S_);
  link_elk_desc_t d = {parts == 3 ? LINK_OP_COSX : LINK_OP_COS, w / parts, w / parts, r, 1.0f, 1e-6f};
  const bg_scatter_t sc = {blk_start, perm, out};
  return block_gather_impl(S_, blk_coords, cell_blk, grid, hdr, &d, m, nullptr, 0, denom, stream, sc);
}

// ---------------------------------------------------------------------------------------------
// one-call R_core
// ---------------------------------------------------------------------------------------------
extern "C" int link_elk_core_forward(const link_elk_buffers_t *b, const link_grid_t *grid,
                                     const link_elk_desc_t *desc, int64_t n, int64_t m_cap,
                                     int32_t build_index, void *stream) {
  if (!b || !grid || check_desc(desc) != LINK_OK) return LINK_ERR_ARG;
  int rc;
  if (build_index == 2) {                                      // a numbering, not the reference's: scan over the voxels
    if (!b->cell_pair) return LINK_ERR_ARG;
    rc = link_index_build_first(b->coords, n, grid, b->cell_pair, b->scratch, b->scratch_bytes, b->cell_blk,
                                b->vox_blk, b->idx_query, b->perm, b->vox_sorted, b->pos_blk, b->blk_start, b->blk_coords,
                                b->counts, b->hdr, stream);
    if (rc != LINK_OK) return rc;
  } else if (build_index) {
    rc = link_index_build(b->coords, n, grid, b->cell_counts, b->scratch, b->scratch_bytes, b->cell_blk,
                          b->vox_blk, b->idx_query, b->perm, b->vox_sorted, b->pos_blk, b->blk_start,
                          b->blk_coords, b->counts, b->hdr, stream);
    if (rc != LINK_OK) return rc;
  }
  if (desc->flags & LINK_ELK_TILES) {
    rc = link_elk_premix_modsum_tiles_io(b->feats, b->io_dtype, b->vox_sorted, b->pos_blk, b->blk_start, b->hdr, b->w_pre, b->pre_ln_w,
                                         b->pre_ln_b, b->w_pos, b->alpha, desc, n, m_cap, b->S, b->s_bytes, b->fin, stream);
    if (rc != LINK_OK) return rc;
    return link_elk_gather_demod_tiles_io(b->S, b->fin, b->vox_sorted, b->pos_blk, b->blk_coords, b->cell_blk, grid, b->hdr,
                                          b->w_pos, b->alpha, b->ln_w, b->ln_b, desc, n, m_cap, b->out, b->io_dtype, stream);
  }
  if (b->io_dtype != LINK_IO_F32) return LINK_ERR_ARG;         // the four-kernel form is fp32
  const float *feats32 = static_cast<const float *>(b->feats);
  float *out32 = static_cast<float *>(b->out);
  rc = link_premix_ln(feats32, b->w_pre, b->pre_ln_w, b->pre_ln_b, n, desc->c, desc->eps, b->fin, stream);
  if (rc != LINK_OK) return rc;
  rc = link_modulate_block_sum(b->fin, b->vox_sorted, b->w_pos, b->alpha, b->blk_start, b->hdr, desc, n,
                               m_cap, b->S, stream);
  if (rc != LINK_OK) return rc;
  if (!(desc->flags & LINK_ELK_FUSED_GATHER) && b->A && b->pos_blk && (desc->c & 3) == 0 && desc->r <= 3) {
    rc = link_block_gather(b->S, b->blk_coords, b->cell_blk, grid, b->hdr, desc, m_cap, b->A, stream);
    if (rc != LINK_OK) return rc;
    return link_voxel_demod_ln(b->A, b->fin, b->vox_sorted, b->pos_blk, b->w_pos, b->alpha, b->ln_w, b->ln_b,
                               b->hdr, desc, n, out32, stream);
  }
  return link_gather_demod_ln(b->S, b->fin, b->vox_sorted, b->w_pos, b->alpha, b->ln_w, b->ln_b,
                              b->blk_start, b->blk_coords, b->cell_blk, grid, b->hdr, desc, n, m_cap, out32,
                              stream);
}
