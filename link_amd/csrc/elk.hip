// link_amd/csrc/elk.hip -- section C of include/link_amd.h: the fused R_core of ELKBlock.forward
// (segmentation/core/models/semantic_kitti/linkunet.py:124-185; detection/det3d/models/utils/
// ts_elk.py:144-230) for gfx950.
//
//   k_premix_ln        fin = LayerNorm(F @ Wpre^T): the only GEMM-shaped step -> f32 MFMA
//                      (v_mfma_f32_16x16x4_f32, exact f32), W staged once per workgroup in LDS
//                      (row stride padded by 4 dwords: conflict-free ds_read_b128), D = W * F^T so
//                      that a lane ends up holding 16t+4g+r channels of ONE voxel: the LayerNorm
//                      reduction is 16 in-lane adds + 2 cross-lane steps, and stores are 16 B/lane.
//   k_modulate_sum     theta / sincos / modulate / per-block pre-aggregation: one wave per block,
//                      lanes = channels, voxels of the block visited in ascending id, sums kept in
//                      registers, ONE non-atomic row write per block.
//   k_gather_demod_ln  r^3 neighbour-block sum (ids looked up in the dense cell table by lanes
//                      0..K-1, broadcast with readlane, rows summed in get_kernel_offsets order),
//                      normalise, then per voxel of the block: de-modulate + LayerNorm + store.
//
// S row layout: [part0 C | part1 C | (part2 C) | count, 3 pad] fp32 -> row stride P*C+4 (16-B rows).
#include "common.h"

using namespace link;

typedef float floatx4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// pre_mix + LayerNorm (MFMA path, C in {16,32,48,...,128}, C % 16 == 0)
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ void __launch_bounds__(256) k_premix_ln_mfma(const float *__restrict__ feats,
                                                        const float *__restrict__ w_pre,
                                                        const float *__restrict__ ln_w,
                                                        const float *__restrict__ ln_b, int64_t n,
                                                        float eps, float *__restrict__ fin) {
  constexpr int T = C / 16;       // 16-wide tiles along channels (rows of D) and along k
  constexpr int LDW = C + 4;      // padded LDS row stride (dwords)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  // stage W (row-major [C][C]) into LDS, 16 B per thread per step
  for (int e = tid * 4; e < C * C; e += 256 * 4) {
    int r = e / C, col = e - r * C;
    *reinterpret_cast<float4 *>(&w_lds[r * LDW + col]) = *reinterpret_cast<const float4 *>(&w_pre[e]);
  }
  // per-lane LayerNorm affine for its channels 16t+4g+r
  float4 lw[T], lb[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
    lw[t] = *reinterpret_cast<const float4 *>(&ln_w[16 * t + 4 * g]);
    lb[t] = *reinterpret_cast<const float4 *>(&ln_b[16 * t + 4 * g]);
  }
  __syncthreads();
  const int64_t tiles = (n + 15) / 16;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < tiles; tile += (int64_t)gridDim.x * 4) {
    const int64_t v = tile * 16 + li;
    const bool ok = v < n;
    // B operand source: this lane's voxel row, k = 16t + 4g + e
    float4 f[T];
#pragma unroll
    for (int t = 0; t < T; t++)
      f[t] = ok ? *reinterpret_cast<const float4 *>(&feats[v * C + 16 * t + 4 * g])
                : make_float4(0.f, 0.f, 0.f, 0.f);
    floatx4 acc[T];
#pragma unroll
    for (int tp = 0; tp < T; tp++) acc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; t++) {
#pragma unroll
      for (int tp = 0; tp < T; tp++) {
        // A operand: W[16tp + li][16t + 4g + e]
        float4 a = *reinterpret_cast<const float4 *>(&w_lds[(16 * tp + li) * LDW + 16 * t + 4 * g]);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, f[t].x, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, f[t].y, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, f[t].z, acc[tp], 0, 0, 0);
        acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, f[t].w, acc[tp], 0, 0, 0);
      }
    }
    // lane holds channels {16tp + 4g + r} of voxel li: LayerNorm over all C channels
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
        float4 o;
        o.x = (acc[tp][0] - mean) * rstd * lw[tp].x + lb[tp].x;
        o.y = (acc[tp][1] - mean) * rstd * lw[tp].y + lb[tp].y;
        o.z = (acc[tp][2] - mean) * rstd * lw[tp].z + lb[tp].z;
        o.w = (acc[tp][3] - mean) * rstd * lw[tp].w + lb[tp].w;
        *reinterpret_cast<float4 *>(&fin[v * C + 16 * tp + 4 * g]) = o;
      }
    }
  }
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

template <int C>
static int launch_premix_mfma(const float *feats, const float *w_pre, const float *ln_w,
                              const float *ln_b, int64_t n, float eps, float *fin, hipStream_t st) {
  size_t lds = (size_t)C * (C + 4) * sizeof(float);
  int64_t tiles = (n + 15) / 16;
  int64_t wgs = (tiles + 3) / 4;
  int64_t cap = 256 * 4;  // 4 workgroups per CU (LDS: 4 * 17 KB at C=64)
  if (lds > 40 * 1024) cap = 256 * 2;
  if (wgs > cap) wgs = cap;
  hipLaunchKernelGGL(k_premix_ln_mfma<C>, dim3((unsigned)wgs), dim3(256), lds, st, feats, w_pre, ln_w,
                     ln_b, n, eps, fin);
  return check_launch("link_premix_ln");
}

extern "C" int link_premix_ln(const float *feats, const float *w_pre, const float *ln_w,
                              const float *ln_b, int64_t n, int32_t c, float eps, float *fin,
                              void *stream) {
  if (n < 0 || c <= 0 || c > 256) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!feats || !w_pre || !ln_w || !ln_b || !fin) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (c) {
    case 16: return launch_premix_mfma<16>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 32: return launch_premix_mfma<32>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 48: return launch_premix_mfma<48>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 64: return launch_premix_mfma<64>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 96: return launch_premix_mfma<96>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
    case 128: return launch_premix_mfma<128>(feats, w_pre, ln_w, ln_b, n, eps, fin, st);
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

// ---------------------------------------------------------------------------------------------
// theta for one channel of one voxel.  theta = ((x/div)*w0 + (y/div)*w1) + (z/div)*w2 evaluated as an
// fma chain in x,y,z order (nn.Linear(3, cg, bias=False) on float coords, linkunet.py:151), then
// * alpha for cos_x (linkunet.py:165).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float theta_of(float x, float y, float z, float w0, float w1, float w2,
                                          float alpha) {
  float t = fmaf(z, w2, fmaf(y, w1, x * w0));
  return t * alpha;
}

template <int CPL, int OP>
__global__ void __launch_bounds__(256) k_modulate_sum(const float *__restrict__ fin,
                                                      const int4 *__restrict__ coords,
                                                      const float *__restrict__ w_pos,
                                                      const float *__restrict__ alpha,
                                                      const int32_t *__restrict__ perm,
                                                      const int32_t *__restrict__ blk_start,
                                                      const int32_t *__restrict__ hdr, int c, int cg,
                                                      float coord_div, float *__restrict__ S) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (b >= hdr[LINK_HDR_M]) return;
  const int st = blk_start[b], en = blk_start[b + 1];
  const int rs = P * c + 4;
  float w0[CPL], w1[CPL], w2[CPL], al[CPL];
  float a0[CPL], a1[CPL], a2[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    int ch = lane + 64 * q;
    int tc = (ch < c) ? ch % cg : 0;
    w0[q] = w_pos[3 * tc + 0]; w1[q] = w_pos[3 * tc + 1]; w2[q] = w_pos[3 * tc + 2];
    al[q] = alpha ? alpha[tc] : 1.0f;
    a0[q] = a1[q] = a2[q] = 0.f;
  }
  for (int p = st; p < en; p++) {
    const int i = perm[p];
    const int4 cd = coords[i];
    float x = (float)cd.x, y = (float)cd.y, z = (float)cd.z;
    if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      int ch = lane + 64 * q;
      if (ch < c) {
        float f = fin[(int64_t)i * c + ch];
        float th = theta_of(x, y, z, w0[q], w1[q], w2[q], al[q]);
        float sn, cs;
        sincosf(th, &sn, &cs);
        if (OP == LINK_OP_SIN) { a0[q] += f * sn; a1[q] += f * cs; }
        else { a0[q] += f * cs; a1[q] += f * sn; }
        if (OP == LINK_OP_COSX) a2[q] += f * th;
      }
    }
  }
  float *row = S + b * (int64_t)rs;
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    int ch = lane + 64 * q;
    if (ch < c) {
      row[ch] = a0[q];
      row[c + ch] = a1[q];
      if (OP == LINK_OP_COSX) row[2 * c + ch] = a2[q];
    }
  }
  if (lane == 0) row[P * c] = (float)(en - st);
}

template <int CPL>
static void launch_modsum(int op, dim3 grid, hipStream_t st, const float *fin, const int4 *coords,
                          const float *w_pos, const float *alpha, const int32_t *perm,
                          const int32_t *blk_start, const int32_t *hdr, int c, int cg, float div,
                          float *S) {
  dim3 block(256);
  if (op == LINK_OP_COS)
    hipLaunchKernelGGL((k_modulate_sum<CPL, LINK_OP_COS>), grid, block, 0, st, fin, coords, w_pos, alpha, perm, blk_start, hdr, c, cg, div, S);
  else if (op == LINK_OP_SIN)
    hipLaunchKernelGGL((k_modulate_sum<CPL, LINK_OP_SIN>), grid, block, 0, st, fin, coords, w_pos, alpha, perm, blk_start, hdr, c, cg, div, S);
  else
    hipLaunchKernelGGL((k_modulate_sum<CPL, LINK_OP_COSX>), grid, block, 0, st, fin, coords, w_pos, alpha, perm, blk_start, hdr, c, cg, div, S);
}

static int check_desc(const link_elk_desc_t *d) {
  if (!d) return LINK_ERR_ARG;
  if (d->op < 0 || d->op > 2 || d->c <= 0 || d->c > 256 || d->cg <= 0 || d->cg > d->c) return LINK_ERR_ARG;
  if (d->r <= 0 || d->r > 7 || d->coord_div == 0.f) return LINK_ERR_ARG;
  return LINK_OK;
}

extern "C" int link_modulate_block_sum(const float *fin, const int32_t *coords, const float *w_pos,
                                       const float *alpha, const int32_t *perm,
                                       const int32_t *blk_start, const int32_t *hdr,
                                       const link_elk_desc_t *desc, int64_t n, int64_t m_cap, float *S_,
                                       void *stream) {
  if (check_desc(desc) != LINK_OK || n < 0 || m_cap < 0) return LINK_ERR_ARG;
  if (n == 0 || m_cap == 0) return LINK_OK;
  if (!fin || !coords || !w_pos || !perm || !blk_start || !hdr || !S_) return LINK_ERR_ARG;
  dim3 grid(blocks_for(m_cap * 64, 256));
  const int4 *c4 = reinterpret_cast<const int4 *>(coords);
  int cpl = (desc->c + 63) / 64;
  hipStream_t st = S(stream);
  switch (cpl) {
    case 1: launch_modsum<1>(desc->op, grid, st, fin, c4, w_pos, alpha, perm, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_); break;
    case 2: launch_modsum<2>(desc->op, grid, st, fin, c4, w_pos, alpha, perm, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_); break;
    case 3: launch_modsum<3>(desc->op, grid, st, fin, c4, w_pos, alpha, perm, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_); break;
    default: launch_modsum<4>(desc->op, grid, st, fin, c4, w_pos, alpha, perm, blk_start, hdr, desc->c, desc->cg, desc->coord_div, S_); break;
  }
  return check_launch("link_modulate_block_sum");
}

// ---------------------------------------------------------------------------------------------
// neighbour-block sum + normalise + de-modulate + LayerNorm
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void kernel_offset_d(int r, int k, int &ox, int &oy, int &oz) {
  int lo = -((r + 1) / 2) + 1;   // nn/utils/kernel.py:21: arange(-r//2+1, r//2+1)
  int a = k % r, b = (k / r) % r, cc = k / (r * r);
  if ((r & 1) != 0) { ox = lo + a; oy = lo + b; oz = lo + cc; }   // odd volume: x fastest
  else { oz = lo + a; oy = lo + b; ox = lo + cc; }                // even volume: z fastest
}

template <int CPL, int OP>
__global__ void __launch_bounds__(256) k_gather_demod_ln(
    const float *__restrict__ S, const float *__restrict__ fin, const int4 *__restrict__ coords,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, const int32_t *__restrict__ perm,
    const int32_t *__restrict__ blk_start, const int4 *__restrict__ blk_coords,
    const int32_t *__restrict__ cell_blk, link_grid_t g, const int32_t *__restrict__ hdr, int c, int cg,
    int r, float coord_div, float eps, float *__restrict__ out) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (b >= hdr[LINK_HDR_M]) return;
  const int st = blk_start[b], en = blk_start[b + 1];
  const int rs = P * c + 4;
  const int K = r * r * r;
  const int4 bc = blk_coords[b];
  float A0[CPL], A1[CPL], A2[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) A0[q] = A1[q] = A2[q] = 0.f;
  float den = 0.f;
  for (int k0 = 0; k0 < K; k0 += 64) {
    int kk = k0 + lane;
    int32_t nb_l = -1;
    if (kk < K) {
      int ox, oy, oz;
      kernel_offset_d(r, kk, ox, oy, oz);
      int32_t cell = cell_of(g, bc.x + ox, bc.y + oy, bc.z + oz, bc.w);
      if (cell >= 0) nb_l = cell_blk[cell] - 1;
    }
    int lim = (K - k0 < 64) ? (K - k0) : 64;
    for (int t = 0; t < lim; t++) {
      int32_t nb = __shfl(nb_l, t, 64);
      if (nb >= 0) {   // wave-uniform
        const float *row = S + (int64_t)nb * rs;
        den += row[P * c];
#pragma unroll
        for (int q = 0; q < CPL; q++) {
          int ch = lane + 64 * q;
          if (ch < c) {
            A0[q] += row[ch];
            A1[q] += row[c + ch];
            if (OP == LINK_OP_COSX) A2[q] += row[2 * c + ch];
          }
        }
      }
    }
  }
  float w0[CPL], w1[CPL], w2[CPL], al[CPL], gw[CPL], gb[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    int ch = lane + 64 * q;
    int tc = (ch < c) ? ch % cg : 0;
    w0[q] = w_pos[3 * tc + 0]; w1[q] = w_pos[3 * tc + 1]; w2[q] = w_pos[3 * tc + 2];
    al[q] = alpha ? alpha[tc] : 1.0f;
    gw[q] = (ch < c) ? ln_w[ch] : 0.f;
    gb[q] = (ch < c) ? ln_b[ch] : 0.f;
    A0[q] = A0[q] / den; A1[q] = A1[q] / den; A2[q] = A2[q] / den;   // utils.py:80
  }
  for (int p = st; p < en; p++) {
    const int i = perm[p];
    const int4 cd = coords[i];
    float x = (float)cd.x, y = (float)cd.y, z = (float)cd.z;
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
        sincosf(th, &sn, &cs);
        float v;
        if (OP == LINK_OP_SIN) v = __fsub_rn(__fmul_rn(A0[q], cs), __fmul_rn(A1[q], sn));    // linkunet.py:148
        else v = __fadd_rn(__fmul_rn(A0[q], cs), __fmul_rn(A1[q], sn));                       // :162
        if (OP == LINK_OP_COSX) {
          float f = fin[(int64_t)i * c + ch];
          v = __fadd_rn(v, __fsub_rn(A2[q], __fmul_rn(f, th)));                               // :176
        }
        nv[q] = v;
        s += v;
      }
    }
    s = wave_sum(s);
    const float mean = s / c;
    float qq = 0.f;
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      float d = (lane + 64 * q < c) ? nv[q] - mean : 0.f;
      qq += d * d;
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

template <int CPL>
static void launch_gdl(int op, dim3 grid, hipStream_t st, const float *S_, const float *fin,
                       const int4 *coords, const float *w_pos, const float *alpha, const float *ln_w,
                       const float *ln_b, const int32_t *perm, const int32_t *blk_start,
                       const int4 *blk_coords, const int32_t *cell_blk, const link_grid_t &g,
                       const int32_t *hdr, const link_elk_desc_t &d, float *out) {
  dim3 block(256);
  if (op == LINK_OP_COS)
    hipLaunchKernelGGL((k_gather_demod_ln<CPL, LINK_OP_COS>), grid, block, 0, st, S_, fin, coords, w_pos, alpha, ln_w, ln_b, perm, blk_start, blk_coords, cell_blk, g, hdr, d.c, d.cg, d.r, d.coord_div, d.eps, out);
  else if (op == LINK_OP_SIN)
    hipLaunchKernelGGL((k_gather_demod_ln<CPL, LINK_OP_SIN>), grid, block, 0, st, S_, fin, coords, w_pos, alpha, ln_w, ln_b, perm, blk_start, blk_coords, cell_blk, g, hdr, d.c, d.cg, d.r, d.coord_div, d.eps, out);
  else
    hipLaunchKernelGGL((k_gather_demod_ln<CPL, LINK_OP_COSX>), grid, block, 0, st, S_, fin, coords, w_pos, alpha, ln_w, ln_b, perm, blk_start, blk_coords, cell_blk, g, hdr, d.c, d.cg, d.r, d.coord_div, d.eps, out);
}

extern "C" int link_gather_demod_ln(const float *S_, const float *fin, const int32_t *coords,
                                    const float *w_pos, const float *alpha, const float *ln_w,
                                    const float *ln_b, const int32_t *perm, const int32_t *blk_start,
                                    const int32_t *blk_coords, const int32_t *cell_blk,
                                    const link_grid_t *grid, const int32_t *hdr,
                                    const link_elk_desc_t *desc, int64_t n, int64_t m_cap, float *out,
                                    void *stream) {
  if (check_desc(desc) != LINK_OK || !grid || n < 0 || m_cap < 0) return LINK_ERR_ARG;
  if (n == 0 || m_cap == 0) return LINK_OK;
  if (!S_ || !coords || !w_pos || !ln_w || !ln_b || !perm || !blk_start || !blk_coords || !cell_blk ||
      !hdr || !out)
    return LINK_ERR_ARG;
  if (desc->op == LINK_OP_COSX && !fin) return LINK_ERR_ARG;
  dim3 g3(blocks_for(m_cap * 64, 256));
  const int4 *c4 = reinterpret_cast<const int4 *>(coords);
  const int4 *b4 = reinterpret_cast<const int4 *>(blk_coords);
  hipStream_t st = S(stream);
  int cpl = (desc->c + 63) / 64;
  switch (cpl) {
    case 1: launch_gdl<1>(desc->op, g3, st, S_, fin, c4, w_pos, alpha, ln_w, ln_b, perm, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
    case 2: launch_gdl<2>(desc->op, g3, st, S_, fin, c4, w_pos, alpha, ln_w, ln_b, perm, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
    case 3: launch_gdl<3>(desc->op, g3, st, S_, fin, c4, w_pos, alpha, ln_w, ln_b, perm, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
    default: launch_gdl<4>(desc->op, g3, st, S_, fin, c4, w_pos, alpha, ln_w, ln_b, perm, blk_start, b4, cell_blk, *grid, hdr, *desc, out); break;
  }
  return check_launch("link_gather_demod_ln");
}

// ---------------------------------------------------------------------------------------------
// one-call R_core
// ---------------------------------------------------------------------------------------------
extern "C" int link_elk_core_forward(const link_elk_buffers_t *b, const link_grid_t *grid,
                                     const link_elk_desc_t *desc, int64_t n, int64_t m_cap,
                                     int32_t build_index, void *stream) {
  if (!b || !grid || check_desc(desc) != LINK_OK) return LINK_ERR_ARG;
  int rc;
  if (build_index) {
    rc = link_index_build(b->coords, n, grid, b->cell_counts, b->scratch, b->scratch_bytes, b->cell_blk,
                          b->vox_blk, b->idx_query, b->perm, b->blk_start, b->blk_coords, b->counts, b->hdr,
                          stream);
    if (rc != LINK_OK) return rc;
  }
  rc = link_premix_ln(b->feats, b->w_pre, b->pre_ln_w, b->pre_ln_b, n, desc->c, desc->eps, b->fin, stream);
  if (rc != LINK_OK) return rc;
  rc = link_modulate_block_sum(b->fin, b->coords, b->w_pos, b->alpha, b->perm, b->blk_start, b->hdr, desc, n,
                               m_cap, b->S, stream);
  if (rc != LINK_OK) return rc;
  return link_gather_demod_ln(b->S, b->fin, b->coords, b->w_pos, b->alpha, b->ln_w, b->ln_b, b->perm,
                              b->blk_start, b->blk_coords, b->cell_blk, grid, b->hdr, desc, n, m_cap, b->out,
                              stream);
}
