// link_amd/csrc/aggregate.hip -- section B (feature half) of include/link_amd.h: the indexed,
// deterministic forms of spvoxelize / aux_to_voxel (segmentation/core/models/utils.py:52,75-82) and
// their adjoints.  Layout: features row-major fp32 [rows, c]; one 64-lane wave owns one block (or one
// output row) and its lanes stride the channels, so every row access is a contiguous, coalesced
// burst; all reductions run in registers in a fixed order (ascending voxel id / ascending k): no fp
// atomics anywhere on the indexed path.
#include "common.h"

using namespace link;

// ---------------------------------------------------------------------------------------------
// block mean: out[b] = sum_{i in block b} in[i] / count_b   (division first, like voxelize_cuda.cu:21)
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(256) k_block_mean(const float *__restrict__ in,
                                                    const int32_t *__restrict__ perm,
                                                    const int32_t *__restrict__ blk_start,
                                                    const int32_t *__restrict__ hdr, int c,
                                                    float *__restrict__ out) {
  int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (b >= hdr[LINK_HDR_M]) return;
  int st = blk_start[b], en = blk_start[b + 1];
  float fc = (float)(en - st);
  for (int j0 = 0; j0 < c; j0 += 64 * VEC) {
    int j = j0 + lane * VEC;
    float acc[VEC];
#pragma unroll
    for (int v = 0; v < VEC; v++) acc[v] = 0.f;
    // the block's voxel ids 64 at a time (one load per lane, then lane broadcasts), rows four in flight: the chain per
    // voxel was id -> row -> add, one after the other (same additions in the same order: bitwise the same means)
    const int jj = j < c ? j : 0;
    for (int p0 = st; p0 < en; p0 += 64) {
      const int np = en - p0 < 64 ? en - p0 : 64;
      const int my = perm[p0 + (lane < np ? lane : 0)];
      for (int q = 0; q < np; q += 4) {
        float x[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int id = __shfl(my, q + u < np ? q + u : q, 64);
          const float *src = in + (int64_t)id * c + jj;
          if (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(src);
            x[u][0] = t.x; x[u][1 % VEC] = t.y; x[u][2 % VEC] = t.z; x[u][3 % VEC] = t.w;
          } else if (VEC == 2) {
            const float2 t = *reinterpret_cast<const float2 *>(src);
            x[u][0] = t.x; x[u][1 % VEC] = t.y;
          } else {
            x[u][0] = src[0];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (q + u < np) {                            // wave-uniform
#pragma unroll
            for (int v = 0; v < VEC; v++) acc[v] += x[u][v] / fc;
          }
      }
    }
    if (j < c) {
      float *dst = out + b * (int64_t)c + j;
#pragma unroll
      for (int v = 0; v < VEC; v++) dst[v] = acc[v];
    }
  }
}

extern "C" int link_block_mean(const float *in, const int32_t *perm, const int32_t *blk_start,
                               const int32_t *hdr, int64_t n, int64_t c, int64_t m_cap, float *out,
                               void *stream) {
  if (n < 0 || c < 0 || m_cap < 0 || c > (1 << 20)) return LINK_ERR_ARG;
  if (m_cap == 0 || c == 0) return LINK_OK;
  if (!in || !perm || !blk_start || !hdr || !out) return LINK_ERR_ARG;
  dim3 grid(blocks_for(m_cap * 64, 256)), block(256);
  if (c % 4 == 0 && c >= 256)
    hipLaunchKernelGGL(k_block_mean<4>, grid, block, 0, S(stream), in, perm, blk_start, hdr, (int)c, out);
  else if (c % 2 == 0 && c >= 128)
    hipLaunchKernelGGL(k_block_mean<2>, grid, block, 0, S(stream), in, perm, blk_start, hdr, (int)c, out);
  else
    hipLaunchKernelGGL(k_block_mean<1>, grid, block, 0, S(stream), in, perm, blk_start, hdr, (int)c, out);
  return check_launch("link_block_mean");
}

// ---------------------------------------------------------------------------------------------
// aux_to_voxel forward: new[m] = sum_k F[nbr]*cnt[nbr] / sum_k cnt[nbr];  out[i] = new[idx[i]]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_nbr_reduce(const float *__restrict__ small_f,
                                                    const int32_t *__restrict__ counts,
                                                    const int32_t *__restrict__ nbr, int64_t m, int c,
                                                    int K, float *__restrict__ new_feat,
                                                    float *__restrict__ denom) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= m) return;
  for (int j0 = 0; j0 < c; j0 += 64) {
    int j = j0 + lane;
    float acc = 0.f, den = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
      int kk = k0 + lane;
      int32_t q_l = (kk < K) ? nbr[row * K + kk] : -1;
      float c_l = (q_l >= 0) ? (float)counts[q_l] : 0.f;
      int lim = (K - k0 < 64) ? (K - k0) : 64;
      for (int t = 0; t < lim; t++) {
        int32_t q = __shfl(q_l, t, 64);
        float cq = __shfl(c_l, t, 64);
        if (q >= 0) {                       // wave-uniform branch
          den += cq;                        // sum of f[:, -1] = 1*count   (utils.py:75-76)
          if (j < c) acc += small_f[(int64_t)q * c + j] * cq;   // f = F*count, weight 1
        }
      }
    }
    if (j < c) new_feat[row * (int64_t)c + j] = acc / den;      // utils.py:80
    if (j0 == 0 && lane == 0) denom[row] = den;
  }
}

__global__ void __launch_bounds__(256) k_row_gather64(const float *__restrict__ src,
                                                      const int64_t *__restrict__ idx, int64_t n, int c,
                                                      float *__restrict__ out) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  int64_t q = idx[row];
  const float *s = src + q * c;
  float *d = out + row * (int64_t)c;
  if ((c & 3) == 0) {
    for (int j = lane * 4; j < c; j += 256)
      *reinterpret_cast<float4 *>(d + j) = *reinterpret_cast<const float4 *>(s + j);
  } else {
    for (int j = lane; j < c; j += 64) d[j] = s[j];
  }
}

extern "C" int link_aux_to_voxel_forward(const float *small_f, const int32_t *counts,
                                         const int32_t *nbr, const int64_t *idx, int64_t n, int64_t m,
                                         int64_t c, int64_t k, float *new_feat, float *denom, float *out,
                                         void *stream) {
  if (n < 0 || m < 0 || c < 0 || k <= 0 || c > (1 << 20) || k > (1 << 15)) return LINK_ERR_ARG;
  if (m == 0 || c == 0) return LINK_OK;
  if (!small_f || !counts || !nbr || !new_feat || !denom) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_nbr_reduce, dim3(blocks_for(m * 64, 256)), dim3(256), 0, S(stream), small_f,
                     counts, nbr, m, (int)c, (int)k, new_feat, denom);
  if (n > 0) {
    if (!idx || !out) return LINK_ERR_ARG;
    hipLaunchKernelGGL(k_row_gather64, dim3(blocks_for(n * 64, 256)), dim3(256), 0, S(stream), new_feat,
                       idx, n, (int)c, out);
  }
  return check_launch("link_aux_to_voxel_forward");
}

// ---------------------------------------------------------------------------------------------
// aux_to_voxel backward
//   g_new[m]   = sum_{i in block m} g_out[i]                  (segment sum, ascending voxel id)
//   g_small[j] = cnt[j] * sum_{m in nbr_t(j)} g_new[m]/denom[m]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_seg_sum(const float *__restrict__ g_out,
                                                 const int32_t *__restrict__ perm,
                                                 const int32_t *__restrict__ blk_start, int64_t m,
                                                 int c, float *__restrict__ g_new) {
  int64_t b = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (b >= m) return;
  int st = blk_start[b], en = blk_start[b + 1];
  for (int j = lane; j < c; j += 64) {
    float acc = 0.f;
    for (int p = st; p < en; p++) acc += g_out[(int64_t)perm[p] * c + j];
    g_new[b * (int64_t)c + j] = acc;
  }
}

__global__ void __launch_bounds__(256) k_nbr_reduce_t(const float *__restrict__ g_new,
                                                      const float *__restrict__ denom,
                                                      const int32_t *__restrict__ counts,
                                                      const int32_t *__restrict__ nbr_t, int64_t m,
                                                      int c, int K, float *__restrict__ g_small) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= m) return;
  float cnt = (float)counts[row];
  for (int j0 = 0; j0 < c; j0 += 64) {
    int j = j0 + lane;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
      int kk = k0 + lane;
      int32_t q_l = (kk < K) ? nbr_t[row * K + kk] : -1;
      float d_l = (q_l >= 0) ? denom[q_l] : 1.f;
      int lim = (K - k0 < 64) ? (K - k0) : 64;
      for (int t = 0; t < lim; t++) {
        int32_t q = __shfl(q_l, t, 64);
        float dq = __shfl(d_l, t, 64);
        if (q >= 0 && j < c) acc += g_new[(int64_t)q * c + j] / dq;
      }
    }
    if (j < c) g_small[row * (int64_t)c + j] = acc * cnt;
  }
}

extern "C" int link_aux_to_voxel_backward(const float *g_out, const int32_t *perm,
                                          const int32_t *blk_start, const int32_t *counts,
                                          const int32_t *nbr_t, const float *denom, int64_t n,
                                          int64_t m, int64_t c, int64_t k, float *g_new, float *g_small,
                                          void *stream) {
  if (n < 0 || m < 0 || c < 0 || k <= 0 || c > (1 << 20) || k > (1 << 15)) return LINK_ERR_ARG;
  if (m == 0 || c == 0) return LINK_OK;
  if (!g_out || !perm || !blk_start || !counts || !nbr_t || !denom || !g_new || !g_small)
    return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_seg_sum, dim3(blocks_for(m * 64, 256)), dim3(256), 0, S(stream), g_out, perm,
                     blk_start, m, (int)c, g_new);
  hipLaunchKernelGGL(k_nbr_reduce_t, dim3(blocks_for(m * 64, 256)), dim3(256), 0, S(stream), g_new, denom,
                     counts, nbr_t, m, (int)c, (int)k, g_small);
  return check_launch("link_aux_to_voxel_backward");
}
