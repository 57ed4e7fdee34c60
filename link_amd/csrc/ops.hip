// link_amd/csrc/ops.hip -- section A of include/link_amd.h: gfx950 replacements for the
// `torchsparse.backend` functions on the LinK path (reference:
// segmentation/torchsparse-u/torchsparse/backend/{hash,others,hashmap,voxelize,devoxelize}).
// Written for CDNA4 directly: 64-wide waves, row-per-wave feature kernels with 16-byte lanes where
// the width allows, no host syncs, no allocation.
#include <string.h>

#include <string>

#include "common.h"

namespace link {
static thread_local std::string g_err;
void set_error(const char *what, hipError_t e) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
}
}  // namespace link

using namespace link;

extern "C" int link_abi_version(void) { return 12; }
extern "C" int32_t link_abi_struct_size(int32_t which) {
  switch (which) {
    case 0: return (int32_t)sizeof(link_grid_t);
    case 1: return (int32_t)sizeof(link_elk_desc_t);
    case 2: return (int32_t)sizeof(link_elk_buffers_t);
    case 3: return (int32_t)sizeof(link_dc_grid_t);
    case 4: return (int32_t)sizeof(link_dc_tuning_t);
    case 5: return (int32_t)sizeof(link_dc_buffers_t);
    case 6: return (int32_t)sizeof(link_lean_buffers_t);
    case 7: return (int32_t)sizeof(link_block_args_t);
    default: return -1;
  }
}
extern "C" const char *link_last_error(void) { return link::g_err.c_str(); }

// ---------------------------------------------------------------------------------------------
// hash / kernel_hash: one thread per row; rows are 16 B so a wave reads 1 KiB contiguous (int4).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_hash(const int4 *__restrict__ coords, int64_t n,
                                              int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int4 c = coords[i];
    out[i] = fnv4(c.x, c.y, c.z, c.w);
  }
}

extern "C" int link_hash(const int32_t *coords, int64_t n, int64_t *out, void *stream) {
  if (n < 0 || (n > 0 && (!coords || !out))) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  hipLaunchKernelGGL(k_hash, dim3(blocks_for(n, 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const int4 *>(coords), n, out);
  return check_launch("link_hash");
}

// out is k-major [K,N]: thread -> (k = blockIdx.y, i) so that stores are coalesced along i and the
// row is re-read from L2 K times (16 B/row, negligible) instead of scattering 8-byte stores.
__global__ void __launch_bounds__(256) k_kernel_hash(const int4 *__restrict__ coords, int64_t n,
                                                     const int32_t *__restrict__ offsets,
                                                     int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int k = blockIdx.y;
  if (i < n) {
    int4 c = coords[i];
    int ox = offsets[3 * k + 0], oy = offsets[3 * k + 1], oz = offsets[3 * k + 2];
    out[(int64_t)k * n + i] = fnv4(c.x + ox, c.y + oy, c.z + oz, c.w);
  }
}

extern "C" int link_kernel_hash(const int32_t *coords, int64_t n, const int32_t *offsets, int64_t k,
                                int64_t *out, void *stream) {
  if (n < 0 || k < 0 || k > 65535) return LINK_ERR_ARG;
  if (n == 0 || k == 0) return LINK_OK;
  if (!coords || !offsets || !out) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_kernel_hash, dim3(blocks_for(n, 256), (unsigned)k), dim3(256), 0, S(stream),
                     reinterpret_cast<const int4 *>(coords), n, offsets, out);
  return check_launch("link_kernel_hash");
}

// ---------------------------------------------------------------------------------------------
// hash_query: open addressing, linear probing, power-of-two table; WAIT-FREE insert.
//   keys[cap] u64 (SENT = empty), pos[cap] u32 (smallest target position holding that key).
//   insert = one 64-bit CAS on the key word (claims an empty slot or finds the key already there)
//   followed by atomicMin on pos (first duplicate wins, query_cpu.cpp:24 insert-if-absent).  No lane
//   ever waits for another lane (a spin on a same-wave lane's store can deadlock under SIMT).
//   The one key equal to the empty sentinel is kept in a dedicated side slot, so there is NO reserved
//   key (the reference reserves 0: hashmap_cuda.cuh:13, query_cpu.cpp:19).
// Three kernels (clear, insert, lookup) on one stream; kernel boundaries order them.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long Q_SENT = 0x8000000000000000ULL;

static inline uint64_t query_cap(int64_t n) {
  uint64_t cap = 64;
  while (cap < (uint64_t)(2 * n + 1)) cap <<= 1;
  return cap;
}

// workspace layout: keys u64[cap] | pos u32[cap] | sent_pos u32 (+pad)
extern "C" size_t link_hash_query_workspace_bytes(int64_t n_target) {
  uint64_t cap = query_cap(n_target < 0 ? 0 : n_target);
  return (size_t)cap * 12 + 64;
}

__global__ void __launch_bounds__(256) k_q_clear(unsigned long long *__restrict__ keys,
                                                 unsigned int *__restrict__ pos, uint64_t cap) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cap) { keys[i] = Q_SENT; pos[i] = 0xFFFFFFFFu; }
  if (i == 0) pos[cap] = 0xFFFFFFFFu;  // side slot for the sentinel-valued key
}

__global__ void __launch_bounds__(256) k_q_insert(const int64_t *__restrict__ target, int64_t n,
                                                  unsigned long long *keys, unsigned int *pos,
                                                  uint64_t mask) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long key = (unsigned long long)target[i];
  if (key == Q_SENT) { atomicMin(&pos[mask + 1], (unsigned int)i); return; }
  uint64_t s = mix64(key) & mask;
  for (;;) {
    unsigned long long old = atomicCAS(&keys[s], Q_SENT, key);
    if (old == Q_SENT || old == key) { atomicMin(&pos[s], (unsigned int)i); return; }
    s = (s + 1) & mask;
  }
}

__global__ void __launch_bounds__(256) k_q_lookup(const int64_t *__restrict__ query, int64_t n1,
                                                  const int64_t *__restrict__ target_idx,
                                                  const unsigned long long *__restrict__ keys,
                                                  const unsigned int *__restrict__ pos, uint64_t mask,
                                                  int64_t *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n1) return;
  unsigned long long key = (unsigned long long)query[i];
  int64_t r = 0;
  if (key == Q_SENT) {
    unsigned int p = pos[mask + 1];
    if (p != 0xFFFFFFFFu) r = target_idx[p] + 1;
  } else {
    uint64_t s = mix64(key) & mask;
    for (;;) {
      unsigned long long k = keys[s];
      if (k == Q_SENT) break;
      if (k == key) { r = target_idx[pos[s]] + 1; break; }
      s = (s + 1) & mask;
    }
  }
  out[i] = r;
}

extern "C" int link_hash_query(const int64_t *query, int64_t n1, const int64_t *target,
                               const int64_t *target_idx, int64_t n, int64_t *out, void *workspace,
                               size_t workspace_bytes, void *stream) {
  if (n1 < 0 || n < 0 || n >= (1LL << 31)) return LINK_ERR_ARG;
  if (n1 == 0) return LINK_OK;
  if (!query || !out || !workspace || (n > 0 && (!target || !target_idx))) return LINK_ERR_ARG;
  uint64_t cap = query_cap(n);
  if (workspace_bytes < link_hash_query_workspace_bytes(n)) return LINK_ERR_WORKSPACE;
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(workspace);
  unsigned int *pos = reinterpret_cast<unsigned int *>(keys + cap);
  hipLaunchKernelGGL(k_q_clear, dim3(blocks_for((int64_t)cap, 256)), dim3(256), 0, S(stream), keys, pos, cap);
  if (n > 0)
    hipLaunchKernelGGL(k_q_insert, dim3(blocks_for(n, 256)), dim3(256), 0, S(stream), target, n, keys, pos,
                       cap - 1);
  hipLaunchKernelGGL(k_q_lookup, dim3(blocks_for(n1, 256)), dim3(256), 0, S(stream), query, n1,
                     target_idx, keys, pos, cap - 1, out);
  return check_launch("link_hash_query");
}

// ---------------------------------------------------------------------------------------------
// count
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_count(const int32_t *__restrict__ idx, int64_t n,
                                               int32_t *out, int64_t s) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    int32_t v = idx[i];
    if (v >= 0 && v < s) atomicAdd(&out[v], 1);
  }
}

extern "C" int link_count(const int32_t *idx, int64_t n, int32_t *out, int64_t s, void *stream) {
  if (n < 0 || s < 0) return LINK_ERR_ARG;
  if (s == 0) return LINK_OK;
  if (!out || (n > 0 && !idx)) return LINK_ERR_ARG;
  if (hipMemsetAsync(out, 0, (size_t)s * sizeof(int32_t), S(stream)) != hipSuccess)
    return check_launch("link_count memset");
  if (n > 0)
    hipLaunchKernelGGL(k_count, dim3(blocks_for(n, 256)), dim3(256), 0, S(stream), idx, n, out, s);
  return check_launch("link_count");
}

// ---------------------------------------------------------------------------------------------
// trilinear weights of the 8 corner voxels of a point (calc_ti_weights, nn/functional/devoxelize.py:10-48): one thread
// per point; the reference composes ~30 elementwise launches over [8, P] temporaries
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_ti_weights(const float *__restrict__ pts, const int64_t *__restrict__ idx, int64_t p,
                                                    float scale, float inv_div, bool scaled, float *__restrict__ w) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= p) return;
  const float4 q = *reinterpret_cast<const float4 *>(pts + 4 * i);
  const float c[3] = {q.x, q.y, q.z};
  float near_[3], far_[3];
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float lo = scaled ? __fmul_rn(floorf(__fdiv_rn(c[a], scale)), scale) : floorf(c[a]);
    const float hi = __fadd_rn(lo, scale);
    near_[a] = __fsub_rn(hi, c[a]);                    // weight of the low corner on this axis
    far_[a] = __fsub_rn(c[a], lo);
  }
  float v[8], sum = 0.f;
#pragma unroll
  for (int k = 0; k < 8; k++) {                        // corner k = 4 dx + 2 dy + dz: the order of get_kernel_offsets(2)
    float t = __fmul_rn(__fmul_rn((k & 4) ? far_[0] : near_[0], (k & 2) ? far_[1] : near_[1]), (k & 1) ? far_[2] : near_[2]);
    if (scaled) t = __fdiv_rn(t, inv_div);
    if (idx[(int64_t)k * p + i] == -1) t = 0.f;
    v[k] = t;
    sum = __fadd_rn(sum, t);
  }
  const float den = __fadd_rn(sum, 1e-8f);
#pragma unroll
  for (int k = 0; k < 8; k++) w[(int64_t)k * p + i] = __fdiv_rn(v[k], den);
}

extern "C" int link_ti_weights(const float *pts, const int64_t *idx_query, int64_t p, float scale, float *w, void *stream) {
  if (p < 0 || !(scale > 0.f)) return LINK_ERR_ARG;
  if (p == 0) return LINK_OK;
  if (!pts || !idx_query || !w) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_ti_weights, dim3(blocks_for(p, 256)), dim3(256), 0, S(stream), pts, idx_query, p, scale,
                     (float)((double)scale * scale * scale), scale != 1.0f, w);
  return check_launch("link_ti_weights");
}

// ---------------------------------------------------------------------------------------------
// voxelize forward (generic idx): one wave per input row, lanes stride the channels -> coalesced
// row reads; fp32 atomics into the output row like the reference kernel.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_voxelize_fwd(const float *__restrict__ in,
                                                      const int32_t *__restrict__ idx,
                                                      const int32_t *__restrict__ counts, int64_t n,
                                                      int c, int64_t n1, float *out) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  int32_t pos = idx[row];
  if (pos < 0 || pos >= n1) return;
  int32_t cn = counts[pos];
  if (cn == 0) return;
  float fc = (float)cn;
  const float *src = in + row * (int64_t)c;
  float *dst = out + (int64_t)pos * c;
  for (int j = lane; j < c; j += 64) atomicAdd(&dst[j], src[j] / fc);
}

extern "C" int link_voxelize_forward(const float *in, const int32_t *idx, const int32_t *counts,
                                     int64_t n, int64_t c, int64_t n1, float *out, void *stream) {
  if (n < 0 || c < 0 || n1 < 0 || c > (1 << 20)) return LINK_ERR_ARG;
  if (n1 == 0 || c == 0) return LINK_OK;
  if (!out || (n > 0 && (!in || !idx || !counts))) return LINK_ERR_ARG;
  if (hipMemsetAsync(out, 0, (size_t)(n1 * c) * sizeof(float), S(stream)) != hipSuccess)
    return check_launch("link_voxelize_forward memset");
  if (n > 0)
    hipLaunchKernelGGL(k_voxelize_fwd, dim3(blocks_for(n * 64, 256)), dim3(256), 0, S(stream), in,
                       idx, counts, n, (int)c, n1, out);
  return check_launch("link_voxelize_forward");
}

// voxelize backward: pure row gather with a per-row scale.
__global__ void __launch_bounds__(256) k_voxelize_bwd(const float *__restrict__ top,
                                                      const int32_t *__restrict__ idx,
                                                      const int32_t *__restrict__ counts, int64_t n,
                                                      int c, float *__restrict__ bottom) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= n) return;
  int32_t pos = idx[row];
  float *dst = bottom + row * (int64_t)c;
  int32_t cn = (pos >= 0) ? counts[pos] : 0;
  if (cn == 0) {
    for (int j = lane; j < c; j += 64) dst[j] = 0.f;
    return;
  }
  float fc = (float)cn;
  const float *src = top + (int64_t)pos * c;
  for (int j = lane; j < c; j += 64) dst[j] = src[j] / fc;
}

extern "C" int link_voxelize_backward(const float *top, const int32_t *idx, const int32_t *counts,
                                      int64_t n, int64_t c, float *bottom, void *stream) {
  if (n < 0 || c < 0 || c > (1 << 20)) return LINK_ERR_ARG;
  if (n == 0 || c == 0) return LINK_OK;
  if (!top || !idx || !counts || !bottom) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_voxelize_bwd, dim3(blocks_for(n * 64, 256)), dim3(256), 0, S(stream), top, idx,
                     counts, n, (int)c, bottom);
  return check_launch("link_voxelize_backward");
}

// ---------------------------------------------------------------------------------------------
// devoxelize forward: one wave per query row; the K indices/weights are loaded once by lanes 0..K-1
// and broadcast with readlane; accumulate in registers in k order (deterministic), one store.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_devox_fwd(const float *__restrict__ feat,
                                                   const int32_t *__restrict__ ind,
                                                   const float *__restrict__ w, int64_t nq, int c,
                                                   int K, float *__restrict__ out) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= nq) return;
  for (int j0 = 0; j0 < c; j0 += 64) {
    int j = j0 + lane;
    float acc = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
      int kk = k0 + lane;
      int32_t my_i = (kk < K) ? ind[row * K + kk] : -1;
      float my_w = (kk < K) ? w[row * K + kk] : 0.f;
      int lim = (K - k0 < 64) ? (K - k0) : 64;
      for (int t = 0; t < lim; t++) {
        int32_t q = __shfl(my_i, t, 64);
        float wk = __shfl(my_w, t, 64);
        float cur = 0.f;
        if (q >= 0 && j < c) cur = feat[(int64_t)q * c + j];
        acc += wk * cur;
      }
    }
    if (j < c) out[row * (int64_t)c + j] = acc;
  }
}

extern "C" int link_devoxelize_forward(const float *feat, const int32_t *ind, const float *w,
                                       int64_t nq, int64_t c, int64_t k, float *out, void *stream) {
  if (nq < 0 || c < 0 || k < 0 || c > (1 << 20) || k > (1 << 20)) return LINK_ERR_ARG;
  if (nq == 0 || c == 0) return LINK_OK;
  if (!out || (k > 0 && (!feat || !ind || !w))) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_devox_fwd, dim3(blocks_for(nq * 64, 256)), dim3(256), 0, S(stream), feat, ind, w,
                     nq, (int)c, (int)k, out);
  return check_launch("link_devoxelize_forward");
}

__global__ void __launch_bounds__(256) k_devox_bwd(const float *__restrict__ top,
                                                   const int32_t *__restrict__ ind,
                                                   const float *__restrict__ w, int64_t nq, int c,
                                                   int K, float *bottom) {
  int64_t row = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  int lane = threadIdx.x & 63;
  if (row >= nq) return;
  for (int k = 0; k < K; k++) {
    int32_t q = ind[row * K + k];
    if (q < 0) continue;
    float wk = w[row * K + k];
    for (int j = lane; j < c; j += 64) atomicAdd(&bottom[(int64_t)q * c + j], wk * top[row * (int64_t)c + j]);
  }
}

extern "C" int link_devoxelize_backward(const float *top, const int32_t *ind, const float *w,
                                        int64_t nq, int64_t n, int64_t c, int64_t k, float *bottom,
                                        void *stream) {
  if (nq < 0 || n < 0 || c < 0 || k < 0 || c > (1 << 20)) return LINK_ERR_ARG;
  if (n == 0 || c == 0) return LINK_OK;
  if (!bottom) return LINK_ERR_ARG;
  if (hipMemsetAsync(bottom, 0, (size_t)(n * c) * sizeof(float), S(stream)) != hipSuccess)
    return check_launch("link_devoxelize_backward memset");
  if (nq > 0 && k > 0) {
    if (!top || !ind || !w) return LINK_ERR_ARG;
    hipLaunchKernelGGL(k_devox_bwd, dim3(blocks_for(nq * 64, 256)), dim3(256), 0, S(stream), top, ind, w,
                       nq, (int)c, (int)k, bottom);
  }
  return check_launch("link_devoxelize_backward");
}
