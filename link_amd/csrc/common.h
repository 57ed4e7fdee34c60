// link_amd/csrc/common.h -- shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "link_amd.h"

namespace link {

constexpr int WAVE = 64;  // CDNA wavefront width (hard-coded per the gfx950 guide)

void set_error(const char *what, hipError_t e);

inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error(what, e);
    return LINK_ERR_LAUNCH;
  }
  return LINK_OK;
}

inline hipStream_t S(void *stream) { return reinterpret_cast<hipStream_t>(stream); }

inline unsigned blocks_for(int64_t n, int threads) {
  int64_t b = (n + threads - 1) / threads;
  return (unsigned)(b < 1 ? 1 : b);
}

// FNV-1a over the four 32-bit words of a row, folded to 60 bits
// (reference: backend/hash/hash_cuda.cu:10-23).
__device__ __forceinline__ int64_t fnv4(int32_t x, int32_t y, int32_t z, int32_t b) {
  uint64_t h = 14695981039346656037ULL;
  h ^= (uint32_t)x; h *= 1099511628211ULL;
  h ^= (uint32_t)y; h *= 1099511628211ULL;
  h ^= (uint32_t)z; h *= 1099511628211ULL;
  h ^= (uint32_t)b; h *= 1099511628211ULL;
  h = (h >> 60) ^ (h & 0x0FFFFFFFFFFFFFFFULL);
  return (int64_t)h;
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}

// floor division (torch.div(..., rounding_mode='floor') on int32; utils.py:45)
__device__ __host__ __forceinline__ int32_t floordiv(int32_t a, int32_t b) {
  int32_t q = a / b, r = a % b;
  return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

// cell index of a block coordinate in the dense grid, or -1 when outside
__device__ __forceinline__ int32_t cell_of(const link_grid_t &g, int32_t bx, int32_t by, int32_t bz,
                                           int32_t bb) {
  uint32_t ux = (uint32_t)(bx - g.lo[0]), uy = (uint32_t)(by - g.lo[1]);
  uint32_t uz = (uint32_t)(bz - g.lo[2]), ub = (uint32_t)(bb - g.lo[3]);
  if (ux >= (uint32_t)g.dim[0] || uy >= (uint32_t)g.dim[1] || uz >= (uint32_t)g.dim[2] ||
      ub >= (uint32_t)g.dim[3])
    return -1;
  return (int32_t)(((ux * (uint32_t)g.dim[1] + uy) * (uint32_t)g.dim[2] + uz) * (uint32_t)g.dim[3] + ub);
}

// wave-wide sum via DPP-friendly xor shuffles (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over an aligned group of G lanes (G power of two <= 64)
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Output stores of the streaming kernels: write-through (sc1) so the lines leave the XCD's L2 while the
// kernel is still running instead of waiting, dirty, for the write-back at the kernel boundary
// (MI355X_MICROARCH.md: boundary "+ B / 6 TB/s when the predecessor leaves B bytes dirty"; "publish-large":
// plain stores + flush 8.2 us vs sc1 write-through 3.0 us).  The next kernel runs on other XCDs anyway.
// Inline asm stores are invisible to the compiler's vmcnt bookkeeping; that only makes its later
// waits conservative (vmcnt retires in order), never too weak.  g_wt_stores toggles it for A/B tests.
typedef float v4f_t __attribute__((ext_vector_type(4)));
// The asm form below was validated (repeat-bitwise tests + stale-cache tests) with the toolchain of ROCm 7.2
// (clang 22).  It carries a hand-placed hazard pad the compiler cannot see, so any OTHER compiler major falls
// back to an ordinary store until it has been re-validated -- the dense-cell kernels (dense_common.h) already
// use the compiler-visible form, __builtin_amdgcn_raw_buffer_store_b128 with aux = sc1.
#define LINK_STORE_WT_VALIDATED_CLANG 22
__device__ __forceinline__ void store_wt(float4 *p, float4 v) {
#if defined(__clang_major__) && __clang_major__ == LINK_STORE_WT_VALIDATED_CLANG
  const v4f_t x = {v.x, v.y, v.z, v.w};
  // s_nop 1: a VMEM store of more than 8 bytes needs wait states before its data VGPRs are overwritten;
  // hipcc pads that hazard for its own instructions but cannot see inside an asm statement.
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(x) : "memory");
#else
  *p = v;
#endif
}
__device__ __forceinline__ void store_out(float4 *p, float4 v, bool wt) {
  if (wt) store_wt(p, v);
  else *p = v;
}

}  // namespace link
