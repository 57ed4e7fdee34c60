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
// The store is the buffer-store builtin with aux = sc1 (as in the dense-cell kernels, dense_common.h): visible to the
// compiler's hazard and vmcnt bookkeeping.  It addresses base + 32-bit byte offset, so the launchers ask for it only for
// tables below 4 GiB (link::wt_ok); larger ones take ordinary stores.
typedef int v4i_wt_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_out(float *base, int64_t elem, float4 v, bool wt) {
  if (wt) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0xFFFFFFFFu, 0x00020000);
    const v4i_wt_t x = {__builtin_bit_cast(int, v.x), __builtin_bit_cast(int, v.y), __builtin_bit_cast(int, v.z),
                        __builtin_bit_cast(int, v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(x, r, (uint32_t)(elem * 4), 0, 16);
  } else {
    *reinterpret_cast<float4 *>(base + elem) = v;
  }
}
static inline bool wt_ok(int64_t rows, int64_t row_floats) { return rows * row_floats * 4 < (1LL << 32); }

// a * b as an fp32 VALUE (one rounding), whatever follows.  hipcc compiles with -ffp-contract=fast and __fmul_rn is a plain
// multiply to it, so `x - __fmul_rn(a, b)` becomes ONE fma with the product unrounded.  That matters where the reference forms a
// product, ROUNDS it, and uses it twice: the cos_x linear part fin * theta (linkunet.py:164-176) enters the block sums as a rounded
// product and is subtracted again as "v - fin * theta" -- for a voxel that is alone in its neighbourhood the two cancel EXACTLY in
// the reference, while an fma on one side leaves the product's rounding error, ~6e-5 at theta ~ 1500, in the row (round 5,
// tools/core_dbg.py: 1.5e-5 of max|out| on the sparse S-kitti stages against 4.7e-7 with the rounded product).  The empty asm
// makes the product exist in a register before anything consumes it.
__device__ __forceinline__ float link_mul_rn(float a, float b) {
  float p = a * b;
  asm("" : "+v"(p));
  return p;
}

}  // namespace link
