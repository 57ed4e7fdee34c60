// link_amd/csrc/dense_common.h -- helpers shared by the dense-cell kernels (dense.hip, dense_fused.hip).
#pragma once
#include <limits.h>

#include "elk_common.h"

namespace link {

typedef int v4i_t __attribute__((ext_vector_type(4)));

// All table stores go through buffer descriptors: write-through (sc1: the lines leave the XCD's L2 while the
// kernel runs; the consumer kernel runs on other XCDs anyway), and -- the point -- a lane that must NOT
// store passes an out-of-range offset, which the hardware drops.  No branch around any store, so every
// s_waitcnt vmcnt the compiler emits is an exact count instead of the vmcnt(0) a conditional VMEM op forces.
// Tables handled here are < 4 GiB (checked by the launchers).
#define DC_OOB 0xFFFFFFF0u
#ifndef DC_ST_AUX
#define DC_ST_AUX 16      /* sc1: write-through; 0 = ordinary write-back stores (A/B: LINK_AMD_CXXFLAGS=-DDC_ST_AUX=0) */
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dc_rsrc(const void *base, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ void st16(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, float4 v) {
  const v4i_t x = {__builtin_bit_cast(int, v.x), __builtin_bit_cast(int, v.y), __builtin_bit_cast(int, v.z),
                   __builtin_bit_cast(int, v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, byte_off, 0, DC_ST_AUX);
}
__device__ __forceinline__ void st16i(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, int4 v) {
  const v4i_t x = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, byte_off, 0, 0);
}
__device__ __forceinline__ void st4i(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, int v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, byte_off, 0, 0);
}

// Occupancy statistics of an insert pass (the STATS variants of the index kernels).  Every lane keeps its own three numbers
// over the kernel's loop; at the end the workgroup reduces them (wave butterfly, four waves through LDS) and one thread adds
// them to one of 16 slots, each on its own 64-byte line: stats i32[16][16], slot [k][0] += voxels inside the grid,
// [k][1] += cells that got their first voxel, [k][2] = max(count of a cell); the reader sums / maxes the 16 slots.  (One
// atomic per wave and pass on a single line made the kernel 59 us instead of 8: same-address atomics serialise in the L2.)
#define DC_STATS_SLOTS 16
__device__ __forceinline__ void dc_index_stats_flush(int32_t *stats, int n_in, int n_first, int mx) {
  __shared__ int s_part[16][3];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n_in += __shfl_xor(n_in, o, 64);
    n_first += __shfl_xor(n_first, o, 64);
    mx = max(mx, __shfl_xor(mx, o, 64));
  }
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  if (lane == 0) { s_part[wave][0] = n_in; s_part[wave][1] = n_first; s_part[wave][2] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    int a = 0, f = 0, m = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) { a += s_part[w][0]; f += s_part[w][1]; m = max(m, s_part[w][2]); }
    int32_t *slot = stats + (blockIdx.x % DC_STATS_SLOTS) * 16;
    if (a) atomicAdd(&slot[0], a);
    if (f) atomicAdd(&slot[1], f);
    if (m) atomicMax(&slot[2], m);
  }
}

__device__ __forceinline__ int dc_cell(const link_dc_grid_t &g, int ux, int uy, int uz, int ub) {
  return ((ub * g.pdim[0] + ux + 1) * g.pdim[1] + uy + 1) * g.pdim[2] + uz + 1;
}

// Slot lists.  `slots` holds vp*k records (x,y,z,id): first the INLINE region, DC_INL records per cell
// (64 contiguous bytes: what the per-cell kernels stream), then the overflow region with the remaining
// k - DC_INL records per cell (touched only by cells with more than DC_INL voxels).
#define DC_INL 4
__device__ __forceinline__ uint32_t dc_slot(const link_dc_grid_t &g, int pcell, int rank) {     // record index
  return rank < DC_INL ? (uint32_t)pcell * DC_INL + (uint32_t)rank
                       : (uint32_t)g.vp * DC_INL + (uint32_t)pcell * (uint32_t)(g.k - DC_INL) + (uint32_t)(rank - DC_INL);
}

// Id lists of the tile form (link_dc_buffers_t::sid): DC_SID_INL voxel ids inline per cell (32 contiguous bytes: what a
// lane of the fused pre_mix kernel reads for its cell), then the overflow region with k - DC_SID_INL ids per cell.
#define DC_SID_INL 8
__device__ __forceinline__ uint32_t dc_sid_off(const link_dc_grid_t &g, int pcell, int rank) {      // u32 index
  const int kk = g.k > DC_SID_INL ? g.k - DC_SID_INL : 0;
  return rank < DC_SID_INL ? (uint32_t)pcell * DC_SID_INL + (uint32_t)rank
                           : (uint32_t)g.vp * DC_SID_INL + (uint32_t)pcell * (uint32_t)kk + (uint32_t)(rank - DC_SID_INL);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

static inline bool dc_width_ok(int c) { return c == 16 || c == 32 || c == 64 || c == 128; }

}  // namespace link
