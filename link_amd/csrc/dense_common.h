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
// Per-wave phase timers (link_dc_tuning_t::k1_dbg / k2_dbg -> tools/k1prof.py, k1mprof.py, k2prof.py) are compiled
// in only with -DDC_PROF=1 (python tools/mkvariant.py PROF "-DDC_PROF=1").  In the default build the pointer is forced to null at
// the top of every kernel body, so the `if (dbg)` tests -- a scalar test + branch each, five to eight per plane step / tile in
// every wave: ~0.3 M scalar instructions per frame on cfg2 -- and the timers fold away (round 5).
#ifndef DC_PROF
#define DC_PROF 0
#endif
#define DC_PROF_PTR(p) do { if (!DC_PROF) (p) = nullptr; } while (0)
// the timers' clock: s_memtime (shader-clock ticks: fine phase resolution, but every XCD counts on its own -- start stamps of waves
// on different XCDs cannot be compared) or, with -DDC_PROF=2, s_memrealtime (the 100 MHz reference counter all XCDs share: 10 ns
// steps -- what tools/slot_timeline.py needs to lay the waves of a launch out on ONE time axis; round 6)
#define DC_NOW() (DC_PROF == 2 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime())
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
// Hand-offs INSIDE a launch (round 6, the persistent batch kernels of dense_batch.hip): a table another workgroup reads before the
// launch ends is stored write-through (aux 16 = sc1: the bytes leave the XCD's L2, so the publishing wave needs only `s_waitcnt
// vmcnt(0)` before its arrival atomic, no `buffer_wbl2`) and, where the reader does not run an agent-scope acquire, loaded with
// sc1 as well (served by the L2, never by the CU's L1) -- MI355X_MICROARCH.md "inter-workgroup visibility".  COH = false: the plain
// forms the stand-alone kernels have always used (a kernel boundary orders them).
template <bool COH>
__device__ __forceinline__ void st16i_c(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, int4 v) {
  const v4i_t x = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(x, r, byte_off, 0, COH ? 16 : 0);
}
template <bool COH>
__device__ __forceinline__ void st4i_c(__amdgpu_buffer_rsrc_t r, uint32_t byte_off, int v) {
  __builtin_amdgcn_raw_buffer_store_b32(v, r, byte_off, 0, COH ? 16 : 0);
}
template <bool COH>
__device__ __forceinline__ int4 ld16i_c(__amdgpu_buffer_rsrc_t r, const int4 *base, uint32_t idx) {      // record `idx` of the table
  if constexpr (COH) {
    const v4i_t x = __builtin_amdgcn_raw_buffer_load_b128(r, idx * 16u, 0, 16);
    return make_int4(x.x, x.y, x.z, x.w);
  } else {
    return base[idx];
  }
}
template <bool COH>
__device__ __forceinline__ int ld4i_c(__amdgpu_buffer_rsrc_t r, const uint32_t *base, uint32_t idx) {
  if constexpr (COH) return __builtin_amdgcn_raw_buffer_load_b32(r, idx * 4u, 0, 16);
  else return (int)base[idx];
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

// Largest |coordinate| (after the optional division) a voxel INSIDE the plan's bounds can have, per axis: the grid covers
// blocks lo .. lo + dim - 1 of edge s, and a voxel outside it never reaches the fused kernels (the slot insert drops it and
// raises status bit 0).  With the theta weights this bounds |theta| for the whole launch, so "does any argument leave the fast
// sincos range" is decided ONCE per kernel from (grid, weights) instead of per tile / per plane round from the thetas
// (DC_THETA_BOUND; round 5: 8 compares + a vote less per round -- and, unlike the vote, independent of what the records hold).
#ifndef DC_THETA_BOUND
#define DC_THETA_BOUND 1
#endif
__device__ __forceinline__ void dc_coord_absmax(const link_dc_grid_t &g, float coord_div, float &ax, float &ay, float &az) {
  float a[3];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float lo = (float)g.lo[k] * (float)g.s, hi = ((float)g.lo[k] + (float)g.dim[k]) * (float)g.s;
    a[k] = fmaxf(fabsf(lo), fabsf(hi)) / coord_div;
  }
  ax = a[0]; ay = a[1]; az = a[2];
}
// |theta| bound of one channel (1 % slack for the roundings of the bound itself); NaN / inf weights give "not small"
__device__ __forceinline__ bool dc_theta_leaves_fast_range(float ax, float ay, float az, float w0, float w1, float w2, float alpha) {
  const float b = (fabsf(w0) * ax + fabsf(w1) * ay + fabsf(w2) * az) * fabsf(alpha) * 1.01f;
  return !(b < 32768.0f);
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

// The slot insert of the dense-cell index as a device function of (workgroup number, workgroups, threads per workgroup):
// k_dc_index (dense_fused.hip) runs it over its own grid (the persistent batch insert, dense_batch.hip, restates it on 32-bit voxel
// numbers).  rank = cnt[cell]++, slots[cell][rank] = (x, y, z, id), vcell[id] = cell; a voxel outside the grid or past a
// full cell sets its bit of the status accumulator and is left out.
template <bool STATS, bool COH = false>
__device__ __forceinline__ void dc_index_body(const int4 *__restrict__ coords, int64_t n, const link_dc_grid_t &g,
                                              uint32_t *__restrict__ cnt, int4 *__restrict__ slots,
                                              int32_t *__restrict__ vcell, int32_t *__restrict__ hdr, int bid, int nblk,
                                              int nthr, int &st_in, int &st_first, int &st_max) {
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  if (bid == 0 && threadIdx.x == 0) hdr[LINK_HDR_NVALID] = (int32_t)n;
  for (int64_t v = (int64_t)bid * nthr + threadIdx.x; v < n; v += (int64_t)nblk * nthr) {
    const int4 rc = coords[v];
    const unsigned ux = (unsigned)(floordiv(rc.x, g.s) - g.lo[0]), uy = (unsigned)(floordiv(rc.y, g.s) - g.lo[1]);
    const unsigned uz = (unsigned)(floordiv(rc.z, g.s) - g.lo[2]), ub = (unsigned)(rc.w - g.lo[3]);
    const bool inside = ux < (unsigned)g.dim[0] && uy < (unsigned)g.dim[1] && uz < (unsigned)g.dim[2] &&
                        ub < (unsigned)g.dim[3];
    if (!inside) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 1);
    const int pcell = inside ? dc_cell(g, (int)ux, (int)uy, (int)uz, (int)ub) : 0;
    const int rank = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_cnt, pcell ? (uint32_t)pcell * 4u : DC_OOB, 0, 0);
    const bool full = pcell != 0 && rank >= g.k;
    if (full) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 2);
    const bool keep = pcell != 0 && !full;
    st16i_c<COH>(r_slots, keep ? dc_slot(g, pcell, rank) * 16u : DC_OOB, make_int4(rc.x, rc.y, rc.z, (int)v));
    vcell[v] = keep ? pcell : 0;
    if (STATS) { st_in += pcell != 0; st_first += (pcell != 0 && rank == 0); st_max = max(st_max, pcell != 0 ? rank + 1 : 0); }
  }
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
