// link_amd/csrc/dense_fused.hip -- the fused kernels of the dense-cell layout (include/link_amd.h section E).
//
// Why fuse: counters on cfg2 (profiles/r02_*) show pre_mix bound by the f32 MFMA pipe with the VALU idle, and
// the per-cell modulate+sum bound by VALU issue with the MFMA pipe idle -- and `fin` making a 51 MB round trip
// between them.  k_dc_premix_modsum does both on the same 16-voxel tile: MFMA of one wave runs beside the
// sincos/modulate VALU work of the other wave on its SIMD, `fin` never leaves registers, and the per-cell
// sums are formed from an LDS image of the tile.
//
//   k_dc_index            coords -> cell, rank = cnt[cell]++, slots[cell][rank] = (x,y,z,id); vcell[i] = cell
//   k_dc_premix_modsum    a wave owns a range of cells: counts -> wave prefix scan -> flat, id-ordered voxel
//                         list in LDS -> tiles of 16 voxels: gather F rows, LayerNorm(F Wpre^T) on MFMA, theta /
//                         sincos / modulate in the MFMA layout (a lane holds 16 channels of ONE voxel: no lane
//                         is ever idle, whatever the cell sizes), X tile -> LDS -> per-cell sums -> S rows
//   k_dc_demod            per voxel pair, original order, persistent + software-pipelined: A[cell] row,
//                         de-modulate, LayerNorm, store
#define DC_IO 0
#define DC_IO_NS dcio_f32
#include "dense_fused_impl.h"

using namespace link;

namespace link {
// tile form of the fused pre_mix kernel (dense_tiles.hip)
int dc_tiles_modsum(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d, int64_t n, bool warm, hipStream_t st);
}

// the fp16 / bf16 instantiations live in their own translation units (dense_fused_f16.hip, dense_fused_bf16.hip)
#define DC_DECL_IO(NS)                                                                                                    \
  namespace NS {                                                                                                           \
  int run_premix_modsum(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t, bool,        \
                        hipStream_t);                                                                                      \
  int run_demod(const float *, const float *, const int32_t *, const int32_t *, const float *, const float *,             \
                const float *, const float *, const link_elk_desc_t &, const link_dc_grid_t &, int64_t, void *,            \
                hipStream_t);                                                                                              \
  int run_gather_demod(const link_dc_buffers_t *, const link_dc_grid_t &, const link_elk_desc_t &, int64_t, hipStream_t); \
  }
DC_DECL_IO(dcio_f16)
DC_DECL_IO(dcio_bf16)
#undef DC_DECL_IO

// ---------------------------------------------------------------------------------------------
// index: slot insert
// ---------------------------------------------------------------------------------------------
template <bool STATS>
#ifndef DC_INDEX_THREADS
#define DC_INDEX_THREADS 256     /* 128 / 512 / 1024: no difference (A/B on one box: 37.0-38.2 us/frame, index 10.0-10.5 us with events) */
#endif
__global__ void __launch_bounds__(DC_INDEX_THREADS) k_dc_index(const int4 *__restrict__ coords, int64_t n, link_dc_grid_t g,
                                                  uint32_t *__restrict__ cnt, int4 *__restrict__ slots,
                                                  int32_t *__restrict__ vcell, int32_t *__restrict__ hdr,
                                                  int32_t *__restrict__ stats) {
  int st_in = 0, st_first = 0, st_max = 0;
  dc_index_body<STATS>(coords, n, g, cnt, slots, vcell, hdr, (int)blockIdx.x, (int)gridDim.x, DC_INDEX_THREADS, st_in, st_first, st_max);
  if (STATS) dc_index_stats_flush(stats, st_in, st_first, st_max);
}

extern "C" int link_dc_index(const int32_t *coords, int64_t n, const link_dc_grid_t *g, uint32_t *cnt,
                             int32_t *slots, int32_t *vcell, int32_t *hdr, void *stream) {
  if (n < 0 || !g) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!coords || !cnt || !slots || !vcell || !hdr) return LINK_ERR_ARG;
  if (g->k < DC_INL || g->vp * (int64_t)g->k * 16 >= (1LL << 32) || n >= (1LL << 29)) return LINK_ERR_ARG;
  int64_t wgs = (n + DC_INDEX_THREADS - 1) / DC_INDEX_THREADS;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index<false>, dim3((unsigned)wgs), dim3(DC_INDEX_THREADS), 0, S(stream), reinterpret_cast<const int4 *>(coords), n,
                     *g, cnt, reinterpret_cast<int4 *>(slots), vcell, hdr, (int32_t *)nullptr);
  return check_launch("link_dc_index");
}

// ---------------------------------------------------------------------------------------------
// slot insert + occupancy counters + bounding box, published to HOST memory by the last workgroup (section G)
// ---------------------------------------------------------------------------------------------
// The block driver's first launch (csrc/block.hip): what link_coords_bbox + link_dc_index_probe + a device-to-host copy did in
// three launches and a stream synchronisation.  `cur` i32[272]: the top-level ticket at 8, 16 result slots of 16 words at 16
// (counters, arrivals, coordinate minima / maxima: dc_probe_scratch_word) -- every workgroup adds its partial results to one slot,
// and the LAST one to finish (two-level ticket) reduces the slots, writes bbox -> host[0..8) and (voxels inside, occupied cells,
// fullest cell) -> host[16..19) (mapped, coherent memory) and, behind a system-scope fence, `seq` into host[8]: the host polls that word instead of synchronising the stream.  Workgroup 0 also lays out `next` (the scratch
// of the following call) so that no initialiser launch is needed.
__device__ __host__ inline int dc_probe_scratch_word(int i) {        // initial value of word i of the 272-word scratch
  const int j = i >= 16 ? (i - 16) & 15 : -1;
  return (j >= 4 && j < 8) ? INT_MAX : ((j >= 8 && j < 12) ? INT_MIN : 0);
}
__global__ void k_dc_probe_scratch_init(int32_t *w) { w[threadIdx.x] = dc_probe_scratch_word((int)threadIdx.x); }
__global__ void __launch_bounds__(256) k_dc_index_probe_bbox(const int4 *__restrict__ coords, int64_t n, link_dc_grid_t g,
                                                             uint32_t *__restrict__ cnt, int4 *__restrict__ slots,
                                                             int32_t *__restrict__ vcell, int32_t *__restrict__ hdr,
                                                             int32_t *__restrict__ cur, int32_t *__restrict__ next,
                                                             int32_t *__restrict__ host, int seq) {
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  if (blockIdx.x == 0) {
    if (threadIdx.x == 0) hdr[LINK_HDR_NVALID] = (int32_t)n;
    for (int i = threadIdx.x; i < 272; i += 256) next[i] = dc_probe_scratch_word(i);
  }
  int st_in = 0, st_first = 0, st_max = 0;
  int mn[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX}, mx[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
    const int4 rc = coords[v];
    mn[0] = min(mn[0], rc.x); mx[0] = max(mx[0], rc.x); mn[1] = min(mn[1], rc.y); mx[1] = max(mx[1], rc.y);
    mn[2] = min(mn[2], rc.z); mx[2] = max(mx[2], rc.z); mn[3] = min(mn[3], rc.w); mx[3] = max(mx[3], rc.w);
    const unsigned ux = (unsigned)(floordiv(rc.x, g.s) - g.lo[0]), uy = (unsigned)(floordiv(rc.y, g.s) - g.lo[1]);
    const unsigned uz = (unsigned)(floordiv(rc.z, g.s) - g.lo[2]), ub = (unsigned)(rc.w - g.lo[3]);
    const bool inside = ux < (unsigned)g.dim[0] && uy < (unsigned)g.dim[1] && uz < (unsigned)g.dim[2] && ub < (unsigned)g.dim[3];
    if (!inside) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 1);
    const int pcell = inside ? dc_cell(g, (int)ux, (int)uy, (int)uz, (int)ub) : 0;
    const int rank = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_cnt, pcell ? (uint32_t)pcell * 4u : DC_OOB, 0, 0);
    const bool full = pcell != 0 && rank >= g.k;
    if (full) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 2);
    const bool keep = pcell != 0 && !full;
    st16i(r_slots, keep ? dc_slot(g, pcell, rank) * 16u : DC_OOB, make_int4(rc.x, rc.y, rc.z, (int)v));
    vcell[v] = keep ? pcell : 0;
    st_in += pcell != 0; st_first += (pcell != 0 && rank == 0); st_max = max(st_max, pcell != 0 ? rank + 1 : 0);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    st_in += __shfl_xor(st_in, o, 64); st_first += __shfl_xor(st_first, o, 64); st_max = max(st_max, __shfl_xor(st_max, o, 64));
#pragma unroll
    for (int a = 0; a < 4; a++) { mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64)); mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64)); }
  }
  __shared__ int s_part[4][11];
  __shared__ int s_last;
  const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
  if (lane == 0) {
    s_part[wave][0] = st_in; s_part[wave][1] = st_first; s_part[wave][2] = st_max;
    for (int a = 0; a < 4; a++) { s_part[wave][3 + a] = mn[a]; s_part[wave][7 + a] = mx[a]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // one of 16 result slots per workgroup, each on its own 64-byte line: [0] voxels inside, [1] occupied cells, [2] fullest cell,
    // [3] arrivals, [4..8) minima, [8..12) maxima (same-address atomics serialise in the L2: 391 workgroups on one line made this
    // kernel 64 us instead of 9).  All of them are RETURNING atomics whose results feed the ticket below, so the slot is complete
    // when the ticket is drawn -- no device-wide fence (an L2 write-back per workgroup) is needed for words only atomics touch.
    int a_ = 0, f_ = 0, m_ = 0;
    for (int w = 0; w < 4; w++) { a_ += s_part[w][0]; f_ += s_part[w][1]; m_ = max(m_, s_part[w][2]); }
    int32_t *slot = cur + 16 + (blockIdx.x % DC_STATS_SLOTS) * 16;
    int dep = atomicAdd(&slot[0], a_) & 0;
    dep |= atomicAdd(&slot[1], f_) & 0;
    dep |= atomicMax(&slot[2], m_) & 0;
    for (int a = 0; a < 4; a++) {
      dep |= atomicMin(&slot[4 + a], min(min(s_part[0][3 + a], s_part[1][3 + a]), min(s_part[2][3 + a], s_part[3][3 + a]))) & 0;
      dep |= atomicMax(&slot[8 + a], max(max(s_part[0][7 + a], s_part[1][7 + a]), max(s_part[2][7 + a], s_part[3][7 + a]))) & 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // two-level ticket: the last arrival of a slot (its share of the workgroups) draws the top-level ticket
    const int share = ((int)gridDim.x - 1 - (int)(blockIdx.x % DC_STATS_SLOTS)) / DC_STATS_SLOTS + 1;
    int last = 0;
    if (atomicAdd(&slot[3], 1 + dep) == share - 1) {
      const int live = (int)gridDim.x < DC_STATS_SLOTS ? (int)gridDim.x : DC_STATS_SLOTS;
      last = atomicAdd(&cur[8], 1) == live - 1;
    }
    s_last = last;
  }
  __syncthreads();
  if (!s_last) return;
  // the last workgroup: reduce the 16 slots and publish to the host
  if (threadIdx.x < 64) {
    const int k = (int)threadIdx.x & 15;
    int32_t *slot = cur + 16 + k * 16;
    int vin = __hip_atomic_load(&slot[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int vfi = __hip_atomic_load(&slot[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int vmx = __hip_atomic_load(&slot[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int bmn[4], bmx[4];
#pragma unroll
    for (int a = 0; a < 4; a++) {
      bmn[a] = __hip_atomic_load(&slot[4 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bmx[a] = __hip_atomic_load(&slot[8 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      vin += __shfl_xor(vin, o, 64); vfi += __shfl_xor(vfi, o, 64); vmx = max(vmx, __shfl_xor(vmx, o, 64));
#pragma unroll
      for (int a = 0; a < 4; a++) { bmn[a] = min(bmn[a], __shfl_xor(bmn[a], o, 64)); bmx[a] = max(bmx[a], __shfl_xor(bmx[a], o, 64)); }
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int a = 0; a < 4; a++) {
        __hip_atomic_store(&host[a], bmn[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&host[4 + a], bmx[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      __hip_atomic_store(&host[16], vin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host[17], vfi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&host[18], vmx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __threadfence_system();
      __hip_atomic_store(&host[8], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

namespace link {
int dc_probe_scratch_init_run(int32_t *w, hipStream_t st) {
  hipLaunchKernelGGL(k_dc_probe_scratch_init, dim3(1), dim3(272), 0, st, w);
  return check_launch("link_elk_block_forward (init)");
}
int dc_index_probe_bbox_run(const link_dc_buffers_t *b, const link_dc_grid_t *g, int64_t n, int32_t *cur, int32_t *next,
                            int32_t *host_dev, int seq, hipStream_t st) {
  if (!b->coords || !b->cnt || !b->slots || !b->vcell || !b->hdr || !cur || !next || !host_dev) return LINK_ERR_ARG;
  if (g->k < DC_INL || g->vp * (int64_t)g->k * 16 >= (1LL << 32) || n >= (1LL << 29) || n <= 0) return LINK_ERR_ARG;
  int64_t wgs = (n + 255) / 256;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index_probe_bbox, dim3((unsigned)wgs), dim3(256), 0, st, reinterpret_cast<const int4 *>(b->coords), n, *g, b->cnt,
                     reinterpret_cast<int4 *>(b->slots), b->vcell, b->hdr, cur, next, host_dev, seq);
  return check_launch("link_elk_block_forward (probe)");
}
}  // namespace link

// ---------------------------------------------------------------------------------------------
// 3x3x3 neighbour table of a frame from its freshly inserted slot lists (section G: the one-call block driver)
// ---------------------------------------------------------------------------------------------
// nbr[i][k] = row of the voxel at coords[i] + offset_k * step, -1 absent; offsets in get_kernel_offsets(3) order (x fastest:
// nn/utils/kernel.py:11-32), i.e. the table link_cell_table_build + link_neighbor_map give -- the reference's sphash ->
// sphashquery kernel map (nn/functional/conv.py:103-113) -- without the voxel-resolution cell table: the target's block cell
// is arithmetic, its slot list (cnt[cell] records (x, y, z, id), arrival order: between the insert and the pre_mix kernel)
// is scanned for the coordinates.  The tables it reads are the frame's own index, a few MB that stay in the L2s (cfg2: 0.26 MB
// of counters, 3.2 MB of inline records), where the cell table is 67 MB of random 4-byte reads behind a build and a clear.
// Duplicate coordinates: the smallest row wins (link_cell_table_build's rule).
// floor(a / s) through the float reciprocal, exact for |a| < 2^22 (the float quotient is within one of the true one; the
// remainder corrects it) -- the integer division hipcc emits is ~25 instructions, and a thread needs three of them
__device__ __forceinline__ int dc_floordiv_fast(int a, int s, float inv) {
  if (__builtin_expect(a >= (1 << 22) || a <= -(1 << 22), 0)) return link::floordiv(a, s);
  int q = (int)floorf((float)a * inv);
  const int r = a - q * s;
  q += (r >= s ? 1 : 0) - (r < 0 ? 1 : 0);
  return q;
}
// A thread owns (voxel, dx) -- three (dx, dy) columns of the 3 x 3 x 3 neighbourhood, nine entries; z is the fastest axis of the
// cell numbering, so a column's three targets lie in one cell (one scan of that cell's records answers all three) unless the
// column crosses a block boundary.  A workgroup of 768 threads owns 256 table rows -- the unit of the pair plan's count pass
// (conv_pairs.hip: k_pair_plan<false>) -- and lays them out in LDS: the table leaves in coalesced 16-byte stores, and with COUNT the
// workgroup also does that count pass on the tile it holds (pairs per offset -> wg_counts[blockIdx][28], rows whose centre entry
// is not the row itself in column 27; row_info[i] = #valid | centre valid << 16), so the pair plan needs no pass of its own over
// the table before its layout.
template <bool COUNT>
__global__ void __launch_bounds__(768) k_dc_neighbor_map(const int4 *__restrict__ coords, int64_t n, link_dc_grid_t g,
                                                         const uint32_t *__restrict__ cnt, const int4 *__restrict__ slots, int step,
                                                         int32_t *__restrict__ nbr, int32_t *__restrict__ wg_counts,
                                                         int32_t *__restrict__ row_info) {
  __shared__ int32_t tile[256 * 27];
  __shared__ int32_t wcnt[4 * 28];
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int rows = (int)((n - row0 < 256) ? n - row0 : 256);
  const int r = (int)threadIdx.x / 3, kx = (int)threadIdx.x - 3 * r;
  if (r < rows) {
    const int4 c = coords[row0 + r];
    const float inv = 1.0f / (float)g.s;
    const int tx = c.x + (kx - 1) * step;
    const unsigned ux = (unsigned)(dc_floordiv_fast(tx, g.s, inv) - g.lo[0]), ub = (unsigned)(c.w - g.lo[3]);
    const bool okx = ux < (unsigned)g.dim[0] && ub < (unsigned)g.dim[3];
    const int bz = dc_floordiv_fast(c.z, g.s, inv);
    const int rz = c.z - bz * g.s;                     // 0 .. s-1
    // A column's three targets (z - step, z, z + step) lie in the voxel's own z-block A and at most in the blocks below (L) and
    // above (U) it; a record of any of these cells is a target iff its (x, y) match and its z differs from the voxel's by -step,
    // 0 or +step -- no need to ask which cell a target falls into.  A and one neighbour block B (L or U: 2 of 7 z-positions cross
    // a boundary at s = 7, step 1) are looked up TOGETHER -- two counts and two 64-byte lines of inline records in flight per
    // column: one memory round trip -- and U on its own only when a column crosses both ways (block edge < 2 x step: rare).
    // (Versions before: cell after cell, record after record, 38 us; a one-cell path NEXT TO a three-cells path, both of which
    // every wave ran: 30 us, 943 vector instructions and 86 loads per wave -- tools/_pmc counters in docs/experiments.md.)
    const bool lowc = rz - step < 0, upc = rz + step >= g.s;
    const unsigned uzA = (unsigned)(bz - g.lo[2]);
    const unsigned uzB = (unsigned)((lowc ? dc_floordiv_fast(c.z - step, g.s, inv) : dc_floordiv_fast(c.z + step, g.s, inv)) - g.lo[2]);
    const unsigned uzU = (unsigned)(dc_floordiv_fast(c.z + step, g.s, inv) - g.lo[2]);
    const bool hasB = lowc || upc, both = lowc && upc;
    const bool any_both = __any(both);                   // wave-uniform
#pragma unroll
    for (int ky = 0; ky < 3; ky++) {
      const int ty = c.y + (ky - 1) * step;
      const unsigned uy = (unsigned)(dc_floordiv_fast(ty, g.s, inv) - g.lo[1]);
      int f[3] = {-1, -1, -1};
      const bool okxy = okx && uy < (unsigned)g.dim[1];
      auto take = [&](const int4 &q) {
        if (q.x == tx && q.y == ty) {
          const int d = q.z - c.z;
          if (d == -step && (f[0] < 0 || q.w < f[0])) f[0] = q.w;
          if (d == 0 && (f[1] < 0 || q.w < f[1])) f[1] = q.w;
          if (d == step && (f[2] < 0 || q.w < f[2])) f[2] = q.w;
        }
      };
      // cell 0 is a padding cell: its count is 0 and its line is always there to be read
      const int pcA = (okxy && uzA < (unsigned)g.dim[2]) ? link::dc_cell(g, (int)ux, (int)uy, (int)uzA, (int)ub) : 0;
      const int pcB = (okxy && hasB && uzB < (unsigned)g.dim[2]) ? link::dc_cell(g, (int)ux, (int)uy, (int)uzB, (int)ub) : 0;
      int nA = (int)cnt[pcA], nB = (int)cnt[pcB];
      const int4 *iA = slots + (int64_t)pcA * DC_INL, *iB = slots + (int64_t)pcB * DC_INL;
      const int4 rA[4] = {iA[0], iA[1], iA[2], iA[3]}, rB[4] = {iB[0], iB[1], iB[2], iB[3]};
      nA = pcA ? (nA < g.k ? nA : g.k) : 0;
      nB = pcB ? (nB < g.k ? nB : g.k) : 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (j < nA) take(rA[j]);
        if (j < nB) take(rB[j]);
      }
      for (int j = DC_INL; j < nA; j++) take(slots[link::dc_slot(g, pcA, j)]);
      for (int j = DC_INL; j < nB; j++) take(slots[link::dc_slot(g, pcB, j)]);
      if (any_both) {
        const int pcU = (okxy && both && uzU < (unsigned)g.dim[2]) ? link::dc_cell(g, (int)ux, (int)uy, (int)uzU, (int)ub) : 0;
        int nU = pcU ? (int)cnt[pcU] : 0;
        nU = nU < g.k ? nU : g.k;
        for (int j = 0; j < nU; j++) take(slots[link::dc_slot(g, pcU, j)]);
      }
      int32_t *o = tile + r * 27 + kx + 3 * ky;
      o[0] = f[0]; o[9] = f[1]; o[18] = f[2];
    }
  }
  __syncthreads();
  {                                                    // the tile leaves as it lies: rows * 27 contiguous ints
    const int tot = rows * 27;
    int32_t *dst = nbr + row0 * 27;                    // 256 * 27 * 4 bytes per workgroup: 16-byte aligned when nbr is
    if ((reinterpret_cast<uintptr_t>(nbr) & 15) == 0) {
      for (int e = threadIdx.x; e < (tot >> 2); e += 768) reinterpret_cast<int4 *>(dst)[e] = reinterpret_cast<const int4 *>(tile)[e];
      for (int e = (tot & ~3) + threadIdx.x; e < tot; e += 768) dst[e] = tile[e];
    } else {
      for (int e = threadIdx.x; e < tot; e += 768) dst[e] = tile[e];
    }
  }
  if (!COUNT) return;
  // the pair plan's count pass on this tile (k_pair_plan<false>, conv_pairs.hip): thread = row, the first four waves
  const int rr = (int)threadIdx.x;
  if (rr < 256) {
    const bool live = rr < rows;
    const int32_t *mine = tile + (live ? rr : 0) * 27;
    const int lane = rr & 63, wave = rr >> 6;
    int nvalid = 0, cvalid = 0;
    for (int k = 0; k < 27; k++) {
      const int v = live ? mine[k] : -1;
      const bool valid = v >= 0;
      if (k == 13) {
        cvalid = valid ? 1 : 0;
        const unsigned long long bad = __ballot(live && v != (int)(row0 + rr));
        if (lane == 0) wcnt[wave * 28 + 27] = __popcll(bad);
      }
      const unsigned long long m = __ballot(valid);
      if (lane == 0) wcnt[wave * 28 + k] = __popcll(m);
      nvalid += valid ? 1 : 0;
    }
    if (live) row_info[row0 + rr] = nvalid | (cvalid << 16);
  }
  __syncthreads();
  if (rr <= 27) wg_counts[(int64_t)blockIdx.x * 28 + rr] = wcnt[rr] + wcnt[28 + rr] + wcnt[56 + rr] + wcnt[84 + rr];
}

static int dc_neighbor_map_run(const int32_t *coords, int64_t n, const link_dc_grid_t *g, const uint32_t *cnt, const int32_t *slots,
                               int32_t step, int32_t *nbr, int32_t *wg_counts, int32_t *row_info, hipStream_t st) {
  if (n < 0 || !g || step <= 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!coords || !cnt || !slots || !nbr || n * 27 >= (1LL << 31)) return LINK_ERR_ARG;
  const unsigned wgs = (unsigned)((n + 255) / 256);
  if (wg_counts && row_info)
    hipLaunchKernelGGL(k_dc_neighbor_map<true>, dim3(wgs), dim3(768), 0, st, reinterpret_cast<const int4 *>(coords), n, *g, cnt,
                       reinterpret_cast<const int4 *>(slots), (int)step, nbr, wg_counts, row_info);
  else
    hipLaunchKernelGGL(k_dc_neighbor_map<false>, dim3(wgs), dim3(768), 0, st, reinterpret_cast<const int4 *>(coords), n, *g, cnt,
                       reinterpret_cast<const int4 *>(slots), (int)step, nbr, (int32_t *)nullptr, (int32_t *)nullptr);
  return link::check_launch("link_dc_neighbor_map");
}
extern "C" int link_dc_neighbor_map(const int32_t *coords, int64_t n, const link_dc_grid_t *g, const uint32_t *cnt,
                                    const int32_t *slots, int32_t step, int32_t *nbr, void *stream) {
  return dc_neighbor_map_run(coords, n, g, cnt, slots, step, nbr, nullptr, nullptr, link::S(stream));
}
namespace link {
// ... with the pair plan's count pass done on the way (wg_counts i32[ceil(n/256)][28], row_info i32[n]: link_pair_plan_count's outputs)
int dc_neighbor_map_count_run(const int32_t *coords, int64_t n, const link_dc_grid_t *g, const uint32_t *cnt, const int32_t *slots,
                              int32_t step, int32_t *nbr, int32_t *wg_counts, int32_t *row_info, hipStream_t st) {
  if (!wg_counts || !row_info) return LINK_ERR_ARG;
  return dc_neighbor_map_run(coords, n, g, cnt, slots, step, nbr, wg_counts, row_info, st);
}
}  // namespace link

namespace link {
// the insert of link_dc_index + occupancy statistics (behind link_dc_index_probe, dense.hip)
int dc_index_stats_run(const link_dc_buffers_t *b, const link_dc_grid_t *g, int64_t n, int32_t *stats, hipStream_t st) {
  if (!b->coords || !b->cnt || !b->slots || !b->vcell || !b->hdr) return LINK_ERR_ARG;
  if (g->k < DC_INL || g->vp * (int64_t)g->k * 16 >= (1LL << 32) || n >= (1LL << 29)) return LINK_ERR_ARG;
  int64_t wgs = (n + DC_INDEX_THREADS - 1) / DC_INDEX_THREADS;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index<true>, dim3((unsigned)wgs), dim3(DC_INDEX_THREADS), 0, st, reinterpret_cast<const int4 *>(b->coords), n, *g, b->cnt,
                     reinterpret_cast<int4 *>(b->slots), b->vcell, b->hdr, stats);
  return check_launch("link_dc_index_probe");
}
}  // namespace link

static int dc_common_ok(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d, int64_t n) {
  if (!b || !g || !d || n < 0) return LINK_ERR_ARG;
  if (b->io_dtype < 0 || b->io_dtype > 2) return LINK_ERR_ARG;
  if (d->op < 0 || d->op > 2 || d->cg <= 0 || d->c % d->cg != 0 || g->k < DC_INL) return LINK_ERR_ARG;
  const int parts = d->op == LINK_OP_COSX ? 3 : 2;
  if ((g->vp + 1) * (int64_t)parts * d->c * 4 >= (1LL << 32) || g->vp * (int64_t)g->k * 16 >= (1LL << 32)) return LINK_ERR_ARG;
  if (n * (int64_t)d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  return LINK_OK;
}

extern "C" int link_dc_premix_modsum(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                     int64_t n, int32_t warm, void *stream) {
  if (dc_common_ok(b, g, d, n) != LINK_OK) return LINK_ERR_ARG;
  if (d->c != 16 && d->c != 32 && d->c != 64) return LINK_ERR_ARG;
  if (g->k > 352) return LINK_ERR_ARG;                 // a cell's records must fit the wave's LDS list (LCAP)
  if (d->op == LINK_OP_COSX && !b->fin) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!b->feats || !b->slots || !b->cnt || !b->cell_n || !b->w_pre || !b->pre_ln_w || !b->pre_ln_b || !b->w_pos ||
      !b->S || !b->hdr)
    return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (b->tune.k1_form == 1) return dc_tiles_modsum(b, g, d, n, warm != 0, st);
  switch (b->io_dtype) {
    case 1: return dcio_f16::run_premix_modsum(b, *g, *d, n, warm != 0, st);
    case 2: return dcio_bf16::run_premix_modsum(b, *g, *d, n, warm != 0, st);
    default: return dcio_f32::run_premix_modsum(b, *g, *d, n, warm != 0, st);
  }
}

extern "C" int link_dc_demod(const float *A, const float *fin, const int32_t *coords, const int32_t *vcell,
                             const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                             const link_elk_desc_t *d, const link_dc_grid_t *g, int64_t n, void *out, int32_t io_dtype,
                             void *stream) {
  if (!d || !g || n < 0 || !dc_width_ok(d->c) || d->op < 0 || d->op > 2 || d->cg <= 0 || d->c % d->cg != 0) return LINK_ERR_ARG;
  if (io_dtype < 0 || io_dtype > 2) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!A || !coords || !vcell || !w_pos || !ln_w || !ln_b || !out || (d->op == LINK_OP_COSX && !fin)) return LINK_ERR_ARG;
  const int parts = d->op == LINK_OP_COSX ? 3 : 2;
  if ((g->vp + 1) * (int64_t)parts * d->c * 4 >= (1LL << 32) || n * (int64_t)d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (io_dtype) {
    case 1: return dcio_f16::run_demod(A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, *d, *g, n, out, st);
    case 2: return dcio_bf16::run_demod(A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, *d, *g, n, out, st);
    default: return dcio_f32::run_demod(A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, *d, *g, n, out, st);
  }
}

extern "C" int link_dc_gather_demod(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                    int64_t n, void *stream) {
  if (dc_common_ok(b, g, d, n) != LINK_OK || !dc_width_ok(d->c) || (d->r != 2 && d->r != 3)) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!b->S || !b->cell_n || !b->slots || !b->w_pos || !b->ln_w || !b->ln_b || !b->out || (d->op == LINK_OP_COSX && !b->fin))
    return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (d->c != 64) return LINK_ERR_ARG;               // the fused gather + de-modulate kernels are built for C = 64; other widths: link_dc_gather + link_dc_demod
  switch (b->io_dtype) {
    case 1: return dcio_f16::run_gather_demod(b, *g, *d, n, st);
    case 2: return dcio_bf16::run_gather_demod(b, *g, *d, n, st);
    default: return dcio_f32::run_gather_demod(b, *g, *d, n, st);
  }
}
