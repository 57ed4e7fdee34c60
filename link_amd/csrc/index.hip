// link_amd/csrc/index.hip -- section B (index half) of include/link_amd.h.
//
// Replaces the reference's  sphash -> torch.unique(dim=0) -> sphash -> sphashquery -> spcount  chain
// (segmentation/core/models/utils.py:45-51) and the neighbour query sphash(C,offsets) -> sphashquery
// (utils.py:65-73) by direct addressing into a dense block grid:
//
//   k_cell_count   one thread per voxel: block coordinate -> cell, rank = atomicAdd(cell_counts[cell])
//   k_cell_scan    ONE decoupled-look-back scan over the V cells: exclusive prefix of (occupied, count)
//                  gives at once the block id of every occupied cell -- its rank in x-major cell order,
//                  which is torch.unique's lexicographic row order -- and the start of the block's
//                  voxel segment; it also clears the counters it consumed (self-cleaning scratch)
//   k_place        scatter voxel ids to blk_start[blk] + rank
//   k_sort_seg     order each block's segment by voxel id (ranks came from atomics, so their order
//                  is arbitrary; downstream reductions must not depend on it)
//
// No hashing, no sort passes, no host sync; 4 launches, O(N + V) bytes.
#include <limits.h>

#include "common.h"

using namespace link;

// ---------------------------------------------------------------------------------------------
// host helper
// ---------------------------------------------------------------------------------------------
extern "C" int64_t link_grid_from_bounds(const int32_t lo[4], const int32_t hi[4], int32_t s,
                                         link_grid_t *g) {
  if (!lo || !hi || !g || s <= 0) return -1;
  g->s = s;
  int64_t v = 1;
  for (int a = 0; a < 4; a++) {
    if (hi[a] < lo[a]) return -1;
    int32_t blo = (a < 3) ? floordiv(lo[a], s) : lo[a];
    int32_t bhi = (a < 3) ? floordiv(hi[a], s) : hi[a];
    g->lo[a] = blo;
    int64_t d = (int64_t)bhi - (int64_t)blo + 1;
    if (d <= 0 || d >= (1LL << 30)) return -1;
    g->dim[a] = (int32_t)d;
    v *= d;
    if (v >= (1LL << 30)) return -1;
  }
  return v;
}

// ---------------------------------------------------------------------------------------------
// bbox
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_bbox(const int4 *__restrict__ coords, int64_t n, int32_t *bbox) {
  int mn[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
  int mx[4] = {INT_MIN, INT_MIN, INT_MIN, INT_MIN};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    int4 c = coords[i];
    mn[0] = min(mn[0], c.x); mx[0] = max(mx[0], c.x);
    mn[1] = min(mn[1], c.y); mx[1] = max(mx[1], c.y);
    mn[2] = min(mn[2], c.z); mx[2] = max(mx[2], c.z);
    mn[3] = min(mn[3], c.w); mx[3] = max(mx[3], c.w);
  }
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn[a] = min(mn[a], __shfl_xor(mn[a], o, 64));
      mx[a] = max(mx[a], __shfl_xor(mx[a], o, 64));
    }
  __shared__ int smn[4][4], smx[4][4];
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0)
    for (int a = 0; a < 4; a++) { smn[wave][a] = mn[a]; smx[wave][a] = mx[a]; }
  __syncthreads();
  if (threadIdx.x < 4) {
    int a = threadIdx.x;
    int m0 = min(min(smn[0][a], smn[1][a]), min(smn[2][a], smn[3][a]));
    int m1 = max(max(smx[0][a], smx[1][a]), max(smx[2][a], smx[3][a]));
    atomicMin(&bbox[a], m0);
    atomicMax(&bbox[4 + a], m1);
  }
}

extern "C" int link_coords_bbox(const int32_t *coords, int64_t n, int32_t *bbox, void *stream) {
  if (n < 0 || !bbox || (n > 0 && !coords)) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  unsigned nb = blocks_for(n, 256 * 8);
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(k_bbox, dim3(nb), dim3(256), 0, S(stream), reinterpret_cast<const int4 *>(coords),
                     n, bbox);
  return check_launch("link_coords_bbox");
}

// ---------------------------------------------------------------------------------------------
// index build
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
#ifndef LINK_SCAN_ITEMS
#define LINK_SCAN_ITEMS 8
#endif
constexpr int SCAN_ITEMS = LINK_SCAN_ITEMS;           // cells per thread (a multiple of 4).  16 / 32 (fewer tiles in the look-back chain) measured
                                                      // slower on every LiDAR stage frame: index rebuilt + 2 … + 8 us (A/B on one box, round 3)
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;  // cells per workgroup
static_assert(SCAN_ITEMS % 4 == 0, "uint4 pieces");

struct IndexScratch {
  int32_t *vox_cell;   // [n]
  int32_t *vox_rank;   // [n]
  int32_t *perm_tmp;   // [n]
  int32_t *pos_tmp;    // [n]  pos_blk stand-in when the caller passes NULL
  int32_t *cell_start; // [v]  first sorted position of each occupied cell's segment
  unsigned long long *desc;  // [tiles]
  unsigned int *ticket;      // [1] (+pad)
};

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static inline int64_t scan_tiles(int64_t v) { return (v + SCAN_TILE - 1) / SCAN_TILE; }

extern "C" size_t link_index_scratch_bytes(int64_t n, int64_t v) {
  if (n < 0) n = 0;
  if (v < 0) v = 0;
  return 4 * align256((size_t)n * 4) + align256((size_t)v * 4) + align256((size_t)scan_tiles(v > n ? v : n) * 8) + 256;
}

static IndexScratch carve(void *scratch, int64_t n, int64_t v) {
  char *p = reinterpret_cast<char *>(scratch);
  IndexScratch s;
  s.vox_cell = reinterpret_cast<int32_t *>(p); p += align256((size_t)n * 4);
  s.vox_rank = reinterpret_cast<int32_t *>(p); p += align256((size_t)n * 4);
  s.perm_tmp = reinterpret_cast<int32_t *>(p); p += align256((size_t)n * 4);
  s.pos_tmp = reinterpret_cast<int32_t *>(p); p += align256((size_t)n * 4);
  s.cell_start = reinterpret_cast<int32_t *>(p); p += align256((size_t)v * 4);
  s.desc = reinterpret_cast<unsigned long long *>(p); p += align256((size_t)scan_tiles(v > n ? v : n) * 8);
  s.ticket = reinterpret_cast<unsigned int *>(p);
  return s;
}

__global__ void __launch_bounds__(256) k_cell_count(const int4 *__restrict__ coords, int64_t n,
                                                    link_grid_t g, unsigned int *cell_counts,
                                                    int32_t *__restrict__ vox_cell,
                                                    int32_t *__restrict__ vox_rank,
                                                    unsigned long long *desc, int64_t tiles,
                                                    unsigned int *ticket) {
  int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // side job: reset the scan's tile descriptors and ticket (ordered before k_cell_scan by the
  // kernel boundary)
  for (int64_t t = gid; t < tiles; t += (int64_t)gridDim.x * blockDim.x) desc[t] = 0ULL;
  if (gid == 0) *ticket = 0u;
  const bool live = gid < n;
  int32_t cell = -1;
  if (live) {
    const int4 c = coords[gid];
    cell = cell_of(g, floordiv(c.x, g.s), floordiv(c.y, g.s), floordiv(c.z, g.s), c.w);
    vox_cell[gid] = cell;
  }
  // The lanes of a wave that fall into the same cell make ONE atomic between them.  LiDAR frames keep spatially close
  // voxels close in memory (sensor scan order, or the lexicographic order of torch.unique), so the ~30 atomics per counter
  // of a big block -- which serialise in the L2 -- become a handful; a frame in random order pays one pass of the loop per
  // lane (64 x ~8 scalar / vector ops) and issues the same atomics as before.
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(cell >= 0), mine = 0;
  while (todo) {
    const int cj = __builtin_amdgcn_readlane(cell, __builtin_ctzll(todo));
    const unsigned long long m = __ballot(cell == cj);
    if (cell == cj) mine = m;
    todo &= ~m;
  }
  const int leader = mine ? __builtin_ctzll(mine) : lane;
  unsigned int base = 0;
  if (cell >= 0 && lane == leader) base = atomicAdd(&cell_counts[cell], (unsigned int)__popcll(mine));
  base = __shfl(base, leader, 64);
  if (cell >= 0) vox_rank[gid] = (int32_t)(base + (unsigned int)__popcll(mine & ((1ull << lane) - 1ull)));
}

// descriptor: [63:62] flag (0 invalid, 1 aggregate, 2 inclusive prefix) [61:31] occupied [30:0] voxels
__device__ __forceinline__ unsigned long long pack_desc(unsigned flag, unsigned occ, unsigned cnt) {
  return ((unsigned long long)flag << 62) | ((unsigned long long)occ << 31) | (unsigned long long)cnt;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_cell_scan(
    unsigned int *cell_counts, int64_t v, link_grid_t g, unsigned long long *desc, unsigned int *ticket,
    int64_t tiles, int32_t *__restrict__ cell_blk, int32_t *__restrict__ cell_start,
    int32_t *__restrict__ blk_start,
    int32_t *__restrict__ blk_coords, int32_t *__restrict__ counts, int32_t *__restrict__ hdr) {
  __shared__ unsigned int s_tile;
  __shared__ unsigned long long s_wave[SCAN_THREADS / 64];
  __shared__ unsigned long long s_excl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);  // dynamic tile id: predecessors have all started
  __syncthreads();
  const int64_t tile = s_tile;
  const int64_t base = tile * SCAN_TILE + (int64_t)tid * SCAN_ITEMS;

  unsigned int cnt[SCAN_ITEMS];
  if (base + SCAN_ITEMS <= v) {
    const uint4 *p = reinterpret_cast<const uint4 *>(cell_counts + base);
    uint4 a[SCAN_ITEMS / 4];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) a[i] = p[i];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) { cnt[4 * i] = a[i].x; cnt[4 * i + 1] = a[i].y; cnt[4 * i + 2] = a[i].z; cnt[4 * i + 3] = a[i].w; }
    uint4 z = make_uint4(0, 0, 0, 0);
    uint4 *q = reinterpret_cast<uint4 *>(cell_counts + base);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; i++) q[i] = z;  // self-clean
  } else {
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
      cnt[k] = (base + k < v) ? cell_counts[base + k] : 0u;
      if (base + k < v) cell_counts[base + k] = 0u;
    }
  }
  // thread aggregate as (occ << 32 | voxels)
  unsigned long long mine = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) mine += ((unsigned long long)(cnt[k] != 0u) << 32) | cnt[k];
  // inclusive wave scan
  unsigned long long incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    unsigned long long x = s_wave[w];
    if (w < wave) wave_off += x;
    total += x;
  }
  // ---- decoupled look-back by wave 0
  if (wave == 0) {
    const unsigned t_occ = (unsigned)(total >> 32), t_cnt = (unsigned)(total & 0xFFFFFFFFu);
    unsigned long long excl = 0;  // (occ << 32 | cnt) of all earlier tiles
    if (tile == 0) {
      if (lane == 0)
        __hip_atomic_store(&desc[0], pack_desc(2u, t_occ, t_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0)
        __hip_atomic_store(&desc[tile], pack_desc(1u, t_occ, t_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t look = tile - 1;
      for (;;) {
        int64_t t = look - lane;
        unsigned long long d;
        if (t >= 0) {
          do {
            d = __hip_atomic_load(&desc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((d >> 62) == 0ULL);
        } else {
          d = pack_desc(2u, 0u, 0u);  // virtual tiles before tile 0: prefix 0
        }
        unsigned flag = (unsigned)(d >> 62);
        unsigned long long val = (((d >> 31) & 0x7FFFFFFFULL) << 32) | (d & 0x7FFFFFFFULL);
        unsigned long long pmask = __ballot(flag == 2u);
        if (pmask != 0ULL) {
          int first = __ffsll((long long)pmask) - 1;  // nearest tile holding an inclusive prefix
          unsigned long long contrib = (lane <= first) ? val : 0ULL;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
          excl += contrib;
          break;
        }
        unsigned long long contrib = val;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
        excl += contrib;
        look -= 64;
      }
      if (lane == 0) {
        unsigned long long inc = excl + total;
        __hip_atomic_store(&desc[tile], pack_desc(2u, (unsigned)(inc >> 32), (unsigned)(inc & 0xFFFFFFFFu)),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) {
      s_excl = excl;
      if (tile == tiles - 1) {  // grand totals
        unsigned long long inc = excl + total;
        int32_t m = (int32_t)(inc >> 32), nv = (int32_t)(inc & 0xFFFFFFFFu);
        hdr[LINK_HDR_M] = m;
        hdr[LINK_HDR_STATUS] = 0;
        hdr[LINK_HDR_NVALID] = nv;
        blk_start[m] = nv;
      }
    }
  }
  __syncthreads();
  unsigned long long pre = s_excl + wave_off + (incl - mine);
  unsigned occ = (unsigned)(pre >> 32), vox = (unsigned)(pre & 0xFFFFFFFFu);
  const uint32_t d3 = (uint32_t)g.dim[3], d2 = (uint32_t)g.dim[2], d1 = (uint32_t)g.dim[1];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) {
    int64_t cell = base + k;
    if (cell < v) {
      if (cnt[k] != 0u) {
        cell_blk[cell] = (int32_t)occ + 1;
        cell_start[cell] = (int32_t)vox;
        blk_start[occ] = (int32_t)vox;
        counts[occ] = (int32_t)cnt[k];
        uint32_t cc = (uint32_t)cell;
        uint32_t ub = cc % d3; cc /= d3;
        uint32_t uz = cc % d2; cc /= d2;
        uint32_t uy = cc % d1; cc /= d1;
        int4 bc = make_int4((int32_t)cc + g.lo[0], (int32_t)uy + g.lo[1], (int32_t)uz + g.lo[2],
                            (int32_t)ub + g.lo[3]);
        reinterpret_cast<int4 *>(blk_coords)[occ] = bc;
        occ++;
        vox += cnt[k];
      } else {
        cell_blk[cell] = 0;
      }
    }
  }
}

__global__ void __launch_bounds__(256) k_place(const int32_t *__restrict__ vox_cell,
                                               const int32_t *__restrict__ vox_rank, int64_t n,
                                               const int32_t *__restrict__ cell_blk,
                                               const int32_t *__restrict__ cell_start,
                                               int32_t *__restrict__ perm_tmp,
                                               int32_t *__restrict__ pos_blk_out,
                                               int32_t *__restrict__ vox_blk,
                                               int64_t *__restrict__ idx_query, int32_t *hdr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int32_t cell = vox_cell[i];
  int32_t blk = -1;
  if (cell >= 0) {
    blk = cell_blk[cell] - 1;                       // one lookup level: both come from the cell
    const int32_t pos = cell_start[cell] + vox_rank[i];
    perm_tmp[pos] = (int32_t)i;
    pos_blk_out[pos] = blk;                         // block of a sorted position (constant over the segment)
  } else {
    atomicOr(&hdr[LINK_HDR_STATUS], 1);
  }
  vox_blk[i] = blk;
  if (idx_query) idx_query[i] = (int64_t)blk;
}

// Order every block's segment by voxel id.  Segment lengths are tiny (<= s^3 for unique voxels), so
// each position counts its smaller neighbours directly: sum of n_b^2 loads, all L1/L2 hits.
constexpr int SORT_MAX_SEG = 8192;
__global__ void __launch_bounds__(256) k_sort_seg(const int32_t *__restrict__ perm_tmp,
                                                  const int32_t *__restrict__ pos_blk_in,
                                                  const int32_t *__restrict__ blk_start,
                                                  const int32_t *__restrict__ hdr, int64_t n,
                                                  const int4 *__restrict__ coords,
                                                  int32_t *__restrict__ perm,
                                                  int4 *__restrict__ vox_sorted,
                                                  unsigned long long *cell_pair = nullptr,
                                                  const int4 *__restrict__ blk_coords = nullptr,
                                                  link_grid_t g = link_grid_t{}) {
  int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n || p >= hdr[LINK_HDR_NVALID]) return;
  const int32_t i = perm_tmp[p];
  const int32_t b = pos_blk_in[p];                  // same hop as perm_tmp (no voxel -> block chase)
  int32_t st = blk_start[b], en = blk_start[b + 1];
  if (cell_pair && p == st) {                       // first-voxel numbering: the block's pair word has been read by every voxel
    const int4 bc = blk_coords[b];                  // of it (k_place_first): back to zero for the next frame's atomics
    const int32_t cell = cell_of(g, bc.x, bc.y, bc.z, bc.w);
    if (cell >= 0) cell_pair[cell] = 0ULL;
  }
  int32_t len = en - st;
  int64_t dst = p;
  if (len > 1 && len <= SORT_MAX_SEG) {
    int32_t r = 0, q = st;
    // big blocks (LiDAR: tens to hundreds of voxels): 16-byte pieces between the unaligned ends
    for (; q < en && (reinterpret_cast<uintptr_t>(perm_tmp + q) & 15); q++) r += (perm_tmp[q] < i);
    for (; q + 4 <= en; q += 4) {
      const int4 v = *reinterpret_cast<const int4 *>(perm_tmp + q);
      r += (v.x < i) + (v.y < i) + (v.z < i) + (v.w < i);
    }
    for (; q < en; q++) r += (perm_tmp[q] < i);
    dst = st + r;
  }
  perm[dst] = i;
  if (vox_sorted) {
    int4 c = coords[i];
    vox_sorted[dst] = make_int4(c.x, c.y, c.z, i);
  }
}

extern "C" int link_index_build(const int32_t *coords, int64_t n, const link_grid_t *grid,
                                uint32_t *cell_counts, void *scratch, size_t scratch_bytes,
                                int32_t *cell_blk, int32_t *vox_blk, int64_t *idx_query, int32_t *perm,
                                int32_t *vox_sorted, int32_t *pos_blk, int32_t *blk_start,
                                int32_t *blk_coords, int32_t *counts, int32_t *hdr, void *stream) {
  if (n < 0 || n >= (1LL << 31) || !grid || !hdr) return LINK_ERR_ARG;
  int64_t v = 1;
  for (int a = 0; a < 4; a++) {
    if (grid->dim[a] <= 0) return LINK_ERR_ARG;
    v *= grid->dim[a];
    if (v >= (1LL << 30)) return LINK_ERR_ARG;
  }
  if (grid->s <= 0) return LINK_ERR_ARG;
  if (!cell_counts || !scratch || !cell_blk || !blk_start || !blk_coords || !counts) return LINK_ERR_ARG;
  if (n > 0 && (!coords || !vox_blk || !perm)) return LINK_ERR_ARG;
  if (scratch_bytes < link_index_scratch_bytes(n, v)) return LINK_ERR_WORKSPACE;
  IndexScratch sc = carve(scratch, n, v);
  const int64_t tiles = scan_tiles(v);
  hipStream_t st = S(stream);
  int64_t cc_threads = n > tiles ? n : tiles;
  if (cc_threads > (1 << 22)) cc_threads = (n > (1 << 22)) ? n : (1 << 22);
  hipLaunchKernelGGL(k_cell_count, dim3(blocks_for(cc_threads, 256)), dim3(256), 0, st,
                     reinterpret_cast<const int4 *>(coords), n, *grid, cell_counts, sc.vox_cell,
                     sc.vox_rank, sc.desc, tiles, sc.ticket);
  hipLaunchKernelGGL(k_cell_scan, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, st, cell_counts, v,
                     *grid, sc.desc, sc.ticket, tiles, cell_blk, sc.cell_start, blk_start, blk_coords, counts, hdr);
  if (n > 0) {
    int32_t *pb = pos_blk ? pos_blk : sc.pos_tmp;   // needed internally even if the caller does not want it
    hipLaunchKernelGGL(k_place, dim3(blocks_for(n, 256)), dim3(256), 0, st, sc.vox_cell, sc.vox_rank, n,
                       cell_blk, sc.cell_start, sc.perm_tmp, pb, vox_blk, idx_query, hdr);
    hipLaunchKernelGGL(k_sort_seg, dim3(blocks_for(n, 256)), dim3(256), 0, st, sc.perm_tmp, pb,
                       blk_start, hdr, n, reinterpret_cast<const int4 *>(coords), perm,
                       reinterpret_cast<int4 *>(vox_sorted), (unsigned long long *)nullptr, (const int4 *)nullptr, *grid);
  }
  return check_launch("link_index_build");
}

// ---------------------------------------------------------------------------------------------
// first-voxel numbering: the index of a frame that occupies a small part of a big grid (LiDAR)
// ---------------------------------------------------------------------------------------------
// link_index_build numbers the blocks in cell order (what torch.unique gives the reference: utils.py:50-58) with a scan over ALL
// V cells of the grid -- 10-17 us of the 28 us an S-kitti stage pays for its index, for 1-2 % occupied cells.  A caller that only
// needs A numbering (ElkCorePlan: the step's output is per voxel, block order never leaves the arena) gets one from a scan over
// the N voxels instead: block b = the b-th voxel, in id order, that is the smallest id of its cell.  Deterministic (ids, not
// insertion order, decide), same tables as link_index_build except that blocks are in first-voxel order.  One 8-byte word per
// cell carries the frame through the three kernels -- (count | code of the smallest id) while the voxels are counted, (segment
// start | block + 1) once the scan has numbered the cell, zero again when the last kernel is through -- so every kernel reaches
// what it needs in two dependent loads.  The previous frame's cells leave cell_blk through its block list (hdr[LINK_HDR_M] rows
// of blk_coords): nothing here is proportional to V.
constexpr int VSCAN_ITEMS = SCAN_ITEMS;             // tiles over n voxels <= the descriptor slots carved for max(n, v) cells
constexpr int VSCAN_TILE = SCAN_THREADS * VSCAN_ITEMS;
constexpr int VSCAN_RESIDENT = 1024;                // tiles that are certainly co-resident (256 CUs x >= 4 such workgroups)
static inline int64_t vscan_tiles(int64_t n) { return (n + VSCAN_TILE - 1) / VSCAN_TILE; }
#define LINK_FIRST_CODE(i) (0x7FFFFFFFu - (unsigned int)(i))     /* atomicMax over a zero-initialised word: the smallest id wins */

__global__ void __launch_bounds__(256) k_cell_count_first(const int4 *__restrict__ coords, int64_t n, link_grid_t g,
                                                          unsigned int *cell_pair /* [V][2]: count, code */,
                                                          int32_t *__restrict__ vox_cell, int32_t *__restrict__ vox_rank,
                                                          unsigned long long *desc, int64_t tiles, unsigned int *ticket,
                                                          int32_t *cell_blk, const int4 *__restrict__ blk_coords_prev,
                                                          int32_t *hdr) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthr = (int64_t)gridDim.x * blockDim.x;
  int4 c = make_int4(0, 0, 0, 0);
  if (gid < n) c = coords[gid];
  for (int64_t t = gid; t < tiles; t += nthr) desc[t] = 0ULL;
  if (gid == 0) *ticket = 0u;
  {                                                  // the previous frame's blocks leave the cell table (their pair words were
    const int64_t m_prev = hdr[LINK_HDR_M];           // zeroed by that frame's k_sort_seg); hdr: what the previous call on
    for (int64_t b = gid; b < m_prev; b += nthr) {    // these buffers wrote -- 0 in a zero-filled hdr
      const int4 pc = blk_coords_prev[b];
      const int32_t pcell = cell_of(g, pc.x, pc.y, pc.z, pc.w);
      if (pcell >= 0) cell_blk[pcell] = 0;
    }
  }
  int32_t cell = -1;
  if (gid < n) {
    cell = cell_of(g, floordiv(c.x, g.s), floordiv(c.y, g.s), floordiv(c.z, g.s), c.w);
    vox_cell[gid] = cell;
    if (cell < 0) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 1);      // published (and reset) by the scan's last tile
  }
  // lanes of a wave in the same cell: one pair of atomics between them (see k_cell_count); the group's lowest lane holds
  // its smallest voxel id
  const int lane = threadIdx.x & 63;
  unsigned long long todo = __ballot(cell >= 0), mine = 0;
  while (todo) {
    const int cj = __builtin_amdgcn_readlane(cell, __builtin_ctzll(todo));
    const unsigned long long m = __ballot(cell == cj);
    if (cell == cj) mine = m;
    todo &= ~m;
  }
  const int leader = mine ? __builtin_ctzll(mine) : lane;
  unsigned int base = 0;
  if (cell >= 0 && lane == leader) {
    base = atomicAdd(&cell_pair[2 * (int64_t)cell], (unsigned int)__popcll(mine));
    atomicMax(&cell_pair[2 * (int64_t)cell + 1], LINK_FIRST_CODE(gid));
  }
  base = __shfl(base, leader, 64);
  if (cell >= 0) vox_rank[gid] = (int32_t)(base + (unsigned int)__popcll(mine & ((1ull << lane) - 1ull)));
}

// One decoupled-look-back scan over the N voxels of (is the smallest id of its cell, voxels of that cell).
// (Placing the voxels in the same pass -- every voxel waiting for its cell's pair word to turn into (start | block + 1) -- was
// tried: one launch fewer, 2-4 us MORE per stage frame; the placement stores then sit behind the look-back chain.)
template <bool TICKET>
__global__ void __launch_bounds__(SCAN_THREADS) k_vox_scan(
    const int32_t *__restrict__ vox_cell, int64_t n, link_grid_t g, unsigned long long *cell_pair, unsigned long long *desc,
    unsigned int *ticket, int64_t tiles, int32_t *__restrict__ cell_blk, int32_t *__restrict__ blk_start,
    int32_t *__restrict__ blk_coords, int32_t *__restrict__ counts, int32_t *__restrict__ hdr) {
  __shared__ unsigned int s_tile;
  __shared__ unsigned long long s_wave[SCAN_THREADS / 64];
  __shared__ unsigned long long s_excl;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t tile = blockIdx.x;                          // few tiles: all resident, a predecessor is always running
  if (TICKET) {
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);     // dynamic tile id: predecessors have all started
    __syncthreads();
    tile = s_tile;
  }
  const int64_t base = tile * VSCAN_TILE + (int64_t)tid * VSCAN_ITEMS;
  int32_t cell[VSCAN_ITEMS];
  unsigned int cnt[VSCAN_ITEMS];
  if (base + VSCAN_ITEMS <= n) {
#pragma unroll
    for (int k = 0; k < VSCAN_ITEMS; k += 4) {
      const int4 q = *reinterpret_cast<const int4 *>(vox_cell + base + k);
      cell[k] = q.x; cell[k + 1] = q.y; cell[k + 2] = q.z; cell[k + 3] = q.w;
    }
  } else {
#pragma unroll
    for (int k = 0; k < VSCAN_ITEMS; k++) cell[k] = (base + k < n) ? vox_cell[base + k] : -1;
  }
  unsigned long long pr[VSCAN_ITEMS];
#pragma unroll
  for (int k = 0; k < VSCAN_ITEMS; k++) pr[k] = cell[k] >= 0 ? cell_pair[cell[k]] : 0ULL;
#pragma unroll
  for (int k = 0; k < VSCAN_ITEMS; k++) {
    // (a cell already numbered by its first voxel's thread shows (start | block + 1): block + 1 is no voxel's code)
    const bool first = cell[k] >= 0 && (unsigned int)(pr[k] >> 32) == LINK_FIRST_CODE(base + k);
    cnt[k] = first ? (unsigned int)pr[k] : 0u;
    if (!first) cell[k] = -1;
  }
  unsigned long long mine = 0;
#pragma unroll
  for (int k = 0; k < VSCAN_ITEMS; k++) mine += ((unsigned long long)(cell[k] >= 0) << 32) | cnt[k];
  unsigned long long incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    unsigned long long t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  unsigned long long wave_off = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; w++) {
    unsigned long long x = s_wave[w];
    if (w < wave) wave_off += x;
    total += x;
  }
  if (wave == 0) {
    const unsigned t_occ = (unsigned)(total >> 32), t_cnt = (unsigned)(total & 0xFFFFFFFFu);
    unsigned long long excl = 0;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&desc[0], pack_desc(2u, t_occ, t_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&desc[tile], pack_desc(1u, t_occ, t_cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t look = tile - 1;
      for (;;) {
        const int64_t t = look - lane;
        unsigned long long d;
        if (t >= 0) {
          do {
            d = __hip_atomic_load(&desc[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } while ((d >> 62) == 0ULL);
        } else {
          d = pack_desc(2u, 0u, 0u);
        }
        const unsigned flag = (unsigned)(d >> 62);
        const unsigned long long val = (((d >> 31) & 0x7FFFFFFFULL) << 32) | (d & 0x7FFFFFFFULL);
        const unsigned long long pmask = __ballot(flag == 2u);
        unsigned long long contrib = val;
        if (pmask != 0ULL) contrib = (lane <= __ffsll((long long)pmask) - 1) ? val : 0ULL;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) contrib += __shfl_xor(contrib, o, 64);
        excl += contrib;
        if (pmask != 0ULL) break;
        look -= 64;
      }
      if (lane == 0) {
        const unsigned long long inc = excl + total;
        __hip_atomic_store(&desc[tile], pack_desc(2u, (unsigned)(inc >> 32), (unsigned)(inc & 0xFFFFFFFFu)), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (lane == 0) {
      s_excl = excl;
      if (tile == tiles - 1) {
        const unsigned long long inc = excl + total;
        const int32_t m = (int32_t)(inc >> 32), nv = (int32_t)(inc & 0xFFFFFFFFu);
        hdr[LINK_HDR_M] = m;
        hdr[LINK_HDR_STATUS] = hdr[LINK_HDR_STATUS_ACC];       // bit 0: a voxel outside the grid (k_cell_count_first)
        hdr[LINK_HDR_STATUS_ACC] = 0;
        hdr[LINK_HDR_NVALID] = nv;
        blk_start[m] = nv;
      }
    }
  }
  __syncthreads();
  const unsigned long long pre = s_excl + wave_off + (incl - mine);
  unsigned occ = (unsigned)(pre >> 32), vox = (unsigned)(pre & 0xFFFFFFFFu);
  const uint32_t d3 = (uint32_t)g.dim[3], d2 = (uint32_t)g.dim[2], d1 = (uint32_t)g.dim[1];
#pragma unroll
  for (int k = 0; k < VSCAN_ITEMS; k++) {
    if (cell[k] >= 0) {
      cell_blk[cell[k]] = (int32_t)occ + 1;
      cell_pair[cell[k]] = ((unsigned long long)(occ + 1u) << 32) | (unsigned long long)vox;      // (start | block + 1)
      blk_start[occ] = (int32_t)vox;
      counts[occ] = (int32_t)cnt[k];
      uint32_t cc = (uint32_t)cell[k];
      const uint32_t ub = cc % d3; cc /= d3;
      const uint32_t uz = cc % d2; cc /= d2;
      const uint32_t uy = cc % d1; cc /= d1;
      reinterpret_cast<int4 *>(blk_coords)[occ] =
          make_int4((int32_t)cc + g.lo[0], (int32_t)uy + g.lo[1], (int32_t)uz + g.lo[2], (int32_t)ub + g.lo[3]);
      occ++;
      vox += cnt[k];
    }
  }
}

__global__ void __launch_bounds__(256) k_place_first(const int32_t *__restrict__ vox_cell, const int32_t *__restrict__ vox_rank,
                                                     int64_t n, const unsigned long long *__restrict__ cell_pair,
                                                     int32_t *__restrict__ perm_tmp, int32_t *__restrict__ pos_blk_out,
                                                     int32_t *__restrict__ vox_blk, int64_t *__restrict__ idx_query) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t cell = vox_cell[i];
  const int32_t rank = vox_rank[i];
  int32_t blk = -1;
  if (cell >= 0) {
    const unsigned long long p = cell_pair[cell];     // (segment start | block + 1)
    blk = (int32_t)(p >> 32) - 1;
    const int32_t pos = (int32_t)(unsigned int)p + rank;
    perm_tmp[pos] = (int32_t)i;
    pos_blk_out[pos] = blk;
  }
  vox_blk[i] = blk;
  if (idx_query) idx_query[i] = (int64_t)blk;
}

extern "C" int link_index_build_first(const int32_t *coords, int64_t n, const link_grid_t *grid, uint64_t *cell_pair,
                                      void *scratch, size_t scratch_bytes, int32_t *cell_blk, int32_t *vox_blk,
                                      int64_t *idx_query, int32_t *perm, int32_t *vox_sorted, int32_t *pos_blk,
                                      int32_t *blk_start, int32_t *blk_coords, int32_t *counts, int32_t *hdr, void *stream) {
  if (n < 0 || n >= (1LL << 30) || !grid || !hdr) return LINK_ERR_ARG;
  int64_t v = 1;
  for (int a = 0; a < 4; a++) {
    if (grid->dim[a] <= 0) return LINK_ERR_ARG;
    v *= grid->dim[a];
    if (v >= (1LL << 30)) return LINK_ERR_ARG;
  }
  if (grid->s <= 0) return LINK_ERR_ARG;
  if (!cell_pair || !scratch || !cell_blk || !blk_start || !blk_coords || !counts) return LINK_ERR_ARG;
  if (n > 0 && (!coords || !vox_blk || !perm)) return LINK_ERR_ARG;
  if (scratch_bytes < link_index_scratch_bytes(n, v)) return LINK_ERR_WORKSPACE;
  IndexScratch sc = carve(scratch, n, v);
  const int64_t tiles = vscan_tiles(n);              // <= the descriptor slots carved for max(n, v) cells
  hipStream_t st = S(stream);
  unsigned long long *pair = reinterpret_cast<unsigned long long *>(cell_pair);
  int64_t cc_threads = n > tiles ? n : tiles;
  if (cc_threads < 65536) cc_threads = 65536;        // (also the threads that walk the previous frame's block list)
  hipLaunchKernelGGL(k_cell_count_first, dim3(blocks_for(cc_threads, 256)), dim3(256), 0, st,
                     reinterpret_cast<const int4 *>(coords), n, *grid, reinterpret_cast<unsigned int *>(cell_pair), sc.vox_cell,
                     sc.vox_rank, sc.desc, tiles, sc.ticket, cell_blk, reinterpret_cast<const int4 *>(blk_coords), hdr);
  if (n == 0) {                                      // an empty frame: M = 0 (k_vox_scan has no tile to write the totals from)
    (void)hipMemsetAsync(hdr, 0, 4 * LINK_HDR_WORDS, st);
    (void)hipMemsetAsync(blk_start, 0, 4, st);
    return check_launch("link_index_build_first");
  }
  if (tiles <= VSCAN_RESIDENT)
    hipLaunchKernelGGL(k_vox_scan<false>, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, st, sc.vox_cell, n, *grid, pair, sc.desc,
                       sc.ticket, tiles, cell_blk, blk_start, blk_coords, counts, hdr);
  else
    hipLaunchKernelGGL(k_vox_scan<true>, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, st, sc.vox_cell, n, *grid, pair, sc.desc,
                       sc.ticket, tiles, cell_blk, blk_start, blk_coords, counts, hdr);
  int32_t *pb = pos_blk ? pos_blk : sc.pos_tmp;
  hipLaunchKernelGGL(k_place_first, dim3(blocks_for(n, 256)), dim3(256), 0, st, sc.vox_cell, sc.vox_rank, n, pair, sc.perm_tmp, pb,
                     vox_blk, idx_query);
  hipLaunchKernelGGL(k_sort_seg, dim3(blocks_for(n, 256)), dim3(256), 0, st, sc.perm_tmp, pb, blk_start, hdr, n,
                     reinterpret_cast<const int4 *>(coords), perm, reinterpret_cast<int4 *>(vox_sorted), pair,
                     reinterpret_cast<const int4 *>(blk_coords), *grid);
  return check_launch("link_index_build_first");
}

// The first half of link_index_build alone: which cells of the grid are occupied, in cell order -- sorted unique
// block coordinates, their counts and the cell table, without placing / ordering the voxels.  (Output sites of a
// site-creating convolution: the "voxels" are candidate rows with many duplicates, only the set matters.)
extern "C" int link_index_cells(const int32_t *coords, int64_t n, const link_grid_t *grid, uint32_t *cell_counts,
                                void *scratch, size_t scratch_bytes, int32_t *cell_blk, int32_t *blk_start,
                                int32_t *blk_coords, int32_t *counts, int32_t *hdr, void *stream) {
  if (n < 0 || n >= (1LL << 31) || !grid || !hdr) return LINK_ERR_ARG;
  int64_t v = 1;
  for (int a = 0; a < 4; a++) {
    if (grid->dim[a] <= 0) return LINK_ERR_ARG;
    v *= grid->dim[a];
    if (v >= (1LL << 30)) return LINK_ERR_ARG;
  }
  if (grid->s <= 0) return LINK_ERR_ARG;
  if (!cell_counts || !scratch || !cell_blk || !blk_start || !blk_coords || !counts || (n > 0 && !coords)) return LINK_ERR_ARG;
  if (scratch_bytes < link_index_scratch_bytes(n, v)) return LINK_ERR_WORKSPACE;
  IndexScratch sc = carve(scratch, n, v);
  const int64_t tiles = scan_tiles(v);
  hipStream_t st = S(stream);
  int64_t cc_threads = n > tiles ? n : tiles;
  if (cc_threads > (1 << 22)) cc_threads = (n > (1 << 22)) ? n : (1 << 22);
  hipLaunchKernelGGL(k_cell_count, dim3(blocks_for(cc_threads, 256)), dim3(256), 0, st,
                     reinterpret_cast<const int4 *>(coords), n, *grid, cell_counts, sc.vox_cell,
                     sc.vox_rank, sc.desc, tiles, sc.ticket);
  hipLaunchKernelGGL(k_cell_scan, dim3((unsigned)tiles), dim3(SCAN_THREADS), 0, st, cell_counts, v,
                     *grid, sc.desc, sc.ticket, tiles, cell_blk, sc.cell_start, blk_start, blk_coords, counts, hdr);
  return check_launch("link_index_cells");
}

// ---------------------------------------------------------------------------------------------
// neighbour map
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void kernel_offset(int r, int k, int &ox, int &oy, int &oz) {
  // get_kernel_offsets(r, 1, 1) (nn/utils/kernel.py:11-32): per axis arange(-r//2+1, r//2+1);
  // odd volume: x fastest (z outer); even volume: z fastest (x outer).
  int lo = -((r + 1) / 2) + 1;
  int a = k % r, b = (k / r) % r, c = k / (r * r);
  if ((r & 1) != 0) { ox = lo + a; oy = lo + b; oz = lo + c; }
  else { oz = lo + a; oy = lo + b; ox = lo + c; }
}

__global__ void __launch_bounds__(256) k_neighbor_map(const int4 *__restrict__ blk_coords,
                                                      const int32_t *__restrict__ cell_blk,
                                                      link_grid_t g, const int32_t *__restrict__ hdr,
                                                      int64_t m_cap, int r, int K, int sign,
                                                      int32_t *__restrict__ nbr) {  // sign = +-step
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t m = hdr ? (int64_t)hdr[LINK_HDR_M] : m_cap;
  if (m > m_cap) m = m_cap;
  if (t >= m * K) return;
  int64_t row = t / K;
  int k = (int)(t - row * K);
  int4 c = blk_coords[row];
  int ox, oy, oz;
  kernel_offset(r, k, ox, oy, oz);
  int32_t cell = cell_of(g, c.x + sign * ox, c.y + sign * oy, c.z + sign * oz, c.w);
  nbr[t] = (cell >= 0) ? cell_blk[cell] - 1 : -1;
}

extern "C" int link_neighbor_map(const int32_t *blk_coords, const int32_t *cell_blk,
                                 const link_grid_t *grid, const int32_t *hdr, int64_t m, int32_t r,
                                 int32_t step, int32_t transpose, int32_t *nbr, void *stream) {
  if (m < 0 || r <= 0 || r > 15 || step <= 0 || !grid) return LINK_ERR_ARG;
  if (m == 0) return LINK_OK;
  if (!blk_coords || !cell_blk || !nbr) return LINK_ERR_ARG;
  int K = r * r * r;
  hipLaunchKernelGGL(k_neighbor_map, dim3(blocks_for(m * K, 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const int4 *>(blk_coords), cell_blk, *grid, hdr, m, (int)r, K,
                     transpose ? -step : step, nbr);
  return check_launch("link_neighbor_map");
}

__global__ void __launch_bounds__(256) k_cell_table(const int4 *__restrict__ rows, int64_t m,
                                                    link_grid_t g, unsigned int *cell_blk, int32_t *hdr) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  int4 c = rows[i];
  int32_t cell = cell_of(g, c.x, c.y, c.z, c.w);
  if (cell < 0) { if (hdr) atomicOr(&hdr[LINK_HDR_STATUS], 1); return; }
  unsigned v = (unsigned)i + 1u;  // first (smallest) row wins, like the reference's insert-if-absent
  unsigned old = atomicCAS(&cell_blk[cell], 0u, v);
  while (old != 0u && old > v) {
    unsigned prev = atomicCAS(&cell_blk[cell], old, v);
    if (prev == old) break;
    old = prev;
  }
}

// undo link_cell_table_build on the same rows: the table is all zero again after m scattered stores instead of a memset
// over every cell (a LiDAR grid: 85 M cells = 340 MB for 150 k rows)
__global__ void __launch_bounds__(256) k_cell_table_clear(const int4 *__restrict__ rows, int64_t m, link_grid_t g,
                                                          unsigned int *cell_blk) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const int4 c = rows[i];
  const int32_t cell = cell_of(g, c.x, c.y, c.z, c.w);
  if (cell >= 0) cell_blk[cell] = 0u;
}

extern "C" int link_cell_table_clear(const int32_t *rows, int64_t m, const link_grid_t *grid, int32_t *cell_blk,
                                     void *stream) {
  if (m < 0 || !grid) return LINK_ERR_ARG;
  if (m == 0) return LINK_OK;
  if (!rows || !cell_blk) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_cell_table_clear, dim3(blocks_for(m, 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const int4 *>(rows), m, *grid, reinterpret_cast<unsigned int *>(cell_blk));
  return check_launch("link_cell_table_clear");
}

extern "C" int link_cell_table_build(const int32_t *rows, int64_t m, const link_grid_t *grid,
                                     int32_t *cell_blk, int32_t *hdr, void *stream) {
  if (m < 0 || !grid) return LINK_ERR_ARG;
  if (m == 0) return LINK_OK;
  if (!rows || !cell_blk) return LINK_ERR_ARG;
  hipLaunchKernelGGL(k_cell_table, dim3(blocks_for(m, 256)), dim3(256), 0, S(stream),
                     reinterpret_cast<const int4 *>(rows), m, *grid,
                     reinterpret_cast<unsigned int *>(cell_blk), hdr);
  return check_launch("link_cell_table_build");
}
