// link_amd/csrc/dense_batch.hip -- R_core of a BATCH of independent frames on the dense-cell layout as THREE launches, two of them
// persistent and queue-fed (include/link_amd.h section H; round 6, VERDICT round 5 "next 1").
//
// What it replaces.  A frame's R_core is three dependent launches (slot insert, K1 = fused pre_mix + modulate + per-cell sums,
// K2 = fused box sum + de-modulate); BASELINE.json configs[3] is "a batch of 8 independent frames", and until this round the
// product's answer to several frames per GPU was "the caller keeps three plans on three HIP streams" -- the overlap of the stages
// of different frames was whatever the hardware queues happened to interleave, and every frame paid two stream-ordered launch
// boundaries and the fill / drain of two grids.  Here the batch is ONE call:
//
//   stream C   k_dc_batch_insert   PERSISTENT, four single-wave workgroups per CU: (frame, chunk of 256 voxels) items off one cursor,
//                                  frames in order; the records are stored write-through (sc1); one arrival on ins_done[frame]
//                                  per item
//   stream A   k_dc_batch_k1       PERSISTENT, one 4-wave workgroup per CU: stages W / LayerNorm / theta parameters ONCE, then every
//                                  wave draws (frame, range of cells) items off its XCD's cursor: dc_k1_range (the body of the
//                                  stand-alone kernel) reading the insert's counts / records with sc1 loads and publishing S rows /
//                                  counts / sorted records with sc1 stores; one arrival on k1_done[frame] per item
//   stream B   k_dc_batch_k2       PERSISTENT, one 8-wave workgroup per CU: draws (frame, tile) items off its XCD's cursor (tiles of
//                                  a frame keep the XCD they have in the stand-alone kernel: halo planes stay in one L2), ONE relaxed
//                                  poll of k1_done[frame], then dc_k2q_body (the stand-alone kernel's body, every global read an sc1
//                                  load) on that tile
//
// No workgroup ever waits on a workgroup of its OWN kernel and every role draws its work off cursors, so whichever workgroups are
// resident finish the batch: insert waits on nothing, K1 on insert arrivals, K2 on K1 arrivals; launch order = dependence order.
// Co-residency is shaped, not required: LDS is handed out in 128 granules of 1 280 bytes per CU (tools/coresidency_probe.hip) -- K1
// padded to 65, K2 63, so one of each fills a CU and two K1 workgroups do not fit; registers per SIMD 208 + 2 x 128 + 32 of 512
// (tests/test_cpu_abi.py::test_batch_kernels_resource_shape reads them off the built code object); the insert has no LDS at all.
// Every spin is bounded (DC_BT_TIMEOUT_TICKS of the 100 MHz clock) and watches a shared error word: a violated assumption ends the
// call with LINK_BATCH_TIMEOUT in link_dc_batch_status, not with a hung GPU.
//
// Visibility inside the launches follows MI355X_MICROARCH.md "inter-workgroup visibility": producers store write-through (sc1)
// and have their stores acknowledged (vmcnt) before their arrival atomic; consumers read with sc1 loads (served by the L2, never by
// the CU's L1) -- no buffer_wbl2 / buffer_inv on the path.  The atomic counters of the insert (cnt) are device-scope atomics and live
// at the memory side.
//
// Results: bit for bit those of link_elk_core_dense_forward per frame (same device bodies, same arithmetic; the launch
// geometry -- cells per K1 item, z-segments of K2 -- does not enter any sum's order).  C = 64, cg = 32, cos / sin, r in {2, 3},
// coord_div = 1, no alpha, fp32 rows, slot capacity <= 352: what the quad-consumer K2 serves; LINK_ERR_ARG otherwise (the caller
// runs the frames one by one).  Measurements, dead ends and the timeline of a call: DESIGN.md section 4i.
#define DC_IO 0
#define DC_IO_NS dcb_f32
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "dense_gather.h"
#include "dense_io.h"

#ifndef DC_BT_MAX
#define DC_BT_MAX 48                 /* frames per launch set (the frame table travels as a kernel argument: 48 x 80 B of the 4 KB a launch takes) */
#endif
#ifndef DC_BT_TIMEOUT_TICKS
#define DC_BT_TIMEOUT_TICKS 200000000ull   /* 2 s of the 100 MHz s_memrealtime clock */
#endif
#ifndef DC_BT_PROF
#define DC_BT_PROF 0                /* 1 (python tools/mkvariant.py BTPROF "-DDC_BT_PROF=1" dense_batch.hip): every K1 / K2 item leaves a row of 100 MHz
                                       timestamps in the buffers given to link_dc_batch_set_debug (tools/batch_timeline.py) */
#endif
#ifndef DC_BT_INS_LEAD
#define DC_BT_INS_LEAD 1000         /* pacing of the insert: at most this many frames (+ the one in progress) ahead of K1.  OFF (1000): paced at 2 with two waves per CU the insert ran beside K1 / K2 all call long and the batch went 36.9 -> 45.0 us / frame (B = 24 x 2 sets); unpaced it floods the first ~200 us of a call and leaves the rest clean */
#endif
#ifndef DC_BT_INS_WAVES
#define DC_BT_INS_WAVES 4           /* single-wave insert workgroups per CU */
#endif
#ifndef DC_BT_RELEASE_FENCE
#define DC_BT_RELEASE_FENCE 0       /* 1: K1 publishes with an agent-scope release fence (buffer_wbl2) in front of its arrival as well
                                       -- belt and braces for A/B; every published table is stored write-through already */
#endif

namespace DC_IO_NS {
using namespace link;

#include "dense_k1_impl.h"
#include "dense_k2_cfg.h"
#include "dense_gather_quad_impl.h"

static_assert(DC_ST_AUX == 16, "the S rows must leave the L2 (sc1) for the in-launch hand-off K1 -> K2");

struct dc_bt_frame_t {               // one frame's buffers (device pointers), 80 bytes
  const void *feats; const int4 *coords; int4 *slots; uint32_t *cnt; int32_t *cell_n; int32_t *vcell; float *S; int32_t *hdr;
  void *out; int64_t n;
};
struct dc_bt_frames_t { dc_bt_frame_t f[DC_BT_MAX]; };
struct dc_bt_par_t {
  const float *w_pre, *pre_ln_w, *pre_ln_b, *w_pos, *ln_w, *ln_b;
  int cg; float eps;
  unsigned long long *dbg1, *dbg2;     // DC_BT_PROF rows (word 0 = rows appended so far, word 1 = capacity), or NULL
};
// one row of eight 64-bit words per item, at the item's own place `r` (no returning atomic: a wave that waits for one drains every
// store it has in flight, and the first version of these timers measured mostly that)
__device__ __forceinline__ void bt_row(unsigned long long *dbg, unsigned long long r, unsigned long long a, unsigned long long b,
                                       unsigned long long c, unsigned long long d, unsigned long long e, unsigned long long f) {
  if (!DC_BT_PROF || !dbg) return;
  if (r + 1 >= dbg[1]) return;
  unsigned long long *o = dbg + 8 * (r + 1);
  o[0] = a; o[1] = b; o[2] = c; o[3] = d; o[4] = e; o[5] = f; o[6] = 1;
}
// sync words of one call (int32, zeroed before the launches): every counter on its own 64-byte line
__host__ __device__ constexpr int bt_err() { return 0; }
__host__ __device__ constexpr int bt_cursor(int xcd) { return 16 * (1 + xcd); }          // K2's item cursors, one per XCD queue
__host__ __device__ constexpr int bt_k1cur(int xcd) { return 16 * (9 + xcd); }          // K1's item cursors, one per XCD slab
__host__ __device__ constexpr int bt_inscur() { return 16 * 17; }                       // the insert's item cursor
__host__ __device__ constexpr int bt_ins(int f) { return 16 * (18 + 2 * f); }           // arrivals of the frame's insert chunks
__host__ __device__ constexpr int bt_k1(int f) { return 16 * (19 + 2 * f); }            // arrivals of the frame's K1 ranges
constexpr int BT_SYNC_WORDS = 16 * (18 + 2 * DC_BT_MAX);

// Lane 0 polls *p until it reaches `target` (relaxed agent-scope loads: sc1, L2-served), sleeping between polls, giving up when the
// call's error word is set or after DC_BT_TIMEOUT_TICKS; returns (wave-uniform) whether the target was reached.
template <typename P>
__device__ __forceinline__ bool bt_wait_ge(P sync, int word, int target, const int nap = 8) {
  int ok = 1;
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int spins = 0;
    while (__hip_atomic_load(&sync[word], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      if (nap > 8) __builtin_amdgcn_s_sleep(64); else __builtin_amdgcn_s_sleep(8);
      if ((++spins & 15) == 0) {
        if (__hip_atomic_load(&sync[bt_err()], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > DC_BT_TIMEOUT_TICKS) {
          __hip_atomic_store(&sync[bt_err()], 1 + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
  }
  return __builtin_amdgcn_readfirstlane(ok) != 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// slot insert of every frame of the batch: workgroup -> (frame, chunk); write-through records; one arrival per workgroup
// ---------------------------------------------------------------------------------------------------------------------------------
// PERSISTENT too: single-wave workgroups, four per CU (a wave per SIMD), 34 registers, NO LDS and no barrier -- a CU's 128 LDS
// granules are taken by one K1 + one K2 workgroup, so a kernel that asks for a single byte of LDS finds no room beside them.  (As an
// ordinary grid of 9 400 workgroups the batch's insert took every free wave slot and register of the chip at the start of a call,
// and K1 / K2 workgroups of the same call found no room until it had drained: tools/batch_timeline.py, first version -- the first
// frame's last K1 range ended 176 us into the call.)  A resident wave per SIMD inserts a frame in ~10 us, twice as fast as K1
// consumes them.
struct dc_bt_ins_args_t { dc_bt_frames_t fr; link_dc_grid_t g; int nframes, wpf; int32_t *sync; };   // the kernel's only argument
__global__ void __launch_bounds__(64) k_dc_batch_insert(dc_bt_ins_args_t args_by_value) {
  (void)args_by_value;
#if defined(__HIP_DEVICE_COMPILE__)
  // Items (frame, chunk of 256 voxels) off ONE cursor, frames in order; an item's arrival goes to ins_done[frame] (target: the
  // chunks of a frame).  Drawn, not dealt: whichever of this kernel's waves are resident insert the whole batch -- nothing waits
  // for a workgroup that has not found room yet.  (Arguments through the laundered kernarg pointer, as in the K2 role: hoisted out
  // of the loop the grid's fields and the frame's descriptors cost registers this kernel does not have -- it must fit the 48 per
  // SIMD the other two leave.)
  typedef const __attribute__((address_space(4))) dc_bt_ins_args_t *args_ptr_t;
  for (;;) {
    args_ptr_t a = (args_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    int t = 0;
    if (threadIdx.x == 0) t = __hip_atomic_fetch_add(&a->sync[bt_inscur()], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t = __builtin_amdgcn_readfirstlane(t);
    const int wpf = a->wpf;
    if (t >= a->nframes * wpf) break;
    const int f = t / wpf, j = t - f * wpf;
    // Paced: a frame is inserted when K1 has reached the frame DC_BT_INS_LEAD before it.  Unpaced, the insert of all 24 frames ran
    // flat out through the first ~200 us of a call -- 2.4 M counter atomics and as many scattered 16-byte write-through stores -- and
    // every K1 item that started in that window took 80-90 us instead of 27 (tools/batch_timeline.py, "first-loads wait" 57 us).
    if (f >= DC_BT_INS_LEAD && !bt_wait_ge(a->sync, bt_k1(f - DC_BT_INS_LEAD), 1, 64)) break;
    const link_dc_grid_t g = a->g;
    // the statements of dc_index_body (dense_common.h) on 32-bit voxel numbers, one pass of 64 voxels at a time: that body's 64-bit
    // grid-stride loop cost 42 registers here (49-50 unrolled), and this kernel has 40 (208 + 2 x 128 + 40 of a SIMD's 512)
    const int n = (int)a->fr.f[f].n;
    const int4 *__restrict__ coords = a->fr.f[f].coords;
    int32_t *__restrict__ vcell = a->fr.f[f].vcell;
    int32_t *__restrict__ hdr = a->fr.f[f].hdr;
    const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(a->fr.f[f].slots, (uint32_t)((int64_t)g.vp * g.k * 16));
    const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(a->fr.f[f].cnt, (uint32_t)(g.vp * 4));
    if (j == 0 && threadIdx.x == 0) hdr[LINK_HDR_NVALID] = n;
    _Pragma("unroll 1") for (int k = 0; k < 4; k++) {
      const int v = (4 * j + k) * 64 + (int)threadIdx.x;
      if (v >= n) break;
      const int4 rc = coords[v];
      const unsigned ux = (unsigned)(floordiv(rc.x, g.s) - g.lo[0]), uy = (unsigned)(floordiv(rc.y, g.s) - g.lo[1]);
      const unsigned uz = (unsigned)(floordiv(rc.z, g.s) - g.lo[2]), ub = (unsigned)(rc.w - g.lo[3]);
      const bool inside = ux < (unsigned)g.dim[0] && uy < (unsigned)g.dim[1] && uz < (unsigned)g.dim[2] && ub < (unsigned)g.dim[3];
      if (!inside) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 1);
      const int pcell = inside ? dc_cell(g, (int)ux, (int)uy, (int)uz, (int)ub) : 0;
      const int rank = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_cnt, pcell ? (uint32_t)pcell * 4u : DC_OOB, 0, 0);
      const bool full = pcell != 0 && rank >= g.k;
      if (full) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 2);
      const bool keep = pcell != 0 && !full;
      st16i_c<true>(r_slots, keep ? dc_slot(g, pcell, rank) * 16u : DC_OOB, make_int4(rc.x, rc.y, rc.z, v));
      vcell[v] = keep ? pcell : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's write-through records (and its counter atomics) have left
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&a->sync[bt_ins(f)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K1 role: parameters staged once, then (frame, range of cells) items in frame order
// ---------------------------------------------------------------------------------------------------------------------------------
// A frame's cells are cut into `nranges` ranges of `cpw` cells, dealt to eight per-XCD cursors as contiguous slabs of `per`
// ranges; item t of XCD x's cursor is range x * per + t % per of frame t / per.  A wave pulls items off its XCD's cursor until
// the cursor runs past the batch.  Every global round trip costs 1-3 us under load, so none of the loop's own is left exposed:
//   * the NEXT item is drawn (returning atomic) before the current range is worked on -- its result is there when it is needed;
//   * an item's arrival on k1_done is posted when the NEXT item's first loads have been waited for (memory operations retire in
//     order: the item's write-through stores have been acknowledged by then) instead of behind a drain of its own;
//   * the frame's insert arrivals are read with the item's first loads; only a wave that finds them incomplete polls.
// (History, same box, 24 frames x 2 sets: wave w owning range w of every frame 48.7 us / frame -- K2 of a frame waits for the frame's
// LAST range, so the batch advanced at the pace of its slowest wave while the fast ones ran frames ahead for nothing; dynamic items
// with blocking draws, a drain per item and eight slab probes at every frame's end 64.8, two / four ranges per wave and frame 83.6 /
// 116: ~17 us of exposed round trips per item.  Three plans on three streams: 34.7.)
template <int OP, int NB>
__global__ void __launch_bounds__(64 * DC_K1_NW, DC_K1_WAVES) k_dc_batch_k1(dc_bt_frames_t fr, dc_bt_par_t p, link_dc_grid_t g, int nframes,
                                                                           int cpw, int nranges, int wpf, int32_t *__restrict__ sync) {
  constexpr int C = 64;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63;
  bool w_big = false, th_big = false;
  dc_k1_stage<C, OP>(smem_raw, p.w_pre, p.pre_ln_w, p.pre_ln_b, p.w_pos, nullptr, p.cg, 1.0f, g, tid, w_big, th_big);
  w_big = DC_K1_SPLIT ? (__syncthreads_or(w_big) != 0) : (__syncthreads(), false);
  const bool th_slow = DC_THETA_BOUND ? __syncthreads_or(th_big) != 0 : false;
  const int Vi = g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  const int x = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u);
  const int per = (nranges + 7) >> 3;                  // ranges of one XCD slab
  const int lo = x * per;
  const int mine = (lo + per < nranges ? lo + per : nranges) - lo;     // ranges of this XCD's slab (<= 0: none)
  if (mine <= 0) return;
  const int total = nframes * mine;
  int32_t *cur = &sync[bt_k1cur(x)];
  int t_next = 0;
  if (lane == 0) t_next = __hip_atomic_fetch_add(cur, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  int pend_f = -1;                                     // frame of the item whose arrival has not been posted yet
  int ins_known = -1;                                  // frames 0 .. ins_known have all their insert arrivals
  for (;;) {
    const int t = __builtin_amdgcn_readfirstlane(t_next);
    if (t >= total) break;
    const unsigned long long tp0 = DC_BT_PROF ? __builtin_amdgcn_s_memrealtime() : 0;
    const int f = t / mine, i = lo + (t - f * mine);
    if (lane == 0) t_next = __hip_atomic_fetch_add(cur, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the item after this one
    const dc_bt_frame_t &F = fr.f[f];
    // insert arrivals of this frame (if not known yet) and of the NEXT one: a wave works on about one range per frame, so without
    // the look-ahead every item would begin with "are the records there?" -> first loads, two dependent round trips
    int ins_a = wpf, ins_b = wpf;
    if (lane == 0) {
      if (f > ins_known) ins_a = __hip_atomic_load(&sync[bt_ins(f)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (f + 1 < nframes && f + 1 > ins_known) ins_b = __hip_atomic_load(&sync[bt_ins(f + 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const int c_begin = i * cpw;
    const int c_end = (c_begin + cpw < Vi) ? c_begin + cpw : Vi;
    int pc_f, nv_f;
    int4 rf0, rf1, rf2, rf3;
    if (f <= ins_known) dc_k1_prefetch<true>(pc_f, nv_f, rf0, rf1, rf2, rf3, g, F.slots, F.cnt, c_begin, c_end, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // ... which the range needs at once; everything older has retired too
    if (pend_f >= 0 && lane == 0) __hip_atomic_fetch_add(&sync[bt_k1(pend_f)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    pend_f = f;
    if (f > ins_known) {                               // not known in advance (first frame, or the insert is only just ahead)
      if (__builtin_amdgcn_readfirstlane(ins_a) < wpf && !bt_wait_ge(sync, bt_ins(f), wpf)) return;
      ins_known = f;
      dc_k1_prefetch<true>(pc_f, nv_f, rf0, rf1, rf2, rf3, g, F.slots, F.cnt, c_begin, c_end, lane);
    }
    if (f + 1 < nframes && f + 1 > ins_known && __builtin_amdgcn_readfirstlane(ins_b) >= wpf) ins_known = f + 1;
    if (i == 0 && lane == 0) {                         // publish the frame's status word (collected by the insert's atomics)
      F.hdr[LINK_HDR_STATUS] = __hip_atomic_load(&F.hdr[LINK_HDR_STATUS_ACC], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&F.hdr[LINK_HDR_STATUS_ACC], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const unsigned long long tp1 = DC_BT_PROF ? __builtin_amdgcn_s_memrealtime() : 0;
    dc_k1_range<C, OP, NB, false, true>(smem_raw, F.feats, F.slots, F.cnt, F.cell_n, p.w_pre, 1.0f, p.eps, F.n, g, false, F.S, nullptr,
                                        nullptr, c_begin, c_end, pc_f, nv_f, rf0, rf1, rf2, rf3, w_big, th_slow, i, 0, 0);
    if (DC_BT_PROF && lane == 0) bt_row(p.dbg1, (unsigned long long)f * nranges + i, (unsigned long long)f, (unsigned long long)i, tp0, tp1, __builtin_amdgcn_s_memrealtime(), (unsigned long long)(blockIdx.x * 4 + (tid >> 6)));
  }
  // the last item's arrival: every table K2 reads was stored write-through, so once the stores are acknowledged they are in memory
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if DC_BT_RELEASE_FENCE
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  if (pend_f >= 0 && lane == 0) __hip_atomic_fetch_add(&sync[bt_k1(pend_f)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// K2 role: (frame, tile) items off per-XCD cursors
// ---------------------------------------------------------------------------------------------------------------------------------
struct dc_bt_k2_args_t {                               // the K2-role kernel's ONLY argument: the kernarg segment is this struct
  dc_bt_frames_t fr;
  dc_bt_par_t p;
  link_dc_grid_t g;
  int nframes, txn, tyn, zsplit, nwg, k1_target;
  int32_t *sync;
};

// Item t of XCD x's cursor is tile j = t % per of frame t / per, run as workgroup number j * 8 + x of the stand-alone kernel (which
// maps it to L = x * per + j: a frame's tiles keep the XCD -- the L2 -- they have there).  The loop's control costs no exposed round
// trip either: the mapper wave (it lays out voxel maps two planes ahead and is idle most of a plane step) draws the NEXT item before
// the tile starts and, once its own plane loop is through, looks at the next item's k1_done -- by the tile's closing barrier the
// answer is in LDS; only when K1 of that frame has really not arrived does it poll.  No acquire: K1 stored write-through, the body
// reads with sc1 loads (COH).
template <int OP, int R>
__global__ void __launch_bounds__(256 + 64 * (DC_K2Q_CW + (DC_K2Q_PMAP ? 1 : 0)), 4) k_dc_batch_k2(dc_bt_k2_args_t args_by_value) {
  // Nothing may stay live in scalar registers from one item to the next: the tile body (dc_k2q_body) runs at the scalar-register
  // limit on its own, and as loop invariants the block's parameters, the grid and the geometry (~40 scalars) were kept across it
  // -- spilled into vector registers, then 200+ bytes of scratch, whose traffic breaks the producers' counted vmcnt waits.  So the
  // arguments are read through the kernarg segment pointer, passed through an opaque asm every iteration (the loads cannot be
  // hoisted: 118-123 registers, no scratch, like the stand-alone kernel), the loop control lives in LDS, and the thread number the
  // body derives its roles from is opaque per iteration as well.
  (void)args_by_value;
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(DC_K2Q_PMAP == 1, "the mapper wave draws the items");
  typedef const __attribute__((address_space(4))) dc_bt_k2_args_t *args_ptr_t;
  constexpr unsigned CTL_THREAD = 256 + 64 * DC_K2Q_CW;  // lane 0 of the mapper wave
  __shared__ int s_ctl[2];                             // [0] frame of the item (-1: done), [1] bid for dc_k2q_body
  const int x = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u);
  int t_next = 0;                                      // (control thread only) the item after the current one
  if (threadIdx.x == CTL_THREAD) {
    args_ptr_t a = (args_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    int32_t *sy = a->sync;
    const int per = (a->nwg + 7) >> 3;
    if (DC_BT_PROF)                                    // when this workgroup became resident (row behind the items' rows; kind 7)
      bt_row(a->p.dbg2, (unsigned long long)a->nframes * (8 * per) + blockIdx.x, 9999, 0, __builtin_amdgcn_s_memrealtime(), 0, 0, blockIdx.x);
    const int t = __hip_atomic_fetch_add(&sy[bt_cursor(x)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    t_next = __hip_atomic_fetch_add(&sy[bt_cursor(x)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int f = t < a->nframes * per ? t / per : -1;
    if (f >= 0 && !bt_wait_ge(sy, bt_k1(f), a->k1_target)) f = -1;
    s_ctl[0] = f; s_ctl[1] = f >= 0 ? (t - f * per) * 8 + x : 0;
  }
  __syncthreads();
  for (;;) {
    args_ptr_t a = (args_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a));
    const int f = __builtin_amdgcn_readfirstlane(s_ctl[0]), bid = __builtin_amdgcn_readfirstlane(s_ctl[1]);
    if (f < 0) break;
    const unsigned long long tq0 = DC_BT_PROF ? __builtin_amdgcn_s_memrealtime() : 0;
    __syncthreads();                                   // everyone has read the item: the control thread may lay out the next
    {
      const link_dc_grid_t g = a->g;
      unsigned tidx = threadIdx.x;                     // nothing derived from the thread number may be hoisted in front of the loop (lane roles,
      asm volatile("" : "+v"(tidx));                   // tile columns, LDS addresses were live across every role of every item)
      dc_k2q_body<OP, R, false, true>(a->fr.f[f].S, a->fr.f[f].cell_n, a->fr.f[f].slots, a->p.w_pos, nullptr, a->p.ln_w, a->p.ln_b, a->p.cg,
                                      1.0f, a->p.eps, a->fr.f[f].n, g, a->txn, a->tyn, a->zsplit, a->nwg, a->fr.f[f].out, nullptr, bid, tidx);
    }
    if (threadIdx.x == CTL_THREAD) {                   // the mapper's plane loop is through: the next item
      const unsigned long long tq1 = DC_BT_PROF ? __builtin_amdgcn_s_memrealtime() : 0;
      int32_t *sy = a->sync;
      const int per = (a->nwg + 7) >> 3;
      const int t = t_next;
      int fn = t < a->nframes * per ? t / per : -1;
      if (fn >= 0) {
        t_next = __hip_atomic_fetch_add(&sy[bt_cursor(x)], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!bt_wait_ge(sy, bt_k1(fn), a->k1_target)) fn = -1;
      }
      s_ctl[0] = fn; s_ctl[1] = fn >= 0 ? (t - fn * per) * 8 + x : 0;
      if (DC_BT_PROF) bt_row(a->p.dbg2, (unsigned long long)f * (8 * per) + bid, (unsigned long long)f, (unsigned long long)bid, tq0, tq1, __builtin_amdgcn_s_memrealtime(), (unsigned long long)blockIdx.x);
    }
    __syncthreads();                                   // the tile's LDS images are free again, the next item is laid out
  }
#endif
}

}  // namespace DC_IO_NS

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
using namespace link;
using namespace dcb_f32;

static constexpr int BT_RING = 4;
// Do two HIP streams sit on ONE hardware queue?  The runtime multiplexes streams onto a few queues (GPU_MAX_HW_QUEUES, 4 by default);
// kernels of two streams that share a queue still start side by side -- but an EVENT RECORD on one of them is a packet with the barrier
// bit, and every later packet of that queue, whichever stream it belongs to, waits behind it.  A call of the batch entry point records
// events behind its role kernels; with the pre_mix and the gather stream on one queue, the next call's pre_mix kernel sits behind the
// gather kernel's completion event and the calls run one after the other (tools/batch_overlap.py: 48 instead of 35 us / frame).
// The test: a 150 us spin kernel + an event record on `x`, then a stamp kernel on `y`; *us = y's start - x's start (a few us either
// way on separate queues); false = a HIP call failed.
__global__ void k_dc_batch_spin(unsigned long long *stamp, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) *stamp = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(16);
}
static bool bt_queue_delay_us(hipStream_t x, hipStream_t y, unsigned long long *scratch /* device, 2 words */, hipEvent_t ev, double *us) {
  unsigned long long h[2] = {0, 0};
  *us = 0.0;
  if (hipStreamSynchronize(x) != hipSuccess || hipStreamSynchronize(y) != hipSuccess) return false;
  hipLaunchKernelGGL(k_dc_batch_spin, dim3(1), dim3(64), 0, x, scratch, 15000);
  if (hipEventRecord(ev, x) != hipSuccess) return false;
  hipLaunchKernelGGL(k_dc_batch_spin, dim3(1), dim3(64), 0, y, scratch + 1, 0);
  if (hipStreamSynchronize(x) != hipSuccess || hipStreamSynchronize(y) != hipSuccess ||
      hipMemcpy(h, scratch, sizeof h, hipMemcpyDeviceToHost) != hipSuccess)
    return false;
  *us = ((double)h[1] - (double)h[0]) / 100.0;
  return true;
}
static bool bt_share_queue(hipStream_t x, hipStream_t y, unsigned long long *scratch, hipEvent_t ev) {
  double d = 0.0;
  return !bt_queue_delay_us(x, y, scratch, ev, &d) || d > 75.0;      // (a failed probe counts as shared: the candidate is passed over)
}
struct link_dc_batch {
  int device, cus;
  hipStream_t sa, sb, sc;                              // K1 role, K2 role, insert
  std::vector<hipStream_t> pool;                       // default-priority streams found on pairwise DIFFERENT hardware queues (sa, sb are two of them)
  int ia, ib;                                          // sa = pool[ia], sb = pool[ib]
  std::vector<std::pair<hipStream_t, unsigned>> callers;   // caller streams seen so far -> bit i: shares a hardware queue with pool[i]
  unsigned long long *probe_scratch;                   // device, 2 words
  hipEvent_t probe_ev;
  int32_t *sync;                                       // BT_RING x BT_SYNC_WORDS
  hipEvent_t ev_in[BT_RING], ev_ms[BT_RING], ev_a[BT_RING], ev_c[BT_RING], ev_out[BT_RING];
  std::vector<const void *> bufs[BT_RING];             // S pointers of the frames the ring slot's call worked on
  bool used[BT_RING];
  long long calls;
  unsigned long long *dbg1, *dbg2;                     // DC_BT_PROF rows (link_dc_batch_set_debug)
};

extern "C" int link_dc_batch_create(link_dc_batch_t **out) {
  if (!out) return LINK_ERR_ARG;
  link_dc_batch *c = new link_dc_batch();
  hipDeviceProp_t pr;
  if (hipGetDevice(&c->device) != hipSuccess || hipGetDeviceProperties(&pr, c->device) != hipSuccess) { delete c; return LINK_ERR_LAUNCH; }
  c->cus = pr.multiProcessorCount;
  // Launch order inside a call is insert -> pre_mix -> gather, the order of the dependences.
  int pr_lo = 0, pr_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);                 // numerically: least = greatest number
  // The insert's stream has the HIGHEST priority: its waves must find their slot before the two roles that wait for them fill the CUs.
  // The pre_mix and the gather stream must sit on DIFFERENT hardware queues, and on other queues than the insert's and the caller's: an
  // event record is a barrier packet for its whole queue, so two of these on one queue run the calls one after the other (48 instead
  // of 35 us / frame; the runtime gave two streams created back to back the same queue in five contexts out of six).  (This removes
  // ONE cause of calls running one after the other; tools/batch_overlap.py still finds contexts in that state whose streams pass every
  // test here -- rocprofv3 queue ids, tools/rocpd_batch_gaps.py: it went with the pre_mix queue's id equal to the insert's or the
  // caller's modulo 4, a dispatch-pipe relation this code has no handle on.  A caller that cares measures: DESIGN.md 4i.)  Candidates are
  // created until up to four sit on pairwise different queues (bt_share_queue); the rejected ones are destroyed afterwards -- alive,
  // they keep their queue's use count up and steer the next candidate elsewhere.
  bool ok = hipStreamCreateWithPriority(&c->sc, hipStreamNonBlocking, pr_hi) == hipSuccess &&
            hipMalloc(reinterpret_cast<void **>(&c->probe_scratch), 16) == hipSuccess &&
            hipEventCreateWithFlags(&c->probe_ev, hipEventDisableTiming) == hipSuccess;
  std::vector<hipStream_t> rejected;
  const bool dbg_place = getenv("LINK_DC_BATCH_DEBUG") != nullptr;
  for (int cand = 0; ok && cand < 12 && c->pool.size() < 4; cand++) {
    hipStream_t s_ = nullptr;
    if (hipStreamCreateWithFlags(&s_, hipStreamNonBlocking) != hipSuccess) { ok = false; break; }
    bool bad = bt_share_queue(c->sc, s_, c->probe_scratch, c->probe_ev);
    for (size_t i = 0; i < c->pool.size() && !bad; i++) bad = bt_share_queue(c->pool[i], s_, c->probe_scratch, c->probe_ev);
    (bad ? rejected : c->pool).push_back(s_);
  }
  while (ok && c->pool.size() < 2 && !rejected.empty()) { c->pool.push_back(rejected.back()); rejected.pop_back(); }   // (not two clean ones to be had: slower, not wrong)
  for (hipStream_t s_ : rejected) (void)hipStreamDestroy(s_);
  ok = ok && c->pool.size() >= 2;
  if (ok) { c->ia = 0; c->ib = 1; c->sa = c->pool[0]; c->sb = c->pool[1]; }
  if (dbg_place)
    fprintf(stderr, "link_dc_batch_create: %zu streams on hardware queues of their own, %zu candidates rejected\n",
            c->pool.size(), rejected.size());
  ok = ok && hipMalloc(reinterpret_cast<void **>(&c->sync), sizeof(int32_t) * BT_RING * BT_SYNC_WORDS) == hipSuccess &&
            hipMemset(c->sync, 0, sizeof(int32_t) * BT_RING * BT_SYNC_WORDS) == hipSuccess;
  for (int i = 0; i < BT_RING && ok; i++) {
    ok = hipEventCreateWithFlags(&c->ev_in[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_ms[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_a[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_c[i], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_out[i], hipEventDisableTiming) == hipSuccess;
    c->used[i] = false;
  }
  c->calls = 0;
  c->dbg1 = c->dbg2 = nullptr;
  if (!ok) { (void)hipGetLastError(); delete c; return LINK_ERR_LAUNCH; }   // (a failed create leaks what it had made: process-fatal anyway)
  *out = c;
  return LINK_OK;
}

extern "C" int link_dc_batch_destroy(link_dc_batch_t *c) {
  if (!c) return LINK_OK;
  (void)hipStreamSynchronize(c->sa); (void)hipStreamSynchronize(c->sb); (void)hipStreamSynchronize(c->sc);
  for (int i = 0; i < BT_RING; i++) {
    (void)hipEventDestroy(c->ev_in[i]); (void)hipEventDestroy(c->ev_ms[i]); (void)hipEventDestroy(c->ev_a[i]); (void)hipEventDestroy(c->ev_c[i]);
    (void)hipEventDestroy(c->ev_out[i]);
  }
  (void)hipFree(c->sync);
  (void)hipFree(c->probe_scratch);
  (void)hipEventDestroy(c->probe_ev);
  for (hipStream_t s_ : c->pool) (void)hipStreamDestroy(s_);
  (void)hipStreamDestroy(c->sc);
  delete c;
  return LINK_OK;
}

// Profiling hook (tools only): device buffers of 8-word rows the K1 / K2 items append their timestamps to -- word 0 of a buffer = rows
// so far (zeroed by the caller), word 1 = capacity in rows; honoured by a -DDC_BT_PROF=1 build, ignored otherwise.
extern "C" int link_dc_batch_set_debug(link_dc_batch_t *c, uint64_t *k1_rows, uint64_t *k2_rows) {
  if (!c) return LINK_ERR_ARG;
  c->dbg1 = reinterpret_cast<unsigned long long *>(k1_rows);
  c->dbg2 = reinterpret_cast<unsigned long long *>(k2_rows);
  return DC_BT_PROF ? LINK_OK : 1;
}

// The same test for any two streams of the caller (a serving loop that keeps frames in flight on several streams wants them on hardware
// queues of their own: two of bench.py's six candidate triples ran at the single-stream rate, tools/stream_placement.py).
// *delay_us: a kernel on `b` behind a 150 us kernel + event record on `a` -- ~5 = separate queues, >= 150 = one queue.  Synchronises both.
extern "C" int link_streams_share_queue(void *a, void *b, double *delay_us) {
  if (!delay_us) return LINK_ERR_ARG;
  unsigned long long *st = nullptr;
  hipEvent_t ev;
  if (hipMalloc(reinterpret_cast<void **>(&st), 16) != hipSuccess) return LINK_ERR_LAUNCH;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipFree(st); return LINK_ERR_LAUNCH; }
  const bool ok = bt_queue_delay_us(S(a), S(b), st, ev, delay_us);
  (void)hipEventDestroy(ev);
  (void)hipFree(st);
  if (!ok) { (void)hipGetLastError(); return LINK_ERR_LAUNCH; }
  return LINK_OK;
}

// Diagnostic (tools, bench.py): delays_us[6] = the test above for (pre_mix -> gather), (pre_mix -> insert), (gather -> insert),
// (caller -> pre_mix), (caller -> gather), (caller -> insert) streams: ~10 = separate hardware queues, >= 150 = one queue.
extern "C" int link_dc_batch_probe_streams(link_dc_batch_t *c, hipStream_t caller, double *delays_us /* host [12] */) {
  if (!c || !delays_us) return LINK_ERR_ARG;
  unsigned long long *st = nullptr;
  hipEvent_t ev;
  if (hipMalloc(reinterpret_cast<void **>(&st), 16) != hipSuccess) return LINK_ERR_LAUNCH;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { (void)hipFree(st); return LINK_ERR_LAUNCH; }
  hipStream_t xs[12] = {c->sa, c->sa, c->sb, caller, caller, caller, c->sb, c->sc, c->sc, c->sa, c->sb, c->sc},
              ys[12] = {c->sb, c->sc, c->sc, c->sa, c->sb, c->sc, c->sa, c->sa, c->sb, caller, caller, caller};
  bool ok = true;
  for (int i = 0; i < 12; i++) ok = bt_queue_delay_us(xs[i], ys[i], st, ev, &delays_us[i]) && ok;
  (void)hipEventDestroy(ev);
  (void)hipFree(st);
  if (!ok) { (void)hipGetLastError(); return LINK_ERR_LAUNCH; }
  return LINK_OK;
}

// status of the calls made so far (synchronises the context's streams): out[0] = first non-zero error word of the ring (0 = none;
// 1 + the sync word a spin gave up on), out[1] = calls made
extern "C" int link_dc_batch_status(link_dc_batch_t *c, int32_t *out) {
  if (!c || !out) return LINK_ERR_ARG;
  if (hipStreamSynchronize(c->sa) != hipSuccess || hipStreamSynchronize(c->sb) != hipSuccess || hipStreamSynchronize(c->sc) != hipSuccess)
    return LINK_ERR_LAUNCH;
  out[0] = 0;
  out[1] = (int32_t)c->calls;
  for (int i = 0; i < BT_RING; i++) {
    int32_t e = 0;
    if (hipMemcpy(&e, c->sync + (size_t)i * BT_SYNC_WORDS + bt_err(), sizeof(e), hipMemcpyDeviceToHost) != hipSuccess) return LINK_ERR_LAUNCH;
    if (e && !out[0]) out[0] = e;
  }
  return out[0] ? LINK_BATCH_TIMEOUT : LINK_OK;
}

template <int OP, int R>
static int batch_launch(link_dc_batch *c, int q, const dc_bt_frames_t &fr, const dc_bt_par_t &p, const link_dc_grid_t &g, const link_elk_desc_t &d,
                        int nframes, int64_t nmax) {
  using K1 = dc_k1_cfg<64, OP>;
  using KQ = dc_k2q_cfg<OP, R>;
  using KG = typename dc_k2_cfg<OP, R>::G;
  // LDS is handed out in 128 granules of 1 280 bytes per CU (tools/coresidency_probe.hip: 81 152 + 81 920 bytes share a CU, 81 152 +
  // 82 048 do not; 82 944 + 80 304 do, 82 944 + 80 896 do not).  With their static LDS (K1: 256 B behind __syncthreads_or; K2: 16 B of
  // loop control) K1 is padded to 65 granules and K2 takes 63: one of each fills a CU, two K1 workgroups do not fit (the mix bench.py's
  // stream geometry was tuned to -- k1_lds_pad 2 048).  Two K2 workgroups DO fit where no K1 workgroup sits; nothing depends on that
  // not happening: every role draws its work from cursors, so whichever workgroups are resident finish the batch.
  // (First versions padded K2 so that two of them would not fit -- 82 400 bytes = 65 granules next to K1's 64: the pair did not fit,
  // the K2 role became resident as the K1 role's workgroups left, and the batch ran its two roles one after the other: 44-48 us / frame.)
  constexpr int LDS_GRAN = 1280, LDS_GRANS = 128, K1_STATIC = 256, K2_STATIC = 16;
  constexpr int k1_lds = 65 * LDS_GRAN - K1_STATIC;
  constexpr int k2_lds = KQ::LDS_BYTES;
  static_assert(k1_lds >= K1::LDS_BYTES && (k2_lds + K2_STATIC + LDS_GRAN - 1) / LDS_GRAN <= LDS_GRANS - 65, "LDS shaping of the two persistent roles");
  int32_t *sync = c->sync + (size_t)q * BT_SYNC_WORDS;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  const int k1_wgs = c->cus;
  // ranges per K1 wave and frame (1: a range is ~6.6 tiles of 16 voxels on cfg2); LINK_DC_BATCH_RPW (experiments only) cuts finer
  static const int rpw = [] { const char *e = getenv("LINK_DC_BATCH_RPW"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 16 ? 16 : v); }();
  int cpw = (int)((vi + (int64_t)k1_wgs * K1::NW * rpw - 1) / ((int64_t)k1_wgs * K1::NW * rpw));
  if (cpw < 1) cpw = 1;
  const int k1_target = (int)((vi + cpw - 1) / cpw);    // ranges of a frame
  const int txn = (g.dim[0] + KG::TX - 1) / KG::TX, tyn = (g.dim[1] + KG::TY - 1) / KG::TY;
  // z-segments of a K2 tile: 1 = whole columns.  The stream geometry cuts columns in two so that ONE frame's tiles fill the chip; here
  // the K2 role's workgroups draw tiles of several frames, and whole columns are 5 % fewer plane steps (B = 32 x 2 sets, one box: 33.9
  // us / frame against 34.7 with two segments, 35.4 with three).  LINK_DC_BATCH_ZSPLIT (experiments only) overrides.
  static const int zs_env = [] { const char *e = getenv("LINK_DC_BATCH_ZSPLIT"); const int v = e ? atoi(e) : 1; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
  int zsplit = zs_env;
  if (zsplit > g.dim[2]) zsplit = g.dim[2];
  const int64_t nwg = (int64_t)txn * tyn * g.dim[3] * zsplit;
  if (nwg > (1 << 20)) return LINK_ERR_ARG;
  int wpf = (int)((nmax + 255) / 256);
  if (wpf > 2048) wpf = 2048;
  if (wpf < 1) wpf = 1;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_batch_k1<OP, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, k1_lds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_batch_k2<OP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, k2_lds);
  // the arrivals a frame's insert posts = its chunks of 256 voxels (wpf).
  // Launch order = dependence order (insert -> K1 -> K2): should two of the streams share a hardware queue after all, the later
  // kernel waits for the earlier one to END -- slow, but never a kernel spinning on one that sits behind it in its own queue.
  const int ins_wgs = c->cus;
  const dc_bt_ins_args_t a0{fr, g, nframes, wpf, sync};
  hipLaunchKernelGGL(k_dc_batch_insert, dim3((unsigned)ins_wgs * DC_BT_INS_WAVES), dim3(64), 0, c->sc, a0);
  int rc = check_launch("link_elk_core_dense_forward_batch (insert)");
  if (rc != LINK_OK) return rc;
  hipLaunchKernelGGL((k_dc_batch_k1<OP, 2>), dim3((unsigned)k1_wgs), dim3(64 * K1::NW), k1_lds, c->sa, fr, p, g, nframes, cpw, k1_target, wpf, sync);
  rc = check_launch("link_elk_core_dense_forward_batch (K1)");
  if (rc != LINK_OK) return rc;
  const dc_bt_k2_args_t a2{fr, p, g, nframes, txn, tyn, zsplit, (int)nwg, k1_target, sync};
  static_assert(sizeof(dc_bt_k2_args_t) <= 4096, "kernel arguments");
  hipLaunchKernelGGL((k_dc_batch_k2<OP, R>), dim3((unsigned)c->cus), dim3(KQ::THREADS), k2_lds, c->sb, a2);
  return check_launch("link_elk_core_dense_forward_batch (K2)");
}

// The call's counters are cleared by a kernel of this file, not by hipMemsetAsync (the runtime's fill goes through its blit path).
__global__ void k_dc_batch_clear(int32_t *sync, int words) {
  for (int i = threadIdx.x; i < words; i += blockDim.x) sync[i] = 0;
}

// A caller stream seen for the first time is tested against the pool (once: ~0.2 ms per pair, and it synchronises that stream); if it
// shares a hardware queue with the pre_mix or the gather stream -- the join's waits in the caller's queue would then sit in front of
// the next call's role kernels -- the roles move to two pool streams it does not share a queue with, behind a drain of the old ones
// (the roles' kernels of consecutive calls must stay in stream order: one pre_mix and one gather workgroup per CU).
static void bt_mind_caller(link_dc_batch *c, hipStream_t st) {
  unsigned mask = 0;
  bool known = false;
  for (auto &e : c->callers) if (e.first == st) { mask = e.second; known = true; break; }
  if (!known) {
    for (size_t i = 0; i < c->pool.size(); i++)
      if (bt_share_queue(st, c->pool[i], c->probe_scratch, c->probe_ev)) mask |= 1u << i;
    if (c->callers.size() >= 16) c->callers.erase(c->callers.begin());
    c->callers.emplace_back(st, mask);
    if (getenv("LINK_DC_BATCH_DEBUG")) fprintf(stderr, "link_dc_batch: caller stream %p shares a hardware queue with pool streams mask 0x%x\n", (void *)st, mask);
  }
  if (!(mask & (1u << c->ia)) && !(mask & (1u << c->ib))) return;
  int na = -1, nb = -1;
  for (int i = 0; i < (int)c->pool.size(); i++)
    if (!(mask & (1u << i))) { if (na < 0) na = i; else if (nb < 0) nb = i; }
  if (na < 0 || nb < 0) return;                          // not two free queues: stay
  (void)hipStreamSynchronize(c->sa); (void)hipStreamSynchronize(c->sb); (void)hipStreamSynchronize(c->sc);
  c->ia = na; c->ib = nb; c->sa = c->pool[na]; c->sb = c->pool[nb];
}

// submit: everything of a call except making the caller's stream wait for its rows (link_dc_batch_join).  *ticket = the number of
// the call's last launch set; the sets of a call run in order on the context's streams, so that set's three events cover the call.
extern "C" int link_dc_batch_submit(link_dc_batch_t *c, const link_dc_buffers_t *frames, const int64_t *n, int32_t nframes,
                                    const link_dc_grid_t *g, const link_elk_desc_t *d, void *stream, int64_t *ticket) {
  if (ticket) *ticket = -1;
  if (!c || !g || !d || nframes < 0 || (nframes > 0 && (!frames || !n))) return LINK_ERR_ARG;
  if (nframes == 0) return LINK_OK;
  // what the quad-consumer K2 and the cell-range K1 serve (the caller runs anything else frame by frame)
  if (d->c != 64 || d->cg != 32 || (d->op != LINK_OP_COS && d->op != LINK_OP_SIN) || (d->r != 2 && d->r != 3) || d->coord_div != 1.0f) return LINK_ERR_ARG;
  if (g->k < DC_INL || g->k > 352 || g->vp * (int64_t)g->k * 16 >= (1LL << 32) || (g->vp + 1) * (int64_t)2 * 64 * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  if constexpr (!(dc_k2q_cfg<LINK_OP_COS, 3>::FITS && dc_k2q_cfg<LINK_OP_COS, 2>::FITS)) return LINK_ERR_ARG;
  const link_dc_buffers_t &b0 = frames[0];
  for (int i = 0; i < nframes; i++) {
    const link_dc_buffers_t &b = frames[i];
    if (n[i] <= 0 || n[i] >= (1LL << 29) || n[i] * 64 * 4 >= (1LL << 32) || b.io_dtype != LINK_IO_F32 || b.alpha) return LINK_ERR_ARG;
    if (!b.feats || !b.coords || !b.slots || !b.cnt || !b.cell_n || !b.vcell || !b.S || !b.hdr || !b.out) return LINK_ERR_ARG;
    if (b.w_pre != b0.w_pre || b.pre_ln_w != b0.pre_ln_w || b.pre_ln_b != b0.pre_ln_b || b.w_pos != b0.w_pos || b.ln_w != b0.ln_w || b.ln_b != b0.ln_b)
      return LINK_ERR_ARG;                               // one block's parameters for the whole batch
    for (int k = 0; k < i; k++)
      if (frames[k].S == b.S || frames[k].cnt == b.cnt || frames[k].out == b.out) return LINK_ERR_ARG;   // a frame needs its own buffers
  }
  if (!b0.w_pre || !b0.pre_ln_w || !b0.pre_ln_b || !b0.w_pos || !b0.ln_w || !b0.ln_b) return LINK_ERR_ARG;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev != c->device) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  bt_mind_caller(c, st);
  const dc_bt_par_t p{b0.w_pre, b0.pre_ln_w, b0.pre_ln_b, b0.w_pos, b0.ln_w, b0.ln_b, d->cg, d->eps, c->dbg1, c->dbg2};
  for (int base = 0; base < nframes; base += DC_BT_MAX) {
    const int nb = nframes - base < DC_BT_MAX ? nframes - base : DC_BT_MAX;
    const int q = (int)(c->calls % BT_RING);
    dc_bt_frames_t fr{};
    int64_t nmax = 0;
    std::vector<const void *> keys;
    for (int i = 0; i < nb; i++) {
      const link_dc_buffers_t &b = frames[base + i];
      fr.f[i] = dc_bt_frame_t{b.feats, reinterpret_cast<const int4 *>(b.coords), reinterpret_cast<int4 *>(b.slots), b.cnt, b.cell_n, b.vcell, b.S,
                              b.hdr, b.out, n[base + i]};
      nmax = n[base + i] > nmax ? n[base + i] : nmax;
      keys.push_back(b.S);
    }
    // order: behind everything already in the caller's stream; behind the ring slot's previous call (its sync words are about to be
    // cleared); behind every earlier call that worked on one of these frames' buffers (K1 overwrites what that call's K2 read, the
    // insert counts into counters that call's K1 zeroed).  Calls on disjoint buffers overlap: K1 of this call starts when K1 of the
    // previous one ends, under the previous call's K2.
    bool fail = hipEventRecord(c->ev_in[q], st) != hipSuccess;
    fail = fail || hipStreamWaitEvent(c->sa, c->ev_in[q], 0) != hipSuccess || hipStreamWaitEvent(c->sb, c->ev_in[q], 0) != hipSuccess ||
           hipStreamWaitEvent(c->sc, c->ev_in[q], 0) != hipSuccess;
    for (int r = 0; r < BT_RING && !fail; r++) {
      if (!c->used[r]) continue;
      bool dep = r == q;
      for (size_t i = 0; i < keys.size() && !dep; i++)
        for (size_t k = 0; k < c->bufs[r].size() && !dep; k++) dep = keys[i] == c->bufs[r][k];
      if (dep)
        fail = hipStreamWaitEvent(c->sa, c->ev_out[r], 0) != hipSuccess || hipStreamWaitEvent(c->sc, c->ev_out[r], 0) != hipSuccess ||
               hipStreamWaitEvent(c->sa, c->ev_a[r], 0) != hipSuccess || hipStreamWaitEvent(c->sc, c->ev_a[r], 0) != hipSuccess;
    }
    int32_t *sync = c->sync + (size_t)q * BT_SYNC_WORDS;
    if (!fail) hipLaunchKernelGGL(k_dc_batch_clear, dim3(1), dim3(256), 0, c->sa, sync, (int)BT_SYNC_WORDS);   // (not hipMemsetAsync: see k_dc_batch_clear)
    fail = fail || hipEventRecord(c->ev_ms[q], c->sa) != hipSuccess ||
           hipStreamWaitEvent(c->sb, c->ev_ms[q], 0) != hipSuccess || hipStreamWaitEvent(c->sc, c->ev_ms[q], 0) != hipSuccess;
    if (fail) { (void)hipGetLastError(); return LINK_ERR_LAUNCH; }
    int rc;
    if (d->op == LINK_OP_COS) rc = d->r == 3 ? batch_launch<LINK_OP_COS, 3>(c, q, fr, p, *g, *d, nb, nmax) : batch_launch<LINK_OP_COS, 2>(c, q, fr, p, *g, *d, nb, nmax);
    else rc = d->r == 3 ? batch_launch<LINK_OP_SIN, 3>(c, q, fr, p, *g, *d, nb, nmax) : batch_launch<LINK_OP_SIN, 2>(c, q, fr, p, *g, *d, nb, nmax);
    // the set's completion events (what link_dc_batch_join, later calls on the same buffers and the ring slot's next user wait for)
    fail = hipEventRecord(c->ev_a[q], c->sa) != hipSuccess || hipEventRecord(c->ev_c[q], c->sc) != hipSuccess || hipEventRecord(c->ev_out[q], c->sb) != hipSuccess;
    c->bufs[q] = keys;
    c->used[q] = true;
    if (ticket) *ticket = c->calls;
    c->calls++;
    if (rc != LINK_OK || fail) {                         // whatever was launched is joined: the caller's stream must not run ahead of it
      (void)hipStreamWaitEvent(st, c->ev_a[q], 0); (void)hipStreamWaitEvent(st, c->ev_c[q], 0); (void)hipStreamWaitEvent(st, c->ev_out[q], 0);
      if (rc != LINK_OK) return rc;
      (void)hipGetLastError();
      return LINK_ERR_LAUNCH;
    }
  }
  return LINK_OK;
}

// join: `stream` waits for the rows of the call `ticket` names (and of every earlier call of the context: its streams run their sets in
// order).  A ticket older than the ring (BT_RING later sets have been submitted) names events a later set has re-recorded: waiting for
// those waits for more, never for less.
extern "C" int link_dc_batch_join(link_dc_batch_t *c, int64_t ticket, void *stream) {
  if (!c || ticket < 0 || ticket >= c->calls) return LINK_ERR_ARG;
  const int q = (int)(ticket % BT_RING);
  hipStream_t st = S(stream);
  if (hipStreamWaitEvent(st, c->ev_a[q], 0) != hipSuccess || hipStreamWaitEvent(st, c->ev_c[q], 0) != hipSuccess ||
      hipStreamWaitEvent(st, c->ev_out[q], 0) != hipSuccess) {
    (void)hipGetLastError();
    return LINK_ERR_LAUNCH;
  }
  return LINK_OK;
}

extern "C" int link_elk_core_dense_forward_batch(link_dc_batch_t *c, const link_dc_buffers_t *frames, const int64_t *n, int32_t nframes,
                                                 const link_dc_grid_t *g, const link_elk_desc_t *d, void *stream) {
  int64_t ticket = -1;
  const int rc = link_dc_batch_submit(c, frames, n, nframes, g, d, stream, &ticket);
  if (rc != LINK_OK || ticket < 0) return rc;            // (an error path has joined what it launched; nframes == 0 launched nothing)
  return link_dc_batch_join(c, ticket, stream);
}
