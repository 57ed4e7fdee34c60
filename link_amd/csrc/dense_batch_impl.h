// link_amd/csrc/dense_batch_impl.h -- the kernels of the batch entry point and the launch of one set, compiled once per feature-row type
// (the including translation unit defines DC_IO / DC_IO_NS: dense_batch.hip fp32, dense_batch_f16.hip, dense_batch_bf16.hip).
#include <stdlib.h>

#include <type_traits>
#include <vector>

#include "dense_batch_common.h"
#include "dense_gather.h"
#include "dense_io.h"

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


template <int OP, int R>
static int batch_launch(const dc_bt_host_t &c, const dc_bt_frames_t &fr, const dc_bt_par_t &p, const link_dc_grid_t &g, const link_elk_desc_t &d,
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
  int32_t *sync = c.sync;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  const int k1_wgs = c.cus;
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
  const int ins_wgs = c.cus;
  const dc_bt_ins_args_t a0{fr, g, nframes, wpf, sync};
  hipLaunchKernelGGL(k_dc_batch_insert, dim3((unsigned)ins_wgs * DC_BT_INS_WAVES), dim3(64), 0, c.sc, a0);
  int rc = check_launch("link_elk_core_dense_forward_batch (insert)");
  if (rc != LINK_OK) return rc;
  hipLaunchKernelGGL((k_dc_batch_k1<OP, 2>), dim3((unsigned)k1_wgs), dim3(64 * K1::NW), k1_lds, c.sa, fr, p, g, nframes, cpw, k1_target, wpf, sync);
  rc = check_launch("link_elk_core_dense_forward_batch (K1)");
  if (rc != LINK_OK) return rc;
  const dc_bt_k2_args_t a2{fr, p, g, nframes, txn, tyn, zsplit, (int)nwg, k1_target, sync};
  static_assert(sizeof(dc_bt_k2_args_t) <= 4096, "kernel arguments");
  hipLaunchKernelGGL((k_dc_batch_k2<OP, R>), dim3((unsigned)c.cus), dim3(KQ::THREADS), k2_lds, c.sb, a2);
  return check_launch("link_elk_core_dense_forward_batch (K2)");
}

// One launch set: `nb` frames (<= DC_BT_MAX), rows of THIS translation unit's type.  External linkage: dense_batch.hip calls the three.
int batch_set_launch(const dc_bt_host_t &c, const link_dc_buffers_t *frames, const int64_t *n, int nb, const link_dc_grid_t &g,
                     const link_elk_desc_t &d) {
  const link_dc_buffers_t &b0 = frames[0];
  const dc_bt_par_t p{b0.w_pre, b0.pre_ln_w, b0.pre_ln_b, b0.w_pos, b0.ln_w, b0.ln_b, d.cg, d.eps, c.dbg1, c.dbg2};
  dc_bt_frames_t fr{};
  int64_t nmax = 0;
  for (int i = 0; i < nb; i++) {
    const link_dc_buffers_t &b = frames[i];
    fr.f[i] = dc_bt_frame_t{b.feats, reinterpret_cast<const int4 *>(b.coords), reinterpret_cast<int4 *>(b.slots), b.cnt, b.cell_n, b.vcell, b.S,
                            b.hdr, b.out, n[i]};
    nmax = n[i] > nmax ? n[i] : nmax;
  }
  if (d.op == LINK_OP_COS) return d.r == 3 ? batch_launch<LINK_OP_COS, 3>(c, fr, p, g, d, nb, nmax) : batch_launch<LINK_OP_COS, 2>(c, fr, p, g, d, nb, nmax);
  return d.r == 3 ? batch_launch<LINK_OP_SIN, 3>(c, fr, p, g, d, nb, nmax) : batch_launch<LINK_OP_SIN, 2>(c, fr, p, g, d, nb, nmax);
}
}  // namespace DC_IO_NS
