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
// coord_div = 1, no alpha, fp32 / fp16 / bf16 rows (one type per call), slot capacity <= 352: what the quad-consumer K2 serves; LINK_ERR_ARG otherwise (the caller
// runs the frames one by one).  Measurements, dead ends and the timeline of a call: DESIGN.md section 4i.
// The kernels live in dense_batch_impl.h, compiled once per feature-row type (this file: fp32; dense_batch_f16.hip, dense_batch_bf16.hip).
#define DC_IO 0
#define DC_IO_NS dcb_f32
#include "dense_batch_impl.h"

namespace dcb_f16 { int batch_set_launch(const link::dc_bt_host_t &, const link_dc_buffers_t *, const int64_t *, int, const link_dc_grid_t &, const link_elk_desc_t &); }
namespace dcb_bf16 { int batch_set_launch(const link::dc_bt_host_t &, const link_dc_buffers_t *, const int64_t *, int, const link_dc_grid_t &, const link_elk_desc_t &); }

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
  bool timing;                                         // link_dc_batch_set_timing: bracket the three launches of every set with timed events
  hipEvent_t tb[BT_RING][3], te[BT_RING][3];           // (insert, pre_mix, gather) start / end of the ring slot's set; created on first use
  bool timed[BT_RING];
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
#ifndef DC_BT_EVENT_FLAGS
#define DC_BT_EVENT_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#endif
    // The context's events order DEVICE work only (stream against stream); nothing the host reads hangs on them -- a caller that reads
    // rows on the host synchronises its own stream, which releases to system scope.  Without hipEventDisableSystemFence every record is
    // a packet with a SYSTEM-scope release: the whole L2 written back behind each role kernel, in front of the queue's next packet.
    ok = hipEventCreateWithFlags(&c->ev_in[i], DC_BT_EVENT_FLAGS) == hipSuccess && hipEventCreateWithFlags(&c->ev_ms[i], DC_BT_EVENT_FLAGS) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_a[i], DC_BT_EVENT_FLAGS) == hipSuccess && hipEventCreateWithFlags(&c->ev_c[i], DC_BT_EVENT_FLAGS) == hipSuccess &&
         hipEventCreateWithFlags(&c->ev_out[i], DC_BT_EVENT_FLAGS) == hipSuccess;
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
  for (int i = 0; i < BT_RING; i++)
    for (int k = 0; k < 3; k++) { if (c->tb[i][k]) (void)hipEventDestroy(c->tb[i][k]); if (c->te[i][k]) (void)hipEventDestroy(c->te[i][k]); }
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

// Measurement hook (bench.py): with timing on, every launch set's three kernels are bracketed by timed events on their own streams;
// link_dc_batch_kernel_times waits for the set `ticket` names and returns the (insert, pre_mix, gather) brackets in ms.  The ring holds
// the last BT_RING sets only: LINK_ERR_ARG for an older ticket or a set submitted with timing off.
extern "C" int link_dc_batch_set_timing(link_dc_batch_t *c, int32_t on) {
  if (!c) return LINK_ERR_ARG;
  c->timing = on != 0;
  return LINK_OK;
}
extern "C" int link_dc_batch_kernel_times(link_dc_batch_t *c, int64_t ticket, float *ms /* host [3] */) {
  if (!c || !ms || ticket < 0 || ticket >= c->calls || c->calls - ticket > BT_RING) return LINK_ERR_ARG;
  const int q = (int)(ticket % BT_RING);
  if (!c->timed[q]) return LINK_ERR_ARG;
  for (int i = 0; i < 3; i++)
    if (hipEventSynchronize(c->te[q][i]) != hipSuccess || hipEventElapsedTime(&ms[i], c->tb[q][i], c->te[q][i]) != hipSuccess) {
      (void)hipGetLastError();
      return LINK_ERR_LAUNCH;
    }
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
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {   // (the test's kernels must not end up in a graph)
      (void)hipGetLastError();
      return;
    }
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
    if (n[i] <= 0 || n[i] >= (1LL << 29) || n[i] * 64 * 4 >= (1LL << 32) || b.io_dtype != b0.io_dtype || b.alpha) return LINK_ERR_ARG;   // one row type per call
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
  for (int base = 0; base < nframes; base += DC_BT_MAX) {
    const int nb = nframes - base < DC_BT_MAX ? nframes - base : DC_BT_MAX;
    const int q = (int)(c->calls % BT_RING);
    std::vector<const void *> keys;
    for (int i = 0; i < nb; i++) keys.push_back(frames[base + i].S);
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
#ifndef DC_BT_CLEAR_ON_INSERT
#define DC_BT_CLEAR_ON_INSERT 1
#endif
    // The set's counters are cleared on the INSERT's stream (1): the insert of this call then starts as soon as the previous call's
    // insert has ended -- under the previous call's roles -- instead of behind the previous call's pre_mix kernel (0: cleared on the
    // pre_mix stream), where its flood of atomics and scattered stores slowed this call's first pre_mix items threefold.
    hipStream_t s_clear = DC_BT_CLEAR_ON_INSERT ? c->sc : c->sa, s_o1 = DC_BT_CLEAR_ON_INSERT ? c->sa : c->sc;
    if (!fail) hipLaunchKernelGGL(k_dc_batch_clear, dim3(1), dim3(256), 0, s_clear, sync, (int)BT_SYNC_WORDS);   // (not hipMemsetAsync: see k_dc_batch_clear)
    fail = fail || hipEventRecord(c->ev_ms[q], s_clear) != hipSuccess ||
           hipStreamWaitEvent(c->sb, c->ev_ms[q], 0) != hipSuccess || hipStreamWaitEvent(s_o1, c->ev_ms[q], 0) != hipSuccess;
    if (fail) { (void)hipGetLastError(); return LINK_ERR_LAUNCH; }
    int rc;
    hipStream_t role3[3] = {c->sc, c->sa, c->sb};
    c->timed[q] = false;
    if (c->timing) {
      bool okt = true;
      for (int i = 0; i < 3 && okt; i++) {
        if (!c->tb[q][i]) okt = hipEventCreate(&c->tb[q][i]) == hipSuccess && hipEventCreate(&c->te[q][i]) == hipSuccess;
        okt = okt && hipEventRecord(c->tb[q][i], role3[i]) == hipSuccess;
      }
      c->timed[q] = okt;
    }
    const dc_bt_host_t hset{c->sa, c->sb, c->sc, c->cus, sync, c->dbg1, c->dbg2};
    if (b0.io_dtype == LINK_IO_F16) rc = dcb_f16::batch_set_launch(hset, frames + base, n + base, nb, *g, *d);
    else if (b0.io_dtype == LINK_IO_BF16) rc = dcb_bf16::batch_set_launch(hset, frames + base, n + base, nb, *g, *d);
    else rc = dcb_f32::batch_set_launch(hset, frames + base, n + base, nb, *g, *d);
    if (c->timed[q])
      for (int i = 0; i < 3; i++) c->timed[q] = hipEventRecord(c->te[q][i], role3[i]) == hipSuccess && c->timed[q];
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
