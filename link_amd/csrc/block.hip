// link_amd/csrc/block.hip -- section G of include/link_amd.h: ONE host call per LinK block on a NEW coordinate set
// (round 5; VERDICT r3 / r4 "native per-block driver").
//
// The reference runs ELKBlock.forward as one Python call whose maps are rebuilt from scratch for every coordinate set
// (segmentation/core/models/semantic_kitti/linkunet.py:124-185: voxel_to_aux's hash / unique / query chain, utils.py:44-52, and
// the convolution's kernel map with `nbsizes.cpu()`, nn/functional/conv.py:103-122).  Through rounds 2-4 link_amd's module path
// did the same work as ~12 FFI calls driven from Python: on cfg2 the call was HOST-bound (tools/cprof_cold.py: ~410 Python
// function calls, 210 us per block against ~130 us of kernels).  This file is that sequence as one C entry:
//
//   main stream   slot insert with occupancy counters + bounding box, ONE launch whose last workgroup writes the results into
//                 mapped host memory  ->  the host polls one word (the call's only round trip)  ->  verdict
//   main stream   27-neighbour table read off the slot lists just filled (link_dc_neighbor_map: no voxel-resolution cell table)
//   side stream   pair plan laid out on the device -> pair GEMM
//   main stream   R_core on the dense-cell layout (the insert above is its index)  ->  wait(side)  ->  centre offset +
//                 pair sums + LayerNorm + add(R_core) + ReLU
//
// The pair plan and the pair GEMM depend on the coordinates and the input rows only, not on R_core: forked onto the context's
// side stream they run underneath the two fused R_core kernels instead of behind them (first version, with the cell-table
// neighbour map on the side stream too: its chain was 100 us against R_core's 51 -- tools/block_timeline.sh).  Everything launched
// here is a kernel the other sections export (and test), plus two of dense_fused.hip written for it: the fused first launch
// (k_dc_index_probe_bbox) and the neighbour table read off the slot lists (k_dc_neighbor_map).
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "common.h"

namespace link {
// dense_fused.hip: the initial image of a 272-word scratch half (only before the first call / after a change of stream); slot insert + occupancy counters + bounding box in one launch, published to host memory by its last workgroup
int dc_probe_scratch_init_run(int32_t *w, hipStream_t st);
// dense_fused.hip: the neighbour table off the slot lists, with the pair plan's count pass done on the tile; conv_pairs.hip: layout + fill
int dc_neighbor_map_count_run(const int32_t *coords, int64_t n, const link_dc_grid_t *g, const uint32_t *cnt, const int32_t *slots,
                              int32_t step, int32_t *nbr, int32_t *wg_counts, int32_t *row_info, hipStream_t st);
int pair_plan_build_counted(const int32_t *nbr, int64_t n, int32_t kvol, int32_t skip_centre, int64_t gran_cap, const int32_t *wg_counts,
                            int32_t *base_k, int32_t *wg_base, int32_t *gran_start, int32_t *wg_ext, int32_t *wg_k, int32_t *hdr,
                            int32_t *ext_start, int32_t *pair_in, int32_t *pair_out, int32_t *ext_list, hipStream_t st);
int dc_index_probe_bbox_run(const link_dc_buffers_t *b, const link_dc_grid_t *g, int64_t n, int32_t *cur, int32_t *next,
                            int32_t *host_dev, int seq, hipStream_t st);
}  // namespace link

struct link_block_ctx {
  int device;
  hipStream_t side;
  hipEvent_t fork, join;
  int32_t *scratch;        // device: 2 x 272 words (bbox at 0, ticket at 8, occupancy counters at 16: 16 partial slots of 16 ints on
                           // their own lines); call i works in half i & 1 and lays the other half out for call i + 1
  int32_t *host;           // pinned + mapped: the 272 words of the last call; word 8 = the call's sequence number once they are there
  int32_t *host_dev;       // the device's address of `host`
  int seq;                 // calls so far
  hipStream_t last_stream; // the stream whose probe laid out the half the next call works in
  bool primed;
};

static constexpr int BLOCK_WORDS = 16 + 256;

extern "C" int link_block_ctx_create(link_block_ctx_t **out) {
  if (!out) return LINK_ERR_ARG;
  *out = nullptr;
  link_block_ctx *c = new link_block_ctx();
  memset(c, 0, sizeof(*c));
  hipError_t e = hipGetDevice(&c->device);
  // (LINK_BLOCK_SIDE_PRIO=1, A/B only: the side stream at the highest stream priority)
  if (e == hipSuccess) {
    if (getenv("LINK_BLOCK_SIDE_PRIO")) {
      int pr_least = 0, pr_greatest = 0;
      e = hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest);
      if (e == hipSuccess) e = hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, pr_greatest);
    } else {
      e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    }
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->fork, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->join, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->scratch), 2 * BLOCK_WORDS * sizeof(int32_t));
  if (e == hipSuccess)
    e = hipHostMalloc(reinterpret_cast<void **>(&c->host), BLOCK_WORDS * sizeof(int32_t), hipHostMallocMapped | hipHostMallocCoherent);
  if (e == hipSuccess) e = hipHostGetDevicePointer(reinterpret_cast<void **>(&c->host_dev), c->host, 0);
  if (e == hipSuccess) memset(c->host, 0, BLOCK_WORDS * sizeof(int32_t));
  if (e != hipSuccess) {
    link::set_error("link_block_ctx_create", e);
    link_block_ctx_destroy(c);
    return LINK_ERR_LAUNCH;
  }
  *out = c;
  return LINK_OK;
}

extern "C" int link_block_ctx_destroy(link_block_ctx_t *c) {
  if (!c) return LINK_OK;
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->fork) (void)hipEventDestroy(c->fork);
  if (c->join) (void)hipEventDestroy(c->join);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->host) (void)hipHostFree(c->host);
  delete c;
  return LINK_OK;
}

// Piece offsets (in 32-bit words, 16-byte aligned) of the arena link_pair_plan_build works in when every list is sized for its
// capacity: wg_counts | row_info | base_k + wg_base + gran_start | wg_ext | wg_k | hdr | ext_start | pair_in | pair_out | ext_list.
// offs[10] = total words; returns the granule capacity (rows_pad = 128 x that), or -1.
extern "C" int64_t link_pair_plan_arena(int64_t n, int32_t kvol, int32_t skip_centre, int64_t offs[11]) {
  if (n <= 0 || kvol <= 0 || kvol > 64 || !offs) return -1;
  const int64_t nwg = (n + 255) / 256;
  const int64_t cap_pairs = n * (int64_t)(kvol - (skip_centre ? 1 : 0));
  const int64_t gran_cap = (cap_pairs + 127 * (int64_t)kvol + 127) / 128;
  const int64_t sizes[10] = {nwg * (kvol + 1), n, kvol + nwg * kvol + kvol + 1, nwg, gran_cap, 8, n + 1, gran_cap * 128, gran_cap * 128,
                             cap_pairs > 0 ? cap_pairs : 1};
  offs[0] = 0;
  for (int i = 0; i < 10; i++) offs[i + 1] = offs[i] + ((sizes[i] + 3) & ~(int64_t)3);
  return gran_cap;
}

// LINK_BLOCK_PROF=1: host-side phase times of link_elk_block_forward (us since entry), printed every 64th call
struct block_prof {
  bool on;
  std::chrono::steady_clock::time_point t0;
  double t[8];
  int n;
  block_prof() : on(getenv("LINK_BLOCK_PROF") != nullptr), n(0) {}
  void start() { if (on) { t0 = std::chrono::steady_clock::now(); n = 0; } }
  void mark() { if (on && n < 8) t[n++] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); }
};

static int block_fail(const char *what, hipError_t e) {
  link::set_error(what, e);
  return LINK_ERR_LAUNCH;
}

// the frame leaves nothing behind in the plan it was tried on: counters and status word back to zero
static int block_unprobe(const link_block_args_t *a, hipStream_t st) {
  hipError_t e = hipMemsetAsync(a->buf->cnt, 0, (size_t)a->g->vp * 4, st);
  if (e == hipSuccess) e = hipMemsetAsync(a->buf->hdr, 0, LINK_HDR_WORDS * 4, st);
  return e == hipSuccess ? LINK_OK : block_fail("link_elk_block_forward (unprobe)", e);
}

extern "C" int link_streams_share_queue(void *a, void *b, double *delay_us);
static void block_side_off_the_callers_queue(link_block_ctx *c, hipStream_t st) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return; }
  hipStream_t rejected[8];
  int nrej = 0;
  for (int tries = 0; tries < 8; tries++) {
    double d = 0.0;
    if (link_streams_share_queue(st, c->side, &d) != LINK_OK || d <= 75.0) break;
    hipStream_t s_new = nullptr;
    if (hipStreamCreateWithFlags(&s_new, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); break; }
    rejected[nrej++] = c->side;                        // kept alive until the search ends: it holds its queue's use count up
    c->side = s_new;
  }
  for (int i = 0; i < nrej; i++) { (void)hipStreamSynchronize(rejected[i]); (void)hipStreamDestroy(rejected[i]); }
}

extern "C" int link_elk_block_forward(link_block_ctx_t *c, link_block_args_t *a, void *stream) {
  if (!c || !a || !a->buf || !a->g || !a->desc || a->n <= 0) return LINK_ERR_ARG;
  const link_dc_buffers_t *b = a->buf;
  const int C = a->desc->c;
  if (!b->feats || !b->coords || !b->out || !a->out || !a->nbr || !a->pair_arena || !a->contrib || !a->w || !a->nl_w || !a->nl_b ||
      a->ts <= 0 || !a->subm)
    return LINK_ERR_ARG;
  if (b->io_dtype != LINK_IO_F32) return LINK_ERR_ARG;                     // (half rows: the module path's own sequence)
  if (!link_conv_pairs_supported(C, C)) return LINK_ERR_ARG;
  const int64_t n = a->n;
  const int kvol = 27;
  int64_t po[11];
  const int64_t gran_cap = link_pair_plan_arena(n, kvol, 1, po);
  if (gran_cap < 0 || po[10] > a->pair_arena_words || a->contrib_rows < gran_cap * 128) return LINK_ERR_WORKSPACE;
  hipStream_t st = link::S(stream);
  a->verdict = LINK_BLOCK_MISS;
  static thread_local block_prof prof;
  static thread_local unsigned prof_calls = 0;
  prof.start();

  // ---- 1. slot insert with occupancy counters + bounding box: ONE launch whose last workgroup writes the 272 result words into
  //         mapped host memory and then the call's sequence number into word 8 -- the host polls that word (a stream
  //         synchronisation costs ~10 us more than the store takes to arrive; it remains the fallback after 2 ms) ----
  const int half = c->seq & 1;
  int32_t *cur = c->scratch + half * BLOCK_WORDS, *next = c->scratch + (half ^ 1) * BLOCK_WORDS;
  int rc;
  if (!c->primed || c->last_stream != st) {            // first call / another stream: nobody laid this half out in stream order
    if (c->primed) (void)hipStreamSynchronize(c->last_stream);
    // The side stream must not sit on the caller's hardware queue (the runtime multiplexes streams onto a few): the fork / join event
    // records would then be barriers between the two chains this call runs side by side (R_block on a new coordinate set read 151 or
    // 281 us from process to process).  Tested once per caller stream (link_streams_share_queue, dense_batch.hip); a side stream that
    // shares is replaced, up to eight times.
    block_side_off_the_callers_queue(c, st);
    rc = link::dc_probe_scratch_init_run(cur, st);
    if (rc != LINK_OK) return rc;
  }
  const int seq = ++c->seq;
  __atomic_store_n(&c->host[8], 0, __ATOMIC_RELEASE);
  rc = link::dc_index_probe_bbox_run(b, a->g, n, cur, next, c->host_dev, seq, st);
  if (rc != LINK_OK) { c->primed = false; return rc; }
  c->primed = true;
  c->last_stream = st;
  prof.mark();                                         // 0: probe launched
  {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    bool arrived = true;
    while (__atomic_load_n(&c->host[8], __ATOMIC_ACQUIRE) != seq) {
      __builtin_ia32_pause();                          // (the core's sibling thread and the memory pipeline get the cycles)
      if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
        const hipError_t es = hipStreamSynchronize(st);
        if (es != hipSuccess) { c->primed = false; return block_fail("link_elk_block_forward (round trip)", es); }
        arrived = __atomic_load_n(&c->host[8], __ATOMIC_ACQUIRE) == seq;
        break;
      }
    }
    if (!arrived) { c->primed = false; return block_fail("link_elk_block_forward (results never arrived)", hipErrorUnknown); }
  }
  hipError_t e = hipSuccess;
  prof.mark();                                         // 1: results arrived
  const int32_t *h = c->host;
  const int64_t n_in = h[16], m = h[17];
  const int32_t mx = h[18];
  for (int i = 0; i < 8; i++) a->bbox[i] = h[i];
  a->stats[0] = (int32_t)n_in; a->stats[1] = (int32_t)m; a->stats[2] = mx; a->stats[3] = 0;

  // ---- 2. verdict: every voxel inside the plan's grid, occupancy the dense-cell kernels are built for ----
  const link_dc_grid_t *g = a->g;
  bool inside = n_in == n;
  for (int ax = 0; ax < 4 && inside; ax++) {
    const int s = ax < 3 ? g->s : 1;
    const int lo = link::floordiv(h[ax], s), hi = link::floordiv(h[4 + ax], s);
    inside = lo >= g->lo[ax] && hi < g->lo[ax] + g->dim[ax];
  }
  const int cell_cap = a->cell_max < g->k ? a->cell_max : g->k;
  const bool dense_ok = m > 0 && (double)n_in <= (double)a->mean_max * (double)m && mx <= cell_cap;
  if (!inside || !dense_ok) {
    a->stats[3] = (!inside ? 1 : 0) | (!dense_ok ? 2 : 0);
    rc = block_unprobe(a, st);
    return rc != LINK_OK ? rc : LINK_BLOCK_MISS;
  }

  // ---- 3. main stream: the 27-neighbour table read off the slot lists the probe has just filled (before the pre_mix kernel
  //         re-orders them), then R_core on that insert.  The launches are issued in the order the device needs them: the
  //         side stream's chain (~50 us) and R_core (~50 us) start together behind the neighbour table ----
  int32_t *pa = a->pair_arena;
  int32_t *wg_counts = pa + po[0], *row_info = pa + po[1], *meta = pa + po[2], *wg_ext = pa + po[3], *wg_k = pa + po[4], *phdr = pa + po[5];
  int32_t *ext_start = pa + po[6], *pair_in = pa + po[7], *pair_out = pa + po[8], *ext_list = pa + po[9];
  // From here on the frame sits in the plan (counters, slot lists, status word): EVERY error return below takes it out again
  // (block_unprobe), so that a caller who catches the error does not run its next frame on top of non-zero counters.
  auto fail = [&](int code) { (void)block_unprobe(a, st); return code; };
  rc = link::dc_neighbor_map_count_run(b->coords, n, g, b->cnt, reinterpret_cast<const int32_t *>(b->slots), a->ts, a->nbr, wg_counts, row_info, st);
  if (rc != LINK_OK) return fail(rc);
  prof.mark();                                         // 2: neighbour table launched
  hipStream_t sd = c->side;
  e = hipEventRecord(c->fork, st);
  if (e != hipSuccess) return fail(block_fail("link_elk_block_forward (fork)", e));
  rc = link_elk_core_dense_forward(b, g, a->desc, n, 2, stream);
  if (rc != LINK_OK) return fail(rc);                  // (argument validation refused the step: its kernels have not consumed the insert)

  prof.mark();                                         // 3: fork recorded + R_core launched
  // ---- 4. side stream: pair plan laid out on the device + the pair GEMM (they need the table and the input rows only) ----
  e = hipStreamWaitEvent(sd, c->fork, 0);
  const int64_t nwg = (n + 255) / 256;
  int rs = e == hipSuccess ? LINK_OK : LINK_ERR_LAUNCH;
  if (rs == LINK_OK)
    rs = link::pair_plan_build_counted(a->nbr, n, kvol, 1, gran_cap, wg_counts, meta, meta + kvol, meta + kvol + nwg * kvol, wg_ext, wg_k, phdr,
                                       ext_start, pair_in, pair_out, ext_list, sd);
  if (rs == LINK_OK) {
    if (a->ws && a->w_big)
      rs = link_conv_pairs_gemm_split(reinterpret_cast<const float *>(b->feats), pair_in, wg_k, gran_cap * 128, a->ws, a->w, a->w_big, C, C,
                                      a->contrib, sd);
    else
      rs = link_conv_pairs_gemm_io(b->feats, LINK_IO_F32, pair_in, wg_k, gran_cap * 128, a->w, C, C, a->contrib, sd);
  }
  prof.mark();                                         // 4: side chain launched
  const hipError_t e1 = hipEventRecord(c->join, sd);
  const hipError_t e2 = hipStreamWaitEvent(st, c->join, 0);        // joined whatever happened: nothing of this call outlives it unordered
  if (e != hipSuccess) return block_fail("link_elk_block_forward (fork)", e);
  if (rs != LINK_OK) return rs;
  if (e1 != hipSuccess || e2 != hipSuccess) return block_fail("link_elk_block_forward (join)", e1 != hipSuccess ? e1 : e2);
  if (rc != LINK_OK) return rc;

  // ---- 5. the convolution's finish with R_core as the addend ----
  rc = link_conv_centre_sum_io(b->feats, a->w, kvol / 2, a->contrib, gran_cap * 128, ext_start, ext_list, n, C, C, nullptr, a->nl_w,
                               a->nl_b, a->nl_eps, b->out, a->flags, a->out, LINK_IO_F32, stream);
  if (rc != LINK_OK) return rc;
  prof.mark();                                         // 5: join + finish launched
  if (prof.on && (++prof_calls & 63u) == 0)
    fprintf(stderr, "link_elk_block_forward host us: probe launched %.1f | results %.1f | table %.1f | fork + R_core %.1f | side chain %.1f | join + finish %.1f\n",
            prof.t[0], prof.t[1], prof.t[2], prof.t[3], prof.t[4], prof.t[5]);
  a->verdict = LINK_BLOCK_DONE;
  return LINK_BLOCK_DONE;
}
