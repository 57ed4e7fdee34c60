// link_amd/csrc/dense_batch_common.h -- what the host side of the batch entry point (dense_batch.hip) and its kernels, compiled once per
// feature-row type (dense_batch_impl.h: dense_batch.hip fp32, dense_batch_f16.hip, dense_batch_bf16.hip), share.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "link_amd.h"

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

namespace link {
// sync words of one launch set (int32, zeroed before the launches): every counter on its own 64-byte line
__host__ __device__ constexpr int bt_err() { return 0; }
__host__ __device__ constexpr int bt_cursor(int xcd) { return 16 * (1 + xcd); }          // K2's item cursors, one per XCD queue
__host__ __device__ constexpr int bt_k1cur(int xcd) { return 16 * (9 + xcd); }          // K1's item cursors, one per XCD slab
__host__ __device__ constexpr int bt_inscur() { return 16 * 17; }                       // the insert's item cursor
__host__ __device__ constexpr int bt_ins(int f) { return 16 * (18 + 2 * f); }           // arrivals of the frame's insert chunks
__host__ __device__ constexpr int bt_k1(int f) { return 16 * (19 + 2 * f); }            // arrivals of the frame's K1 ranges
constexpr int BT_SYNC_WORDS = 16 * (18 + 2 * DC_BT_MAX);
// what a launch set needs of the context: the three role streams, the CU count, the set's sync words, the profiling rows
struct dc_bt_host_t {
  hipStream_t sa, sb, sc;
  int cus;
  int32_t *sync;
  unsigned long long *dbg1, *dbg2;
};
}  // namespace link
