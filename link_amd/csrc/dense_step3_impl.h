// link_amd/csrc/dense_step3_impl.h -- three frames in flight in ONE launch per step (included by dense_fused_impl.h inside
// its IO namespace, after the kernels whose bodies it runs).
//
// A frame's R_core is three dependent launches -- slot insert, K1 (pre_mix + modulate + per-cell sums), K2 (box gather +
// de-modulate) -- and none of them fills the chip on its own: K1 is bound by the latency of its record -> row -> MFMA chain at
// 16 waves per CU, K2 by its plane ring at two workgroups per CU.  Frames issued on separate HIP streams overlap by what the
// hardware queues happen to interleave; stage streams chained by events are worse (54 us / frame: a cross-queue event costs
// ~14 us, tools/pipe3.py).  The step kernel makes the overlap a property of ONE stream and ONE launch per frame: its grid is
// three ranges of workgroups,
//
//     [ K2 of frame t-2 | K1 of frame t-1 | slot insert of frame t ]
//
// each running the SAME device body as the stand-alone kernel of its stage (dc_k2q_body, dc_k1m_body, dc_index_body), on the
// buffers of its own frame.  The dependences of a frame are between launches (stream order); inside a launch the three
// ranges touch three different frames, so no workgroup waits on another.  Every CU holds two workgroups of 512 threads
// whatever their roles; the K2 range is first in the grid so that its workgroups -- the longest -- are placed first, the
// insert is last and its few workgroups take the slots the other two leave.
//
// Measured (cfg2, three different frames in rotation, tools/step3.py, profiles/r04_*_step_kernel_vs_streams.txt): 36.5-42.9 us
// per frame against 32.9-39.9 for three plans on three streams on the same boxes -- the launch lasts as long as its K2 range
// (70 k shader ticks; K1's waves end at 47 k on average, 62 k at the latest), and with every wave slot of the chip taken either
// way the wave-time of the two kernels is what both arrangements are bound by.  So bench.py keeps the streams; the step
// kernel is for a caller that has ONE stream (a sensor loop inside a larger graph) and wants the overlap anyway: against one
// plan on one stream (52 us / frame) it is the faster form.
//
// A role whose frame pointer is null is absent (pipeline fill and drain).  Two-part rows at C = 64 with the pair form
// (cg = 32), no alpha, coord_div = 1: the configuration the quad kernel serves.
#pragma once

#ifndef DC_S3_K2_PRIO
#define DC_S3_K2_PRIO 3    /* wave priority of the gather range (the longer one; one frame over and over: 34.0-36.7 us / frame against
                              36.0-40.1 at priority 0, A/B on one box) / of the pre_mix range */
#endif
#ifndef DC_S3_K1_PRIO
#define DC_S3_K1_PRIO 0
#endif
struct dc_s3_k2_t {
  const float *S; const int32_t *cell_n; const int4 *slots; void *out; unsigned long long *dbg; int64_t n;
  int txn, tyn, zsplit, nwg;
};
struct dc_s3_k1_t {
  const void *feats; int4 *slots; uint32_t *cnt; int32_t *cell_n; float *S; float *fin; int32_t *hdr; unsigned long long *dbg;
  int64_t n; int cpw;
};
struct dc_s3_ix_t {
  const int4 *coords; uint32_t *cnt; int4 *slots; int32_t *vcell; int32_t *hdr; int64_t n;
};
struct dc_s3_par_t {
  const float *w_pre, *pre_ln_w, *pre_ln_b, *w_pos, *ln_w, *ln_b;
  int cg; float eps;
};

template <int OP, int R>
__global__ void __launch_bounds__(512, 4) k_dc_step3(dc_s3_k2_t a2, dc_s3_k1_t a1, dc_s3_ix_t a0, dc_s3_par_t p, link_dc_grid_t g,
                                                      int n_k2, int n_k1, int n_ix) {
  constexpr int C = 64, NB = 2;
  int bid = (int)blockIdx.x;
  if (bid < n_k2) {
    if (DC_S3_K2_PRIO) __builtin_amdgcn_s_setprio(DC_S3_K2_PRIO);
    dc_k2q_body<OP, R, false>(a2.S, a2.cell_n, a2.slots, p.w_pos, nullptr, p.ln_w, p.ln_b, p.cg, 1.0f, p.eps, a2.n, g, a2.txn,
                              a2.tyn, a2.zsplit, a2.nwg, a2.out, a2.dbg, bid, threadIdx.x);
    return;
  }
  bid -= n_k2;
  if (bid < n_k1) {
    if (DC_S3_K1_PRIO) __builtin_amdgcn_s_setprio(DC_S3_K1_PRIO);
    dc_k1m_body<C, OP, NB, false>(a1.feats, a1.slots, a1.cnt, a1.cell_n, p.w_pre, p.pre_ln_w, p.pre_ln_b, p.w_pos, nullptr, p.cg,
                                  1.0f, p.eps, a1.n, g, a1.cpw, false, a1.S, a1.fin, a1.hdr, a1.dbg, nullptr, bid);
    return;
  }
  bid -= n_k1;
  int s0 = 0, s1 = 0, s2 = 0;
  dc_index_body<false>(a0.coords, a0.n, g, a0.cnt, a0.slots, a0.vcell, a0.hdr, bid, n_ix, 512, s0, s1, s2);
}

template <int OP, int R>
static int launch_step3(const link_dc_buffers_t *b0, const link_dc_buffers_t *b1, const link_dc_buffers_t *b2,
                        const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n0, int64_t n1, int64_t n2, int ix_wgs,
                        hipStream_t st) {
  using KQ = dc_k2q_cfg<OP, R>;
  using K1 = dc_k1m_cfg<64, OP>;
  using KG = typename dc_k2_cfg<OP, R>::G;
  static_assert(K1::NW == 8 && KQ::THREADS == 512, "the three roles share 512-thread workgroups");
  const link_dc_buffers_t *bp = b2 ? b2 : (b1 ? b1 : b0);
  dc_s3_par_t p{bp->w_pre, bp->pre_ln_w, bp->pre_ln_b, bp->w_pos, bp->ln_w, bp->ln_b, d.cg, d.eps};
  dc_s3_k2_t a2{};
  dc_s3_k1_t a1{};
  dc_s3_ix_t a0{};
  int n_k2 = 0, n_k1 = 0, n_ix = 0;
  if (b2 && n2 > 0) {
    const int txn = (g.dim[0] + KG::TX - 1) / KG::TX, tyn = (g.dim[1] + KG::TY - 1) / KG::TY;
    int zsplit = b2->tune.k2_zsplit;
    if (zsplit <= 0) {
      const int64_t tiles = (int64_t)txn * tyn * g.dim[3];
      zsplit = (int)(256 / tiles);                       // half the chip's workgroup slots: the other half is K1's
      if (zsplit < 1) zsplit = 1;
    }
    if (zsplit > g.dim[2]) zsplit = g.dim[2];
    const int64_t nwg = (int64_t)txn * tyn * g.dim[3] * zsplit;
    n_k2 = (int)((nwg + 7) / 8 * 8);
    a2 = dc_s3_k2_t{b2->S, b2->cell_n, reinterpret_cast<const int4 *>(b2->slots), b2->out,
                    reinterpret_cast<unsigned long long *>(b2->tune.k2_dbg), n2, txn, tyn, zsplit, (int)nwg};
  }
  if (b1 && n1 > 0) {
    const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
    int64_t waves = (int64_t)(b1->tune.k1_wgs > 0 ? b1->tune.k1_wgs : 512) * 4;
    int cpw = (int)((vi + waves - 1) / waves);
    if (cpw < 1) cpw = 1;
    n_k1 = (int)((vi + (int64_t)cpw * K1::NW - 1) / ((int64_t)cpw * K1::NW));
    n_k1 = (n_k1 + 7) / 8 * 8;                           // the ranges keep the workgroup -> XCD map of a stand-alone grid
    a1 = dc_s3_k1_t{b1->feats, reinterpret_cast<int4 *>(b1->slots), b1->cnt, b1->cell_n, b1->S, b1->fin, b1->hdr,
                    reinterpret_cast<unsigned long long *>(b1->tune.k1_dbg), n1, cpw};
  }
  if (b0 && n0 > 0) {
    int64_t wgs = (n0 + 2047) / 2048;                    // four voxels a thread: the insert rides in the slots the others leave
    if (ix_wgs > 0) wgs = ix_wgs;
    if (wgs > 1024) wgs = 1024;
    n_ix = (int)wgs;
    a0 = dc_s3_ix_t{reinterpret_cast<const int4 *>(b0->coords), b0->cnt, reinterpret_cast<int4 *>(b0->slots), b0->vcell, b0->hdr, n0};
  }
  const int grid = n_k2 + n_k1 + n_ix;
  if (grid == 0) return LINK_OK;
  constexpr int lds = KQ::LDS_BYTES > K1::LDS_BYTES ? KQ::LDS_BYTES : K1::LDS_BYTES;
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_step3<OP, R>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k_dc_step3<OP, R>), dim3((unsigned)grid), dim3(512), lds, st, a2, a1, a0, p, g, n_k2, n_k1, n_ix);
  return check_launch("link_elk_core_dense_step3");
}

// 0 if this build serves (op, r) with the step kernel
int run_step3(const link_dc_buffers_t *b0, const link_dc_buffers_t *b1, const link_dc_buffers_t *b2, const link_dc_grid_t &g,
              const link_elk_desc_t &d, int64_t n0, int64_t n1, int64_t n2, int ix_wgs, hipStream_t st) {
  if (d.c != 64 || d.cg != 32 || d.coord_div != 1.0f || (d.r != 2 && d.r != 3)) return LINK_ERR_ARG;
  if (d.op == LINK_OP_COS) {
    if constexpr (dc_k2q_cfg<LINK_OP_COS, 3>::FITS && dc_k2q_cfg<LINK_OP_COS, 2>::FITS)
      return d.r == 3 ? launch_step3<LINK_OP_COS, 3>(b0, b1, b2, g, d, n0, n1, n2, ix_wgs, st)
                      : launch_step3<LINK_OP_COS, 2>(b0, b1, b2, g, d, n0, n1, n2, ix_wgs, st);
  } else if (d.op == LINK_OP_SIN) {
    if constexpr (dc_k2q_cfg<LINK_OP_SIN, 3>::FITS && dc_k2q_cfg<LINK_OP_SIN, 2>::FITS)
      return d.r == 3 ? launch_step3<LINK_OP_SIN, 3>(b0, b1, b2, g, d, n0, n1, n2, ix_wgs, st)
                      : launch_step3<LINK_OP_SIN, 2>(b0, b1, b2, g, d, n0, n1, n2, ix_wgs, st);
  }
  return LINK_ERR_ARG;
}
