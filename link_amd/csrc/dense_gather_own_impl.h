// link_amd/csrc/dense_gather_own_impl.h -- "own-cell" form of the fused box sum + de-modulate kernel (C = 64; round 3,
// link_dc_tuning_t::k2_form bit 2).  Included inside DC_IO_NS by dense_fused_impl.h after the K2 configuration structs.
//
// tools/k2prof.py on the producer / consumer form: a plane step is ~4600 ticks, of which the producer waves spend 1930 in
// three dependent LDS round trips of the box sum and 844 in the barrier, the consumer waves 2277 in the pair loop (scan /
// ballot / ds_bpermute dealing + LDS reads of records and A rows) and 1447 in the barrier; 79 KB of LDS and 8 waves of 128
// registers per workgroup admit two workgroups per CU.  This form trades the balanced dealing for resources:
//   * a lane group de-modulates the voxels of ITS OWN cell, straight from the A row it has just summed (registers): no A
//     image, no count image, no dealing, no second barrier; the cost is idle groups (a plane step lasts as long as the
//     fullest of a wave's four cells: ~2 pair rounds instead of ~1.4);
//   * a FIFTH wave does nothing but the LDS-DMA of the plane ring.  It issues no stores, so its vmcnt waits are exact and a
//     ring of TWO buffers with prefetch distance 1 is enough (the compute waves' output stores would otherwise sit between the
//     DMAs on the same counter: the reason the other forms need three buffers);
//   * the plane's three x-offsets arrive as one pipelined LDS request (dense_gather.h);
//   => 39.4 KB of LDS and 5 waves per workgroup: three to four workgroups per CU instead of two.
#pragma once

template <int OP, int R>
struct dc_k2o_cfg {
  static constexpr int C = 64, P = (OP == LINK_OP_COSX) ? 3 : 2;
  using G = dc_gather_cfg<C, P, R>;
  static constexpr int RB = P * C * 4;
  static constexpr int NPW = (G::NPC + 63) / 64;       // LDS-DMA instructions of one plane (one wave: 64 pieces of 16 B each)
  static constexpr int PLANE = NPW * 1024;
  static constexpr int CNT_OFF = PLANE, REC_OFF = PLANE + 256;
  static constexpr int BUF_BYTES = REC_OFF + 16 * DC_INL * 16;
#ifndef DC_K2O_RING
#define DC_K2O_RING 2      /* plane ring slots: 2 = prefetch distance 1 (39 KB), 3 = distance 2 (59 KB) */
#endif
  static constexpr int RING = DC_K2O_RING;
  static constexpr int NI = NPW + 2;                   // DMA instructions per plane
  static constexpr int LDS_BYTES = RING * BUF_BYTES;
  static_assert(G::NCOL <= 64, "count image: one wave-instruction");
  static_assert(G::NG == 16, "16 columns per workgroup");
};

template <int OP, int R, bool PAIR, bool DIV>
__global__ void __launch_bounds__(320, 3) k_dc_gather_demod_own(
    const float *__restrict__ S_, const int32_t *__restrict__ cell_n, const int4 *__restrict__ slots,
    const float *__restrict__ fin, const float *__restrict__ w_pos, const float *__restrict__ alpha,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, int cg, float coord_div, float eps, int64_t n,
    link_dc_grid_t g, int txn, int tyn, int zsplit, int nwg, void *__restrict__ out, unsigned long long *__restrict__ dbg) {
  using KO = dc_k2o_cfg<OP, R>;
  unsigned long long tq0 = dbg ? __builtin_amdgcn_s_memtime() : 0, tq_bar = 0, tq_box = 0, tq_pairs = 0, tq_dma = 0;
  int tq_rounds = 0;
  using K = typename KO::G;
  constexpr int C = 64, P = KO::P, LPR = 16, TY = K::TY, TX = K::TX, HY = K::HY, HLO = K::HLO;
  constexpr int RB = P * C * 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int per = (nwg + 7) >> 3;
  const int L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (L >= nwg) return;
  int t = L;
  const int zseg = t % zsplit; t /= zsplit;
  const int ty = t % tyn; t /= tyn;
  const int tx = t % txn;
  const int b = t / txn;
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int PDx = g.pdim[0], PDy = g.pdim[1], PDz = g.pdim[2];
  const int x0 = tx * TX, y0 = ty * TY;
  const int zs = (int)(((long long)Dz * zseg) / zsplit), ze = (int)(((long long)Dz * (zseg + 1)) / zsplit);
  if (zs >= ze) return;
  const int nplanes = (ze - zs) + R - 1;
  const int pz0 = zs + 1 - HLO;
  auto col_cell0 = [&](int hx, int hy) {             // padded cell id of (haloed column, z = 0), clamped into the grid
    int px = x0 + 1 - HLO + hx, py = y0 + 1 - HLO + hy;
    px = px < PDx - 1 ? px : PDx - 1;
    py = py < PDy - 1 ? py : PDy - 1;
    return (uint32_t)(((b * PDx + px) * PDy + py) * PDz);
  };
  if (wave == 4) {
    // ---- the DMA wave: plane i+1 -> ring slot (i+1) & 1 while the compute waves work on plane i ----
    uint32_t src_off[KO::NPW];
#pragma unroll
    for (int i = 0; i < KO::NPW; i++) {
      int pid = i * 64 + lane;
      if (pid >= K::NPC) pid = K::NPC - 1;
      const int col = pid / K::RP, pcs = pid % K::RP;
      src_off[i] = col_cell0(col / HY, col % HY) * (uint32_t)RB + (uint32_t)pcs * 16u;
    }
    const int ce = lane < K::NCOL ? lane : K::NCOL - 1;
    const uint32_t cnt_cell0 = col_cell0(ce / HY, ce % HY);
    const int rcol = lane >> 2, rec_k = lane & 3;      // inline slot records of the 16 interior cells of an output plane
    const uint32_t rec_cell0 = col_cell0(rcol / TY + HLO, rcol % TY + HLO);
    const char *Sb = reinterpret_cast<const char *>(S_);
    auto issue = [&](int plane) {
      int pz = pz0 + plane;
      pz = pz < PDz - 1 ? pz : PDz - 1;
      char *buf = lds + (plane % KO::RING) * KO::BUF_BYTES;
#pragma unroll
      for (int i = 0; i < KO::NPW; i++) {
        const char *src = Sb + (size_t)src_off[i] + (size_t)pz * RB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(buf + i * 1024), 16, 0, 0);
      }
      const int32_t *csrc = cell_n + cnt_cell0 + pz;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)csrc,
                                       (__attribute__((address_space(3))) void *)(buf + KO::CNT_OFF), 4, 0, 0);
      int po = pz0 + plane - (R - 1) + HLO;             // output plane closed by this plane
      po = po < 0 ? 0 : (po < PDz - 1 ? po : PDz - 1);
      const int4 *rsrc = slots + ((size_t)(rec_cell0 + po) * DC_INL + rec_k);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)rsrc,
                                       (__attribute__((address_space(3))) void *)(buf + KO::REC_OFF), 16, 0, 0);
    };
    if constexpr (KO::RING == 2) {
      issue(0);
      wait_vmcnt<0>();
      asm volatile("s_barrier" ::: "memory");
      for (int i = 0; i < nplanes; i++) {
        unsigned long long tqa = dbg ? __builtin_amdgcn_s_memtime() : 0;
        if (i + 1 < nplanes) {                          // ring slot (i+1) & 1 was read during step i-1: free since the last barrier
          issue(i + 1);
          wait_vmcnt<0>();
        }
        if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_dma += tqb - tqa; tqa = tqb; }
        asm volatile("s_barrier" ::: "memory");
        if (dbg) tq_bar += __builtin_amdgcn_s_memtime() - tqa;
      }
    } else {
      issue(0);
      if (nplanes > 1) issue(1);
      if (nplanes > 1) wait_vmcnt<KO::NI>(); else wait_vmcnt<0>();      // plane 0 landed (this wave issues nothing but DMAs: exact counts)
      asm volatile("s_barrier" ::: "memory");
      for (int i = 0; i < nplanes; i++) {
        if (i + 2 < nplanes) issue(i + 2);              // ring slot (i+2) % 3 was read during step i-1
        if (i + 1 < nplanes) {                          // plane i+1 must be in before the next barrier
          if (i + 2 < nplanes) wait_vmcnt<KO::NI>(); else wait_vmcnt<0>();
        }
        asm volatile("s_barrier" ::: "memory");
      }
    }
    if (dbg && lane == 0) {
      unsigned long long *d = dbg + ((size_t)L * 8 + 4) * 8;
      d[0] = __builtin_amdgcn_s_memtime() - tq0; d[1] = tq_dma; d[2] = tq_bar; d[6] = nplanes; d[7] = 3;
    }
    return;
  }
  // ---- compute waves: this group's column ----
  const int grp = tid >> 4, li = tid & 15;
  const int ix = grp / TY, iy = grp % TY;
  const bool col_ok = (x0 + ix < Dx) && (y0 + iy < Dy);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)lds;
  const uint32_t row_lane = (uint32_t)((ix * HY + iy) * RB + li * 16);
  const uint32_t cnt_lane = (uint32_t)((ix * HY + iy) * 4);
  const int ch0 = 4 * li;
  const bool hi = PAIR && li >= 8;
  const __amdgpu_buffer_rsrc_t r_out = dc_rsrc(out, (uint32_t)(n * C * IO_BYTES));
  float w0[4], w1[4], w2[4], al[4], gw[4], gb[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int ch = ch0 + e, tc = ch % cg;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
    gw[e] = ln_w[ch]; gb[e] = ln_b[ch];
  }
  float4 r0[P], r1[P];
  float c0 = 0.f, c1 = 0.f;
  int n_prev = 0;                                      // voxels in this group's cell of the previous plane
#pragma unroll
  for (int pp = 0; pp < P; pp++) r0[pp] = r1[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
  asm volatile("s_barrier" ::: "memory");              // plane 0 is in ring slot 0
  for (int i = 0; i < nplanes; i++) {
    unsigned long long tqa = dbg ? __builtin_amdgcn_s_memtime() : 0;
    const uint32_t bufa = lds_base + (uint32_t)((i % KO::RING) * KO::BUF_BYTES);
    float4 cur[P];
    float cc = 0.f;
#pragma unroll
    for (int pp = 0; pp < P; pp++) cur[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const uint32_t ra = bufa + row_lane, ca = bufa + (uint32_t)KO::CNT_OFF + cnt_lane;
      if constexpr (P == 2 && R == 3) {
        dc_read_plane_p2r3<C>(ra, ca, cur, cc);
      } else {
        dc_read_dx<C, P, R, 0>(ra, ca, cur, cc);
        dc_read_dx<C, P, R, 1>(ra, ca, cur, cc);
        if (R == 3) dc_read_dx<C, P, R, R == 3 ? 2 : 1>(ra, ca, cur, cc);
      }
    }
    const int n_here = lds_rd_b32(bufa + (uint32_t)KO::CNT_OFF + cnt_lane + (uint32_t)((HLO * HY + HLO) * 4));
    if (dbg) { asm volatile("" :: "v"(cur[0].x)); const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_box += tqb - tqa; tqa = tqb; }
    if (i >= R - 1) {
      const int po = pz0 + i - (R - 1) + HLO;
      float4 a[P];
      float den;
      if (R == 3) {
        den = (c0 + c1) + cc;
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          a[pp].x = (r0[pp].x + r1[pp].x) + cur[pp].x; a[pp].y = (r0[pp].y + r1[pp].y) + cur[pp].y;
          a[pp].z = (r0[pp].z + r1[pp].z) + cur[pp].z; a[pp].w = (r0[pp].w + r1[pp].w) + cur[pp].w;
        }
      } else {
        den = c1 + cc;
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          a[pp].x = r1[pp].x + cur[pp].x; a[pp].y = r1[pp].y + cur[pp].y;
          a[pp].z = r1[pp].z + cur[pp].z; a[pp].w = r1[pp].w + cur[pp].w;
        }
      }
      const float inv = den > 0.f ? 1.0f / den : 0.f;
      float Av[P][4];                                   // this cell's normalised neighbour sums: the lane's four channels of every part
#pragma unroll
      for (int pp = 0; pp < P; pp++) { Av[pp][0] = a[pp].x * inv; Av[pp][1] = a[pp].y * inv; Av[pp][2] = a[pp].z * inv; Av[pp][3] = a[pp].w * inv; }
      const int n_own = col_ok ? n_prev : 0;            // the plane that closed is the previous one for both R
      const uint32_t recb = bufa + (uint32_t)KO::REC_OFF + (uint32_t)(grp * DC_INL * 16);
      for (int k = 0; k < n_own; k += 2) {              // the cell's voxels two at a time (ascending id: the fused pre_mix kernel ordered them)
        if (dbg) tq_rounds++;
        const bool hasB = k + 1 < n_own;
        const int kA = k, kB = hasB ? k + 1 : k;
        v4f_t qa, qb;
        lds_rd2_b128(recb + (uint32_t)((kA < DC_INL ? kA : 0) * 16), recb + (uint32_t)((kB < DC_INL ? kB : 0) * 16), qa, qb);
        int4 recA = make_int4(__float_as_int(qa.x), __float_as_int(qa.y), __float_as_int(qa.z), __float_as_int(qa.w));
        int4 recB = make_int4(__float_as_int(qb.x), __float_as_int(qb.y), __float_as_int(qb.z), __float_as_int(qb.w));
        if (kB >= DC_INL) {                             // overflow records (cells with more than DC_INL voxels): ordinary loads
          const int pcell = ((b * PDx + x0 + ix + 1) * PDy + y0 + iy + 1) * PDz + po;
          if (kA >= DC_INL) recA = slots[dc_slot(g, pcell, kA)];
          recB = slots[dc_slot(g, pcell, kB)];
        }
        float4 fx = make_float4(0.f, 0.f, 0.f, 0.f), fy = fx;
        if (OP == LINK_OP_COSX) {
          fx = *reinterpret_cast<const float4 *>(&fin[(int64_t)recA.w * C + ch0]);
          fy = *reinterpret_cast<const float4 *>(&fin[(int64_t)recB.w * C + ch0]);
        }
        // ---- theta / sincos / de-modulate / LayerNorm / store (k_dc_demod's body; both voxels share the A row) ----
        float thA[4], thB[4];
        bool big = false;
        {
          const bool swapped = PAIR && hi && hasB;
          float xa = (float)(swapped ? recB.x : recA.x), ya = (float)(swapped ? recB.y : recA.y), za = (float)(swapped ? recB.z : recA.z);
          float xb = (float)recB.x, yb = (float)recB.y, zb = (float)recB.z;
          if (DIV) { xa = xa / coord_div; ya = ya / coord_div; za = za / coord_div; xb = xb / coord_div; yb = yb / coord_div; zb = zb / coord_div; }
#pragma unroll
          for (int e = 0; e < 4; e++) {
            thA[e] = theta_of(xa, ya, za, w0[e], w1[e], w2[e], al[e]);
            thB[e] = PAIR ? thA[e] : theta_of(xb, yb, zb, w0[e], w1[e], w2[e], al[e]);
            big |= !(fabsf(thA[e]) < 32768.0f) || !(fabsf(thB[e]) < 32768.0f);
          }
        }
        float snA[4], csA[4], snB[4], csB[4];
        if (__builtin_expect(__any(big), 0)) {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            sincos_nocall(thA[e], snA[e], csA[e]);
            if (PAIR) { snB[e] = snA[e]; csB[e] = csA[e]; } else sincos_nocall(thB[e], snB[e], csB[e]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            sincos_small(thA[e], snA[e], csA[e]);
            if (PAIR) { snB[e] = snA[e]; csB[e] = csA[e]; } else sincos_small(thB[e], snB[e], csB[e]);
          }
        }
        if (PAIR) {                                     // this lane evaluated ONE voxel's theta: swap with the partner half
          const bool swapped = hi && hasB;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const float sn = snA[e], cs = csA[e];
            const float so = partner<LPR>(sn), co = partner<LPR>(cs);
            snA[e] = swapped ? so : sn; csA[e] = swapped ? co : cs;
            snB[e] = hi ? sn : so;      csB[e] = hi ? cs : co;
          }
        }
        const float fxa[4] = {fx.x, fx.y, fx.z, fx.w}, fya[4] = {fy.x, fy.y, fy.z, fy.w};
        float nvA[4], nvB[4], sA = 0.f, sB = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float A0 = Av[0][e], A1 = Av[1][e];
          if (OP == LINK_OP_SIN) {                                                   // linkunet.py:148
            nvA[e] = __fsub_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
            nvB[e] = __fsub_rn(__fmul_rn(A0, csB[e]), __fmul_rn(A1, snB[e]));
          } else {                                                                   // :162
            nvA[e] = __fadd_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
            nvB[e] = __fadd_rn(__fmul_rn(A0, csB[e]), __fmul_rn(A1, snB[e]));
          }
          if (OP == LINK_OP_COSX) {                                                  // :176
            nvA[e] = __fadd_rn(nvA[e], __fsub_rn(Av[P - 1][e], link_mul_rn(fxa[e], thA[e])));
            nvB[e] = __fadd_rn(nvB[e], __fsub_rn(Av[P - 1][e], link_mul_rn(fya[e], thB[e])));
          }
          sA += nvA[e]; sB += nvB[e];
        }
        sA = grp_sum<LPR>(sA);
        sB = grp_sum<LPR>(sB);
        const float meanA = sA * (1.0f / C), meanB = sB * (1.0f / C);
        float qA = 0.f, qB = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float dA = nvA[e] - meanA, dB = nvB[e] - meanB;
          qA += dA * dA; qB += dB * dB;
        }
        qA = grp_sum<LPR>(qA);
        qB = grp_sum<LPR>(qB);
        const float rsA = __builtin_amdgcn_rsqf(qA * (1.0f / C) + eps), rsB = __builtin_amdgcn_rsqf(qB * (1.0f / C) + eps);
        float4 oa, ob;
        oa.x = (nvA[0] - meanA) * rsA * gw[0] + gb[0]; oa.y = (nvA[1] - meanA) * rsA * gw[1] + gb[1];
        oa.z = (nvA[2] - meanA) * rsA * gw[2] + gb[2]; oa.w = (nvA[3] - meanA) * rsA * gw[3] + gb[3];
        ob.x = (nvB[0] - meanB) * rsB * gw[0] + gb[0]; ob.y = (nvB[1] - meanB) * rsB * gw[1] + gb[1];
        ob.z = (nvB[2] - meanB) * rsB * gw[2] + gb[2]; ob.w = (nvB[3] - meanB) * rsB * gw[3] + gb[3];
        io_st4(r_out, (uint32_t)recA.w * (uint32_t)C + (uint32_t)ch0, true, oa);
        io_st4(r_out, (uint32_t)recB.w * (uint32_t)C + (uint32_t)ch0, hasB, ob);
      }
    }
#pragma unroll
    for (int pp = 0; pp < P; pp++) { r0[pp] = r1[pp]; r1[pp] = cur[pp]; }
    c0 = c1; c1 = cc;
    n_prev = n_here;
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_pairs += tqb - tqa; tqa = tqb; }
    asm volatile("s_barrier" ::: "memory");            // every wave is done with ring slot i & 1; plane i+1 is in the other one
    if (dbg) tq_bar += __builtin_amdgcn_s_memtime() - tqa;
  }
  if (dbg && lane == 0) {
    unsigned long long *d = dbg + ((size_t)L * 8 + (threadIdx.x >> 6)) * 8;
    d[0] = __builtin_amdgcn_s_memtime() - tq0; d[1] = 0; d[2] = tq_bar; d[3] = tq_box; d[4] = tq_pairs; d[5] = tq_rounds; d[6] = nplanes; d[7] = 1;
  }
}
