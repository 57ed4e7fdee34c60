// link_amd/csrc/dense_k1_impl.h -- the cell-range form of the fused pre_mix kernel (K1 of the dense-cell layout), as device
// functions + the stand-alone kernel built from them.  Included inside an IO namespace (DC_IO, DC_IO_NS: dense_fused_impl.h) by
// dense_fused_impl.h -- and by dense_batch.hip (round 6), whose persistent K1-role kernel stages the parameters ONCE per launch and
// then runs dc_k1_range over (frame, cell range) items of a batch of frames.  Split in round 6 from the body of
// k_dc_premix_modsum: the statements are the same, in the same order (tools/kernel_regs.py: registers as before the split).
//
//   dc_k1_stage     W (fp16 hi | lo image or fp32), LayerNorm weight / bias, theta weights -> LDS; "a weight outside the fp16 range",
//                   "some theta may leave the fast sincos range" (per thread: the caller votes across the workgroup)
//   dc_k1_prefetch  the first chunk's cell ids, inline records and counts -- requested BEFORE the staging so the two latencies overlap
//   dc_k1_range     one wave's range of cells [c_begin, c_end): counts -> wave prefix scan -> id-ordered voxel list in LDS -> tiles
//                   of 16 voxels (row gather, contraction on the matrix cores, LayerNorm, theta / sincos / modulate, X tile -> LDS,
//                   per-cell sums, one S row store per cell); publishes cell_n, zeroes cnt, writes the sorted records back
#pragma once

#ifndef DC_K1_LCAP
#define DC_K1_LCAP 352
#endif
#ifndef DC_K1_NW
#define DC_K1_NW 4
#endif
#ifndef DC_K1_ABL
#define DC_K1_ABL 0      /* ablation builds for tools/ab_bench.sh (wrong results!): 1 no MFMA, 2 no sincos, 4 no per-cell sums, 8 rows from one address */
#endif
#ifndef DC_K1_SPLIT
#define DC_K1_SPLIT 1    /* pre_mix contraction as an fp16 hi/lo split on the f16 matrix cores (see mfma_tile); 0 = v_mfma_f32_16x16x4_f32 */
#endif
#ifndef DC_K2_ABL
#define DC_K2_ABL 0      /* ablation builds of the split gather kernel (wrong results!): 1 no sincos, 2 no LayerNorm reductions, 4 no pair processing, 8 no output stores */
#endif
#ifndef DC_K2_SECOND_ROUND
#define DC_K2_SECOND_ROUND 1 /* split gather kernel: the producer waves take pairs 16..31 of the previous plane (0: the consumer waves take every pair) */
#endif
#ifndef DC_K2_PIPE_READS
#define DC_K2_PIPE_READS 0   /* split gather kernel, producer half: the plane's three x-offsets as one pipelined LDS request (dense_gather.h) */
#endif
#ifndef DC_K1_MFMA32
#define DC_K1_MFMA32 1   /* pairs of 16-channel blocks on v_mfma_f32_16x16x32_f16 */
#endif
#ifndef DC_K1_SUMB
#define DC_K1_SUMB 8       /* X rows in flight per batch of the per-cell sums */
#endif
#ifndef DC_K1_SWAP_SUMS
#define DC_K1_SWAP_SUMS 1  /* LayerNorm statistics over a voxel's four lane groups through v_permlane32_swap / v_permlane16_swap (VALU) instead
                              of two ds_bpermute each: four dependent LDS round trips per tile leave the wave's chain (round 5) */
#endif
// v + v[lane ^ 16] + v[lane ^ 32] + v[lane ^ 48]
__device__ __forceinline__ float dc_k1_sum_groups(float v) {
#if DC_K1_SWAP_SUMS
  // (inline asm: hipcc 7.2 returns the first result of the permlane swap builtins for both elements -- tile_common.h)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  a += b;
  b = a;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
#else
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
#endif
}

template <int C, int OP>
struct dc_k1_cfg {
  static constexpr int T = C / 16;
  static constexpr int P = op_parts<OP>::value;
  static constexpr int LDW = C + 4;
  static constexpr int RB = P * C * 4;                 // bytes of one X / S row
  static constexpr int XROW = RB + 16;                 // LDS row stride: +4 dwords -> conflict-free b128 writes
  static constexpr int RGL = P * C / 4;                // lanes holding one row (16 B each)
  static constexpr int RGS = RGL <= 8 ? 8 : (RGL <= 16 ? 16 : (RGL <= 32 ? 32 : 64));
  static constexpr int RG = 64 / RGS;                  // rows summed side by side per wave
  static constexpr int LCAP = DC_K1_LCAP;                   // records of one cell range kept in LDS (>= 7^3; two workgroups must fit 160 KB)
  static constexpr int LDH = 2 * C + 8;                // fp16 image of W, row co = [hi(C) | lo(C) | pad]: 4 * odd dwords, conflict-free ds_read_b64
  static constexpr int WIMG_BYTES = DC_K1_SPLIT ? C * LDH * 2 : C * LDW * 4;   // same size as the fp32 image: two workgroups per CU
  static constexpr int W_BYTES = WIMG_BYTES + (2 * C + 4 * C) * 4;   // W, LayerNorm weight / bias, theta weights (w0 | w1 | w2 | alpha per channel)
  static constexpr int LIST_OFF = 0;
  static constexpr int SCELL_OFF = LCAP * 16;          // padded cell id of every list slot
  static constexpr int X_OFF = SCELL_OFF + LCAP * 4;
  static constexpr int WAVE_BYTES = X_OFF + 16 * XROW;
  static constexpr int NW = DC_K1_NW;                  // waves per workgroup (they share one W image)
  static constexpr int LDS_BYTES = W_BYTES + NW * WAVE_BYTES;
};


// NB = number of distinct 16-channel theta blocks of a voxel: channel ch uses theta[ch % cg]; when cg is a
// multiple of 16 the MFMA channel block tp (channels 16 tp + 4 g + r of lane group g) uses theta block
// tp % (cg/16), so a lane evaluates 4*NB sincos per voxel instead of 4*T; otherwise NB = T.
// PIPE: software-pipelined tiles (MFMAs of tile t+1 issued inside tile t's VALU block; two accumulator and
// row sets: ~256 registers, 2 waves per SIMD and nothing else fits beside them) or plain tiles (rows of t+1 in
// flight while t is multiplied, then finished: ~130 registers, so that a second frame's kernels can share the
// SIMDs).  Same arithmetic, bit for bit.
#ifndef DC_K1_WAVES
#define DC_K1_WAVES 2     /* register budget = 512 / this; LDS (80 KB per workgroup) allows 2 workgroups per CU anyway, and at 3 the tile body spills (A/B: LINK_AMD_CXXFLAGS=-DDC_K1_WAVES=3) */
#endif
// padded cell id owned by `lane` in the chunk starting at interior cell number `chunk`
__device__ __forceinline__ int dc_k1_cell_of(const link_dc_grid_t &g, int chunk, int nrem, int lane) {
  const int q = chunk + (lane < nrem ? lane : 0);
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int z = q % Dz;
  int t = q / Dz;
  const int y = t % Dy;
  t /= Dy;
  return dc_cell(g, t % Dx, y, z, t / Dx);
}

// what the first chunk of a range needs, requested early: its lanes' cell ids, inline records and counts (plain values, not a
// struct: as a struct the four records went through 32 bytes of scratch)
// COH (round 6, the persistent batch kernel): counts and records were written by ANOTHER workgroup of a launch that is still
// running (the batch's slot insert) -> sc1 loads (dense_common.h); the stand-alone kernel reads them behind a kernel boundary.
template <bool COH = false>
__device__ __forceinline__ void dc_k1_prefetch(int &pc_f, int &nv_f, int4 &rf0, int4 &rf1, int4 &rf2, int4 &rf3,
                                               const link_dc_grid_t &g, const int4 *__restrict__ slots,
                                               const uint32_t *__restrict__ csrc, int c_begin, int c_end, int lane) {
  pc_f = 0; nv_f = 0;
  rf0 = make_int4(0, 0, 0, 0); rf1 = rf0; rf2 = rf0; rf3 = rf0;
  if (c_begin < c_end) {
    pc_f = dc_k1_cell_of(g, c_begin, (c_end - c_begin < 64) ? c_end - c_begin : 64, lane);
    const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
    const __amdgpu_buffer_rsrc_t r_c = dc_rsrc(csrc, (uint32_t)(g.vp * 4));
    rf0 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc_f * DC_INL + 0); rf1 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc_f * DC_INL + 1);
    rf2 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc_f * DC_INL + 2); rf3 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc_f * DC_INL + 3);
    nv_f = ld4i_c<COH>(r_c, csrc, (uint32_t)pc_f);
  }
}

// stage W and the LayerNorm / theta parameters of the block into the workgroup's LDS image (every thread of the workgroup calls it;
// the caller synchronises).  w_big / th_big: this thread saw a weight outside the fp16 split's range / a channel whose theta may
// leave the fast sincos range on this grid.
template <int C, int OP>
__device__ __forceinline__ void dc_k1_stage(char *smem_raw, const float *__restrict__ w_pre, const float *__restrict__ ln_w,
                                            const float *__restrict__ ln_b, const float *__restrict__ w_pos,
                                            const float *__restrict__ alpha, int cg, float coord_div, const link_dc_grid_t &g,
                                            int tid, bool &w_big, bool &th_big) {
  using K = dc_k1_cfg<C, OP>;
  constexpr int LDW = K::LDW;
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  float *ln_lds = reinterpret_cast<float *>(smem_raw + K::WIMG_BYTES);
  float *pw_lds = ln_lds + 2 * C;                      // read per tile: 32 fewer live registers than per-lane copies
  (void)w_lds;
  {                                                    // stage W and the LayerNorm parameters
    // all loads first, ONE wait, then the LDS writes -- no predicate around the writes (hipcc turns a
    // predicated write into load / wait / write per iteration: four dependent round trips at C = 64)
    constexpr int NF4 = C * C / 4;                     // float4 pieces of W
    constexpr int NT = 64 * K::NW;
    constexpr int NV = (NF4 + NT - 1) / NT;
    float4 wv[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int e = (i * NT + tid) * 4;
      wv[i] = *reinterpret_cast<const float4 *>(&w_pre[(NF4 % NT == 0 || e < C * C) ? e : 0]);
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int e = (i * NT + tid) * 4;
      if (NF4 % NT != 0 && e >= C * C) e = 0;         // C = 16: surplus lanes rewrite piece 0 with its own value
      const int r = e / C, col = e - r * C;
      const float4 wq = (NF4 % NT == 0 || (i * NT + tid) * 4 < C * C) ? wv[i] : *reinterpret_cast<const float4 *>(&w_pre[0]);
      if constexpr (DC_K1_SPLIT) {
        // w = hi + lo with hi = fp16(w), lo = fp16(w - hi): 22 mantissa bits, exact products on the f16 matrix cores
        uint2 hi, lo;
        dc_split4(wq, hi, lo);
        unsigned short *wh = reinterpret_cast<unsigned short *>(smem_raw);
        *reinterpret_cast<uint2 *>(&wh[r * K::LDH + col]) = hi;
        *reinterpret_cast<uint2 *>(&wh[r * K::LDH + C + col]) = lo;
        w_big |= !(fmaxf(fmaxf(fabsf(wq.x), fabsf(wq.y)), fmaxf(fabsf(wq.z), fabsf(wq.w))) < 32768.0f);
      } else {
        *reinterpret_cast<float4 *>(&w_lds[r * LDW + col]) = wq;
      }
    }
    if (tid < C) ln_lds[tid] = ln_w[tid];
    else if (tid < 2 * C) ln_lds[tid] = ln_b[tid - C];
    if (tid < C) {                                     // theta weights of channel tid (channel ch uses theta[ch % cg])
      const int tc = tid % cg;
      const float q0 = w_pos[3 * tc + 0], q1 = w_pos[3 * tc + 1], q2 = w_pos[3 * tc + 2], qa = alpha ? alpha[tc] : 1.0f;
      pw_lds[tid] = q0; pw_lds[C + tid] = q1; pw_lds[2 * C + tid] = q2;
      pw_lds[3 * C + tid] = qa;
      if (DC_THETA_BOUND) {                            // can any theta of this launch leave the fast sincos range? (dense_common.h)
        float ax, ay, az;
        dc_coord_absmax(g, coord_div, ax, ay, az);
        th_big = dc_theta_leaves_fast_range(ax, ay, az, q0, q1, q2, qa);
      }
    }
  }
}

// One wave's range of cells.  pc_f .. rf3: what dc_k1_prefetch requested for the range's first chunk; w_big / th_slow: the
// workgroup-uniform verdicts of the staging; wid: the wave's number in the launch (profiling rows only).
template <int C, int OP, int NB, bool PIPE, bool COH = false>
__device__ __forceinline__ void dc_k1_range(char *smem_raw, const void *__restrict__ feats, int4 *__restrict__ slots,
                                            uint32_t *__restrict__ cnt, int32_t *__restrict__ cell_n,
                                            const float *__restrict__ w_pre, float coord_div, float eps, int64_t n,
                                            const link_dc_grid_t &g, bool warm, float *__restrict__ S_, float *__restrict__ fin,
                                            unsigned long long *__restrict__ dbg, int c_begin,
                                            int c_end, const int pc_f, const int nv_f, const int4 rf0, const int4 rf1, const int4 rf2,
                                            const int4 rf3, bool w_big, bool th_slow, int wid,
                                            unsigned long long tq0, unsigned long long tq1) {
  using K = dc_k1_cfg<C, OP>;
  constexpr int T = K::T, P = K::P, LDW = K::LDW;
  unsigned long long tq_cell = 0, tq_fill = 0, tq_body = 0, tq_sum = 0;
  int tq_tiles = 0;
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  float *ln_lds = reinterpret_cast<float *>(smem_raw + K::WIMG_BYTES);
  float *pw_lds = ln_lds + 2 * C;                      // read per tile: 32 fewer live registers than per-lane copies
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  char *wbase = smem_raw + K::W_BYTES + wave * K::WAVE_BYTES;
  int4 *list = reinterpret_cast<int4 *>(wbase + K::LIST_OFF);
  int *scell = reinterpret_cast<int *>(wbase + K::SCELL_OFF);
  char *xbuf = wbase + K::X_OFF;
  const uint32_t *__restrict__ csrc = warm ? reinterpret_cast<const uint32_t *>(cell_n) : cnt;
  const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, (uint32_t)((g.vp + 1) * K::RB));
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * C * 4));     // written for cos_x only
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_n = dc_rsrc(cell_n, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const int rl = lane;                                 // lane's 16-byte piece of an X / S row
  const bool ract = rl < K::RGL;

  for (int chunk = c_begin; chunk < c_end;) {
    const int nrem = (c_end - chunk < 64) ? c_end - chunk : 64;
    unsigned long long tqa = dbg ? DC_NOW() : 0;
    // ---- cell lanes: count, padded cell id, inline records (all requested before anything is consumed) ----
    int pc, nv;
    int4 r0, r1, r2, r3;
    if (chunk == c_begin) {                             // wave-uniform: the first chunk was requested before the staging
      pc = pc_f; nv = nv_f; r0 = rf0; r1 = rf1; r2 = rf2; r3 = rf3;
    } else {
      pc = dc_k1_cell_of(g, chunk, nrem, lane);
      r0 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc * DC_INL + 0); r1 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc * DC_INL + 1);
      r2 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc * DC_INL + 2); r3 = ld16i_c<COH>(r_slots, slots, (uint32_t)pc * DC_INL + 3);
      nv = ld4i_c<COH>(warm ? r_n : r_cnt, csrc, (uint32_t)pc);
    }
    constexpr int LCAPX = K::LCAP;
    nv = nv < g.k ? nv : g.k;
    nv = nv < LCAPX ? nv : LCAPX;
    if (lane >= nrem) nv = 0;
    int incl = nv;                                      // inclusive prefix over the wave: DPP row scan (zeros shifted in) + the three row
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // totals as scalars (round 5; were six dependent ds_bpermute
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // round trips per chunk: the __shfl_up form)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
    {
      const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47);
      incl += gq == 0 ? 0 : (gq == 1 ? t0 : (gq == 2 ? t0 + t1 : t0 + t1 + t2));
    }
    const unsigned long long fit = __ballot(lane < nrem && incl <= LCAPX);
    const int nfit = __builtin_amdgcn_readfirstlane(__popcll(fit));      // >= 1: a cell never exceeds LCAP
    const int Ttot = __builtin_amdgcn_readlane(incl, nfit - 1);
    if (lane < nfit) {
      const int excl = incl - nv;
      // order the inline records by voxel id: keys id*4+slot through a 5-exchange network
      int k0 = nv > 0 ? r0.w * 4 + 0 : INT_MAX, k1 = nv > 1 ? r1.w * 4 + 1 : INT_MAX;
      int k2 = nv > 2 ? r2.w * 4 + 2 : INT_MAX, k3 = nv > 3 ? r3.w * 4 + 3 : INT_MAX;
      int a, bb;
      a = min(k0, k1); bb = max(k0, k1); k0 = a; k1 = bb;
      a = min(k2, k3); bb = max(k2, k3); k2 = a; k3 = bb;
      a = min(k0, k2); bb = max(k0, k2); k0 = a; k2 = bb;
      a = min(k1, k3); bb = max(k1, k3); k1 = a; k3 = bb;
      a = min(k1, k2); bb = max(k1, k2); k1 = a; k2 = bb;
      const int ks[4] = {k0, k1, k2, k3};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int s = ks[j] & 3;
        int4 r;
        r.x = s == 0 ? r0.x : (s == 1 ? r1.x : (s == 2 ? r2.x : r3.x));
        r.y = s == 0 ? r0.y : (s == 1 ? r1.y : (s == 2 ? r2.y : r3.y));
        r.z = s == 0 ? r0.z : (s == 1 ? r1.z : (s == 2 ? r2.z : r3.z));
        r.w = ks[j] >> 2;
        if (j < nv) { list[excl + j] = r; scell[excl + j] = pc; }
        // the id-ordered records go back to the slot list: the fused gather+demod walks it and must pair the
        // same voxels in every run (rank order from the atomics is not reproducible)
        st16i_c<COH>(r_slots, (j < nv && nv <= DC_INL && !warm) ? ((uint32_t)pc * DC_INL + j) * 16u : DC_OOB, r);
      }
      for (int k = DC_INL; k < nv; k++) {               // overflow records: insertion by id (rare)
        const int4 r = ld16i_c<COH>(r_slots, slots, dc_slot(g, pc, k));
        scell[excl + k] = pc;
        int pos = k;
        while (pos > 0 && list[excl + pos - 1].w > r.w) {
          list[excl + pos] = list[excl + pos - 1];
          pos--;
        }
        list[excl + pos] = r;
      }
      if (nv > DC_INL && !warm)
        for (int k = 0; k < nv; k++) st16i_c<COH>(r_slots, dc_slot(g, pc, k) * 16u, list[excl + k]);
    }
    {                                                   // publish the counts, reset the counters
      const uint32_t coff = (lane < nfit && !warm) ? (uint32_t)pc * 4u : DC_OOB;
      st4i_c<COH>(r_n, coff, nv);
      st4i(r_cnt, coff, 0);
    }
    for (unsigned long long em = __ballot(lane < nfit && nv == 0); em; em &= em - 1) {   // empty cells: zero rows
      const int pcj = __builtin_amdgcn_readlane(pc, __builtin_ctzll(em));
      st16(r_S, ract ? (uint32_t)pcj * (uint32_t)K::RB + (uint32_t)rl * 16u : DC_OOB, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (dbg) { const unsigned long long tqb = DC_NOW(); tq_cell += tqb - tqa; tqa = tqb; }
    // ---- tiles of 16 voxels, software-pipelined: while tile t's accumulators go through LayerNorm /
    // theta / sincos / modulate on the VALU, the 64 MFMAs of tile t+1 are issued from the same instruction
    // stream (one basic block: nothing in it depends on them), and the rows of tile t+2 are in flight ----
    const int ntile = (Ttot + 15) >> 4;
    const int nloop = ntile;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto ld_rows = [&](int tile_idx, int4 &rr, float4 (&ff)[T]) {
      int sl = 16 * tile_idx + li;
      sl = sl < Ttot ? sl : Ttot - 1;
      rr = list[sl];
#pragma unroll
      for (int tt = 0; tt < T; tt++)
        ff[tt] = io_ld4(feats, (int64_t)((DC_K1_ABL & 8) ? li : rr.w) * C + 16 * tt + 4 * gq);
    };
    // pre_mix contraction D[co][voxel] = sum_ci W[co][ci] x[voxel][ci].  DC_K1_SPLIT: both operands as fp16 hi + lo
    // (22 mantissa bits each), products hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x16_f16 -- exact products, fp32
    // accumulation, the dropped lo*lo term is 2^-22 relative -- 48 matrix instructions of 4 passes instead of 64 of 8
    // (the matrix pipe's time is fully exposed in this kernel: without it the launch is 6.8 us shorter).  fp16 rows
    // have lo = 0: two products.  Values outside the fp16 range (|x| or |w| >= 2^15: never on LayerNorm-ed networks)
    // take the fp32 instruction with W read from global memory -- slow, exact, wave-uniform.
    auto mfma_tile = [&](const float4 (&ff)[T], floatx4 (&cc)[T]) {
      if (DC_K1_ABL & 1) {
#pragma unroll
        for (int tp = 0; tp < T; tp++) cc[tp] = (floatx4){ff[tp].x, ff[tp].y, ff[tp].z, ff[tp].w};
        return;
      }
#pragma unroll
      for (int tp = 0; tp < T; tp++) cc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
      if constexpr (DC_K1_SPLIT) {
        uint2 bh[T], bl[T];
        float mx = 0.f;
#pragma unroll
        for (int tt = 0; tt < T; tt++) {
          dc_split4(ff[tt], bh[tt], bl[tt]);
          mx = fmaxf(mx, fmaxf(fmaxf(fabsf(ff[tt].x), fabsf(ff[tt].y)), fmaxf(fabsf(ff[tt].z), fabsf(ff[tt].w))));
        }
        if (__builtin_expect(!(w_big || __any(!(mx < 32768.0f))), 1)) {
          const unsigned short *wh = reinterpret_cast<const unsigned short *>(smem_raw);
          if constexpr (DC_K1_MFMA32 && T % 2 == 0) {
#pragma unroll
            for (int tt = 0; tt < T; tt += 2) {
              uint2 ah[2][T], al[2][T];
#pragma unroll
              for (int h = 0; h < 2; h++)
#pragma unroll
                for (int tp = 0; tp < T; tp++) {
                  ah[h][tp] = *reinterpret_cast<const uint2 *>(&wh[(16 * tp + li) * K::LDH + 16 * (tt + h) + 4 * gq]);
                  al[h][tp] = *reinterpret_cast<const uint2 *>(&wh[(16 * tp + li) * K::LDH + C + 16 * (tt + h) + 4 * gq]);
                }
#pragma unroll
              for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16x2(al[0][tp], al[1][tp], bh[tt], bh[tt + 1], cc[tp]);
              if constexpr (IO != 1) {
#pragma unroll
                for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16x2(ah[0][tp], ah[1][tp], bl[tt], bl[tt + 1], cc[tp]);
              }
#pragma unroll
              for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16x2(ah[0][tp], ah[1][tp], bh[tt], bh[tt + 1], cc[tp]);
            }
            return;
          }
#pragma unroll
          for (int tt = 0; tt < T; tt++) {
            uint2 ah[T], al[T];
#pragma unroll
            for (int tp = 0; tp < T; tp++) {
              ah[tp] = *reinterpret_cast<const uint2 *>(&wh[(16 * tp + li) * K::LDH + 16 * tt + 4 * gq]);
              al[tp] = *reinterpret_cast<const uint2 *>(&wh[(16 * tp + li) * K::LDH + C + 16 * tt + 4 * gq]);
            }
#pragma unroll
            for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16(al[tp], bh[tt], cc[tp]);
            if constexpr (IO != 1) {                      // fp16 rows: lo = 0 exactly
#pragma unroll
              for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16(ah[tp], bl[tt], cc[tp]);
            }
#pragma unroll
            for (int tp = 0; tp < T; tp++) cc[tp] = dc_mfma_f16(ah[tp], bh[tt], cc[tp]);
          }
          return;
        }
      }
#pragma unroll
      for (int tt = 0; tt < T; tt++) {
        float4 a[T];
#pragma unroll
        for (int tp = 0; tp < T; tp++) {
          if constexpr (DC_K1_SPLIT) a[tp] = *reinterpret_cast<const float4 *>(&w_pre[(16 * tp + li) * C + 16 * tt + 4 * gq]);
          else a[tp] = *reinterpret_cast<const float4 *>(&w_lds[(16 * tp + li) * LDW + 16 * tt + 4 * gq]);
        }
#pragma unroll
        for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].x, ff[tt].x, cc[tp], 0, 0, 0);
#pragma unroll
        for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].y, ff[tt].y, cc[tp], 0, 0, 0);
#pragma unroll
        for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].z, ff[tt].z, cc[tp], 0, 0, 0);
#pragma unroll
        for (int tp = 0; tp < T; tp++) cc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].w, ff[tt].w, cc[tp], 0, 0, 0);
      }
    };
    // Two register sets (A/B) for the records, rows and accumulators, used alternately by the two halves of
    // the unrolled loop: nothing is ever copied, so no instruction of a step waits for the loads it issued.
    int4 recA = make_int4(0, 0, 0, 0), recB = recA, recC = recA;
    float4 fA[T], fB[T];
    floatx4 acA[T], acB[T];
    if (PIPE && ntile > 0) {                            // pipeline fill: tile 0 multiplied, tile 1 requested
      ld_rows(0, recA, fB);
      ld_rows(ntile > 1 ? 1 : 0, recB, fA);
      mfma_tile(fB, acA);
    }
    if (dbg) { const unsigned long long tqb = DC_NOW(); tq_fill += tqb - tqa; tqa = tqb; }
    // step t: finishes tile t (record rec, accumulators ac), multiplies tile t+1 (record recn, rows fn) into
    // acn, requests tile t+2 (record rec2, rows f2)
    // `phase`: 0 = finish the tile and form its sums (pipelined variant); 1 = finish only; 2 = sums only.  The plain
    // loop runs the sums of tile t-1 AFTER the MFMAs of tile t were issued: the row stores of a tile are then old by
    // the time the next row loads are waited for (s_waitcnt vmcnt counts stores too, and waiting on just-issued
    // write-through stores every tile was the largest stall of this kernel), and the matrix pipe works through
    // tile t while the VALU adds up tile t-1.
    auto step = [&](auto phase_tag, int t, const int4 &rec, const int4 &recn, int4 &rec2, const float4 (&fn)[T], float4 (&f2)[T],
                    const floatx4 (&ac)[T], floatx4 (&acn)[T]) {
      constexpr int PHASE = decltype(phase_tag)::value;
      if (PHASE != 2 && ntile > 0) {
        const int slot = 16 * t + li;
        // theta of this voxel's blocks; a wave whose arguments all sit below 2^15 takes the branch-free body
        float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
        if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
        float th[NB][4];
        bool big = false;
#pragma unroll
        for (int tb = 0; tb < NB; tb++) {
          const float4 q0 = *reinterpret_cast<const float4 *>(&pw_lds[16 * tb + 4 * gq]);
          const float4 q1 = *reinterpret_cast<const float4 *>(&pw_lds[C + 16 * tb + 4 * gq]);
          const float4 q2 = *reinterpret_cast<const float4 *>(&pw_lds[2 * C + 16 * tb + 4 * gq]);
          const float4 qa = *reinterpret_cast<const float4 *>(&pw_lds[3 * C + 16 * tb + 4 * gq]);
          th[tb][0] = theta_of(x, y, z, q0.x, q1.x, q2.x, qa.x); th[tb][1] = theta_of(x, y, z, q0.y, q1.y, q2.y, qa.y);
          th[tb][2] = theta_of(x, y, z, q0.z, q1.z, q2.z, qa.z); th[tb][3] = theta_of(x, y, z, q0.w, q1.w, q2.w, qa.w);
          if (!DC_THETA_BOUND) {
#pragma unroll
            for (int r = 0; r < 4; r++) big |= !(fabsf(th[tb][r]) < 32768.0f);
          }
        }
        const bool slow = DC_THETA_BOUND ? th_slow : __any(big);
        const bool more = PIPE && t + 1 < ntile;
        (void)recn;
        auto body = [&](auto more_tag, auto slow_tag) {
          constexpr bool MORE = decltype(more_tag)::value, SLOW = decltype(slow_tag)::value;
          if constexpr (MORE) mfma_tile(fn, acn);
          if constexpr (MORE) ld_rows(t + 2 < ntile ? t + 2 : t + 1, rec2, f2);
          float sn[NB][4], cs[NB][4];
#pragma unroll
          for (int tb = 0; tb < NB; tb++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              if (DC_K1_ABL & 2) { sn[tb][r] = th[tb][r]; cs[tb][r] = 1.0f - th[tb][r]; }
              else if constexpr (SLOW) sincos_nocall(th[tb][r], sn[tb][r], cs[tb][r]);
              else sincos_small(th[tb][r], sn[tb][r], cs[tb][r]);
            }
          // LayerNorm over the voxel's C channels: 16 in-lane values + the 4 lane groups
          float s = 0.f;
#pragma unroll
          for (int tp = 0; tp < T; tp++) s += (ac[tp][0] + ac[tp][1]) + (ac[tp][2] + ac[tp][3]);
          s = dc_k1_sum_groups(s);
          const float mean = s * (1.0f / C);
          float qq = 0.f;
#pragma unroll
          for (int tp = 0; tp < T; tp++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const float d = ac[tp][r] - mean;
              qq += d * d;
            }
          qq = dc_k1_sum_groups(qq);
          const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + eps);
#pragma unroll
          for (int tp = 0; tp < T; tp++) {
            const float4 lw = *reinterpret_cast<const float4 *>(&ln_lds[16 * tp + 4 * gq]);
            const float4 lb = *reinterpret_cast<const float4 *>(&ln_lds[C + 16 * tp + 4 * gq]);
            const float fv[4] = {(ac[tp][0] - mean) * rstd * lw.x + lb.x, (ac[tp][1] - mean) * rstd * lw.y + lb.y,
                                 (ac[tp][2] - mean) * rstd * lw.z + lb.z, (ac[tp][3] - mean) * rstd * lw.w + lb.w};
            if (OP == LINK_OP_COSX)                     // the de-modulation of cos_x needs fin (linkunet.py:176)
              st16(r_fin, slot < Ttot ? (uint32_t)rec.w * (uint32_t)(C * 4) + (uint32_t)((16 * tp + 4 * gq) * 4) : DC_OOB,
                   make_float4(fv[0], fv[1], fv[2], fv[3]));
            const int tb = tp % NB;
            float p0[4], p1[4], p2[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
              if (OP == LINK_OP_SIN) { p0[r] = fv[r] * sn[tb][r]; p1[r] = fv[r] * cs[tb][r]; }
              else { p0[r] = fv[r] * cs[tb][r]; p1[r] = fv[r] * sn[tb][r]; }
              p2[r] = fv[r] * th[tb][r];
            }
            char *xr = xbuf + li * K::XROW + (16 * tp + 4 * gq) * 4;
            *reinterpret_cast<float4 *>(xr) = make_float4(p0[0], p0[1], p0[2], p0[3]);
            *reinterpret_cast<float4 *>(xr + C * 4) = make_float4(p1[0], p1[1], p1[2], p1[3]);
            if (P == 3) *reinterpret_cast<float4 *>(xr + 2 * C * 4) = make_float4(p2[0], p2[1], p2[2], p2[3]);
          }
          if constexpr (MORE && !SLOW) {
#pragma unroll
            for (int i = 0; i < T * T * 4; i++) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
              __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // six VALU
            }
          }
        };
        if (__builtin_expect(slow, 0)) {                // never on sane inputs: no overlap, smallest code
          if (more) {
            mfma_tile(fn, acn);
            ld_rows(t + 2 < ntile ? t + 2 : t + 1, rec2, f2);
          }
          body(std::false_type{}, std::true_type{});
        } else {
          if (more) body(std::true_type{}, std::false_type{}); else body(std::false_type{}, std::false_type{});
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      if (PHASE != 2 && dbg) { const unsigned long long tqb = DC_NOW(); tq_body += tqb - tqa; tqa = tqb; tq_tiles++; }
      if (PHASE == 1 || (DC_K1_ABL & 4)) return;
      // ---- per-cell sums of this tile: ONE stream over the 16 rows in slot order (ascending voxel id inside a
      // cell), every lane owning 16 bytes of the row; a row that closes its cell (the next slot belongs to
      // another cell) is followed by the cell's S row store.  All rows are requested up front, the close
      // flags are a 16-bit ballot, the closing cell's id comes by v_readlane: no dependent LDS round trips. ----
      {
        const int slot = 16 * t + li;
        const int myc = slot < Ttot ? scell[slot] : -1;
        const int nxc = slot + 1 < Ttot ? scell[slot + 1] : -2;
        const unsigned closes = (unsigned)__ballot(gq == 0 && slot < Ttot && myc != nxc);
        const char *xrow = xbuf + (ract ? rl : 0) * 16;
#pragma unroll
        for (int h = 0; h < 16; h += DC_K1_SUMB) {
          float4 v[DC_K1_SUMB];
#pragma unroll
          for (int k = 0; k < DC_K1_SUMB; k++) v[k] = *reinterpret_cast<const float4 *>(xrow + (h + k) * K::XROW);
#pragma unroll
          for (int k = 0; k < DC_K1_SUMB; k++) {
            if (16 * t + h + k < Ttot) {                // wave-uniform
              acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w;
              if ((closes >> (h + k)) & 1u) {           // wave-uniform
                const int pcs = __builtin_amdgcn_readlane(myc, h + k);
                st16(r_S, ract ? (uint32_t)pcs * (uint32_t)K::RB + (uint32_t)rl * 16u : DC_OOB, acc);
                acc = make_float4(0.f, 0.f, 0.f, 0.f);
              }
            }
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      if (dbg) { const unsigned long long tqb = DC_NOW(); tq_sum += tqb - tqa; tqa = tqb; }
    };
    // rows of tile t+1 sit in fA for even t and fB for odd t, accumulators of tile t in acA / acB likewise;
    // the three records rotate by plain copies (they come from LDS: no VMEM wait is involved)
    if constexpr (PIPE) {
      for (int t = 0; t < nloop; t += 2) {
        step(std::integral_constant<int, 0>{}, t, recA, recB, recC, fA, fB, acA, acB);
        recA = recB; recB = recC;
        if (t + 1 < nloop) {
          step(std::integral_constant<int, 0>{}, t + 1, recA, recB, recC, fB, fA, acB, acA);
          recA = recB; recB = recC;
        }
      }
    } else {
      // plain tiles: rows of tile t in fA (even t) / fB (odd t), the other set receives tile t+1 meanwhile
      using body_only = std::integral_constant<int, 1>;
      using sums_only = std::integral_constant<int, 2>;
      if (ntile > 0) ld_rows(0, recA, fA);
      for (int t = 0; t < nloop; t += 2) {
        if (t + 1 < ntile) ld_rows(t + 1, recB, fB);
        mfma_tile(fA, acA);
        if (t > 0) step(sums_only{}, t - 1, recB, recB, recC, fB, fB, acA, acB);
        step(body_only{}, t, recA, recB, recC, fB, fB, acA, acB);
        if (t + 1 < nloop) {
          if (t + 2 < ntile) ld_rows(t + 2, recA, fA);
          mfma_tile(fB, acA);                          // acA is free again: tile t was finished above
          step(sums_only{}, t, recA, recA, recC, fA, fA, acA, acB);
          step(body_only{}, t + 1, recB, recA, recC, fA, fA, acA, acB);
        }
      }
      if (nloop > 0) step(sums_only{}, nloop - 1, recA, recA, recC, fA, fA, acA, acB);
    }
    chunk += nfit;
  }
  if (dbg && lane == 0) {
    unsigned long long *d = dbg + (size_t)wid * 8;
    const unsigned long long te = DC_NOW();
    d[0] = tq1 - tq0; d[1] = tq_cell; d[2] = tq_fill; d[3] = tq_body; d[4] = tq_sum; d[5] = te - tq0; d[6] = tq_tiles; d[7] = tq0;
  }
}

template <int C, int OP, int NB, bool PIPE>
__global__ void __launch_bounds__(64 * DC_K1_NW, PIPE ? 2 : DC_K1_WAVES) k_dc_premix_modsum(
    const void *__restrict__ feats, int4 *__restrict__ slots, uint32_t *__restrict__ cnt,
    int32_t *__restrict__ cell_n, const float *__restrict__ w_pre, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, const float *__restrict__ w_pos, const float *__restrict__ alpha, int cg,
    float coord_div, float eps, int64_t n, link_dc_grid_t g, int cpw, bool warm, float *__restrict__ S_,
    float *__restrict__ fin, int32_t *__restrict__ hdr, unsigned long long *__restrict__ dbg) {
  using K = dc_k1_cfg<C, OP>;
  DC_PROF_PTR(dbg);
  // optional phase timing (tools/dcbench.py --phases): per wave 8 slots of s_memtime deltas
  unsigned long long tq0 = dbg ? DC_NOW() : 0, tq1 = 0;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // the first chunk's cell records and counts are requested BEFORE W is staged: the two latencies overlap
  const int Vi = g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  const int wid = blockIdx.x * K::NW + wave;
  const int c_begin = wid * cpw;
  const int c_end = (c_begin + cpw < Vi) ? c_begin + cpw : Vi;
  int pc_f, nv_f;
  int4 rf0, rf1, rf2, rf3;
  dc_k1_prefetch<false>(pc_f, nv_f, rf0, rf1, rf2, rf3, g, slots, warm ? reinterpret_cast<const uint32_t *>(cell_n) : cnt, c_begin, c_end, lane);
  bool w_big = false;                                  // a weight outside the fp16 split's range: fp32 contraction (never on sane models)
  bool th_big = false;
  dc_k1_stage<C, OP>(smem_raw, w_pre, ln_w, ln_b, w_pos, alpha, cg, coord_div, g, tid, w_big, th_big);
  if (blockIdx.x == 0 && tid == 0 && !warm) {          // publish the step's status word
    hdr[LINK_HDR_STATUS] = hdr[LINK_HDR_STATUS_ACC];
    hdr[LINK_HDR_STATUS_ACC] = 0;
  }
  w_big = DC_K1_SPLIT ? (__syncthreads_or(w_big) != 0 || (LINK_COSX_EXACT && OP == LINK_OP_COSX)) : (__syncthreads(), false);   // cos_x: exact contraction (elk_common.h)
  const bool th_slow = DC_THETA_BOUND ? __syncthreads_or(th_big) != 0 : false;     // workgroup-uniform (the same in every workgroup)
  if (dbg) tq1 = DC_NOW();
  if (c_begin >= c_end) return;
  dc_k1_range<C, OP, NB, PIPE>(smem_raw, feats, slots, cnt, cell_n, w_pre, coord_div, eps, n, g, warm, S_, fin, dbg, c_begin,
                                       c_end, pc_f, nv_f, rf0, rf1, rf2, rf3, w_big, th_slow, wid, tq0, tq1);
}
