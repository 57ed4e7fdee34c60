// link_amd/csrc/dense_fused_mm_impl.h -- "matrix-core sums" form of the fused pre_mix + LayerNorm + modulate + per-cell sum
// kernel of the dense-cell layout (round 4; link_dc_tuning_t::k1_form = 2; C = 32 / 64).  Included inside DC_IO_NS by
// dense_fused_impl.h.  Replaces voxelize_cuda.cu:12-25 (fp atomics per (voxel, channel)) + the torch Linear / LayerNorm /
// sin / cos / mul / cat graph in front of it (linkunet.py:132-162) like the cell-range form does; what changed and why:
//
// Counters on the cell-range form (profiles/r03_v9_pmc_counters.txt): 4.26 M VALU + 1.0 M SALU + 0.62 M LDS wave-instructions
// per launch, waves parked 53 % of their life at 2 waves per SIMD (198 registers, 79 KB of LDS per 4-wave workgroup), more LDS
// bank-conflict cycles than LDS instruction cycles; ~1 100 static VALU per 16-voxel tile of which ~400 are arithmetic -- the
// rest moves data between the MFMA accumulator layout (a lane = 16 channels of ONE voxel), the X tile in LDS and the
// row-sum shape (a lane = 4 channels of 16 voxels in turn).  This form never leaves the matrix cores' layouts:
//
//   * the contraction is TRANSPOSED: D[voxel][co] = F[voxel][:] . W[co][:] (A = feature rows, B = W), so a lane holds 4 voxels
//     (accumulator rows 4g + i) x one channel per 16-channel block (column l & 15).  B comes from an LDS image laid out in
//     planes [k-block][hi | lo][lane group] of 64 x 16 bytes: every operand is ONE conflict-free ds_read_b128;
//   * the per-cell sums are a SECOND matrix product on the (otherwise idle, exact) fp32 matrix instruction:
//     S^T[co][cell] = sum_v X^T[co][v] . M[v][cell] with M the 0/1 membership matrix of the tile's voxels in its cells.
//     The A operand of v_mfma_f32_16x16x4_f32 (row l & 15, k = l >> 4) is exactly what the first product left in a lane --
//     X of voxel 4g + j, channel l & 15 is register j of lane l -- so the modulate multiply feeds the matrix core directly:
//     no X tile in LDS (8.4 KB per wave), no transpose, no row-sum loop; the fp32 instruction is bit-for-bit an fmaf chain
//     (MI355X_MICROARCH.md), so a sum is a fixed-order chain of exact adds: bitwise reproducible.  The result lands as
//     a lane = 4 consecutive channels of one cell: the S row leaves in 16-byte stores;
//   * tiles are cut at cell boundaries (a lane per cell holds the counts; one ballot per tile), so a cell's sum is complete
//     when its tile ends; a cell with more than 16 voxels spans tiles of its own and carries its partial sums through 768 B
//     of LDS in a fixed order;
//   * 8 waves share one W image; 6.3 KB of LDS per wave (voxel list, cell of every list entry, cell addresses, carry row),
//     <= 128 registers: 4 waves per SIMD, and two frames' kernels fit a CU next to each other.
//
// Numerics: contraction as in the cell-range form (fp16 hi/lo split of both operands, three products, fp32 accumulation; fp32
// instruction outside the fp16 range); LayerNorm two-pass in fp32 (rstd by v_rsq_f32); theta / sincos as everywhere; per-cell
// sums in fp32, voxel order 0,4,8,12,1,5,... of the tile's id-sorted list.  A non-finite feature row makes its cell's S row NaN
// (as the reference's sums do) without touching the other cells of the tile (0 x NaN is kept out of the product, see `bad`).
#pragma once

#ifndef DC_K1M_NW
#define DC_K1M_NW 8        /* waves per workgroup (one W image) */
#endif
#ifndef DC_K1M_WAVES
#define DC_K1M_WAVES 4     /* register budget: 512 / this per lane */
#endif
#ifndef DC_K1M_PREFETCH
#define DC_K1M_PREFETCH 1  /* the next tile's feature rows are requested right after this tile's contraction (0: at the end of the tile) */
#endif
#ifndef DC_K1M_SUM32
#define DC_K1M_SUM32 0     /* second product on v_mfma_f32_16x16x4_f32 (exact fp32 chain; 32 instructions of 32 cycles per tile: the
                              matrix pipe's time, 4 x what the adds would take on the VALU, ends up on the wave's critical path:
                              2.8 k ticks per tile measured) instead of the fp16 hi/lo split on v_mfma_f32_16x16x16_f16 */
#endif
#ifndef DC_K1M_THETA_BOUND
#define DC_K1M_THETA_BOUND 0   /* the launch-wide |theta| bound of dense_common.h instead of the per-tile vote: in THIS kernel it costs more
                                  than it saves -- the flag lives across the whole tile loop of a kernel that sits exactly at its 128-register
                                  budget, and hipcc spilled 16 more registers (scratch 16 -> 80 B; 23.6 -> 27.4 us, round 5 A/B) */
#endif
#ifndef DC_K1M_ROWSUM_ASM
#define DC_K1M_ROWSUM_ASM 1    /* LayerNorm statistics: the four voxels' sums over a DPP row as 16 v_add_f32_dpp (four interleaved chains, so
                                  the DPP read-after-write hazard is covered by the other chains' instructions) -- hipcc emits v_mov_b32_dpp +
                                  v_pk_add_f32 pairs for the builtin form: 32 instructions more per tile */
#endif
#ifndef DC_K1M_LCAP
#define DC_K1M_LCAP 352    /* voxel records of one chunk of cells kept in LDS (>= 7^3: a whole cell) */
#endif

// the T 16-channel pieces of a lane's feature row: one address register, compile-time element offsets
template <int TT, int T>
__device__ __forceinline__ void dc_ld_row_pieces(__amdgpu_buffer_rsrc_t r, uint32_t ro, float4 (&ff)[T]) {
  if constexpr (TT < T) {
    ff[TT] = io_ldb4<16 * TT>(r, ro);
    dc_ld_row_pieces<TT + 1, T>(r, ro, ff);
  }
}

// sums of four values over the 16 lanes of their DPP rows (every lane gets its row's total): xor 1, xor 2, half mirror, mirror
__device__ __forceinline__ void dc_row16_sum4(float &a, float &b, float &c, float &d) {
#if DC_K1M_ROWSUM_ASM
#define DC_RS4_(CTRL)                                                              \
  "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"              \
  "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"              \
  "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"              \
  "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\t"
  // s_nop 1: a VALU write of an operand right in front of the block -> DPP read (hipcc does not look inside asm); inside the
  // block a value is read again three instructions after it was written
  asm("s_nop 1\n\t" DC_RS4_("quad_perm:[1,0,3,2]") DC_RS4_("quad_perm:[2,3,0,1]") DC_RS4_("row_half_mirror") DC_RS4_("row_mirror")
      "s_nop 1"
      : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef DC_RS4_
#else
  a = grp_sum<16>(a); b = grp_sum<16>(b); c = grp_sum<16>(c); d = grp_sum<16>(d);
#endif
}

template <int C, int OP>
struct dc_k1m_cfg {
  static_assert(C % 32 == 0, "K blocks of 32 input channels");
  static constexpr int T = C / 16;                     // 16-channel output blocks
  static constexpr int KB = C / 32;                    // K blocks of the f16 matrix instruction
  static constexpr int P = op_parts<OP>::value;
  static constexpr int RB = P * C * 4;                 // bytes of one S row
  static constexpr int LCAP = DC_K1M_LCAP;
  static constexpr int PLANE = C * 16;                 // one W plane: C rows (co) x 8 halves
  static constexpr int WIMG_BYTES = KB * 2 * 4 * PLANE;   // [kb][hi | lo][g]
  // per-channel parameter tables, transposed so that a lane (channel l16 of every block) takes ONE 16-byte read per table:
  // LayerNorm weight [l16][cb] | bias [l16][cb] (16 x T floats each) and theta weights [l16][tb][w0 w1 w2 alpha]
  static constexpr int LNW_OFF = WIMG_BYTES, LNB_OFF = LNW_OFF + 16 * T * 4, PW_OFF = LNB_OFF + 16 * T * 4;
  static constexpr int W_BYTES = PW_OFF + 16 * T * 16;
  static constexpr int LIST_OFF = 0;                   // int4 (x, y, z, id) per list entry
  static constexpr int ORD_OFF = LCAP * 16;            // u8: ordinal (among the chunk's occupied cells) of every entry's cell
  static constexpr int PCS_OFF = ORD_OFF + ((LCAP + 15) / 16) * 16;   // i32[64 + 16]: padded cell id by ordinal
  static constexpr int CARRY_OFF = PCS_OFF + 80 * 4;   // P * C floats: partial sums of a cell that spans tiles
  static constexpr int WAVE_BYTES = CARRY_OFF + P * C * 4;
  static constexpr int NW = DC_K1M_NW;
  static constexpr int LDS_BYTES = W_BYTES + NW * WAVE_BYTES;
};

// The kernel body is a device function of the workgroup number `bid` (the three-stage step kernel of round 4 ran it over a range
// of a shared grid; removed in round 6 -- the batch entry point of dense_batch.hip supersedes it).
template <int C, int OP, int NB>
__device__ __forceinline__ void dc_k1m_body(
    const void *__restrict__ feats, int4 *__restrict__ slots, uint32_t *__restrict__ cnt, int32_t *__restrict__ cell_n,
    const float *__restrict__ w_pre, const float *__restrict__ ln_w, const float *__restrict__ ln_b,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, int cg, float coord_div, float eps, int64_t n,
    const link_dc_grid_t &g, int cpw, bool warm, float *__restrict__ S_, float *__restrict__ fin, int32_t *__restrict__ hdr,
    unsigned long long *__restrict__ dbg, const int bid) {
  using K = dc_k1m_cfg<C, OP>;
  constexpr int T = K::T, KB = K::KB, P = K::P, RB = K::RB;
  DC_PROF_PTR(dbg);
  unsigned long long tq0 = dbg ? __builtin_amdgcn_s_memtime() : 0, tq1 = 0, tq_cell = 0, tq_mm = 0, tq_ln = 0, tq_sum = 0;
  int tq_tiles = 0;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, gq = lane >> 4;
  char *wbase = smem_raw + K::W_BYTES + wave * K::WAVE_BYTES;
  int4 *list = reinterpret_cast<int4 *>(wbase + K::LIST_OFF);
  unsigned char *ordof = reinterpret_cast<unsigned char *>(wbase + K::ORD_OFF);
  int *pcs = reinterpret_cast<int *>(wbase + K::PCS_OFF);
  float *carry = reinterpret_cast<float *>(wbase + K::CARRY_OFF);
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int Vi = Dx * Dy * Dz * g.dim[3];
  const int wid = bid * K::NW + wave;
  const int c_begin = wid * cpw;
  const int c_end = (c_begin + cpw < Vi) ? c_begin + cpw : Vi;
  const uint32_t *__restrict__ csrc = warm ? reinterpret_cast<const uint32_t *>(cell_n) : cnt;
  auto cell_of = [&](int chunk, int nrem) {
    const int q = chunk + (lane < nrem ? lane : 0);
    const int z = q % Dz;
    int t = q / Dz;
    const int y = t % Dy;
    t /= Dy;
    return dc_cell(g, t % Dx, y, z, t / Dx);
  };
  // the first chunk's cell records and counts are requested BEFORE W is staged (the two latencies overlap); every later
  // chunk's at the bottom of the chunk loop -- so these registers are live only from a request to the cell section that uses it
  int pc = 0, nv = 0;
  int4 r0 = make_int4(0, 0, 0, 0), r1 = r0, r2 = r0, r3 = r0;
  auto request_chunk = [&](int chunk) {
    pc = cell_of(chunk, (c_end - chunk < 64) ? c_end - chunk : 64);
    r0 = slots[(int64_t)pc * DC_INL + 0]; r1 = slots[(int64_t)pc * DC_INL + 1];
    r2 = slots[(int64_t)pc * DC_INL + 2]; r3 = slots[(int64_t)pc * DC_INL + 3];
    nv = (int)csrc[pc];
  };
  if (c_begin < c_end) request_chunk(c_begin);
  bool w_big = false;                                  // a weight outside the fp16 split's range: fp32 contraction (never on sane models)
  bool th_big = false;
  {
    // W image: the float4 W[co][4p .. 4p+3] is piece g = p % 4 of 16-channel block tt = p / 4; its hi / lo halves go to
    // plane [kb = tt / 2][hi | lo][g] at row co, bytes 8 * (tt % 2): a lane's B operand of v_mfma_f32_16x16x32_f16 for
    // (kb, column co) -- k = 8g + j -> channel 4g + j of block 2 kb (j < 4) / 2 kb + 1 (j >= 4), the same map the A operand uses
    constexpr int NF4 = C * C / 4, NT = 64 * K::NW, NV = (NF4 + NT - 1) / NT;
    float4 wv[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int e = (i * NT + tid) * 4;
      wv[i] = *reinterpret_cast<const float4 *>(&w_pre[(NF4 % NT == 0 || e < C * C) ? e : 0]);
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int f = i * NT + tid;
      if (NF4 % NT != 0 && f >= NF4) f = 0;            // surplus threads rewrite piece 0 with its own value
      const float4 wq = (NF4 % NT == 0 || i * NT + tid < NF4) ? wv[i] : *reinterpret_cast<const float4 *>(&w_pre[0]);
      const int co = f / (C / 4), p = f % (C / 4);
      const int tt = p >> 2, pg = p & 3, kb = tt >> 1, h = tt & 1;
      uint2 hi, lo;
      dc_split4(wq, hi, lo);
      *reinterpret_cast<uint2 *>(smem_raw + ((kb * 2 + 0) * 4 + pg) * K::PLANE + co * 16 + h * 8) = hi;
      *reinterpret_cast<uint2 *>(smem_raw + ((kb * 2 + 1) * 4 + pg) * K::PLANE + co * 16 + h * 8) = lo;
      w_big |= !(fmaxf(fmaxf(fabsf(wq.x), fabsf(wq.y)), fmaxf(fabsf(wq.z), fabsf(wq.w))) < 32768.0f);
    }
  }
  // parameter tables (read per tile: 16 fewer live registers than per-lane copies held for the whole kernel)
  if (tid < C) {
    const int cb = tid >> 4, li = tid & 15;
    reinterpret_cast<float *>(smem_raw + K::LNW_OFF)[li * T + cb] = ln_w[tid];
    reinterpret_cast<float *>(smem_raw + K::LNB_OFF)[li * T + cb] = ln_b[tid];
    const int tc = tid % cg;                           // channel ch uses theta[ch % cg]; block tb < NB holds channels 16 tb + li
    const float4 pwv = make_float4(w_pos[3 * tc + 0], w_pos[3 * tc + 1], w_pos[3 * tc + 2], alpha ? alpha[tc] : 1.0f);
    reinterpret_cast<float4 *>(smem_raw + K::PW_OFF)[li * T + cb] = pwv;
    if (DC_K1M_THETA_BOUND) {                          // can any theta of this launch leave the fast sincos range? (dense_common.h)
      float ax, ay, az;
      dc_coord_absmax(g, coord_div, ax, ay, az);
      th_big = dc_theta_leaves_fast_range(ax, ay, az, pwv.x, pwv.y, pwv.z, pwv.w);
    }
  }
  if (bid == 0 && tid == 0 && !warm) {                 // publish the step's status word
    hdr[LINK_HDR_STATUS] = hdr[LINK_HDR_STATUS_ACC];
    hdr[LINK_HDR_STATUS_ACC] = 0;
  }
  w_big = __syncthreads_or(w_big) != 0 || (LINK_COSX_EXACT && OP == LINK_OP_COSX);   // cos_x: exact contraction (elk_common.h)
  const bool th_slow = DC_K1M_THETA_BOUND ? __syncthreads_or(th_big) != 0 : false;  // workgroup-uniform (the same in every workgroup)
  if (dbg) tq1 = __builtin_amdgcn_s_memtime();
  if (c_begin >= c_end) return;
  const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, (uint32_t)((g.vp + 1) * RB));
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * C * 4));     // written for cos_x only
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_n = dc_rsrc(cell_n, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_feats = dc_rsrc(feats, (uint32_t)(n * C * IO_BYTES));
  const __amdgpu_buffer_rsrc_t r_w = dc_rsrc(w_pre, (uint32_t)(C * C * 4));
  const bool ract = lane < RB / 16;                    // lanes holding a 16-byte piece of an S row (zero rows of empty cells)
  const float inv_c = 1.0f / (float)C;

  for (int chunk = c_begin; chunk < c_end;) {
    const int nrem = (c_end - chunk < 64) ? c_end - chunk : 64;
    unsigned long long tqa = dbg ? __builtin_amdgcn_s_memtime() : 0;
    // ---- cell lanes: count, padded cell id, inline records (requested by request_chunk) ----
    nv = nv < g.k ? nv : g.k;
    nv = nv < K::LCAP ? nv : K::LCAP;
    if (lane >= nrem) nv = 0;
    // inclusive prefix over the wave: DPP row scan (zeros shifted in) + the three row totals as scalars -- no per-lane
    // permute addresses to keep (the __shfl_up form's six address registers were spilled, and a scratch reload waits for
    // every store in flight)
    int incl = nv;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
    {
      const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47);
      incl += gq == 0 ? 0 : (gq == 1 ? t0 : (gq == 2 ? t0 + t1 : t0 + t1 + t2));
    }
    const unsigned long long fit = __ballot(lane < nrem && incl <= K::LCAP);
    const int nfit = __builtin_amdgcn_readfirstlane(__popcll(fit));      // >= 1: a cell never exceeds LCAP
    const int Ttot = __builtin_amdgcn_readlane(incl, nfit - 1);
    const bool mine = lane < nfit;
    const int excl = incl - nv;
    // ordinal of this cell among the chunk's occupied cells: the column (mod 16) its sums take in the tiles' second product
    const unsigned long long occ = __ballot(mine && nv > 0);
    const int ord = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(occ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)occ, 0u));
    if (mine) {
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
        if (j < nv) { list[excl + j] = r; ordof[excl + j] = (unsigned char)ord; }
        // the id-ordered records go back to the slot list: the fused gather + de-modulate kernel walks it and must see the
        // same order in every run (rank order from the atomics is not reproducible)
        st16i(r_slots, (j < nv && nv <= DC_INL && !warm) ? ((uint32_t)pc * DC_INL + j) * 16u : DC_OOB, r);
      }
      for (int k = DC_INL; k < nv; k++) {               // overflow records: insertion by id (rare)
        const v4i_t rv = __builtin_amdgcn_raw_buffer_load_b128(r_slots, dc_slot(g, pc, k) * 16u, 0, 0);
        const int rx = rv.x, ry = rv.y, rz = rv.z, rw = rv.w;      // (scalars: an int4 held across the loop went to the stack)
        ordof[excl + k] = (unsigned char)ord;
        int pos = k;
        while (pos > 0 && list[excl + pos - 1].w > rw) {
          list[excl + pos] = list[excl + pos - 1];
          pos--;
        }
        list[excl + pos] = make_int4(rx, ry, rz, rw);
      }
      if (nv > DC_INL && !warm)
        for (int k = 0; k < nv; k++) slots[dc_slot(g, pc, k)] = list[excl + k];
      if (nv > 0) pcs[ord] = pc;
    }
    {                                                   // publish the counts, reset the counters
      const uint32_t coff = (mine && !warm) ? (uint32_t)pc * 4u : DC_OOB;
      st4i(r_n, coff, nv);
      st4i(r_cnt, coff, 0);
    }
    for (unsigned long long em = __ballot(mine && nv == 0); em; em &= em - 1) {   // empty cells: zero rows
      const int pcj = __builtin_amdgcn_readlane(pc, __builtin_ctzll(em));
      st16(r_S, ract ? (uint32_t)pcj * (uint32_t)RB + (uint32_t)lane * 16u : DC_OOB, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __builtin_amdgcn_wave_barrier();
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_cell += tqb - tqa; tqa = tqb; }

    // ---- tiles: whole cells, at most 16 voxels; a cell with more than 16 voxels spans tiles of its own ----
    // descriptor of the tile that starts at list position `b`: first cell lane, number of voxels, number of cells, whether the
    // first cell continues an earlier tile / the last cell continues into the next one
    struct tile_t { int len, ordF, ncell; bool cont_in, cont_out; };
    auto describe = [&](int b) {
      tile_t t;
      const unsigned long long mF = __ballot(mine && nv > 0 && excl <= b && b < incl);     // exactly one lane
      const int cF = __builtin_ctzll(mF);
      const int inclF = __builtin_amdgcn_readlane(incl, cF), exclF = __builtin_amdgcn_readlane(excl, cF);
      t.ordF = __builtin_amdgcn_readlane(ord, cF);
      t.cont_in = exclF < b;
      if (inclF - b > 16) {
        t.len = 16; t.ncell = 1; t.cont_out = true;
      } else {
        const unsigned long long m = __ballot(mine && nv > 0 && lane >= cF && incl <= b + 16);
        t.ncell = __popcll(m);
        t.len = __builtin_amdgcn_readlane(incl, 63 - __builtin_clzll(m)) - b;
        t.cont_out = false;
      }
      return t;
    };
    auto ld_rows = [&](int b, int len, float4 (&ff)[T]) {
      const int id = list[b + (l16 < len ? l16 : len - 1)].w;
      const uint32_t ro = (uint32_t)id * (uint32_t)(C * IO_BYTES) + (uint32_t)(4 * gq * IO_BYTES);
      dc_ld_row_pieces<0, T>(r_feats, ro, ff);
    };
    float4 ff[T];
    tile_t cur = {0, 0, 0, false, false};
    if (Ttot > 0) { cur = describe(0); ld_rows(0, cur.len, ff); }
    for (int base = 0; base < Ttot;) {
      const tile_t t = cur;
      const int nbase = base + t.len;
      unsigned long long tqt = dbg ? __builtin_amdgcn_s_memtime() : 0;
      // ---- the tile's records: the 4 voxels this lane holds accumulator rows of (4 gq + j); theta and the membership
      // column at once, so that only they stay live through the contraction ----
      float th[NB][4], mj[4];
      int idj[OP == LINK_OP_COSX ? 4 : 1];
      bool big;
      {
        float4 pw[NB];
#pragma unroll
        for (int tb = 0; tb < NB; tb++) pw[tb] = reinterpret_cast<const float4 *>(smem_raw + K::PW_OFF)[l16 * T + tb];
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int vv = 4 * gq + j;
          const int sl = base + (vv < t.len ? vv : t.len - 1);
          const int4 rec = list[sl];
          mj[j] = (vv < t.len && ((int)ordof[sl] & 15) == l16) ? 1.0f : 0.0f;   // B operand of the second product
          if (OP == LINK_OP_COSX) idj[j] = rec.w;
          float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
          if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
          for (int tb = 0; tb < NB; tb++) {
            th[tb][j] = theta_of(x, y, z, pw[tb].x, pw[tb].y, pw[tb].z, pw[tb].w);
            if (!DC_K1M_THETA_BOUND) mx = fmaxf(mx, fabsf(th[tb][j]));
          }
        }
        big = !(mx < 32768.0f);                         // (an infinite theta survives v_max; a NaN gives NaN on either path)
      }
      const uint2 mjh = make_uint2((mj[0] != 0.f ? 0x3C00u : 0u) | (mj[1] != 0.f ? 0x3C000000u : 0u),
                                   (mj[2] != 0.f ? 0x3C00u : 0u) | (mj[3] != 0.f ? 0x3C000000u : 0u));   // 1.0h / 0
      (void)mjh;
      // column role: the cell whose sums this lane's column (l16) receives, its S row, whether the row is complete here
      const int ord_c = t.ordF + ((l16 - t.ordF) & 15);
      const int pc_c = pcs[ord_c];
      const bool col_store = ord_c < t.ordF + t.ncell && !(t.cont_out && ord_c == t.ordF + t.ncell - 1);
      const uint32_t s_off = col_store ? (uint32_t)pc_c * (uint32_t)RB + (uint32_t)(gq * 16) : DC_OOB;
      // ---- pre_mix contraction D[voxel][co]: fp16 hi/lo split of both operands, three products, fp32 accumulation ----
      floatx4 acc[T];
#pragma unroll
      for (int cb = 0; cb < T; cb++) acc[cb] = (floatx4){0.f, 0.f, 0.f, 0.f};
      {
        uint2 ah[T], al[T];
        float mx = 0.f;
#pragma unroll
        for (int tt = 0; tt < T; tt++) {
          dc_split4(ff[tt], ah[tt], al[tt]);
          mx = fmaxf(mx, fmaxf(fmaxf(fabsf(ff[tt].x), fabsf(ff[tt].y)), fmaxf(fabsf(ff[tt].z), fabsf(ff[tt].w))));
        }
        if (__builtin_expect(!(w_big || __any(!(mx < 32768.0f))), 1)) {
          // B operands two blocks at a time, the next pair requested before this pair is multiplied (two register sets)
          constexpr int NG = KB * T / 2;                // groups of two 16-channel blocks
          auto ld_b = [&](int gi, uint4 (&bh)[2], uint4 (&bl)[2]) {
            const int kb = gi / (T / 2), cb0 = 2 * (gi % (T / 2));
#pragma unroll
            for (int u = 0; u < 2; u++) {
              bh[u] = *reinterpret_cast<const uint4 *>(smem_raw + ((kb * 2 + 0) * 4 + gq) * K::PLANE + (16 * (cb0 + u) + l16) * 16);
              bl[u] = *reinterpret_cast<const uint4 *>(smem_raw + ((kb * 2 + 1) * 4 + gq) * K::PLANE + (16 * (cb0 + u) + l16) * 16);
            }
          };
          auto mm_b = [&](int gi, const uint4 (&bh)[2], const uint4 (&bl)[2]) {
            const int kb = gi / (T / 2), cb0 = 2 * (gi % (T / 2));
#pragma unroll
            for (int u = 0; u < 2; u++) {
              const int cb = cb0 + u;
              const uint2 bh0 = make_uint2(bh[u].x, bh[u].y), bh1 = make_uint2(bh[u].z, bh[u].w);
              const uint2 bl0 = make_uint2(bl[u].x, bl[u].y), bl1 = make_uint2(bl[u].z, bl[u].w);
              if constexpr (IO != 1) acc[cb] = dc_mfma_f16x2(al[2 * kb], al[2 * kb + 1], bh0, bh1, acc[cb]);   // fp16 rows: lo = 0 exactly
              acc[cb] = dc_mfma_f16x2(ah[2 * kb], ah[2 * kb + 1], bl0, bl1, acc[cb]);
              acc[cb] = dc_mfma_f16x2(ah[2 * kb], ah[2 * kb + 1], bh0, bh1, acc[cb]);
            }
          };
          uint4 bhA[2], blA[2], bhB[2], blB[2];
          ld_b(0, bhA, blA);
#pragma unroll
          for (int gi = 0; gi < NG; gi += 2) {
            if (gi + 1 < NG) ld_b(gi + 1, bhB, blB);
            mm_b(gi, bhA, blA);
            __builtin_amdgcn_sched_barrier(0);
            if (gi + 1 < NG) {
              if (gi + 2 < NG) ld_b(gi + 2, bhA, blA);
              mm_b(gi + 1, bhB, blB);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        } else {
          // values outside the fp16 range: the fp32 instruction, B from global memory (slow, exact, wave-uniform, rare):
          // A[voxel l16][k = gq] = F[voxel][16 tt + 4 gq + e], B[k = gq][co] = W[co][16 tt + 4 gq + e]
#pragma unroll
          for (int tt = 0; tt < T; tt++)
#pragma unroll
            for (int cb = 0; cb < T; cb++) {
              const v4i_t wi = __builtin_amdgcn_raw_buffer_load_b128(r_w, (uint32_t)(((16 * cb + l16) * C + 16 * tt + 4 * gq) * 4), 0, 0);
              const float4 wq = make_float4(__int_as_float(wi.x), __int_as_float(wi.y), __int_as_float(wi.z), __int_as_float(wi.w));
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ff[tt].x, wq.x, acc[cb], 0, 0, 0);
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ff[tt].y, wq.y, acc[cb], 0, 0, 0);
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ff[tt].z, wq.z, acc[cb], 0, 0, 0);
              acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ff[tt].w, wq.w, acc[cb], 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
      // ---- the next tile's rows are requested now: they fly while this tile goes through the VALU and the second product ----
#if DC_K1M_PREFETCH
      if (nbase < Ttot) { cur = describe(nbase); ld_rows(nbase, cur.len, ff); }
      __builtin_amdgcn_sched_barrier(0);
#endif
      if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_mm += tqb - tqt; tqt = tqb; }
      const bool slow = DC_K1M_THETA_BOUND ? th_slow : __any(big);
      float sn[NB][4], cs[NB][4];
      if (__builtin_expect(slow, 0)) {
#pragma unroll
        for (int tb = 0; tb < NB; tb++)
#pragma unroll
          for (int j = 0; j < 4; j++) sincos_nocall(th[tb][j], sn[tb][j], cs[tb][j]);
      } else {
#pragma unroll
        for (int tb = 0; tb < NB; tb++)
#pragma unroll
          for (int j = 0; j < 4; j++) sincos_small(th[tb][j], sn[tb][j], cs[tb][j]);
      }
      // ---- LayerNorm over each voxel's C channels: T in-lane values + the 16 lanes of the accumulator row ----
      unsigned badm = 0;                                // bit j: voxel 4 gq + j is not finite (its cell's sums become NaN below)
      {
        float lnw[T], lnb[T];
        if constexpr (T == 4) {
          const float4 lw4 = reinterpret_cast<const float4 *>(smem_raw + K::LNW_OFF)[l16];
          const float4 lb4 = reinterpret_cast<const float4 *>(smem_raw + K::LNB_OFF)[l16];
          lnw[0] = lw4.x; lnw[1] = lw4.y; lnw[2] = lw4.z; lnw[3] = lw4.w;
          lnb[0] = lb4.x; lnb[1] = lb4.y; lnb[2] = lb4.z; lnb[3] = lb4.w;
        } else {
#pragma unroll
          for (int cb = 0; cb < T; cb++) {
            lnw[cb] = reinterpret_cast<const float *>(smem_raw + K::LNW_OFF)[l16 * T + cb];
            lnb[cb] = reinterpret_cast<const float *>(smem_raw + K::LNB_OFF)[l16 * T + cb];
          }
        }
        float s4[4], q4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          s4[j] = 0.f;
#pragma unroll
          for (int cb = 0; cb < T; cb++) s4[j] += acc[cb][j];
        }
        dc_row16_sum4(s4[0], s4[1], s4[2], s4[3]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float mean = s4[j] * inv_c;
          q4[j] = 0.f;
#pragma unroll
          for (int cb = 0; cb < T; cb++) {
            const float d = acc[cb][j] - mean;
            acc[cb][j] = d;
            q4[j] = fmaf(d, d, q4[j]);
          }
        }
        dc_row16_sum4(q4[0], q4[1], q4[2], q4[3]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float qq = q4[j];
          const float rstd = __builtin_amdgcn_rsqf(fmaf(qq, inv_c, eps));
          bool bj = !(qq < __builtin_inff());
          if (slow) {
#pragma unroll
            for (int tb = 0; tb < NB; tb++) bj |= !(fabsf(th[tb][j]) < __builtin_inff());
          }
          badm |= bj ? (1u << j) : 0u;
#pragma unroll
          for (int cb = 0; cb < T; cb++) acc[cb][j] = fmaf(acc[cb][j] * rstd, lnw[cb], lnb[cb]);
        }
      }
      if (OP == LINK_OP_COSX) {                         // the de-modulation of cos_x needs fin (linkunet.py:176)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int cb = 0; cb < T; cb++)
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(acc[cb][j]), r_fin,
                                                  (4 * gq + j) < t.len ? (uint32_t)idj[j] * (uint32_t)(C * 4) + (uint32_t)((16 * cb + l16) * 4) : DC_OOB,
                                                  0, DC_ST_AUX);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_ln += tqb - tqt; tqt = tqb; }
      // ---- modulate -> second product -> S rows, part by part.  SAN: the tile holds a non-finite voxel (never on finite
      // inputs): its X is kept out of the product (0 x NaN would reach the other cells' sums) and its cell's rows become NaN ----
      auto parts = [&](auto san_tag) {
        constexpr bool SAN = decltype(san_tag)::value;
        unsigned nanmask = 0;                           // columns (cells) that hold a non-finite voxel
        if constexpr (SAN) {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int vv = 4 * gq + j;
            const int sl = base + (vv < t.len ? vv : t.len - 1);
            nanmask |= (((badm >> j) & 1u) && vv < t.len) ? (1u << ((int)ordof[sl] & 15)) : 0u;
          }
          // (row-uniform: the LayerNorm statistics are) -> the four lane groups' masks as scalars
          nanmask = (unsigned)(__builtin_amdgcn_readlane((int)nanmask, 0) | __builtin_amdgcn_readlane((int)nanmask, 16) |
                               __builtin_amdgcn_readlane((int)nanmask, 32) | __builtin_amdgcn_readlane((int)nanmask, 48));
        }
#pragma unroll
        for (int p = 0; p < P; p++) {
          floatx4 aS[T];
          if (__builtin_expect(t.cont_in, 0)) {         // the first cell continues: its partial sums come from the carry row
            const bool keep = l16 == (t.ordF & 15);
#pragma unroll
            for (int cb = 0; cb < T; cb++) {
              const float4 cv = *reinterpret_cast<const float4 *>(&carry[p * C + 16 * cb + 4 * gq]);
              aS[cb] = keep ? (floatx4){cv.x, cv.y, cv.z, cv.w} : (floatx4){0.f, 0.f, 0.f, 0.f};
            }
          } else {
#pragma unroll
            for (int cb = 0; cb < T; cb++) aS[cb] = (floatx4){0.f, 0.f, 0.f, 0.f};
          }
#pragma unroll
          for (int cb = 0; cb < T; cb++) {
            const int tb = cb % NB;
            float xv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float m = p == 2 ? th[tb][j] : ((p == 0) == (OP != LINK_OP_SIN) ? cs[tb][j] : sn[tb][j]);   // cos|sin (sin: sin|cos), theta
              // The product must exist as an fp32 VALUE before it is split (the empty asm pins it): hipcc otherwise folds this
              // multiply into the split's subtraction (lo = fp16(fma(a, m, -hi)) with hi = fp16(a m) in ONE rounding) while the hi
              // it packs for the matrix core is fp16(fp32(a m)): the two hi differ by an fp16 ulp once in ~10^4 values (double
              // rounding) and the lo no longer matches -- tools/k1mdbg.py found it; __fmul_rn does not stop the fold
              xv[j] = acc[cb][j] * m;
              asm volatile("" : "+v"(xv[j]));
              if constexpr (SAN) xv[j] = ((badm >> j) & 1u) ? 0.f : xv[j];
            }
#if DC_K1M_SUM32
#pragma unroll
            for (int j = 0; j < 4; j++) aS[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[j], mj[j], aS[cb], 0, 0, 0);
#else
            // X = hi + lo (fp16 pairs, 22 bits).  ONE v_mfma_f32_16x16x32_f16 per block: the lane's k = 8 gq .. 8 gq + 7 are
            // [hi of its four voxels | lo of the same four] against [their membership flags | the same flags] as halves (any
            // assignment of k is right as long as both operands use it); products with 0 / 1 are exact, fp32 accumulation.
            // (As two dependent v_mfma_f32_16x16x16_f16 the lo term went missing in ~1 of 10^4 sums: tools/k1mdbg.py.)
            uint2 xh, xl;
            dc_split4(make_float4(xv[0], xv[1], xv[2], xv[3]), xh, xl);
            aS[cb] = dc_mfma_f16x2(xh, xl, mjh, mjh, aS[cb]);
#endif
          }
          if constexpr (SAN) {
            if ((nanmask >> l16) & 1u) {
#pragma unroll
              for (int cb = 0; cb < T; cb++) aS[cb] = (floatx4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
            }
          }
          // lane (column l16, group gq) holds S[cell][p*C + 16 cb + 4 gq .. +3]
#pragma unroll
          for (int cb = 0; cb < T; cb++)
            st16(r_S, s_off + (col_store ? (uint32_t)((p * C + 16 * cb) * 4) : 0u), make_float4(aS[cb][0], aS[cb][1], aS[cb][2], aS[cb][3]));
          __builtin_amdgcn_sched_barrier(0);
          if (__builtin_expect(t.cont_out, 0)) {        // the last cell continues: hand its partial sums to the next tile
            if (l16 == ((t.ordF + t.ncell - 1) & 15)) {
#pragma unroll
              for (int cb = 0; cb < T; cb++)
                *reinterpret_cast<float4 *>(&carry[p * C + 16 * cb + 4 * gq]) = make_float4(aS[cb][0], aS[cb][1], aS[cb][2], aS[cb][3]);
            }
          }
        }
      };
      if (__builtin_expect(__any(badm != 0), 0)) parts(std::true_type{}); else parts(std::false_type{});
      if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_sum += tqb - tqt; }
      if (__builtin_expect(t.cont_out, 0)) __builtin_amdgcn_wave_barrier();
#if !DC_K1M_PREFETCH
      if (nbase < Ttot) { cur = describe(nbase); ld_rows(nbase, cur.len, ff); }
#endif
      base = nbase;
      if (dbg) tq_tiles++;
    }
    __builtin_amdgcn_wave_barrier();
    chunk += nfit;
    if (chunk < c_end) request_chunk(chunk);
  }
  if (dbg && lane == 0) {
    unsigned long long *d = dbg + (size_t)wid * 8;
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    d[0] = tq1 - tq0; d[1] = tq_cell; d[2] = tq_mm; d[3] = tq_ln; d[4] = tq_sum; d[5] = te - tq0; d[6] = tq_tiles; d[7] = tq0;
  }
}

template <int C, int OP, int NB>
__global__ void __launch_bounds__(64 * DC_K1M_NW, DC_K1M_WAVES) k_dc_premix_modsum_mm(
    const void *__restrict__ feats, int4 *__restrict__ slots, uint32_t *__restrict__ cnt, int32_t *__restrict__ cell_n,
    const float *__restrict__ w_pre, const float *__restrict__ ln_w, const float *__restrict__ ln_b,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, int cg, float coord_div, float eps, int64_t n,
    link_dc_grid_t g, int cpw, bool warm, float *__restrict__ S_, float *__restrict__ fin, int32_t *__restrict__ hdr,
    unsigned long long *__restrict__ dbg) {
  dc_k1m_body<C, OP, NB>(feats, slots, cnt, cell_n, w_pre, ln_w, ln_b, w_pos, alpha, cg, coord_div, eps, n, g, cpw, warm,
                         S_, fin, hdr, dbg, (int)blockIdx.x);
}

template <int C, int OP, int NB>
static int launch_k1m(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n, bool warm,
                      hipStream_t st) {
  using K = dc_k1m_cfg<C, OP>;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  // tune.k1_wgs counts 256-thread workgroups of the cell-range form (4 waves): the same number of waves here
  int64_t waves = (int64_t)(b->tune.k1_wgs > 0 ? b->tune.k1_wgs : 512) * 4;
  int cpw = (int)((vi + waves - 1) / waves);
  if (cpw < 1) cpw = 1;
  const int64_t wgs = (vi + (int64_t)cpw * K::NW - 1) / ((int64_t)cpw * K::NW);
  int pad = b->tune.k1_lds_pad;
  pad = pad < 0 ? 0 : (pad > 16384 ? 16384 : pad);
  const int lds = K::LDS_BYTES + pad <= 160 * 1024 ? K::LDS_BYTES + pad : K::LDS_BYTES;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_premix_modsum_mm<C, OP, NB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k_dc_premix_modsum_mm<C, OP, NB>), dim3((unsigned)wgs), dim3(64 * K::NW), lds, st, b->feats,
                     reinterpret_cast<int4 *>(b->slots), b->cnt, b->cell_n, b->w_pre, b->pre_ln_w, b->pre_ln_b, b->w_pos,
                     b->alpha, d.cg, d.coord_div, d.eps, n, g, cpw, warm, b->S, b->fin, b->hdr,
                     reinterpret_cast<unsigned long long *>(b->tune.k1_dbg));
  return check_launch("link_dc_premix_modsum");
}

template <int C, int OP>
static int dispatch_k1m_nb(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n, bool warm,
                           hipStream_t st) {
  constexpr int T = C / 16;
  int nb = (d.cg % 16 == 0) ? d.cg / 16 : T;
  if (nb > T) nb = T;
  if (T >= 2 && nb == T / 2) return launch_k1m<C, OP, (T >= 2 ? T / 2 : 1)>(b, g, d, n, warm, st);
  if (T >= 4 && nb == T / 4) return launch_k1m<C, OP, (T >= 4 ? T / 4 : 1)>(b, g, d, n, warm, st);
  return launch_k1m<C, OP, T>(b, g, d, n, warm, st);   // any other grouping: every block evaluates its own theta
}

template <int C>
static int dispatch_k1m_op(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n, bool warm,
                           hipStream_t st) {
  switch (d.op) {
    case LINK_OP_COS: return dispatch_k1m_nb<C, LINK_OP_COS>(b, g, d, n, warm, st);
    case LINK_OP_SIN: return dispatch_k1m_nb<C, LINK_OP_SIN>(b, g, d, n, warm, st);
    default: return dispatch_k1m_nb<C, LINK_OP_COSX>(b, g, d, n, warm, st);
  }
}
