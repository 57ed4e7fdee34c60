// link_amd/csrc/dense_gather_quad_impl.h -- producer / consumer form of the fused box sum + de-modulate kernel with
// QUAD consumers (C = 64, two-part rows, theta shared by channels j and j + 32; round 3).  Included inside DC_IO_NS by
// dense_fused_impl.h after the K2 configuration structs and LDS helpers.
//
// tools/k2prof.py on the round-2 producer / consumer form (k_dc_gather_demod_split): a plane step lasts ~4600 ticks; the
// producer waves spend 1930 of them in three dependent LDS round trips of the box sum, 625 in a second round of pairs and
// 844 in the barrier, the consumer waves 2277 in the pair loop and 1447 in the barrier.  Both halves are chains of
// dependent latencies (LDS round trips, DPP reductions), not pipe time: VALU 40 % busy, LDS array 20 %.  This form
// shortens both chains:
//   * producers and consumers run SEPARATE loops (own live registers: the producers' box sum takes the plane's three
//     x-offsets as one pipelined LDS request, dense_gather.h -- in the common loop of the round-2 kernel that spilled);
//     the producers issue no stores and never take pairs, so their counted DMA waits are exact;
//   * a consumer voxel is handled by a QUAD of lanes (16 channels each) instead of a pair of voxels by 16 lanes (4
//     channels each): the four consumer waves take 64 voxels per round -- a plane holds ~32, so there is ONE round per
//     plane where the pair form needed 1.5 of 16 pairs plus the producers' help; LayerNorm sums are 15 in-lane adds + two
//     quad steps instead of four 16-lane reductions; the voxel -> (cell, slot) map is 15 scalar compares against the
//     plane's count prefix (v_readlane) instead of scan / two ballots / three ds_bpermute per pair.
// Lane q of a quad holds the 16-byte pieces q, q+4, q+8, q+12 of a row (channels 16 j + 4 q + e): pieces p and p + 8
// share theta, so a lane evaluates 8 sincos per voxel.
#pragma once
#ifndef DC_K2_MAP
#define DC_K2_MAP 1      /* tile enumeration by y-halves (0: ty fastest over the whole grid): FETCH_SIZE of the kernel 27.2 -> 23.9 MB
                          (x 2: 54.4 -> 47.9 MB from the memory side) on cfg2, same time */
#endif

#ifndef DC_K2Q_CW
#define DC_K2Q_CW 3      /* consumer waves: 16 voxel slots each.  A cfg2 plane holds ~32 voxels (Poisson: more than 48 in 0.3 % of the
                            planes), so three waves still finish a plane in ONE round and the fourth only issued instructions for empty
                            slots (round 4; 4 = the round-3 geometry) */
#endif
#ifndef DC_K2Q_PMAP
#define DC_K2Q_PMAP 1    /* the voxel -> (cell, slot) map of a plane is laid out by ONE extra wave (the mapper), two steps ahead, from the
                            plane's count image (a lane per voxel slot: 16 broadcast LDS reads, an in-lane prefix, 15 compares), and a
                            consumer lane reads its entry: the consumers' count prefix (LDS read, 4 DPP steps, 16 v_readlane, 45
                            scalar-operand VALU per wave and plane) leaves the step's critical path.  Not a producer's job: with the
                            map in their loop the producers spilled, and scratch traffic would break their counted vmcnt waits.
                            (0 = round-3 consumers) */
#endif
#ifndef DC_K2Q_EARLY
#define DC_K2Q_EARLY 1   /* a consumer wave whose 16 voxel slots all lie beyond the plane's voxel count leaves the round at once (wave-uniform):
                            an interior cfg2 plane holds ~32 voxels, so the third wave is needed by 43 % of the planes only, and the
                            tiles on the grid's rim (1 of 4 columns inside) keep one wave busy -- before, every consumer wave issued
                            the whole round for slots without a voxel (round 5: -19 % of the kernel's VALU instructions) */
#endif
#ifndef DC_K2Q_RIM
#define DC_K2Q_RIM 1
#endif
#ifndef DC_K2Q_FMA
#define DC_K2Q_FMA 1     /* de-modulation A0 cos + A1 sin as mul + fma (0: separate IEEE mul / mul / add like the reference's eager
                            ops -- the difference is one rounding, 6e-8 relative, next to the hardware trig's 4e-7) */
#endif

template <int OP, int R>
struct dc_k2q_cfg {
  using K2 = dc_k2_cfg<OP, R>;
  static constexpr int PAR_OFF = K2::SPLIT_LDS_BYTES;            // LayerNorm weight | bias image (2 x 64 floats)
  // count images run two planes ahead of the row images (the producers look a plane's counts up before they request its
  // rows): a ring of five 256-byte images (the tile's <= 64 haloed columns, requested by every producer wave alike) laid
  // into the three plane slots' 1 KB count areas -- no LDS beyond the round-2 layout, whose size the frames in flight
  // are tuned to
  static constexpr int MAP_OFF = PAR_OFF + 2 * 64 * 4;           // 3 voxel maps: 64 entries (cell << 12 | slot) + the plane's voxel count
  static constexpr int MAP_BYTES = 64 * 4 + 16;
  static constexpr int LDS_BYTES = MAP_OFF + (DC_K2Q_PMAP ? 3 * MAP_BYTES : 0);
  static constexpr int THREADS = 256 + 64 * (DC_K2Q_CW + (DC_K2Q_PMAP ? 1 : 0));     // producers + consumers (+ the mapper wave)
  static constexpr bool FITS = K2::P == 2 && 2 * LDS_BYTES <= 160 * 1024;
};

// record + the eight A-row pieces of a quad lane: nine b128 reads in flight, ONE wait
__device__ __forceinline__ void lds_rd9_b128(uint32_t ar, uint32_t a0, uint32_t a1, v4f_t &r, v4f_t (&x)[2][4]) {
  asm volatile("ds_read_b128 %0, %9\n\t"
               "ds_read_b128 %1, %10\n\tds_read_b128 %2, %10 offset:64\n\tds_read_b128 %3, %10 offset:128\n\tds_read_b128 %4, %10 offset:192\n\t"
               "ds_read_b128 %5, %11\n\tds_read_b128 %6, %11 offset:64\n\tds_read_b128 %7, %11 offset:128\n\tds_read_b128 %8, %11 offset:192\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(r), "=&v"(x[0][0]), "=&v"(x[0][1]), "=&v"(x[0][2]), "=&v"(x[0][3]), "=&v"(x[1][0]), "=&v"(x[1][1]),
                 "=&v"(x[1][2]), "=&v"(x[1][3])
               : "v"(ar), "v"(a0), "v"(a1)
               : "memory");
}
__device__ __forceinline__ void lds_rd8_b128(uint32_t a0, uint32_t a1, v4f_t (&x)[2][4]) {
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\tds_read_b128 %3, %8 offset:192\n\t"
               "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:64\n\tds_read_b128 %6, %9 offset:128\n\tds_read_b128 %7, %9 offset:192\n\t"
               "s_waitcnt lgkmcnt(0)"
               : "=&v"(x[0][0]), "=&v"(x[0][1]), "=&v"(x[0][2]), "=&v"(x[0][3]), "=&v"(x[1][0]), "=&v"(x[1][1]), "=&v"(x[1][2]),
                 "=&v"(x[1][3])
               : "v"(a0), "v"(a1)
               : "memory");
}
#ifndef DC_K2Q_OVERLAP
#define DC_K2Q_OVERLAP 1   /* consumers: theta / sincos run while the eight A-row pieces are still on their way (LDS returns in order:
                              the record is waited for alone), and the LayerNorm weights are requested before the statistics'
                              reductions instead of after them -- two LDS latencies off the round's dependent chain (0: round-3 order) */
#endif
// The same nine reads, but only the FIRST (the record) is waited for: LDS reads return in order, so lgkmcnt(8) means "the record
// is here"; lds_wait_pieces() later makes the eight A pieces valid (guide section 5.7 form (ii): the waiting statement names them "+v").
__device__ __forceinline__ void lds_rd9_b128_rec_first(uint32_t ar, uint32_t a0, uint32_t a1, v4f_t &r, v4f_t (&x)[2][4]) {
  asm volatile("ds_read_b128 %0, %9\n\t"
               "ds_read_b128 %1, %10\n\tds_read_b128 %2, %10 offset:64\n\tds_read_b128 %3, %10 offset:128\n\tds_read_b128 %4, %10 offset:192\n\t"
               "ds_read_b128 %5, %11\n\tds_read_b128 %6, %11 offset:64\n\tds_read_b128 %7, %11 offset:128\n\tds_read_b128 %8, %11 offset:192\n\t"
               "s_waitcnt lgkmcnt(8)"
               : "=&v"(r), "=&v"(x[0][0]), "=&v"(x[0][1]), "=&v"(x[0][2]), "=&v"(x[0][3]), "=&v"(x[1][0]), "=&v"(x[1][1]),
                 "=&v"(x[1][2]), "=&v"(x[1][3])
               : "v"(ar), "v"(a0), "v"(a1)
               : "memory");
}
__device__ __forceinline__ void lds_rd8_b128_nowait(uint32_t a0, uint32_t a1, v4f_t (&x)[2][4]) {
  asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:64\n\tds_read_b128 %2, %8 offset:128\n\tds_read_b128 %3, %8 offset:192\n\t"
               "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:64\n\tds_read_b128 %6, %9 offset:128\n\tds_read_b128 %7, %9 offset:192"
               : "=&v"(x[0][0]), "=&v"(x[0][1]), "=&v"(x[0][2]), "=&v"(x[0][3]), "=&v"(x[1][0]), "=&v"(x[1][1]), "=&v"(x[1][2]),
                 "=&v"(x[1][3])
               : "v"(a0), "v"(a1)
               : "memory");
}
__device__ __forceinline__ void lds_wait_pieces(v4f_t (&x)[2][4]) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[0][2]), "+v"(x[0][3]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[1][2]), "+v"(x[1][3])
               :
               : "memory");
}
// the counts of a lane's N row pieces: N reads in flight, one wait
template <int N>
__device__ __forceinline__ void lds_rd_counts(uint32_t base, const uint32_t (&off)[N], int (&cn)[N]) {
  if constexpr (N == 5) {
    asm volatile("ds_read_b32 %0, %5\n\tds_read_b32 %1, %6\n\tds_read_b32 %2, %7\n\tds_read_b32 %3, %8\n\tds_read_b32 %4, %9\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(cn[0]), "=&v"(cn[1]), "=&v"(cn[2]), "=&v"(cn[3]), "=&v"(cn[4])
                 : "v"(base + off[0]), "v"(base + off[1]), "v"(base + off[2]), "v"(base + off[3]), "v"(base + off[4])
                 : "memory");
  } else if constexpr (N == 4) {
    asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %5\n\tds_read_b32 %2, %6\n\tds_read_b32 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(cn[0]), "=&v"(cn[1]), "=&v"(cn[2]), "=&v"(cn[3])
                 : "v"(base + off[0]), "v"(base + off[1]), "v"(base + off[2]), "v"(base + off[3])
                 : "memory");
  } else {
#pragma unroll
    for (int i = 0; i < N; i++) cn[i] = lds_rd_b32(base + off[i]);
  }
}
// sum over the four lanes of a quad (quad_perm [1,0,3,2] then [2,3,0,1])
__device__ __forceinline__ float dc_quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  return v;
}

// the kernel body as a device function of the workgroup number `bid` (see dc_k1m_body)
// COH (round 6, the persistent batch kernel of dense_batch.hip): the tables were stored write-through by ANOTHER workgroup of a
// launch that is still running -> every global read (plane DMA, counts, records) is an sc1 load (L2-served, never the CU's L1), which
// replaces an agent-scope acquire per tile (MI355X_MICROARCH.md "inter-workgroup visibility").  false: plain loads behind a kernel
// boundary, as ever.
template <int OP, int R, bool DIV, bool COH = false>
__device__ __forceinline__ void dc_k2q_body(
    const float *__restrict__ S_, const int32_t *__restrict__ cell_n, const int4 *__restrict__ slots,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, int cg, float coord_div, float eps, int64_t n, const link_dc_grid_t &g, int txn, int tyn,
    int zsplit, int nwg, void *__restrict__ out, unsigned long long *__restrict__ dbg, const int bid, const unsigned tidx) {
  using K2 = dc_k2_cfg<OP, R>;
  using KQ = dc_k2q_cfg<OP, R>;
  DC_PROF_PTR(dbg);
  // optional per-wave timing (tools/k2prof.py): s_memtime ticks waiting for the plane DMA, in the barrier, in the box sums /
  // the quad round
  unsigned long long tq0 = dbg ? DC_NOW() : 0, tq_dma = 0, tq_bar = 0, tq_work = 0;
  int tq_rounds = 0;
  using K = typename K2::G;
  constexpr int C = 64, P = 2, TY = K::TY, TX = K::TX, HY = K::HY, HLO = K::HLO;
  constexpr int RB = P * C * 4;
  static_assert(K2::P == 2, "two-part rows");
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const bool producer = tidx < 256;
  const int tid = tidx & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((tidx & 255) >> 6);
  const int per = (nwg + 7) >> 3;
  const int L = (bid & 7) * per + (bid >> 3);
  if (L >= nwg) return;
  int t = L;
  const int zseg = t % zsplit; t /= zsplit;
#if DC_K2_MAP
  // tiles of one batch item enumerated y-half by y-half: a contiguous range of L (what one XCD runs) is then ~2.5 x 5 tiles
  // instead of ~1.25 x 10 -- fewer halo rows fetched into two L2s
  const int per_b = txn * tyn;
  const int b = t / per_b;
  int tt2 = t - b * per_b;
  const int H = (tyn + 1) >> 1;
  int tx, ty;
  if (tt2 < txn * H) { tx = tt2 / H; ty = tt2 - tx * H; }
  else { tt2 -= txn * H; const int H2 = tyn - H; tx = tt2 / H2; ty = H + tt2 - tx * H2; }
#else
  const int ty = t % tyn; t /= tyn;
  const int tx = t % txn;
  const int b = t / txn;
#endif
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int PDx = g.pdim[0], PDy = g.pdim[1], PDz = g.pdim[2];
  const int x0 = tx * TX, y0 = ty * TY;
  const int zs = (int)(((long long)Dz * zseg) / zsplit), ze = (int)(((long long)Dz * (zseg + 1)) / zsplit);
  if (zs >= ze) return;
  const int nplanes = (ze - zs) + R - 1;
  const int pz0 = zs + 1 - HLO;
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)lds;
  const uint32_t abuf0 = lds_base + K2::SPLIT_ABUF_OFF, ncnt0 = lds_base + K2::SPLIT_NCNT_OFF;
  if (tidx < 128) {                             // LayerNorm weight | bias image for the consumers
    float *par = reinterpret_cast<float *>(lds + KQ::PAR_OFF);
    par[tidx] = tidx < 64 ? ln_w[tidx] : ln_b[tidx - 64];
  }
  if (producer) {
    // ================= producers: plane ring (LDS-DMA, 3 slots, prefetch distance 2) + box sums -> A rows =================
    auto col_cell0 = [&](int hx, int hy) {           // padded cell id of (haloed column, z = 0), clamped into the grid
      int px = x0 + 1 - HLO + hx, py = y0 + 1 - HLO + hy;
      px = px < PDx - 1 ? px : PDx - 1;
      py = py < PDy - 1 ? py : PDy - 1;
      return (uint32_t)(((b * PDx + px) * PDy + py) * PDz);
    };
    uint32_t src_off[K::PASSES], cnt_off[K::PASSES];
#pragma unroll
    for (int i = 0; i < K::PASSES; i++) {
      int pid = i * 256 + tid;
      if (pid >= K::NPC) pid = K::NPC - 1;
      const int col = pid / K::RP, pcs = pid % K::RP;
      src_off[i] = col_cell0(col / HY, col % HY) * (uint32_t)RB + (uint32_t)pcs * 16u;
      cnt_off[i] = (uint32_t)col * 4u;
    }
    uint32_t cnt_cell0;
    {
      int e = wave * 64 + lane;
      if (e >= K::NCOL) e = K::NCOL - 1;
      cnt_cell0 = col_cell0(e / HY, e % HY);
    }
    const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, (uint32_t)((g.vp + 1) * RB));
    // (ring slots are carried as rotating counters `pm3` = plane % 3: a modulo by 3 is a multiply-high + shift + multiply + subtract
    // on the scalar unit, five of them per plane step and wave)
    auto cnt_slot = [&](int plane, int pm3) -> uint32_t {     // LDS byte offset of the count image of `plane`
      (void)plane;
      return (uint32_t)(pm3 * K2::SPLIT_BUF_BYTES + K2::SPLIT_PLANE);
    };
    auto issue_cnt = [&](int plane, int pm3) {
      int pz = pz0 + plane;
      pz = pz < PDz - 1 ? pz : PDz - 1;
      const int32_t *csrc = cell_n + cnt_cell0 + pz;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)csrc,
                                       (__attribute__((address_space(3))) void *)(lds + cnt_slot(plane, pm3) + wave * 256), 4, 0, COH ? 16 : 0);
    };
    uint32_t rec_cell0;                                // inline slot records of the 16 interior cells of an output plane
    int rec_k;
    {
      const int piece = wave * 16 + (lane & 15);
      const int col = piece >> 2;
      rec_k = piece & 3;
      rec_cell0 = col_cell0(col / TY + HLO, col % TY + HLO);
    }
    const char *Sb = reinterpret_cast<const char *>(S_);
    auto issue = [&](int plane, int pm3, bool) {
      int pz = pz0 + plane;
      pz = pz < PDz - 1 ? pz : PDz - 1;
      char *buf = lds + pm3 * K2::SPLIT_BUF_BYTES;
#pragma unroll
      for (int i = 0; i < K::PASSES; i++) {
        const bool surplus = (i * 256 + wave * 64) >= K::NPC;          // wave-uniform: repeat pass 0 (same data, same place)
        const int ii = surplus ? 0 : i;
        const char *src = Sb + (size_t)(surplus ? src_off[0] : src_off[i]) + (size_t)pz * RB;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(buf + (ii * 256 + wave * 64) * 16), 16, 0, COH ? 16 : 0);
      }
      issue_cnt(plane, pm3);
      int po = pz0 + plane - (R - 1) + HLO;             // output plane closed by this plane
      po = po < 0 ? 0 : (po < PDz - 1 ? po : PDz - 1);
      const int4 *rsrc = slots + ((size_t)(rec_cell0 + po) * DC_INL + rec_k);
      if (lane < 16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)rsrc,
                                         (__attribute__((address_space(3))) void *)(lds + K2::SPLIT_REC_OFF + (plane & 3) * K2::REC_BYTES + wave * 256), 16, 0, COH ? 16 : 0);
    };
    const int grp = tid >> 4, li = tid & 15;
    // Rim tiles (DC_K2Q_RIM, round 5).  The 4 x 4 tiling of a 37 x 37 grid has 19 rim tiles of 100 whose columns beyond the grid
    // hold nothing; their producer waves used to form box sums for them all the same (a quarter of a rim tile's columns are real).
    // A producer wave whose four lane groups are all outside skips the plane's reads, sums and row writes -- it still issues its
    // share of the plane DMA and keeps the barriers.  Wave w owns columns ix = w (groups 4w .. 4w+3 = iy 0..3), which makes whole
    // waves idle on the x rim only; on the y rim the (group -> column) map is transposed (wave w owns iy = w).  The A image and
    // the counts stay addressed by the canonical cell number ix * TY + iy the mapper and the consumers use.
    const bool rim_t = DC_K2Q_RIM && (y0 + TY > Dy) && (x0 + TX <= Dx);      // workgroup-uniform
    const int ix = rim_t ? grp % TX : grp / TY, iy = rim_t ? grp / TX : grp % TY;
    const int cell = ix * TY + iy;
    const bool col_ok = (x0 + ix < Dx) && (y0 + iy < Dy);
    const bool wave_has = (DC_K2Q_RIM && DC_K2Q_PMAP) ? __any(col_ok) : true;   // wave-uniform
    if (!wave_has && li == 0) {                        // its cells hold nothing in any plane: both count images say so once (the
      lds_wr_b32(ncnt0 + (uint32_t)(cell * 4), 0);     // consumers walk them for planes of more than 64 voxels)
      lds_wr_b32(ncnt0 + (uint32_t)(K2::NG * 4 + cell * 4), 0);
    }
    const uint32_t row_lane = (uint32_t)((ix * HY + iy) * RB + li * 16);
    const uint32_t cnt_lane = (uint32_t)((ix * HY + iy) * 4);
    float4 r0[P], r1[P];
    float c0 = 0.f, c1 = 0.f;
    int n_prev = 0;                                    // voxels in this group's cell of the previous plane
#pragma unroll
    for (int pp = 0; pp < P; pp++) r0[pp] = r1[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    issue(0, 0, false);
    if (nplanes > 1) issue(1, 1, false);
    int m3 = 0, m3p2 = 2;                              // i % 3, (i + 2) % 3
    for (int i = 0; i <= nplanes; i++, m3 = m3 == 2 ? 0 : m3 + 1, m3p2 = m3p2 == 2 ? 0 : m3p2 + 1) {
      unsigned long long tqa = dbg ? DC_NOW() : 0;
      if (i < nplanes) {
        if (i + 1 < nplanes) wait_vmcnt<K2::NI>(); else wait_vmcnt<0>();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the A rows of the previous step are in LDS
      if (dbg) { const unsigned long long tqb = DC_NOW(); tq_dma += tqb - tqa; tqa = tqb; }
      asm volatile("s_barrier" ::: "memory");
      if (dbg) { const unsigned long long tqb = DC_NOW(); tq_bar += tqb - tqa; tqa = tqb; }
      if (i >= nplanes) break;
      if (i + 2 < nplanes) issue(i + 2, m3p2, true);
      const uint32_t bufa = lds_base + (uint32_t)(m3 * K2::SPLIT_BUF_BYTES);
      if (!wave_has) continue;                            // rim tile: none of this wave's columns lies inside the grid
      float4 cur[P];
      float cc = 0.f;
#pragma unroll
      for (int pp = 0; pp < P; pp++) cur[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
      {
        const uint32_t ra = bufa + row_lane, ca = lds_base + cnt_slot(i, m3) + cnt_lane;
        if constexpr (R == 3) {
          dc_read_plane_p2r3<C>(ra, ca, cur, cc);
        } else {
          dc_read_dx<C, P, R, 0>(ra, ca, cur, cc);
          dc_read_dx<C, P, R, 1>(ra, ca, cur, cc);
        }
      }
      const int n_here = lds_rd_b32(lds_base + cnt_slot(i, m3) + cnt_lane + (uint32_t)((HLO * HY + HLO) * 4));
      if (i >= R - 1) {
        const uint32_t abuf = abuf0 + (uint32_t)((i & 1) * K2::NG * RB), ncnt = ncnt0 + (uint32_t)((i & 1) * K2::NG * 4);
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
#pragma unroll
        for (int pp = 0; pp < P; pp++)
          lds_wr_b128(abuf + (uint32_t)(cell * RB + pp * C * 4 + li * 16),
                      make_float4(a[pp].x * inv, a[pp].y * inv, a[pp].z * inv, a[pp].w * inv));
        if (li == 0) lds_wr_b32(ncnt + (uint32_t)(cell * 4), col_ok ? n_prev : 0);   // the plane that closed is the previous one for both R
      }
#pragma unroll
      for (int pp = 0; pp < P; pp++) { r0[pp] = r1[pp]; r1[pp] = cur[pp]; }
      c0 = c1; c1 = cc;
      n_prev = n_here;
      if (dbg) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq_work += DC_NOW() - tqa; }
    }
    if (dbg && lane == 0) {
      unsigned long long *d = dbg + ((size_t)L * 8 + (tidx >> 6)) * 8;
      d[0] = DC_NOW() - tq0; d[1] = tq_dma; d[2] = tq_bar; d[3] = tq_work; d[4] = tq0; d[5] = 0; d[6] = nplanes; d[7] = 1;
    }
    return;
  }
  constexpr int CW = DC_K2Q_CW, SLOTS = 16 * CW;
  if (DC_K2Q_PMAP && wave == CW) {
    // ================= mapper: one wave lays out the voxel -> (cell, slot) map of every plane =================
    // After barrier i the count image of plane i is in LDS (the producers waited for its DMA before the barrier; its ring
    // slot is requested again only at step i + 1).  Lane v finds the cell of the plane's v-th voxel -- cells in group order,
    // slots in id order -- from the plane's 16 interior counts (broadcast reads, four per round trip) and writes
    // cell << 12 | slot; the consumers read the entry two barriers later (plane i is the output plane of step i + 1).
    int m3 = 0;                                        // i % 3
    for (int i = 0; i <= nplanes; i++, m3 = m3 == 2 ? 0 : m3 + 1) {
      asm volatile("s_barrier" ::: "memory");
      if (i >= nplanes) break;
      const uint32_t ca = lds_base + (uint32_t)(m3 * K2::SPLIT_BUF_BYTES + K2::SPLIT_PLANE);
      int run = 0, c_ = 0, start = 0;
#define Q_(k) "i"(((((k) / TY) + HLO) * HY + ((k) % TY) + HLO) * 4)
#define DC_MAP4(K0)                                                                                                          \
      {                                                                                                                      \
        int n0, n1, n2, n3;                                                                                                  \
        asm volatile("ds_read_b32 %0, %4 offset:%c5\n\tds_read_b32 %1, %4 offset:%c6\n\tds_read_b32 %2, %4 offset:%c7\n\t"     \
                     "ds_read_b32 %3, %4 offset:%c8\n\ts_waitcnt lgkmcnt(0)"                                                \
                     : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3)                                                            \
                     : "v"(ca), Q_(K0), Q_(K0 + 1), Q_(K0 + 2), Q_(K0 + 3)                                                   \
                     : "memory");                                                                                            \
        const int nn[4] = {n0, n1, n2, n3};                                                                                  \
        _Pragma("unroll") for (int kk = 0; kk < 4; kk++) {                                                                   \
          const int k = K0 + kk;                                                                                             \
          const bool okk = (x0 + k / TY < Dx) && (y0 + k % TY < Dy);   /* wave-uniform: columns beyond the grid hold nothing */ \
          run += okk ? nn[kk] : 0;                                                                                           \
          if (k < 15) {                                                                                                      \
            const bool le = run <= lane;                                                                                     \
            c_ += le ? 1 : 0;                                                                                                \
            start = le ? run : start;                                                                                        \
          }                                                                                                                  \
        }                                                                                                                    \
      }
      DC_MAP4(0) DC_MAP4(4) DC_MAP4(8) DC_MAP4(12)
#undef DC_MAP4
#undef Q_
      const uint32_t mb = lds_base + (uint32_t)(KQ::MAP_OFF + m3 * KQ::MAP_BYTES);
      lds_wr_b32(mb + (uint32_t)(lane * 4), (c_ << 12) | ((lane - start) & 4095));
      if (lane == 0) lds_wr_b32(mb + 256u, run);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    return;
  }
  // ================= consumers: one voxel per quad of lanes =================
  const int q = lane & 3, li16 = lane & 15;
  const int qd = wave * 16 + (lane >> 2);              // 0..SLOTS-1: this quad's slot among the plane's voxels
  const __amdgpu_buffer_rsrc_t r_out = dc_rsrc(out, (uint32_t)(n * C * IO_BYTES));
  // theta weights of the lane's channels 16 j + 4 q + e: blocks j and j + 2 share theta (channels ch and ch + 32)
  // (no alpha here: the launcher sends blocks with a theta scale -- cos_x only in the reference, linkunet.py:165 -- to the other forms)
  float w0[2][4], w1[2][4], w2[2][4];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const int tc = (16 * j + 4 * q + e) % cg;
      w0[j][e] = w_pos[3 * tc + 0]; w1[j][e] = w_pos[3 * tc + 1]; w2[j][e] = w_pos[3 * tc + 2];
    }
  const uint32_t par = lds_base + (uint32_t)KQ::PAR_OFF + (uint32_t)(q * 16);
  bool slow_k = false;                                 // wave-uniform, the same in every wave: some |theta| may leave the fast range
  if (DC_THETA_BOUND) {
    float ax, ay, az;
    dc_coord_absmax(g, DIV ? coord_div : 1.0f, ax, ay, az);
    bool b_ = false;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) b_ |= dc_theta_leaves_fast_range(ax, ay, az, w0[j][e], w1[j][e], w2[j][e], 1.0f);
    slow_k = __any(b_);
  }
  int m3p1 = 1;                                        // (i + 1) % 3 == (jp - 1) % 3
  for (int i = 0; i <= nplanes; i++, m3p1 = m3p1 == 2 ? 0 : m3p1 + 1) {
    unsigned long long tqa = dbg ? DC_NOW() : 0;
    asm volatile("s_barrier" ::: "memory");
    if (dbg) { const unsigned long long tqb = DC_NOW(); tq_bar += tqb - tqa; tqa = tqb; }
    const int jp = i - 1;                               // the output-plane step whose A rows the producers finished last step
    if (jp < R - 1 || jp >= nplanes) continue;
    const uint32_t abuf = abuf0 + (uint32_t)((jp & 1) * K2::NG * RB), ncnt = ncnt0 + (uint32_t)((jp & 1) * K2::NG * 4);
    const uint32_t recb = lds_base + (uint32_t)(K2::SPLIT_REC_OFF + (jp & 3) * K2::REC_BYTES);
    const int po = pz0 + jp - (R - 1) + HLO;
    int Tv, m0 = 0;
    uint32_t mb = 0;
    if (DC_K2Q_PMAP) {
      // the producers laid the plane's voxel map out two steps ago (plane jp - 1): entry = cell << 12 | slot, then the voxel count
      mb = lds_base + (uint32_t)(KQ::MAP_OFF + m3p1 * KQ::MAP_BYTES);
      int tv_;
      asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %3 offset:256\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(m0), "=&v"(tv_) : "v"(mb + (uint32_t)(qd * 4)), "v"(mb) : "memory");
      Tv = __builtin_amdgcn_readfirstlane(tv_);
    } else {
      Tv = 0;
    }
    int e_[16];
    if (!DC_K2Q_PMAP) {
      // inclusive prefix of the 16 cell counts: every DPP row holds all of them, so they become wave-uniform scalars
      int incl = lds_rd_b32(ncnt + (uint32_t)(li16 * 4));
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
#pragma unroll
      for (int k = 0; k < 16; k++) e_[k] = __builtin_amdgcn_readlane(incl, k);
      Tv = e_[15];
    }
    for (int base = 0; base < Tv; base += SLOTS) {      // wave-uniform: one pass unless the plane holds more than SLOTS voxels
      if (DC_K2Q_EARLY && base + wave * 16 >= Tv) break;   // none of this wave's quads has a voxel (wave-uniform)
      if (dbg) tq_rounds++;
      const bool valid = base + qd < Tv;                // quads without a voxel take the plane's first one and store nothing
      int c, k;
      if (DC_K2Q_PMAP) {
        if (Tv <= 64) {                                 // wave-uniform: the map covers the plane (a second round re-reads it)
          const int mm = base == 0 ? m0 : lds_rd_b32(mb + (uint32_t)((valid ? base + qd : 0) * 4));
          c = valid ? (mm >> 12) : 0;
          k = valid ? (mm & 4095) : 0;
        } else {                                        // a plane with more voxels than the map holds (dense cells): walk the counts
          const int v = valid ? base + qd : 0;
          int run = 0, start = 0;
          c = 0;
          for (int kk = 0; kk < 15; kk++) {
            run += lds_rd_b32(ncnt + (uint32_t)(kk * 4));
            const bool le = run <= v;
            c += le ? 1 : 0;
            start = le ? run : start;
          }
          k = v - start;
        }
      } else {
        const int v = valid ? base + qd : Tv - 1;
        c = 0;
        int start = 0;                                  // cell of voxel v = number of prefix entries <= v; start = the last such entry
#pragma unroll
        for (int kk = 0; kk < 15; kk++) {
          const bool le = e_[kk] <= v;                  // e_ is non-decreasing and wave-uniform (scalar operands)
          c += le ? 1 : 0;
          start = le ? e_[kk] : start;
        }
        k = v - start;
      }
      v4f_t rq, Av[2][4];
      if (DC_K2Q_OVERLAP)
        lds_rd9_b128_rec_first(recb + (uint32_t)((c * DC_INL + (k < DC_INL ? k : 0)) * 16), abuf + (uint32_t)(c * RB + q * 16),
                               abuf + (uint32_t)(c * RB + C * 4 + q * 16), rq, Av);
      else
        lds_rd9_b128(recb + (uint32_t)((c * DC_INL + (k < DC_INL ? k : 0)) * 16), abuf + (uint32_t)(c * RB + q * 16),
                     abuf + (uint32_t)(c * RB + C * 4 + q * 16), rq, Av);
      int4 rec = make_int4(__float_as_int(rq.x), __float_as_int(rq.y), __float_as_int(rq.z), __float_as_int(rq.w));
      if (k >= DC_INL) {                                // overflow records (cells with more than DC_INL voxels): ordinary loads
        const int pcell = ((b * PDx + x0 + c / TY + 1) * PDy + y0 + c % TY + 1) * PDz + po;
        rec = ld16i_c<COH>(dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16)), slots, dc_slot(g, pcell, k));
      }
      float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
      if (DIV) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
      float th[2][4], sn[2][4], cs[2][4];
      bool big = false;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          th[j][e] = fmaf(z, w2[j][e], fmaf(y, w1[j][e], x * w0[j][e]));      // theta_of without the scale
        }
      if (!DC_THETA_BOUND) {                            // range of the fast sincos: ONE compare on the largest magnitude (an
        float mx = fmaxf(fmaxf(fabsf(th[0][0]), fabsf(th[0][1])), fabsf(th[0][2]));   // infinity survives v_max; a NaN theta gives
        mx = fmaxf(fmaxf(mx, fabsf(th[0][3])), fabsf(th[1][0]));                       // NaN on either path)
        mx = fmaxf(fmaxf(mx, fabsf(th[1][1])), fabsf(th[1][2]));
        mx = fmaxf(mx, fabsf(th[1][3]));
        big = valid && !(mx < 32768.0f);                // (a quad without a voxel may hold the stale record of an empty cell: it
                                                        // must not pick the wave's path, or results depend on the previous frame)
      }
      if (__builtin_expect(DC_THETA_BOUND ? slow_k : __any(big), 0)) {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) sincos_nocall(th[j][e], sn[j][e], cs[j][e]);
      } else {
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int e = 0; e < 4; e++) sincos_small(th[j][e], sn[j][e], cs[j][e]);
      }
      if (DC_K2Q_OVERLAP) lds_wait_pieces(Av);          // the A pieces arrived while theta / sincos were evaluated
      float nv[4][4], s = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float A0 = Av[0][j][e], A1 = Av[1][j][e], cc_ = cs[j & 1][e], ss_ = sn[j & 1][e];
          if (DC_K2Q_FMA) {
            nv[j][e] = fmaf(OP == LINK_OP_SIN ? -A1 : A1, ss_, A0 * cc_);
          } else {
            if (OP == LINK_OP_SIN) nv[j][e] = __fsub_rn(__fmul_rn(A0, cc_), __fmul_rn(A1, ss_));      // linkunet.py:148
            else nv[j][e] = __fadd_rn(__fmul_rn(A0, cc_), __fmul_rn(A1, ss_));                         // :162
          }
          s += nv[j][e];
        }
      v4f_t gwb[2][4];                                  // LayerNorm weight | bias pieces q, q+4, q+8, q+12
      if (DC_K2Q_OVERLAP) lds_rd8_b128_nowait(par, par + 256u, gwb);      // land during the two reductions below
      s = dc_quad_sum(s);
      const float mean = s * (1.0f / C);
      float qq = 0.f;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float d = nv[j][e] - mean;
          qq += d * d;
        }
      qq = dc_quad_sum(qq);
      const float rs = __builtin_amdgcn_rsqf(qq * (1.0f / C) + eps);
      if (DC_K2Q_OVERLAP) lds_wait_pieces(gwb); else lds_rd8_b128(par, par + 256u, gwb);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        float4 o;
        o.x = (nv[j][0] - mean) * rs * gwb[0][j][0] + gwb[1][j][0]; o.y = (nv[j][1] - mean) * rs * gwb[0][j][1] + gwb[1][j][1];
        o.z = (nv[j][2] - mean) * rs * gwb[0][j][2] + gwb[1][j][2]; o.w = (nv[j][3] - mean) * rs * gwb[0][j][3] + gwb[1][j][3];
        io_st4(r_out, (uint32_t)rec.w * (uint32_t)C + (uint32_t)(16 * j + 4 * q), valid, o);
      }
    }
    if (dbg) tq_work += DC_NOW() - tqa;
  }
  if (dbg && lane == 0) {
    unsigned long long *d = dbg + ((size_t)L * 8 + (tidx >> 6)) * 8;
    d[0] = DC_NOW() - tq0; d[1] = tq0; d[2] = tq_bar; d[3] = 0; d[4] = tq_work; d[5] = tq_rounds; d[6] = nplanes; d[7] = 2;
  }
}

template <int OP, int R, bool DIV>
__global__ void __launch_bounds__(256 + 64 * (DC_K2Q_CW + (DC_K2Q_PMAP ? 1 : 0)), 4) k_dc_gather_demod_quad(
    const float *__restrict__ S_, const int32_t *__restrict__ cell_n, const int4 *__restrict__ slots,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, int cg, float coord_div, float eps, int64_t n, link_dc_grid_t g, int txn, int tyn,
    int zsplit, int nwg, void *__restrict__ out, unsigned long long *__restrict__ dbg) {
  dc_k2q_body<OP, R, DIV>(S_, cell_n, slots, w_pos, alpha, ln_w, ln_b, cg, coord_div, eps, n, g, txn, tyn, zsplit, nwg, out, dbg,
                          (int)blockIdx.x, threadIdx.x);
}
