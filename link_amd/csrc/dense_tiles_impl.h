// link_amd/csrc/dense_tiles_impl.h -- tile form of the fused pre_mix + LayerNorm + modulate + per-cell-sum kernel
// (round 3; include/link_amd.h section E, link_dc_tuning_t::k1_form = 1).  Compiled once per feature I/O type like
// dense_fused_impl.h (DC_IO / DC_IO_NS).
//
// What changed against the cell-range form of round 2 (dense_fused_impl.h, the default, k1_form = 0), and why -- the
// round-2 ablations left 16.7 of 25.6 us in the kernel's frame, not in its arithmetic:
//   * the slot lists hold voxel IDS (4 bytes, 8 inline per cell = one 32-byte piece per cell) instead of 16-byte
//     records: a lane owns a cell, sorts its <= 8 ids in registers (19-exchange network) and scatters them to a
//     flat LDS list by the wave prefix of the counts -- one round trip, no per-record sort loop; cells with more than
//     8 voxels (2e-4 of the cells on cfg2, every cell of a LiDAR frame) are ranked by counting in their own pass;
//   * a tile's per-cell sums are formed IN the matrix-core accumulator layout: a DPP row holds the tile's 16 voxels
//     (same 16 channels each), so the sum over the voxels of a cell is a segmented scan along the row -- four
//     v_fmac_f32_dpp per value with 0/1 segment masks -- and the closing lane of each segment stores its piece of
//     the S row.  No X tile in LDS (8.4 KB per wave, a quarter of the old kernel's LDS cycles were bank conflicts),
//     no second pass over the tile.  A cell that straddles two tiles hands its partial sum over through 512 bytes
//     of LDS;
//   * a wave keeps no voxel records: (x, y, z) come from coords[id] with the row gather; the id-ordered records the
//     gather kernel deals from are written back to `slots` from the tile (one 16-byte store per voxel);
//   * 36 KB of LDS and <= 128 registers instead of 80 KB / 198: four workgroups per CU instead of two, and room
//     beside the gather kernel of another frame.
// Sums are formed in a fixed tree over the id-ordered voxels of a cell: bitwise reproducible, but not the
// left-to-right order of the cell-range form (the two forms agree to rounding; the reference's own GPU kernel
// sums with atomics in arrival order, voxelize_cuda.cu:12-25).
// Non-finite feature rows: 0 * inf inside the segmented scan can spread a NaN to the other cells of the same
// 16-voxel tile (the reference confines it to the cell's r^3 neighbourhood).
#pragma once
#include "tile_common.h"

#ifndef DC_T2_WAVES
#define DC_T2_WAVES 3     /* register budget = 512 / this; at 4 (128 registers) the tile body spills 40 of them */
#endif
#ifndef DC_T2_X1
#define DC_T2_X1 0
#define DC_T2_X2 0
#define DC_T2_X3 0
#endif
#ifndef DC_T2_ABL
#define DC_T2_ABL 0       /* ablation builds (wrong results!): 1 no MFMA, 2 no sincos, 4 no segmented scan, 8 no S stores */
#endif

namespace DC_IO_NS {
using namespace link;

template <int C, int OP>
struct dc_t2_cfg {
  static constexpr int T = C / 16;
  static constexpr int P = op_parts<OP>::value;
  static constexpr int RB = P * C * 4;                 // bytes of one S row
  static constexpr int RGL = P * C / 4;                // lanes holding one row (16 B each)
  static constexpr int LDH = dc_wimg<C>::LDH;
  static constexpr int WIMG_BYTES = dc_wimg<C>::WIMG_BYTES;
  static constexpr int W_BYTES = dc_wimg<C>::W_BYTES;
  static constexpr int LIST_N = 512;                   // 64 cells x 8 inline ids
  static constexpr int DUMP_N = 16;                    // where the writes of empty id slots go
  static constexpr int RAW_N = 352;                    // ids of one heavy cell (k <= 352, as the cell-range form)
  static constexpr int LIST_BYTES = (LIST_N + DUMP_N) * 8 + RAW_N * 4;   // list entries: (chunk lane << 26 | voxel id, padded cell id)
  static constexpr int CARRY_BYTES = P * C * 4;        // open cell at a tile boundary: 4 lane groups x P parts x C/4 values
  static constexpr int WAVE_BYTES = LIST_BYTES + CARRY_BYTES;
  static constexpr int NW = 4;
  static constexpr int LDS_BYTES = W_BYTES + NW * WAVE_BYTES;
};

template <int C, int OP, int NB>
__global__ void __launch_bounds__(256, DC_T2_WAVES) k_dc_tiles(
    const void *__restrict__ feats, const int4 *__restrict__ coords, const uint32_t *__restrict__ sid,
    int4 *__restrict__ slots, uint32_t *__restrict__ cnt, int32_t *__restrict__ cell_n,
    const float *__restrict__ w_pre, const float *__restrict__ ln_w, const float *__restrict__ ln_b,
    const float *__restrict__ w_pos, const float *__restrict__ alpha, int cg, float coord_div, float eps, int64_t n,
    link_dc_grid_t g, int cpw, bool warm, float *__restrict__ S_, float *__restrict__ fin, int32_t *__restrict__ hdr,
    unsigned long long *__restrict__ dbg) {
  using K = dc_t2_cfg<C, OP>;
  // optional phase timing (tools/dcbench.py --phases): per wave 8 slots of s_memtime values
  unsigned long long tq0 = dbg ? __builtin_amdgcn_s_memtime() : 0, tq1 = 0, tq_chunk = 0, tq_mfma = 0, tq_valu = 0, tq_scan = 0;
  int tq_tiles = 0;
  constexpr int T = K::T, P = K::P, NV = 4 * T;        // NV values per part and lane (16 at C = 64)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *ln_lds = reinterpret_cast<float *>(smem_raw + K::WIMG_BYTES);
  float *pw_lds = ln_lds + 2 * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  char *wbase = smem_raw + K::W_BYTES + wave * K::WAVE_BYTES;
  uint2 *list = reinterpret_cast<uint2 *>(wbase);
  uint32_t *raw = reinterpret_cast<uint32_t *>(list + K::LIST_N + K::DUMP_N);
  float *carry = reinterpret_cast<float *>(wbase + K::LIST_BYTES);
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int Vi = Dx * Dy * Dz * g.dim[3];
  const int wid = blockIdx.x * K::NW + wave;
  const int c_begin = wid * cpw;
  const int c_end = (c_begin + cpw < Vi) ? c_begin + cpw : Vi;
  const uint32_t *__restrict__ csrc = warm ? reinterpret_cast<const uint32_t *>(cell_n) : cnt;
  const uint4 *__restrict__ sid4 = reinterpret_cast<const uint4 *>(sid);
  // a chunk = 64 consecutive interior cells, one per lane: padded cell id, count, the 8 inline ids
  auto cell_of = [&](int chunk, int nrem) {
    const int q = chunk + (lane < nrem ? lane : 0);
    const int z = q % Dz;
    int t = q / Dz;
    const int y = t % Dy;
    t /= Dy;
    return dc_cell(g, t % Dx, y, z, t / Dx);
  };
  int pc_n = 0, cn_n = 0;
  uint4 ia_n = make_uint4(0, 0, 0, 0), ib_n = ia_n;
  auto load_chunk = [&](int chunk) {                   // the first chunk's request is issued before W is staged
    const int nrem = (c_end - chunk < 64) ? c_end - chunk : 64;
    pc_n = cell_of(chunk, nrem);
    cn_n = (int)csrc[pc_n];
    ia_n = sid4[(int64_t)pc_n * 2];
    ib_n = sid4[(int64_t)pc_n * 2 + 1];
  };
  if (c_begin < c_end) load_chunk(c_begin);
  // stage W (fp16 hi | lo image) and the parameters; a weight outside the fp16 split's range: fp32 contraction
  bool w_big = dc_stage_weights<C, 64 * K::NW, LINK_TILE_EXACT(OP)>(smem_raw, w_pre, ln_w, ln_b, w_pos, alpha, cg, tid);
  for (int i = lane; i < K::CARRY_BYTES / 4; i += 64) carry[i] = 0.f;     // read unconditionally by every tile (times 0 unless a cell straddles)
  if (blockIdx.x == 0 && tid == 0 && !warm) {          // publish the step's status word
    hdr[LINK_HDR_STATUS] = hdr[LINK_HDR_STATUS_ACC];
    hdr[LINK_HDR_STATUS_ACC] = 0;
  }
  w_big = __syncthreads_or(w_big) != 0 || (LINK_COSX_EXACT && OP == LINK_OP_COSX);   // cos_x: exact contraction (elk_common.h)
  if (dbg) tq1 = __builtin_amdgcn_s_memtime();
  if (c_begin >= c_end) return;
  const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, (uint32_t)((g.vp + 1) * K::RB));
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * C * 4));     // written for cos_x only
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_n = dc_rsrc(cell_n, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_feats = dc_rsrc(feats, (uint32_t)(n * C * IO_BYTES));
  const __amdgpu_buffer_rsrc_t r_coords = dc_rsrc(coords, (uint32_t)(n * 16));
  const unsigned short *wh = reinterpret_cast<const unsigned short *>(smem_raw);

  auto mfma_tile = [&](const float4 (&ff)[T], floatx4 (&cc)[T]) {
    if (DC_T2_ABL & 1) {
#pragma unroll
      for (int tp = 0; tp < T; tp++) cc[tp] = (floatx4){ff[tp].x, ff[tp].y, ff[tp].z, ff[tp].w};
      return;
    }
    dc_premix_tile<C, LINK_TILE_EXACT(OP)>(wh, w_pre, w_big, li, gq, ff, cc);
  };

  for (int chunk = c_begin; chunk < c_end; chunk += 64) {
    unsigned long long tqa = dbg ? __builtin_amdgcn_s_memtime() : 0;
    if (chunk != c_begin) load_chunk(chunk);            // launches normally give a wave one chunk (cpw <= 64)
    const int nrem = (c_end - chunk < 64) ? c_end - chunk : 64;
    const bool act = lane < nrem;
    const int pc = pc_n;
    int cn = act ? cn_n : 0;
    cn = cn < g.k ? cn : g.k;
    uint32_t ks[8] = {ia_n.x, ia_n.y, ia_n.z, ia_n.w, ib_n.x, ib_n.y, ib_n.z, ib_n.w};
    const bool heavy = cn > 8;                          // ranked by counting in its own pass below
    const int nreg = heavy ? 0 : cn;
    // ---- ids of the cell in ascending order (the sums and the gather kernel's pairing must not depend on the arrival
    // order the atomics produced): 19-exchange network, empty slots sort to the end ----
#pragma unroll
    for (int j = 0; j < 8; j++) ks[j] = j < nreg ? (ks[j] & DC_ID_MASK) : 0xFFFFFFFFu;
#define DC_CE(i, j) { const uint32_t lo_ = min(ks[i], ks[j]), hi_ = max(ks[i], ks[j]); ks[i] = lo_; ks[j] = hi_; }
    DC_CE(0, 1) DC_CE(2, 3) DC_CE(4, 5) DC_CE(6, 7)
    DC_CE(0, 2) DC_CE(1, 3) DC_CE(4, 6) DC_CE(5, 7)
    DC_CE(1, 2) DC_CE(5, 6)
    DC_CE(0, 4) DC_CE(1, 5) DC_CE(2, 6) DC_CE(3, 7)
    DC_CE(2, 4) DC_CE(3, 5)
    DC_CE(1, 2) DC_CE(3, 4) DC_CE(5, 6)
#undef DC_CE
    int incl = nreg;                                    // inclusive prefix over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    const int excl = incl - nreg;
    const int total = __builtin_amdgcn_readlane(incl, 63);
#pragma unroll
    for (int j = 0; j < 8; j++)
      list[j < nreg ? excl + j : K::LIST_N + (lane & (K::DUMP_N - 1))] = make_uint2(((uint32_t)lane << DC_ID_BITS) | ks[j], (uint32_t)pc);
    {                                                   // publish the counts, reset the counters
      const uint32_t coff = (act && !warm) ? (uint32_t)pc * 4u : DC_OOB;
      st4i(r_n, coff, cn);
      st4i(r_cnt, coff, 0);
    }
    for (unsigned long long em = __ballot(act && cn == 0); em; em &= em - 1) {   // empty cells: zero rows
      const int pcj = __builtin_amdgcn_readlane(pc, __builtin_ctzll(em));
      st16(r_S, lane < K::RGL ? (uint32_t)pcj * (uint32_t)K::RB + (uint32_t)lane * 16u : DC_OOB, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_chunk += tqb - tqa; tqa = tqb; }
    // ---- passes over the LDS list: the chunk's regular cells first, then one pass per heavy cell ----
    unsigned long long hm = __ballot(act && heavy);
    int ptotal = total, hcell = -1;                     // hcell: chunk lane of the heavy cell this pass belongs to (-1: regular pass)
    for (;;) {
      if (ptotal > 0) {
        const int ntile = (ptotal + 15) >> 4;
        bool cont_prev = false;                         // the tile's first voxels continue the cell the previous tile ended in
        int rank_carry = 0;                             // ... and this many of that cell's voxels came before this tile
        auto ld_rows = [&](int t, uint32_t &key, int &pcell, float4 (&ff)[T], int4 &crd) {
          int sl = 16 * t + li;
          sl = sl < ptotal ? sl : ptotal - 1;
          const uint2 e = list[sl];
          key = e.x;
          pcell = (int)e.y;
          const int id = (int)(key & DC_ID_MASK);
          const v4i_t cr = __builtin_amdgcn_raw_buffer_load_b128(r_coords, (uint32_t)id * 16u, 0, 0);
          crd = make_int4(cr.x, cr.y, cr.z, cr.w);
          const uint32_t ro = ((uint32_t)id * (uint32_t)C + (uint32_t)(4 * gq)) * (uint32_t)IO_BYTES;
          ff[0] = io_ldb4<0>(r_feats, ro);
          if constexpr (T > 1) ff[1] = io_ldb4<16>(r_feats, ro);
          if constexpr (T > 2) { ff[2] = io_ldb4<32>(r_feats, ro); ff[3] = io_ldb4<48>(r_feats, ro); }
        };
        // one tile: the rows are dead once the matrix cores have them, so the NEXT tile's rows are requested into the same
        // registers right after the contraction and land while this tile's LayerNorm / sincos / scan run
        auto tile = [&](int t, uint32_t &key, int &pcell, float4 (&ff)[T], int4 &crd) {
          const int pos = 16 * t + li;
          const bool valid = pos < ptotal;
          const int id = (int)(key & DC_ID_MASK), cl = (int)(key >> DC_ID_BITS);
          unsigned long long tqt = dbg ? __builtin_amdgcn_s_memtime() : 0;
          floatx4 ac[T];
          mfma_tile(ff, ac);
          // the rows are dead once the matrix cores have them: the next tile's rows are requested into the same registers
          // and land while this tile's theta / sincos / LayerNorm / scans run
          uint32_t key_n;
          int pcell_n;
          int4 crd_n;
          ld_rows(t + 1, key_n, pcell_n, ff, crd_n);    // clamped to the list's last entry beyond its end
          if (dbg) { asm volatile("s_nop 0" :: "v"(ac[0][0])); const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_mfma += tqb - tqt; tqt = tqb; }
          // theta of this voxel's blocks
          float x = (float)crd.x, y = (float)crd.y, z = (float)crd.z;
          if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
          float th[NB][4], sn[NB][4], cs[NB][4];
          bool big = false;
#pragma unroll
          for (int tb = 0; tb < NB; tb++) {
            const float4 q0 = *reinterpret_cast<const float4 *>(&pw_lds[16 * tb + 4 * gq]);
            const float4 q1 = *reinterpret_cast<const float4 *>(&pw_lds[C + 16 * tb + 4 * gq]);
            const float4 q2 = *reinterpret_cast<const float4 *>(&pw_lds[2 * C + 16 * tb + 4 * gq]);
            const float4 qa = *reinterpret_cast<const float4 *>(&pw_lds[3 * C + 16 * tb + 4 * gq]);
            th[tb][0] = theta_of(x, y, z, q0.x, q1.x, q2.x, qa.x); th[tb][1] = theta_of(x, y, z, q0.y, q1.y, q2.y, qa.y);
            th[tb][2] = theta_of(x, y, z, q0.z, q1.z, q2.z, qa.z); th[tb][3] = theta_of(x, y, z, q0.w, q1.w, q2.w, qa.w);
#pragma unroll
            for (int r = 0; r < 4; r++) big |= !(fabsf(th[tb][r]) < 32768.0f);
          }
          if (!DC_T2_X2 && __builtin_expect(__any(big), 0)) {        // never on sane inputs
#pragma unroll
            for (int tb = 0; tb < NB; tb++)
#pragma unroll
              for (int r = 0; r < 4; r++) sincos_nocall(th[tb][r], sn[tb][r], cs[tb][r]);
          } else {
#pragma unroll
            for (int tb = 0; tb < NB; tb++)
#pragma unroll
              for (int r = 0; r < 4; r++) {
                if (DC_T2_ABL & 2) { sn[tb][r] = th[tb][r]; cs[tb][r] = 1.0f - th[tb][r]; }
                else sincos_small(th[tb][r], sn[tb][r], cs[tb][r]);
              }
          }
          // LayerNorm over the voxel's C channels: 4T in-lane values + the 4 lane groups
          float s = 0.f;
#pragma unroll
          for (int tp = 0; tp < T; tp++) s += (ac[tp][0] + ac[tp][1]) + (ac[tp][2] + ac[tp][3]);
          s = dc_sum_groups(s);
          const float mean = s * (1.0f / C);
          float qq = 0.f;
#pragma unroll
          for (int tp = 0; tp < T; tp++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const float d = ac[tp][r] - mean;
              qq += d * d;
            }
          qq = dc_sum_groups(qq);
          const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + eps);
          if (dbg) { asm volatile("s_nop 0" :: "v"(rstd)); const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_valu += tqb - tqt; tqt = tqb; }
          // ---- segments of the DPP row: lanes li of one cell are adjacent (the list is cell-major); slots beyond the
          // list's end form segments of their own that are never stored ----
          const int ckey = valid ? cl : 64 + li;
          const int prevk = dc_dpp_i<0x111>(ckey, -1);               // row_shr:1 (lane 0: -1 -> head)
          const int nextk = dc_dpp_i<0x101>(ckey, -2);               // row_shl:1 (lane 15: no source)
          int seg = (prevk != ckey) ? li : 0;                         // segment start, spread by a max-scan
          seg = max(seg, dc_dpp_i<0x111>(seg, 0));
          seg = max(seg, dc_dpp_i<0x112>(seg, 0));
          seg = max(seg, dc_dpp_i<0x114>(seg, 0));
          seg = max(seg, dc_dpp_i<0x118>(seg, 0));
          const int soff = li - seg;                                  // position of this lane's voxel inside its segment
          const float m1 = soff >= 1 ? 1.0f : 0.0f, m2 = soff >= 2 ? 1.0f : 0.0f;
          const float m4 = soff >= 4 ? 1.0f : 0.0f, m8 = soff >= 8 ? 1.0f : 0.0f;
          // most cells hold one to four voxels: the stride-4 / stride-8 steps run only for tiles that have such a segment
          const bool step4 = __any(soff >= 4), step8 = __any(soff >= 8);
          const float cin = (cont_prev && li == 0) ? 1.0f : 0.0f;
          // the id-ordered record of this voxel, where the gather kernel deals from: rank inside the cell = position in the
          // segment (+ what previous tiles held of a cell that straddles)
          const int rank = soff + ((cont_prev && seg == 0) ? rank_carry : 0);
          st16i(r_slots, (valid && gq == 0 && !warm) ? dc_slot(g, pcell, rank) * 16u : DC_OOB, make_int4(crd.x, crd.y, crd.z, id));
          // does the cell of the tile's last voxel go on in the next tile?  (its first entry is lane 0 of the prefetched keys)
          const bool cont_next = 16 * t + 16 < ptotal &&
                                 (int)((uint32_t)__builtin_amdgcn_readlane((int)key_n, 0) >> DC_ID_BITS) == __builtin_amdgcn_readlane(cl, 15);
          const bool closes = valid && (li == 15 ? !cont_next : nextk != ckey);
          const uint32_t srow = closes && !(DC_T2_ABL & 8) ? (uint32_t)pcell * (uint32_t)K::RB + (uint32_t)(16 * gq) : DC_OOB;
          // one 16-channel block (4 values per lane) and one part (cos | sin | theta) at a time: normalise, modulate,
          // carry in, segmented scan, store, carry out -- 8 live values instead of 2 x 4T
          // LayerNorm weights and the carried partial sums of block tp+1 are requested before block tp is worked on: the
          // LDS round trips hide behind the scans instead of heading every block
          float4 lw_n = *reinterpret_cast<const float4 *>(&ln_lds[4 * gq]);
          float4 lb_n = *reinterpret_cast<const float4 *>(&ln_lds[C + 4 * gq]);
          float4 cv_n[P];
#pragma unroll
          for (int pp = 0; pp < P; pp++) cv_n[pp] = *reinterpret_cast<const float4 *>(carry + (pp * 4 + gq) * NV);
#pragma unroll
          for (int tp = 0; tp < T; tp++) {
            const int tb = tp % NB;
            const float4 lw = lw_n, lb = lb_n;
            float4 cv[P];
#pragma unroll
            for (int pp = 0; pp < P; pp++) cv[pp] = cv_n[pp];
            if (tp + 1 < T) {
              lw_n = *reinterpret_cast<const float4 *>(&ln_lds[16 * (tp + 1) + 4 * gq]);
              lb_n = *reinterpret_cast<const float4 *>(&ln_lds[C + 16 * (tp + 1) + 4 * gq]);
#pragma unroll
              for (int pp = 0; pp < P; pp++) cv_n[pp] = *reinterpret_cast<const float4 *>(carry + (pp * 4 + gq) * NV + 4 * (tp + 1));
            }
            const float fv[4] = {(ac[tp][0] - mean) * rstd * lw.x + lb.x, (ac[tp][1] - mean) * rstd * lw.y + lb.y,
                                 (ac[tp][2] - mean) * rstd * lw.z + lb.z, (ac[tp][3] - mean) * rstd * lw.w + lb.w};
            if (OP == LINK_OP_COSX)                     // the de-modulation of cos_x needs fin (linkunet.py:176)
              st16(r_fin, valid ? (uint32_t)id * (uint32_t)(C * 4) + (uint32_t)((16 * tp + 4 * gq) * 4) : DC_OOB,
                   make_float4(fv[0], fv[1], fv[2], fv[3]));
#pragma unroll
            for (int pp = 0; pp < P; pp++) {
              float pv[4];
#pragma unroll
              for (int r = 0; r < 4; r++) {
                if (pp == 2) pv[r] = fv[r] * th[tb][r];
                else if ((pp == 0) == (OP == LINK_OP_SIN)) pv[r] = fv[r] * sn[tb][r];
                else pv[r] = fv[r] * cs[tb][r];
              }
              // lane 0 of the row takes the partial sum the previous tile left open (cin = 0 otherwise; stale LDS content
              // is finite: the buffer is zeroed at kernel start), the scan spreads it over the cell's lanes
              pv[0] = fmaf(cv[pp].x, cin, pv[0]); pv[1] = fmaf(cv[pp].y, cin, pv[1]);
              pv[2] = fmaf(cv[pp].z, cin, pv[2]); pv[3] = fmaf(cv[pp].w, cin, pv[3]);
              if (!(DC_T2_ABL & 4)) {
                DC_SCAN4("row_shr:1", m1, pv[0], pv[1], pv[2], pv[3]);
                DC_SCAN4("row_shr:2", m2, pv[0], pv[1], pv[2], pv[3]);
                if (step4) DC_SCAN4("row_shr:4", m4, pv[0], pv[1], pv[2], pv[3]);
                if (step8) DC_SCAN4("row_shr:8", m8, pv[0], pv[1], pv[2], pv[3]);
              }
              st16(r_S, srow == DC_OOB ? DC_OOB : srow + (uint32_t)(pp * C * 4 + 64 * tp), make_float4(pv[0], pv[1], pv[2], pv[3]));
              if (cont_next && li == 15)                // the open cell's partial sums wait in LDS for the next tile
                *reinterpret_cast<float4 *>(carry + (pp * 4 + gq) * NV + 4 * tp) = make_float4(pv[0], pv[1], pv[2], pv[3]);
            }
          }
          if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_scan += tqb - tqt; tq_tiles++; }
          if (cont_next) {                              // wave-uniform
            const int seg15 = __builtin_amdgcn_readlane(seg, 15);
            rank_carry = 16 - seg15 + ((cont_prev && seg15 == 0) ? rank_carry : 0);
          }
          cont_prev = cont_next;
          key = key_n;
          pcell = pcell_n;
          crd = crd_n;
          __builtin_amdgcn_wave_barrier();
        };
        uint32_t key0 = 0;
        int pcell0 = 0;
        float4 f0[T];
        int4 crd0 = make_int4(0, 0, 0, 0);
        ld_rows(0, key0, pcell0, f0, crd0);
        for (int t = 0; t < ntile; t++) tile(t, key0, pcell0, f0, crd0);
      }
      if (DC_T2_X3 || !hm) break;
      // ---- a heavy cell (more than 8 voxels): its ids -> LDS, rank = number of smaller ids, list = the cell alone ----
      hcell = __builtin_ctzll(hm);
      hm &= hm - 1;
      const int hc = __builtin_amdgcn_readlane(cn, hcell), hpc = __builtin_amdgcn_readlane(pc, hcell);
      __builtin_amdgcn_wave_barrier();
      for (int e = lane; e < hc; e += 64) raw[e] = sid[dc_sid_off(g, hpc, e)] & DC_ID_MASK;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      for (int e = lane; e < hc; e += 64) {
        const uint32_t mine = raw[e];
        int rk = 0;
        for (int j = 0; j < hc; j++) rk += raw[j] < mine ? 1 : 0;
        list[rk] = make_uint2(((uint32_t)hcell << DC_ID_BITS) | mine, (uint32_t)hpc);
      }
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ptotal = hc;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (dbg && lane == 0) {
    unsigned long long *d = dbg + (size_t)wid * 8;
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    d[0] = tq1 - tq0; d[1] = tq_chunk; d[2] = tq_mfma; d[3] = tq_valu; d[4] = tq_scan; d[5] = te - tq0; d[6] = tq_tiles; d[7] = tq0;
  }
}

template <int C, int OP, int NB>
static int launch_t2(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     bool warm, hipStream_t st) {
  using K = dc_t2_cfg<C, OP>;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  const int64_t waves = (int64_t)(b->tune.k1_wgs > 0 ? b->tune.k1_wgs : 512) * K::NW;
  int cpw = (int)((vi + waves - 1) / waves);
  if (cpw < 1) cpw = 1;
  const int64_t wgs = (vi + (int64_t)cpw * K::NW - 1) / ((int64_t)cpw * K::NW);
  int pad = b->tune.k1_lds_pad;
  pad = pad < 0 ? 0 : (pad > 65536 ? 65536 : pad);
  const int lds = K::LDS_BYTES + pad <= 160 * 1024 ? K::LDS_BYTES + pad : K::LDS_BYTES;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_tiles<C, OP, NB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((k_dc_tiles<C, OP, NB>), dim3((unsigned)wgs), dim3(64 * K::NW), lds, st, b->feats,
                     reinterpret_cast<const int4 *>(b->coords), b->sid, reinterpret_cast<int4 *>(b->slots), b->cnt,
                     b->cell_n, b->w_pre, b->pre_ln_w, b->pre_ln_b, b->w_pos, b->alpha, d.cg, d.coord_div, d.eps, n, g, cpw,
                     warm, b->S, b->fin, b->hdr, reinterpret_cast<unsigned long long *>(b->tune.k1_dbg));
  return check_launch("link_dc_premix_modsum");
}

template <int C, int OP>
static int dispatch_t2_nb(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                          bool warm, hipStream_t st) {
  constexpr int T = C / 16;
  int nb = (d.cg % 16 == 0) ? d.cg / 16 : T;
  if (nb > T) nb = T;
  if (T >= 2 && nb == T / 2) return launch_t2<C, OP, (T >= 2 ? T / 2 : 1)>(b, g, d, n, warm, st);
  if (T >= 4 && nb == T / 4) return launch_t2<C, OP, (T >= 4 ? T / 4 : 1)>(b, g, d, n, warm, st);
  return launch_t2<C, OP, T>(b, g, d, n, warm, st);    // any other grouping: every block evaluates its own theta
}

template <int C>
static int dispatch_t2_op(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                          bool warm, hipStream_t st) {
  switch (d.op) {
    case LINK_OP_COS: return dispatch_t2_nb<C, LINK_OP_COS>(b, g, d, n, warm, st);
    case LINK_OP_SIN: return dispatch_t2_nb<C, LINK_OP_SIN>(b, g, d, n, warm, st);
    default: return dispatch_t2_nb<C, LINK_OP_COSX>(b, g, d, n, warm, st);
  }
}

int run_tiles_modsum(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n, bool warm,
                     hipStream_t st) {
  switch (d.c) {
    case 16: return dispatch_t2_op<16>(b, g, d, n, warm, st);
    case 32: return dispatch_t2_op<32>(b, g, d, n, warm, st);
    default: return dispatch_t2_op<64>(b, g, d, n, warm, st);
  }
}

}  // namespace DC_IO_NS
