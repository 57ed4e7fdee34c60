// link_amd/csrc/elk_tiles_impl.h -- tile form of R_core on the GENERAL layout (round 3; include/link_amd.h section C,
// link_elk_premix_modsum_tiles / link_elk_gather_demod_tiles): two launches on a built block index instead of four, for the
// frames the dense-cell layout does not take -- LiDAR-shaped block grids (1-2 % of the cells occupied, 5-50 voxels per
// occupied block, a few hundred in the heaviest), any of the widths C = 16 / 32 / 64 / 128 of the detection backbone
// (reference call sites: linkunet.py:345-363, scn.py:586-607).
//
//   k_elk_tiles       pre_mix (matrix cores, fp16 hi | lo split of both operands) + LayerNorm + theta + sincos + modulate
//                     + per-block sums -> block table S, in ONE pass over the voxels in block order: a wave takes 16 sorted
//                     positions per tile in the accumulator layout of v_mfma_f32_16x16x32_f16 (a DPP row = the tile's 16
//                     voxels, 4 channels each), so the sum over a block's voxels is a segmented scan along the row (as in
//                     the dense-cell tile form, dense_tiles_impl.h).  Big blocks are split: a workgroup owns a range of
//                     sorted positions cut at block boundaries and deals its tiles to its 4 waves; a block that crosses a
//                     wave boundary leaves partial rows (at most 2 per wave) that one wave adds up in wave order after
//                     the workgroup barrier -- fixed order, no atomics, bitwise reproducible.
//   k_elk_gather_tiles  r^3 neighbour sum + count normalisation + de-modulate + LayerNorm, a wave per tile of 16-64
//                     sorted positions: the neighbour rows of a block are fetched once per (block, tile) -- a big block
//                     is shared by as many waves as it has tiles instead of being one lane group's serial loop.
// Replaces k_premix_ln_tlp + k_modulate_sum_g and k_block_gather_g + k_voxel_demod_ln_g (elk.hip) on those frames; the
// fin matrix exists only for cos_x (its de-modulation reads it, linkunet.py:176).
#pragma once
#include "tile_common.h"

#ifdef ELK_T_DBG          /* profiling builds only (tools/lidar_prof.py): per-wave s_memtime phases in a device array */
__device__ unsigned long long elk_t_dbg[8 * 32768];
__device__ unsigned long long elk_g_dbg[8 * 32768];
#define ELK_T_TICK(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#else
#define ELK_T_TICK(v)
#endif

namespace DC_IO_NS {
using namespace link;

template <int C, int OP>
struct elk_t_cfg {
  static constexpr int T = C / 16;
  static constexpr int P = op_parts<OP>::value;
  static constexpr int RB = P * C * 4;                 // bytes of one S row
  static constexpr int RGL = P * C / 4;                // 16-byte pieces of one row
  static constexpr int NW = 4;
  static constexpr int CARRY_BYTES = P * C * 4;        // open block at a tile boundary: 4 lane groups x P parts x C/4 values
  static constexpr int FLAG_OFF = dc_wimg<C>::W_BYTES + NW * CARRY_BYTES;
  static constexpr int LDS_BYTES = FLAG_OFF + 2 * NW * 4;
  // registers: C = 128 holds 8 accumulator blocks + 8 row pieces + their fp16 splits: one wave per SIMD (the unified file of 512)
  static constexpr int WAVES = C <= 64 ? 3 : 1;
};

// byte offset of the partial rows behind the table: rows [0, m_cap], then the m_cap + 1 counts, 16-byte aligned
__host__ __device__ inline int64_t elk_t_part_off(int64_t m_cap, int rs) { return (((m_cap + 1) * (int64_t)(rs + 1) * 4) + 15) & ~(int64_t)15; }

template <int C, int OP, int NB>
__global__ void __launch_bounds__(256, (elk_t_cfg<C, OP>::WAVES)) k_elk_tiles(
    const void *__restrict__ feats, const int4 *__restrict__ vox_sorted, const int32_t *__restrict__ pos_blk,
    const int32_t *__restrict__ blk_start, const int32_t *__restrict__ hdr, const float *__restrict__ w_pre,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, const float *__restrict__ w_pos,
    const float *__restrict__ alpha, int cg, float coord_div, float eps, int64_t n, int64_t m_cap, int span,
    float *__restrict__ S_, uint32_t s_bytes, uint32_t part_off, float *__restrict__ fin) {
  using K = elk_t_cfg<C, OP>;
  constexpr int T = K::T, P = K::P, NV = 4 * T;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *ln_lds = reinterpret_cast<float *>(smem_raw + dc_wimg<C>::WIMG_BYTES);
  float *pw_lds = ln_lds + 2 * C;
  int *flags = reinterpret_cast<int *>(smem_raw + K::FLAG_OFF);       // per wave: bit0 tail open, bit1 through; [NW + w]: block of the tail
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  ELK_T_TICK(tq0);
  float *carry = reinterpret_cast<float *>(smem_raw + dc_wimg<C>::W_BYTES + wave * K::CARRY_BYTES);
  const int nv = hdr[LINK_HDR_NVALID];
  const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, s_bytes);
  if (blockIdx.x == 0 && tid < K::RGL + 1) {           // the all-zero row absent neighbours point at, and its count
    if (tid < K::RGL) st16(r_S, (uint32_t)m_cap * (uint32_t)K::RB + (uint32_t)tid * 16u, make_float4(0.f, 0.f, 0.f, 0.f));
    else st4i(r_S, (uint32_t)(m_cap + 1) * (uint32_t)K::RB + (uint32_t)m_cap * 4u, 0);
  }
  // ---- this workgroup's range of sorted positions: the nominal span, both ends moved up to the next block boundary.
  // Branch-free (clamped buffer loads + selects): the three dependent round trips (header -> block of the position -> that
  // block's end) run while the weights are on their way, there is no branch for the compiler to wait in front of ----
  const __amdgpu_buffer_rsrc_t r_pb = dc_rsrc(pos_blk, (uint32_t)(n * 4));
  const __amdgpu_buffer_rsrc_t r_bs = dc_rsrc(blk_start, (uint32_t)((n + 1) * 4));
  auto boundary = [&](int64_t q) -> int {
    const int qc = q < 1 ? 1 : (q >= nv ? (nv > 1 ? nv - 1 : 1) : (int)q);
    const int bp = __builtin_amdgcn_raw_buffer_load_b32(r_pb, (uint32_t)(qc - 1) * 4u, 0, 0);
    const int bq = __builtin_amdgcn_raw_buffer_load_b32(r_pb, (uint32_t)qc * 4u, 0, 0);      // beyond the array: 0
    const int nx = __builtin_amdgcn_raw_buffer_load_b32(r_bs, (uint32_t)(bq + 1) * 4u, 0, 0);
    return q <= 0 ? 0 : (q >= nv ? nv : (bp != bq ? (int)q : nx));
  };
  const int a = boundary((int64_t)blockIdx.x * span), e = boundary((int64_t)(blockIdx.x + 1) * span);
  ELK_T_TICK(tq1);
  const int tpw = (((e - a + 15) >> 4) + K::NW - 1) / K::NW;        // tiles per wave
  const int wa = a + wave * tpw * 16;
  const int wb = (wa + tpw * 16 < e) ? wa + tpw * 16 : e;
  const bool has = wa < wb;                            // false for every wave of a workgroup whose range is empty (a >= e)
  const int wac = has ? wa : 1, wbc = has ? wb : 1;
  const bool head_open = has && wa > a && __builtin_amdgcn_raw_buffer_load_b32(r_pb, (uint32_t)(wac - 1) * 4u, 0, 0) ==
                                              __builtin_amdgcn_raw_buffer_load_b32(r_pb, (uint32_t)wac * 4u, 0, 0);
  const bool tail_open = has && wb < e && __builtin_amdgcn_raw_buffer_load_b32(r_pb, (uint32_t)(wbc - 1) * 4u, 0, 0) ==
                                              __builtin_amdgcn_raw_buffer_load_b32(r_pb, (uint32_t)wbc * 4u, 0, 0);
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * C * 4));     // written for cos_x only
  const __amdgpu_buffer_rsrc_t r_feats = dc_rsrc(feats, (uint32_t)(n * C * IO_BYTES));
  const __amdgpu_buffer_rsrc_t r_vox = dc_rsrc(vox_sorted, (uint32_t)(n * 16));
  // records of the first two tiles are on their way while W is staged
  auto ld_rec = [&](int t, int4 &rec, int &blk) {
    int pos = wa + 16 * t + li;
    pos = pos < wb ? pos : wb - 1;
    pos = has ? pos : 0;
    const v4i_t r = __builtin_amdgcn_raw_buffer_load_b128(r_vox, (uint32_t)pos * 16u, 0, 0);
    rec = make_int4(r.x, r.y, r.z, r.w);
    blk = __builtin_amdgcn_raw_buffer_load_b32(r_pb, (uint32_t)pos * 4u, 0, 0);
  };
  int4 rec0, rec1;
  int blk0, blk1;
  ld_rec(0, rec0, blk0);
  ld_rec(1, rec1, blk1);
  bool w_big = dc_stage_weights<C, 64 * K::NW, LINK_TILE_EXACT(OP)>(smem_raw, w_pre, ln_w, ln_b, w_pos, alpha, cg, tid);
  for (int i = lane; i < K::CARRY_BYTES / 4; i += 64) carry[i] = 0.f;     // read unconditionally by every tile (times 0 unless a block straddles)
  w_big = __syncthreads_or(w_big) != 0 || (LINK_COSX_EXACT && OP == LINK_OP_COSX);   // cos_x: exact contraction (elk_common.h)
  if (a >= e) return;                                  // nominal span inside one block that an earlier workgroup owns
  ELK_T_TICK(tq2);
#ifdef ELK_T_DBG
  unsigned long long tq_mfma = 0, tq_ln = 0, tq_scan = 0, tq_rows = 0;
#endif
  const unsigned short *wh = reinterpret_cast<const unsigned short *>(smem_raw);
  const uint32_t cnt_off = (uint32_t)(m_cap + 1) * (uint32_t)K::RB;
  const uint32_t my_part = part_off + (uint32_t)((blockIdx.x * K::NW + wave) * 2) * (uint32_t)K::RB;
  bool through = false;

  if (has) {
    const int ntile = (wb - wa + 15) >> 4;
    bool cont_prev = false;                             // the tile's first voxels continue the block the previous tile ended in
    int rank_carry = 0;                                 // ... and this many of that block's voxels came before this tile (in this wave)
    bool head_alive = true;                             // the run that began at the wave's first position is still open
    auto ld_rows = [&](const int4 &rec, float4 (&ff)[T]) {
      const uint32_t ro = ((uint32_t)rec.w * (uint32_t)C + (uint32_t)(4 * gq)) * (uint32_t)IO_BYTES;
      ff[0] = io_ldb4<0>(r_feats, ro);
      if constexpr (T > 1) ff[1] = io_ldb4<16>(r_feats, ro);
      if constexpr (T > 2) { ff[2] = io_ldb4<32>(r_feats, ro); ff[3] = io_ldb4<48>(r_feats, ro); }
      if constexpr (T > 4) { ff[4] = io_ldb4<64>(r_feats, ro); ff[5] = io_ldb4<80>(r_feats, ro); ff[6] = io_ldb4<96>(r_feats, ro); ff[7] = io_ldb4<112>(r_feats, ro); }
    };
    float4 ff[T];
    ld_rows(rec0, ff);
    for (int t = 0; t < ntile; t++) {
      const int pos = wa + 16 * t + li;
      const bool valid = pos < wb;
      const int4 rec = rec0;
      const int blk = blk0;
      floatx4 ac[T];
#ifdef ELK_T_DBG
      ELK_T_TICK(ta);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ELK_T_TICK(tb);
      tq_rows += tb - ta;
#endif
      dc_premix_tile<C, LINK_TILE_EXACT(OP)>(wh, w_pre, w_big, li, gq, ff, ac);
#ifdef ELK_T_DBG
      asm volatile("s_nop 0" ::"v"(ac[0][0]));
      ELK_T_TICK(tc);
      tq_mfma += tc - tb;
#endif
      // the rows are dead once the matrix cores have them: the next tile's rows are requested into the same registers (their
      // record arrived a tile ago) and the records of the tile after that, and land while this tile's LayerNorm / sincos /
      // scans run
      ld_rows(rec1, ff);
      const int blk_next0 = __builtin_amdgcn_readlane(blk1, 0);
      rec0 = rec1;
      blk0 = blk1;
      ld_rec(t + 2, rec1, blk1);
      float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
      if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
      // theta / sincos of the NB distinct 16-channel blocks (channel ch uses theta[ch % cg]); with NB == T each block's
      // values are formed where the block is worked on instead of being held for all of them
      float th[NB][4], sn[NB][4], cs[NB][4];
      auto trig = [&](int tb, float (&th_)[4], float (&sn_)[4], float (&cs_)[4]) {
        const float4 q0 = *reinterpret_cast<const float4 *>(&pw_lds[16 * tb + 4 * gq]);
        const float4 q1 = *reinterpret_cast<const float4 *>(&pw_lds[C + 16 * tb + 4 * gq]);
        const float4 q2 = *reinterpret_cast<const float4 *>(&pw_lds[2 * C + 16 * tb + 4 * gq]);
        const float4 qa = *reinterpret_cast<const float4 *>(&pw_lds[3 * C + 16 * tb + 4 * gq]);
        th_[0] = theta_of(x, y, z, q0.x, q1.x, q2.x, qa.x); th_[1] = theta_of(x, y, z, q0.y, q1.y, q2.y, qa.y);
        th_[2] = theta_of(x, y, z, q0.z, q1.z, q2.z, qa.z); th_[3] = theta_of(x, y, z, q0.w, q1.w, q2.w, qa.w);
        bool big = false;
#pragma unroll
        for (int r = 0; r < 4; r++) big |= !(fabsf(th_[r]) < 32768.0f);
        if (__builtin_expect(__any(big), 0)) {           // never on sane inputs
#pragma unroll
          for (int r = 0; r < 4; r++) sincos_nocall(th_[r], sn_[r], cs_[r]);
        } else {
#pragma unroll
          for (int r = 0; r < 4; r++) sincos_small(th_[r], sn_[r], cs_[r]);
        }
      };
      if constexpr (NB < T) {
#pragma unroll
        for (int tb = 0; tb < NB; tb++) trig(tb, th[tb], sn[tb], cs[tb]);
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
#ifdef ELK_T_DBG
      asm volatile("s_nop 0" ::"v"(rstd));
      ELK_T_TICK(td);
      tq_ln += td - tc;
#endif
      // ---- segments of the DPP row: positions of one block are adjacent; lanes beyond the wave's range form segments of
      // their own that are never stored ----
      const int ckey = valid ? blk : 0x40000000 + li;
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
      const bool step4 = __any(soff >= 4), step8 = __any(soff >= 8);
      const float cin = (cont_prev && li == 0) ? 1.0f : 0.0f;
      const int nrun = soff + 1 + ((cont_prev && seg == 0) ? rank_carry : 0);   // voxels of the run up to and including this lane
      // does the block of the tile's last voxel go on in this wave's next tile?
      const bool cont_next = t + 1 < ntile && blk_next0 == __builtin_amdgcn_readlane(blk, 15);
      const bool last_pos = pos == wb - 1;
      const bool closes = valid && (li == 15 || last_pos ? !cont_next : nextk != ckey);
      // where the closing lane's sums go: the block's row -- or, for the run that came in from the previous wave / goes on
      // in the next one, this wave's partial rows behind the table
      const bool in_head = head_alive && seg == 0;
      const bool to_head = in_head && head_open, to_tail = last_pos && tail_open;
      const uint32_t srow = !closes ? DC_OOB
                            : (to_head ? my_part : (to_tail ? my_part + (uint32_t)K::RB : (uint32_t)blk * (uint32_t)K::RB)) + (uint32_t)(16 * gq);
      st4i(r_S, (closes && !to_head && !to_tail && gq == 0) ? cnt_off + (uint32_t)blk * 4u : DC_OOB, __float_as_int((float)nrun));
      if (t + 1 == ntile) through = __any(last_pos && to_head && tail_open);
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
        float th1[4], sn1[4], cs1[4];
        if constexpr (NB == T) trig(tp, th1, sn1, cs1);
        const float fv[4] = {(ac[tp][0] - mean) * rstd * lw.x + lb.x, (ac[tp][1] - mean) * rstd * lw.y + lb.y,
                             (ac[tp][2] - mean) * rstd * lw.z + lb.z, (ac[tp][3] - mean) * rstd * lw.w + lb.w};
        if (OP == LINK_OP_COSX)                       // the de-modulation of cos_x needs fin (linkunet.py:176)
          st16(r_fin, valid ? (uint32_t)rec.w * (uint32_t)(C * 4) + (uint32_t)((16 * tp + 4 * gq) * 4) : DC_OOB,
               make_float4(fv[0], fv[1], fv[2], fv[3]));
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          float pv[4];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float th_ = NB == T ? th1[r] : th[tb][r], sn_ = NB == T ? sn1[r] : sn[tb][r], cs_ = NB == T ? cs1[r] : cs[tb][r];
            if (pp == 2) pv[r] = fv[r] * th_;
            else if ((pp == 0) == (OP == LINK_OP_SIN)) pv[r] = fv[r] * sn_;
            else pv[r] = fv[r] * cs_;
          }
          // lane 0 of the row takes the partial sum the previous tile left open (cin = 0 otherwise; the buffer is zeroed at
          // kernel start, so what it multiplies is finite), the scan spreads it over the block's lanes
          pv[0] = fmaf(cv[pp].x, cin, pv[0]); pv[1] = fmaf(cv[pp].y, cin, pv[1]);
          pv[2] = fmaf(cv[pp].z, cin, pv[2]); pv[3] = fmaf(cv[pp].w, cin, pv[3]);
          DC_SCAN4("row_shr:1", m1, pv[0], pv[1], pv[2], pv[3]);
          DC_SCAN4("row_shr:2", m2, pv[0], pv[1], pv[2], pv[3]);
          if (step4) DC_SCAN4("row_shr:4", m4, pv[0], pv[1], pv[2], pv[3]);
          if (step8) DC_SCAN4("row_shr:8", m8, pv[0], pv[1], pv[2], pv[3]);
          st16(r_S, srow == DC_OOB ? DC_OOB : srow + (uint32_t)(pp * C * 4 + 64 * tp), make_float4(pv[0], pv[1], pv[2], pv[3]));
          if (cont_next && li == 15)                  // the open block's partial sums wait in LDS for the next tile
            *reinterpret_cast<float4 *>(carry + (pp * 4 + gq) * NV + 4 * tp) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        }
      }
      const int seg15 = __builtin_amdgcn_readlane(seg, 15);
      if (cont_next) rank_carry = 16 - seg15 + ((cont_prev && seg15 == 0) ? rank_carry : 0);
      head_alive = head_alive && cont_next && seg15 == 0;
      cont_prev = cont_next;
      __builtin_amdgcn_wave_barrier();
#ifdef ELK_T_DBG
      ELK_T_TICK(te);
      tq_scan += te - td;
#endif
    }
  }
  ELK_T_TICK(tq3);
  // ---- blocks that cross wave boundaries: the wave such a block STARTS in adds the partial rows in wave order ----
  if (lane == 0) {
    flags[wave] = (tail_open ? 1 : 0) | (through ? 2 : 0);
    flags[K::NW + wave] = tail_open ? pos_blk[wb - 1] : 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the partial rows have left this wave
  __syncthreads();
  if (tail_open && !through) {
    const int b = flags[K::NW + wave];
    int last = wave + 1;                                  // the wave the block ends in
    while (flags[last] & 2) last++;
    for (int pc = lane; pc < K::RGL; pc += 64) {
      const uint32_t po = (uint32_t)pc * 16u;
      v4i_t v = __builtin_amdgcn_raw_buffer_load_b128(r_S, my_part + (uint32_t)K::RB + po, 0, 17);    // sc0 sc1: not from this CU's L1
      float4 acc = make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
      for (int j = wave + 1; j <= last; j++) {
        v = __builtin_amdgcn_raw_buffer_load_b128(r_S, part_off + (uint32_t)((blockIdx.x * K::NW + j) * 2) * (uint32_t)K::RB + po, 0, 17);
        acc.x += __int_as_float(v.x); acc.y += __int_as_float(v.y); acc.z += __int_as_float(v.z); acc.w += __int_as_float(v.w);
      }
      st16(r_S, (uint32_t)b * (uint32_t)K::RB + po, acc);
    }
    if (lane == 0) st4i(r_S, cnt_off + (uint32_t)b * 4u, __float_as_int((float)(blk_start[b + 1] - blk_start[b])));
  }
#ifdef ELK_T_DBG
  if (lane == 0 && blockIdx.x * K::NW + wave < 32768) {
    unsigned long long *d = elk_t_dbg + (size_t)(blockIdx.x * K::NW + wave) * 8;
    const unsigned long long tq4 = __builtin_amdgcn_s_memtime();
    d[0] = tq1 - tq0; d[1] = tq2 - tq1; d[2] = tq_rows; d[3] = tq_mfma; d[4] = tq_ln; d[5] = tq_scan; d[6] = tq4 - tq3; d[7] = tq4 - tq0;
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// neighbour sum + de-modulate + LayerNorm over tiles of sorted positions
// ---------------------------------------------------------------------------------------------
template <int C, int OP, int R>
struct elk_g_cfg {
  static constexpr int LPR = C / 4;                    // lanes of one feature row (16 bytes each)
  static constexpr int G = 64 / LPR;                   // lane groups of a wave
  static constexpr int P = op_parts<OP>::value;
  static constexpr int RS = P * C;                     // floats of one table row
  static constexpr int R3 = R * R * R;
#ifndef ELK_G_STEPS
#define ELK_G_STEPS 4     /* 8: 25.7 against 19.5 us on 59k voxels (C = 64, r = 2), 12.0 against 8.2 us on 3k: the slowest tile sets the time */
#endif
  static constexpr int WP = ELK_G_STEPS * G < 64 ? ELK_G_STEPS * G : 64;   // sorted positions per wave: 4 voxel steps per group
  static constexpr int STEPS = WP / G;
  static constexpr int RMAX = WP / 2 < 16 ? WP / 2 : 16;   // blocks whose neighbour sums sit in LDS at a time
  static constexpr int NW = 4;
  static constexpr int A_BYTES = RMAX * RS * 4;
  static constexpr int NB_BYTES = (RMAX * R3 * 4 + 15) & ~15;
  static constexpr int REC_BYTES = WP * 16;
  static constexpr int WAVE_BYTES = A_BYTES + NB_BYTES + REC_BYTES + RMAX * 4;
  static constexpr int LDS_BYTES = NW * WAVE_BYTES;
};

// A wave takes WP consecutive sorted positions.  The distinct blocks among them (run heads: positions are block-major)
// are resolved to neighbour ids by all 64 lanes (cell arithmetic + cell_blk: one round trip for up to RMAX blocks), their
// normalised neighbour sums A = sum of the r^3 neighbour rows / summed count are formed by the lane groups (a group per
// block, all r^2 rows of a plane in flight) and left in LDS; then the groups deal the positions among themselves: record,
// A row of the position's block, theta, sincos, de-modulation in separate IEEE mul / add like the reference's eager ops
// (linkunet.py:148,162,176), LayerNorm (DPP reductions inside the group), one 16-byte store per lane.
template <int C, int OP, int R>
__global__ void __launch_bounds__(256) k_elk_gather_tiles(
    const float *__restrict__ S, const float *__restrict__ fin, const int4 *__restrict__ vox_sorted,
    const int32_t *__restrict__ pos_blk, const int4 *__restrict__ blk_coords, const int32_t *__restrict__ cell_blk,
    link_grid_t g, const int32_t *__restrict__ hdr, const float *__restrict__ w_pos, const float *__restrict__ alpha,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, int cg, float coord_div, float eps, int64_t m_cap,
    void *__restrict__ out) {
  using K = elk_g_cfg<C, OP, R>;
  constexpr int LPR = K::LPR, G = K::G, P = K::P, RS = K::RS, R2 = R * R, R3 = K::R3, WP = K::WP;
  constexpr int LO = -((R + 1) / 2) + 1;               // nn/utils/kernel.py:21
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & (LPR - 1), grp = lane / LPR;
  ELK_T_TICK(tg0);
#ifdef ELK_T_DBG
  unsigned long long tg_nb = 0, tg_a = 0, tg_v = 0;
#endif
  char *wbase = smem_raw + wave * K::WAVE_BYTES;
  float *A_lds = reinterpret_cast<float *>(wbase);
  int32_t *nb_lds = reinterpret_cast<int32_t *>(wbase + K::A_BYTES);
  int4 *rec_lds = reinterpret_cast<int4 *>(wbase + K::A_BYTES + K::NB_BYTES);
  int32_t *run_blk = reinterpret_cast<int32_t *>(wbase + K::A_BYTES + K::NB_BYTES + K::REC_BYTES);
  const int nv = hdr[LINK_HDR_NVALID];
  // workgroup w runs on XCD w % 8: each XCD takes a contiguous eighth of the sorted positions (blocks are in cell order, so
  // an x-slab of the grid), and its L2 holds that slab's table rows instead of every XCD cycling the whole table
  const int64_t per_xcd = (int64_t)gridDim.x >> 3;      // the launcher rounds the grid up to a multiple of 8
  const int64_t wg = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t p0 = (wg * K::NW + wave) * WP;
  if (p0 >= nv) return;                                // wave-uniform; nothing below is a workgroup barrier
  const int npos = nv - p0 < WP ? (int)(nv - p0) : WP;
  // ---- positions of the tile: block id, record; run heads ----
  const bool valid = lane < npos;
  const int64_t cl = p0 + (valid ? lane : npos - 1);
  const int blk = pos_blk[cl];
  rec_lds[lane < WP ? lane : 0] = vox_sorted[p0 + (lane < npos ? lane : 0)];
  const int prev = __shfl_up(blk, 1, 64);
  const bool head = valid && (lane == 0 || blk != prev);
  const unsigned long long hm = __ballot(head);
  const int nruns = __popcll(hm);
  // parameters of this lane's four channels
  const int ch0 = 4 * li;
  float w0[4], w1[4], w2[4], al[4];
#pragma unroll
  for (int e_ = 0; e_ < 4; e_++) {
    const int tc = (ch0 + e_) % cg;
    w0[e_] = w_pos[3 * tc + 0]; w1[e_] = w_pos[3 * tc + 1]; w2[e_] = w_pos[3 * tc + 2];
    al[e_] = alpha ? alpha[tc] : 1.0f;
  }
  const float4 gw = *reinterpret_cast<const float4 *>(&ln_w[ch0]), gb = *reinterpret_cast<const float4 *>(&ln_b[ch0]);
  const float *__restrict__ Scnt = S + (m_cap + 1) * RS;
  const int my_run = __popcll(hm & ((2ull << lane) - 1ull)) - 1;      // run of this lane's position
  // cos_x: the fin rows of this group's positions (grp, grp + G, ...) are requested now -- they come from HBM (the kernel
  // before wrote them through) and land while the neighbour sums are formed
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float4 fpre[OP == LINK_OP_COSX ? K::STEPS : 1];
  if (OP == LINK_OP_COSX) {
#pragma unroll
    for (int i = 0; i < K::STEPS; i++) {
      const int l = grp + i * G;
      const int id = rec_lds[l < npos ? l : npos - 1].w;
      fpre[i] = *reinterpret_cast<const float4 *>(&fin[(int64_t)id * C + ch0]);
    }
  }

#ifdef ELK_T_DBG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  ELK_T_TICK(tg1);
  for (int rb = 0; rb < nruns; rb += K::RMAX) {
    const int nr = nruns - rb < K::RMAX ? nruns - rb : K::RMAX;
    ELK_T_TICK(tpa);
    if (head && my_run >= rb && my_run < rb + nr) run_blk[my_run - rb] = blk;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // neighbour ids of the pass's blocks; an absent neighbour points at the all-zero row (id m_cap, count 0)
    for (int e_ = lane; e_ < nr * R3; e_ += 64) {
      const int j = e_ / R3, k = e_ - j * R3;
      const int dz = k / R2, t = k - dz * R2;
      const int4 bc = blk_coords[run_blk[j]];
      const int32_t cell = cell_of(g, bc.x + LO + t % R, bc.y + LO + t / R, bc.z + LO + dz, bc.w);
      const int nb = cell >= 0 ? cell_blk[cell] - 1 : -1;
      nb_lds[e_] = nb >= 0 ? nb : (int)m_cap;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    ELK_T_TICK(tpb);
    // ---- A rows: a group per block, a plane of r^2 rows in flight ----
    for (int j = grp; j < nr; j += G) {
      float acc[P][4], den = 0.f;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[pp][q] = 0.f;
      // rows in flight per batch: as many of the r^3 as ~28 16-byte registers hold (r = 2: all 8; r = 3: 14 + 13 with two
      // parts, 3 x 9 with three)
      constexpr int NBAT = (R3 * P + 27) / 28, BR = (R3 + NBAT - 1) / NBAT;
#pragma unroll
      for (int bt = 0; bt < NBAT; bt++) {
        float4 v[BR][P];
        float vd[BR];
#pragma unroll
        for (int t = 0; t < BR; t++) {
          if (bt * BR + t < R3) {
            const int nb = nb_lds[j * R3 + bt * BR + t];
            const float *row = S + (int64_t)nb * RS + ch0;
            vd[t] = Scnt[nb];
#pragma unroll
            for (int pp = 0; pp < P; pp++) v[t][pp] = *reinterpret_cast<const float4 *>(&row[pp * C]);
          }
        }
#pragma unroll
        for (int t = 0; t < BR; t++) {
          if (bt * BR + t < R3) {
            den += vd[t];
#pragma unroll
            for (int pp = 0; pp < P; pp++) {
              acc[pp][0] += v[t][pp].x; acc[pp][1] += v[t][pp].y; acc[pp][2] += v[t][pp].z; acc[pp][3] += v[t][pp].w;
            }
          }
        }
      }
#pragma unroll
      for (int pp = 0; pp < P; pp++)                   // utils.py:80: the neighbourhood mean
        *reinterpret_cast<float4 *>(&A_lds[j * RS + pp * C + ch0]) =
            make_float4(acc[pp][0] / den, acc[pp][1] / den, acc[pp][2] / den, acc[pp][3] / den);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ELK_T_TICK(tpc);
    // ---- voxels of the pass: position grp, grp + G, ... ----
    auto step_ok = [&](int l, int &rl) {
      rl = __popcll(hm & ((2ull << l) - 1ull)) - 1;
      return l < npos && rl >= rb && rl < rb + nr;
    };
    int4 rec_n = rec_lds[grp];
#pragma unroll
    for (int i = 0; i < K::STEPS; i++) {
      const int l = grp + i * G;
      const int4 rec = rec_n;
      const float4 f4 = OP == LINK_OP_COSX ? fpre[i] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (i + 1 < K::STEPS) rec_n = rec_lds[l + G < npos ? l + G : npos - 1];
      int rl;
      const bool ok = step_ok(l, rl);
      const int ra = ok ? rl - rb : 0;
      float4 Av[P];
#pragma unroll
      for (int pp = 0; pp < P; pp++) Av[pp] = *reinterpret_cast<const float4 *>(&A_lds[ra * RS + pp * C + ch0]);
      float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
      if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
      float th[4], sn[4], cs[4];
      bool big = false;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        th[q] = theta_of(x, y, z, w0[q], w1[q], w2[q], al[q]);
        big |= !(fabsf(th[q]) < 32768.0f);
      }
      if (__builtin_expect(__any(big), 0)) {
#pragma unroll
        for (int q = 0; q < 4; q++) sincos_nocall(th[q], sn[q], cs[q]);
      } else {
#pragma unroll
        for (int q = 0; q < 4; q++) sincos_small(th[q], sn[q], cs[q]);
      }
      const float a0[4] = {Av[0].x, Av[0].y, Av[0].z, Av[0].w}, a1[4] = {Av[1].x, Av[1].y, Av[1].z, Av[1].w};
      const float a2[4] = {Av[P - 1].x, Av[P - 1].y, Av[P - 1].z, Av[P - 1].w};
      const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
      float nvv[4], s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float va;
        if (OP == LINK_OP_SIN) va = __fsub_rn(__fmul_rn(a0[q], cs[q]), __fmul_rn(a1[q], sn[q]));
        else va = __fadd_rn(__fmul_rn(a0[q], cs[q]), __fmul_rn(a1[q], sn[q]));
        if (OP == LINK_OP_COSX) va = __fadd_rn(va, __fsub_rn(a2[q], link_mul_rn(fv[q], th[q])));
        nvv[q] = va;
        s += va;
      }
      s = grp_sum<LPR>(s);
      const float mean = s * (1.0f / C);
      float qq = 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float d = nvv[q] - mean;
        qq += d * d;
      }
      qq = grp_sum<LPR>(qq);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + eps);
      if (ok) {
        const float4 o = make_float4((nvv[0] - mean) * rstd * gw.x + gb.x, (nvv[1] - mean) * rstd * gw.y + gb.y,
                                     (nvv[2] - mean) * rstd * gw.z + gb.z, (nvv[3] - mean) * rstd * gw.w + gb.w);
        io_st4_ptr(out, (int64_t)rec.w * C + ch0, o);
      }
    }
    __builtin_amdgcn_wave_barrier();
#ifdef ELK_T_DBG
    ELK_T_TICK(tpd);
    tg_nb += tpb - tpa; tg_a += tpc - tpb; tg_v += tpd - tpc;
#endif
  }
#ifdef ELK_T_DBG
  if (lane == 0 && wg * K::NW + wave < 32768) {
    unsigned long long *d = elk_g_dbg + (size_t)(wg * K::NW + wave) * 8;
    const unsigned long long tg2 = __builtin_amdgcn_s_memtime();
    d[0] = tg1 - tg0; d[1] = tg_nb; d[2] = tg_a; d[3] = tg_v; d[4] = nruns; d[5] = 0; d[6] = 0; d[7] = tg2 - tg0;
  }
#endif
}

}  // namespace DC_IO_NS
