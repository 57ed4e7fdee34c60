// link_amd/csrc/dense_fused_impl.h -- the fused dense-cell kernels, compiled once per feature I/O type:
// the including translation unit defines DC_IO (0 fp32, 1 fp16, 2 bf16) and DC_IO_NS.  Feature rows are read
// (feats) and written (out) in that type; everything in between -- the MFMA contraction, theta, the block
// table, LayerNorm statistics -- is fp32 (the reference's AMP contract: custom_fwd(cast_inputs=torch.half) on
// the voxelize / devoxelize ops, torchsparse/nn/functional/voxelize.py:13, devoxelize.py:54, with fp32
// accumulation; SURVEY.md section 8b "AMP").
#pragma once
#include <type_traits>

#include "dense_gather.h"
#include "dense_io.h"

namespace DC_IO_NS {
using namespace link;

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
#ifndef DC_K1_PRIO
#define DC_K1_PRIO 0
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


// Cell section of the sparse-cell kernels: the records of a chunk's cells into a block-major LDS list, cooperatively.
// LiDAR blocks hold 5-27 voxels: the dense form's per-cell-lane fetch (4 inline records + a serial loop of dependent loads for
// the rest, an insertion sort in LDS) took 40-90 us per wave there.  Here the cell lanes only lay out WHERE each list position
// comes from (scell[p] = its cell, sseg[p] = the cell's first position | count << 16); then every lane takes positions
// p = lane, lane + 64 and loads slot (cell, p - first) -- all records in one round trip -- and, when the slot order is still the
// insert's (SORT: rank order of the atomics), finds its place in the cell's id order by counting smaller ids in LDS and writes
// the record there (and back to the slot list, so that later readers find id order).  At most DC_SP_LCAP records per chunk.
#define DC_SP_LCAP 128
template <bool SORT>
__device__ __forceinline__ void dc_sparse_fetch(const link_dc_grid_t &g, __amdgpu_buffer_rsrc_t r_slots, int lane, bool mine, int pc,
                                                int nv, int excl, int Ttot, int4 *list, int *scell, int *sseg, int *tmp_id) {
  if (mine)
    for (int k = 0; k < nv; k++) { scell[excl + k] = pc; sseg[excl + k] = excl | (nv << 16); }
  __builtin_amdgcn_wave_barrier();
  int4 rec[DC_SP_LCAP / 64];
  int seg[DC_SP_LCAP / 64], pcj[DC_SP_LCAP / 64];
#pragma unroll
  for (int i = 0; i < DC_SP_LCAP / 64; i++) {
    const int p = lane + 64 * i;
    const bool on = p < Ttot;
    pcj[i] = on ? scell[p] : 0;
    seg[i] = on ? sseg[p] : 0;
    const v4i_t rv = __builtin_amdgcn_raw_buffer_load_b128(r_slots, on ? dc_slot(g, pcj[i], p - (seg[i] & 0xFFFF)) * 16u : DC_OOB, 0, 0);
    rec[i] = make_int4(rv.x, rv.y, rv.z, rv.w);
  }
  if constexpr (SORT) {
#pragma unroll
    for (int i = 0; i < DC_SP_LCAP / 64; i++) {
      const int p = lane + 64 * i;
      if (p < Ttot) tmp_id[p] = rec[i].w;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < DC_SP_LCAP / 64; i++) {
      const int p = lane + 64 * i;
      if (p < Ttot) {
        const int st = seg[i] & 0xFFFF, len = seg[i] >> 16;
        int r = 0;
        for (int q = 0; q < len; q++) r += tmp_id[st + q] < rec[i].w;
        list[st + r] = rec[i];
        st16i(r_slots, dc_slot(g, pcj[i], r) * 16u, rec[i]);      // id order goes back to the slot list
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < DC_SP_LCAP / 64; i++) {
      const int p = lane + 64 * i;
      if (p < Ttot) list[p] = rec[i];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

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
// SPARSE (round 4, the sparse-cell layout of LiDAR-shaped frames: dense_gather_sparse_impl.h): the cells a wave owns are
// not a range of the grid but the cells whose FIRST voxel (insert rank 0) has an id in the wave's range of voxel ids --
// `occ[i]` = that cell for voxel i, 0 otherwise (written by k_dc_index_sparse).  Every occupied cell is owned exactly once,
// no cell of the (mostly empty) grid is visited for nothing, no zero rows are written: absent neighbours are recognised by
// cell_n == 0 in the gather kernel.  Which wave owns a cell varies from run to run (the atomic ranks do); what it computes
// for the cell does not (records in id order, sums in that order).
template <int C, int OP, int NB, bool PIPE, bool SPARSE = false>
__global__ void __launch_bounds__(64 * DC_K1_NW, PIPE ? 2 : DC_K1_WAVES) k_dc_premix_modsum(
    const void *__restrict__ feats, int4 *__restrict__ slots, uint32_t *__restrict__ cnt,
    int32_t *__restrict__ cell_n, const float *__restrict__ w_pre, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, const float *__restrict__ w_pos, const float *__restrict__ alpha, int cg,
    float coord_div, float eps, int64_t n, link_dc_grid_t g, int cpw, bool warm, float *__restrict__ S_,
    float *__restrict__ fin, int32_t *__restrict__ hdr, unsigned long long *__restrict__ dbg,
    const int32_t *__restrict__ occ = nullptr) {
  using K = dc_k1_cfg<C, OP>;
  DC_PROF_PTR(dbg);
#if DC_K1_PRIO
  __builtin_amdgcn_s_setprio(DC_K1_PRIO);               // A/B (round 5): wave priority of the pre_mix kernel against a co-resident gather kernel
#endif
  // optional phase timing (tools/dcbench.py --phases): per wave 8 slots of s_memtime deltas
  unsigned long long tq0 = dbg ? __builtin_amdgcn_s_memtime() : 0, tq1 = 0, tq_cell = 0, tq_fill = 0, tq_body = 0, tq_sum = 0;
  int tq_tiles = 0;
  constexpr int T = K::T, P = K::P, LDW = K::LDW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  float *ln_lds = reinterpret_cast<float *>(smem_raw + K::WIMG_BYTES);
  float *pw_lds = ln_lds + 2 * C;                      // read per tile: 32 fewer live registers than per-lane copies
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  char *wbase = smem_raw + K::W_BYTES + wave * K::WAVE_BYTES;
  int4 *list = reinterpret_cast<int4 *>(wbase + K::LIST_OFF);
  int *scell = reinterpret_cast<int *>(wbase + K::SCELL_OFF);
  char *xbuf = wbase + K::X_OFF;
  // the first chunk's cell records and counts are requested BEFORE W is staged: the two latencies overlap
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int Vi = SPARSE ? (int)n : Dx * Dy * Dz * g.dim[3];
  const int wid = blockIdx.x * K::NW + wave;
  const int c_begin = wid * cpw;
  const int c_end = (c_begin + cpw < Vi) ? c_begin + cpw : Vi;
  const uint32_t *__restrict__ csrc = warm ? reinterpret_cast<const uint32_t *>(cell_n) : cnt;
  auto cell_of = [&](int chunk, int nrem) {
    const int q = chunk + (lane < nrem ? lane : 0);
    if constexpr (SPARSE) return (int)occ[q];          // 0 = this voxel is not the first of its cell: an idle lane (count 0)
    const int z = q % Dz;
    int t = q / Dz;
    const int y = t % Dy;
    t /= Dy;
    return dc_cell(g, t % Dx, y, z, t / Dx);
  };
  int pc_f = 0, nv_f = 0;
  int4 rf0 = make_int4(0, 0, 0, 0), rf1 = rf0, rf2 = rf0, rf3 = rf0;
  if (c_begin < c_end) {
    pc_f = cell_of(c_begin, (c_end - c_begin < 64) ? c_end - c_begin : 64);
    if constexpr (!SPARSE) {
      rf0 = slots[(int64_t)pc_f * DC_INL + 0]; rf1 = slots[(int64_t)pc_f * DC_INL + 1];
      rf2 = slots[(int64_t)pc_f * DC_INL + 2]; rf3 = slots[(int64_t)pc_f * DC_INL + 3];
    }
    nv_f = (int)csrc[pc_f];
  }
  bool w_big = false;                                  // a weight outside the fp16 split's range: fp32 contraction (never on sane models)
  bool th_big = false;
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
  if (blockIdx.x == 0 && tid == 0 && !warm) {          // publish the step's status word
    hdr[LINK_HDR_STATUS] = hdr[LINK_HDR_STATUS_ACC];
    hdr[LINK_HDR_STATUS_ACC] = 0;
  }
  w_big = DC_K1_SPLIT ? (__syncthreads_or(w_big) != 0 || (LINK_COSX_EXACT && OP == LINK_OP_COSX)) : (__syncthreads(), false);   // cos_x: exact contraction (elk_common.h)
  const bool th_slow = DC_THETA_BOUND ? __syncthreads_or(th_big) != 0 : false;     // workgroup-uniform (the same in every workgroup)
  if (dbg) tq1 = __builtin_amdgcn_s_memtime();
  if (c_begin >= c_end) return;
  const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, (uint32_t)((g.vp + 1) * K::RB));
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * C * 4));     // written for cos_x only
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_n = dc_rsrc(cell_n, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const int rl = lane;                                 // lane's 16-byte piece of an X / S row
  const bool ract = rl < K::RGL;

  for (int chunk = c_begin; chunk < c_end;) {
    const int nrem = (c_end - chunk < 64) ? c_end - chunk : 64;
    unsigned long long tqa = dbg ? __builtin_amdgcn_s_memtime() : 0;
    // ---- cell lanes: count, padded cell id, inline records (all requested before anything is consumed) ----
    int pc, nv;
    int4 r0, r1, r2, r3;
    if (chunk == c_begin) {                             // wave-uniform: the first chunk was requested before the staging
      pc = pc_f; nv = nv_f; r0 = rf0; r1 = rf1; r2 = rf2; r3 = rf3;
    } else {
      pc = cell_of(chunk, nrem);
      if constexpr (!SPARSE) {
        r0 = slots[(int64_t)pc * DC_INL + 0]; r1 = slots[(int64_t)pc * DC_INL + 1];
        r2 = slots[(int64_t)pc * DC_INL + 2]; r3 = slots[(int64_t)pc * DC_INL + 3];
      }
      nv = (int)csrc[pc];
    }
    constexpr int LCAPX = SPARSE ? DC_SP_LCAP : K::LCAP;
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
    if constexpr (SPARSE) {
      // cooperative fetch (dc_sparse_fetch): list positions [0, 128) of the wave's list, sseg behind scell's first 128 entries,
      // the id scratch behind the list's first 128 records
      int *sseg = scell + DC_SP_LCAP, *tmp_id = reinterpret_cast<int *>(list + DC_SP_LCAP);
      if (warm) dc_sparse_fetch<false>(g, r_slots, lane, lane < nfit, pc, nv, incl - nv, Ttot, list, scell, sseg, tmp_id);
      else dc_sparse_fetch<true>(g, r_slots, lane, lane < nfit, pc, nv, incl - nv, Ttot, list, scell, sseg, tmp_id);
    } else if (lane < nfit) {
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
        st16i(r_slots, (j < nv && nv <= DC_INL && !warm) ? ((uint32_t)pc * DC_INL + j) * 16u : DC_OOB, r);
      }
      for (int k = DC_INL; k < nv; k++) {               // overflow records: insertion by id (rare)
        const int4 r = slots[dc_slot(g, pc, k)];
        scell[excl + k] = pc;
        int pos = k;
        while (pos > 0 && list[excl + pos - 1].w > r.w) {
          list[excl + pos] = list[excl + pos - 1];
          pos--;
        }
        list[excl + pos] = r;
      }
      if (nv > DC_INL && !warm)
        for (int k = 0; k < nv; k++) slots[dc_slot(g, pc, k)] = list[excl + k];
    }
    {                                                   // publish the counts, reset the counters
      const uint32_t coff = (lane < nfit && !warm && (!SPARSE || pc != 0)) ? (uint32_t)pc * 4u : DC_OOB;
      st4i(r_n, coff, nv);
      st4i(r_cnt, coff, 0);
    }
    if constexpr (!SPARSE) {
      for (unsigned long long em = __ballot(lane < nfit && nv == 0); em; em &= em - 1) {   // empty cells: zero rows
        const int pcj = __builtin_amdgcn_readlane(pc, __builtin_ctzll(em));
        st16(r_S, ract ? (uint32_t)pcj * (uint32_t)K::RB + (uint32_t)rl * 16u : DC_OOB, make_float4(0.f, 0.f, 0.f, 0.f));
      }
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_cell += tqb - tqa; tqa = tqb; }
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
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_fill += tqb - tqa; tqa = tqb; }
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
      if (PHASE != 2 && dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_body += tqb - tqa; tqa = tqb; tq_tiles++; }
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
      if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_sum += tqb - tqa; tqa = tqb; }
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
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    d[0] = tq1 - tq0; d[1] = tq_cell; d[2] = tq_fill; d[3] = tq_body; d[4] = tq_sum; d[5] = te - tq0; d[6] = tq_tiles; d[7] = tq0;
  }
}

template <int C, int OP, int NB, bool PIPE>
static int launch_k1p(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     bool warm, hipStream_t st) {
  using K = dc_k1_cfg<C, OP>;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  int64_t waves = (int64_t)(b->tune.k1_wgs > 0 ? b->tune.k1_wgs : 512) * 4;
  int cpw = (int)((vi + waves - 1) / waves);
  if (cpw < 1) cpw = 1;
  const int64_t wgs = (vi + (int64_t)cpw * K::NW - 1) / ((int64_t)cpw * K::NW);
  int k1_pad = b->tune.k1_lds_pad;
  k1_pad = k1_pad < 0 ? 0 : (k1_pad > 16384 ? 16384 : k1_pad);
  // tune.k1_lds_pad: extra dynamic LDS requested on purpose (frames in flight): a workgroup that cannot share its CU with a
  // second one of its own kind shares it with the other frame's gather kernel instead -- the better mix (bench.py)
  const int lds_k1 = K::LDS_BYTES + k1_pad <= 160 * 1024 ? K::LDS_BYTES + k1_pad : K::LDS_BYTES;
  if (lds_k1 > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_premix_modsum<C, OP, NB, PIPE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds_k1);
  hipLaunchKernelGGL((k_dc_premix_modsum<C, OP, NB, PIPE>), dim3((unsigned)wgs), dim3(64 * K::NW), lds_k1, st, b->feats,
                     reinterpret_cast<int4 *>(b->slots), b->cnt, b->cell_n, b->w_pre, b->pre_ln_w, b->pre_ln_b,
                     b->w_pos, b->alpha, d.cg, d.coord_div, d.eps, n, g, cpw, warm, b->S, b->fin, b->hdr,
                     reinterpret_cast<unsigned long long *>(b->tune.k1_dbg));
  return check_launch("link_dc_premix_modsum");
}

// sparse-cell layout: voxel ids per wave (the cells those voxels were first in).  Small frames want many light waves (the
// general layout's tile form runs ONE 16-voxel tile per wave below 32 k voxels for the same reason: a wave is a chain of
// dependent round trips, and 47 waves of 64 voxels took 36 us on a 3 k-voxel frame); tune.k1_wgs > 0 overrides (sweeps)
static inline int dc_sparse_ids_per_wave(const link_dc_buffers_t *b, int64_t n) {
  if (b->tune.k1_wgs > 0 && b->tune.k1_wgs <= 64) return b->tune.k1_wgs;
  return n <= 32768 ? 8 : 16;                          // measured on the S-kitti stage frames (tools/lidar_core.py, IPW sweep)
}
// sparse-cell layout: the fused pre_mix kernel over ranges of voxel ids
template <int C, int OP, int NB>
static int launch_k1_sparse(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                            bool warm, const int32_t *occ, hipStream_t st) {
  using K = dc_k1_cfg<C, OP>;
  const int cpw = dc_sparse_ids_per_wave(b, n);
  const int64_t wgs = (n + (int64_t)cpw * K::NW - 1) / ((int64_t)cpw * K::NW);
  if (K::LDS_BYTES > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_premix_modsum<C, OP, NB, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  hipLaunchKernelGGL((k_dc_premix_modsum<C, OP, NB, false, true>), dim3((unsigned)wgs), dim3(64 * K::NW), K::LDS_BYTES, st,
                     b->feats, reinterpret_cast<int4 *>(b->slots), b->cnt, b->cell_n, b->w_pre, b->pre_ln_w, b->pre_ln_b,
                     b->w_pos, b->alpha, d.cg, d.coord_div, d.eps, n, g, cpw, warm, b->S, b->fin, b->hdr,
                     reinterpret_cast<unsigned long long *>(b->tune.k1_dbg), occ);
  return check_launch("link_dc_premix_modsum(sparse)");
}
template <int C, int OP>
static int dispatch_k1s_nb(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                           bool warm, const int32_t *occ, hipStream_t st) {
  constexpr int T = C / 16;
  int nb = (d.cg % 16 == 0) ? d.cg / 16 : T;
  if (nb > T) nb = T;
  if (T >= 2 && nb == T / 2) return launch_k1_sparse<C, OP, (T >= 2 ? T / 2 : 1)>(b, g, d, n, warm, occ, st);
  if (T >= 4 && nb == T / 4) return launch_k1_sparse<C, OP, (T >= 4 ? T / 4 : 1)>(b, g, d, n, warm, occ, st);
  return launch_k1_sparse<C, OP, T>(b, g, d, n, warm, occ, st);
}
template <int C>
static int dispatch_k1s_op(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                           bool warm, const int32_t *occ, hipStream_t st) {
  switch (d.op) {
    case LINK_OP_COS: return dispatch_k1s_nb<C, LINK_OP_COS>(b, g, d, n, warm, occ, st);
    case LINK_OP_SIN: return dispatch_k1s_nb<C, LINK_OP_SIN>(b, g, d, n, warm, occ, st);
    default: return dispatch_k1s_nb<C, LINK_OP_COSX>(b, g, d, n, warm, occ, st);
  }
}

template <int C, int OP, int NB>
static int launch_k1(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     bool warm, hipStream_t st) {
  return b->tune.k1_pipe ? launch_k1p<C, OP, NB, true>(b, g, d, n, warm, st) : launch_k1p<C, OP, NB, false>(b, g, d, n, warm, st);
}

template <int C, int OP>
static int dispatch_k1_nb(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                          bool warm, hipStream_t st) {
  constexpr int T = C / 16;
  int nb = (d.cg % 16 == 0) ? d.cg / 16 : T;
  if (nb > T) nb = T;
  if (nb == T) return launch_k1<C, OP, T>(b, g, d, n, warm, st);
  if (T >= 2 && nb == T / 2) return launch_k1<C, OP, (T >= 2 ? T / 2 : 1)>(b, g, d, n, warm, st);
  if (T >= 4 && nb == T / 4) return launch_k1<C, OP, (T >= 4 ? T / 4 : 1)>(b, g, d, n, warm, st);
  return launch_k1<C, OP, T>(b, g, d, n, warm, st);    // any other grouping: every block evaluates its own theta
}

template <int C>
static int dispatch_k1_op(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                          bool warm, hipStream_t st) {
  switch (d.op) {
    case LINK_OP_COS: return dispatch_k1_nb<C, LINK_OP_COS>(b, g, d, n, warm, st);
    case LINK_OP_SIN: return dispatch_k1_nb<C, LINK_OP_SIN>(b, g, d, n, warm, st);
    default: return dispatch_k1_nb<C, LINK_OP_COSX>(b, g, d, n, warm, st);
  }
}

#include "dense_fused_mm_impl.h"

// ---------------------------------------------------------------------------------------------
// per-voxel de-modulate + LayerNorm, original voxel order
// ---------------------------------------------------------------------------------------------
// A group of LPR lanes (row of C floats = LPR x float4) per voxel PAIR (2p, 2p+1); a group walks pairs p,
// p + #groups, ... as a three-stage pipeline -- meta(p+2): the two coordinate rows and cell ids; rows(p+1):
// the two A rows (buffer loads, 32-bit offsets); out(p): theta / sincos (shared between channels j and
// j + C/2 when PAIR) / de-modulate / LayerNorm / store.  All parameters live in registers for the whole
// kernel; invalid lanes store to an out-of-range offset.
template <int LPR, int OP, bool PAIR, bool DIV>
__global__ void __launch_bounds__(256) k_dc_demod(const float *__restrict__ A_, const float *__restrict__ fin,
                                                  const int4 *__restrict__ coords, const int32_t *__restrict__ vcell,
                                                  const float *__restrict__ w_pos, const float *__restrict__ alpha,
                                                  const float *__restrict__ ln_w, const float *__restrict__ ln_b, int c,
                                                  int cg, float coord_div, float eps, int64_t n, int64_t a_rows,
                                                  void *__restrict__ out) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int li = lane & (LPR - 1);
  const int ch0 = 4 * li;
  const bool hi = PAIR && (li >= LPR / 2);
  const int ra = P * c * 4;                            // A row bytes
  const __amdgpu_buffer_rsrc_t r_A = dc_rsrc(A_, (uint32_t)(a_rows * ra));
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * c * 4));
  const __amdgpu_buffer_rsrc_t r_out = dc_rsrc(out, (uint32_t)(n * c * IO_BYTES));
  float w0[4], w1[4], w2[4], al[4], gw[4], gb[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int ch = ch0 + e, tc = ch % cg;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
    gw[e] = ln_w[ch]; gb[e] = ln_b[ch];
  }
  const float inv_c = 1.0f / (float)c;
  const int64_t npair = (n + 1) >> 1;
  const int64_t ngroups = (int64_t)gridDim.x * 4 * G;
  int64_t p = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;
  if (p >= npair) return;
  auto ld_meta = [&](int64_t pp, int4 &ca, int4 &cb, int &va, int &vb) {
    const int64_t q = pp < npair ? pp : npair - 1;
    const int64_t ia = 2 * q, ib = (2 * q + 1 < n) ? 2 * q + 1 : 2 * q;
    ca = coords[ia]; cb = coords[ib];
    va = vcell[ia]; vb = vcell[ib];
  };
  auto ld_rows = [&](int va, int vb, v4i_t (&ra_)[P], v4i_t (&rb_)[P]) {
#pragma unroll
    for (int pp = 0; pp < P; pp++) {
      ra_[pp] = __builtin_amdgcn_raw_buffer_load_b128(r_A, (uint32_t)va * (uint32_t)ra + (uint32_t)((pp * c + ch0) * 4), 0, 0);
      rb_[pp] = __builtin_amdgcn_raw_buffer_load_b128(r_A, (uint32_t)vb * (uint32_t)ra + (uint32_t)((pp * c + ch0) * 4), 0, 0);
    }
  };
  int4 c0a, c0b, c1a, c1b;
  int v0a, v0b, v1a, v1b;
  // NOTE: elements of these vectors are converted with __int_as_float (by value); __builtin_bit_cast on a
  // vector-element lvalue reads element 0 whatever the index (clang, ROCm 7.2)
  v4i_t a0[P], b0[P];
  ld_meta(p, c0a, c0b, v0a, v0b);
  ld_rows(v0a, v0b, a0, b0);
  ld_meta(p + ngroups, c1a, c1b, v1a, v1b);
  for (; p < npair; p += ngroups) {
    v4i_t a1[P], b1[P];
    ld_rows(v1a, v1b, a1, b1);
    int4 c2a, c2b;
    int v2a, v2b;
    ld_meta(p + 2 * ngroups, c2a, c2b, v2a, v2b);
    const bool hasB = 2 * p + 1 < n;
    v4i_t fx = {0, 0, 0, 0}, fy = {0, 0, 0, 0};
    if (OP == LINK_OP_COSX) {
      fx = __builtin_amdgcn_raw_buffer_load_b128(r_fin, (uint32_t)(2 * p) * (uint32_t)(c * 4) + (uint32_t)(ch0 * 4), 0, 0);
      fy = __builtin_amdgcn_raw_buffer_load_b128(r_fin, hasB ? (uint32_t)(2 * p + 1) * (uint32_t)(c * 4) + (uint32_t)(ch0 * 4) : DC_OOB, 0, 0);
    }
    float nvA[4], nvB[4], sA = 0.f, sB = 0.f;
    // theta first; a wave whose arguments all sit below 2^15 (always, on sane inputs) evaluates the branch-free
    // float sincos -- left as one `if` inside sincos_nocall, hipcc if-converts the double-precision path and
    // executes its f64 instructions on every iteration
    float thA[4], thB[4];
    bool big = false;
    {
      const bool swapped = PAIR && hi && hasB;
      float xa = (float)(swapped ? c0b.x : c0a.x), ya = (float)(swapped ? c0b.y : c0a.y), za = (float)(swapped ? c0b.z : c0a.z);
      float xb = (float)c0b.x, yb = (float)c0b.y, zb = (float)c0b.z;
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
    if (PAIR) {                                        // this lane evaluated ONE voxel's theta: swap with the partner half
      const bool swapped = hi && hasB;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float sn = snA[e], cs = csA[e];
        const float so = partner<LPR>(sn), co = partner<LPR>(cs);
        snA[e] = swapped ? so : sn; csA[e] = swapped ? co : cs;
        snB[e] = hi ? sn : so;      csB[e] = hi ? cs : co;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float A0 = __int_as_float(a0[0][e]), A1 = __int_as_float(a0[1][e]);
      const float B0 = __int_as_float(b0[0][e]), B1 = __int_as_float(b0[1][e]);
      if (OP == LINK_OP_SIN) {                                                   // linkunet.py:148
        nvA[e] = __fsub_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
        nvB[e] = __fsub_rn(__fmul_rn(B0, csB[e]), __fmul_rn(B1, snB[e]));
      } else {                                                                   // :162
        nvA[e] = __fadd_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
        nvB[e] = __fadd_rn(__fmul_rn(B0, csB[e]), __fmul_rn(B1, snB[e]));
      }
      if (OP == LINK_OP_COSX) {                                                  // :176
        nvA[e] = __fadd_rn(nvA[e], __fsub_rn(__int_as_float(a0[P - 1][e]), link_mul_rn(__int_as_float(fx[e]), thA[e])));
        nvB[e] = __fadd_rn(nvB[e], __fsub_rn(__int_as_float(b0[P - 1][e]), link_mul_rn(__int_as_float(fy[e]), thB[e])));
      }
      sA += nvA[e]; sB += nvB[e];
    }
    sA = grp_sum<LPR>(sA);
    sB = grp_sum<LPR>(sB);
    const float meanA = sA * inv_c, meanB = sB * inv_c;
    float qA = 0.f, qB = 0.f;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float dA = nvA[e] - meanA, dB = nvB[e] - meanB;
      qA += dA * dA; qB += dB * dB;
    }
    qA = grp_sum<LPR>(qA);
    qB = grp_sum<LPR>(qB);
    const float rsA = __builtin_amdgcn_rsqf(qA * inv_c + eps), rsB = __builtin_amdgcn_rsqf(qB * inv_c + eps);
    float4 oa, ob;
    oa.x = (nvA[0] - meanA) * rsA * gw[0] + gb[0]; oa.y = (nvA[1] - meanA) * rsA * gw[1] + gb[1];
    oa.z = (nvA[2] - meanA) * rsA * gw[2] + gb[2]; oa.w = (nvA[3] - meanA) * rsA * gw[3] + gb[3];
    ob.x = (nvB[0] - meanB) * rsB * gw[0] + gb[0]; ob.y = (nvB[1] - meanB) * rsB * gw[1] + gb[1];
    ob.z = (nvB[2] - meanB) * rsB * gw[2] + gb[2]; ob.w = (nvB[3] - meanB) * rsB * gw[3] + gb[3];
    const uint32_t offA = (uint32_t)(2 * p) * (uint32_t)c + (uint32_t)ch0;       // element offset
    io_st4(r_out, offA, v0a != 0, oa);                   // a voxel the index dropped (cell 0) keeps its row untouched
    io_st4(r_out, offA + (uint32_t)c, hasB && v0b != 0, ob);
    c0a = c1a; c0b = c1b; v0a = v1a; v0b = v1b;
    c1a = c2a; c1b = c2b; v1a = v2a; v1b = v2b;
#pragma unroll
    for (int pp = 0; pp < P; pp++) { a0[pp] = a1[pp]; b0[pp] = b1[pp]; }
  }
}

template <int LPR>
static void launch_dc_demod(const link_elk_desc_t &d, int64_t n, int64_t a_rows, hipStream_t st, const float *A,
                            const float *fin, const int32_t *coords, const int32_t *vcell, const float *w_pos,
                            const float *alpha, const float *ln_w, const float *ln_b, void *out) {
  constexpr int G = 64 / LPR;
  const int64_t npair = (n + 1) / 2;
  int64_t wgs = (npair + 4 * G - 1) / (4 * G);
  if (wgs > 1024) wgs = 1024;
  const bool two_part = d.op == LINK_OP_COS || d.op == LINK_OP_SIN;
  const bool pair = LPR >= 2 && d.c == 2 * d.cg && two_part;
  const int4 *co = reinterpret_cast<const int4 *>(coords);
#define LINK_DCDM2(OPP, PP, DD)                                                                                             \
  hipLaunchKernelGGL((k_dc_demod<LPR, OPP, PP, DD>), dim3((unsigned)wgs), dim3(256), 0, st, A, fin, co, vcell, w_pos, alpha, \
                     ln_w, ln_b, d.c, d.cg, d.coord_div, d.eps, n, a_rows, out)
#define LINK_DCDM(OPP, PP)                                        \
  do {                                                            \
    if (d.coord_div != 1.0f) LINK_DCDM2(OPP, PP, true);           \
    else LINK_DCDM2(OPP, PP, false);                              \
  } while (0)
  switch (d.op) {
    case LINK_OP_COS: if (pair) LINK_DCDM(LINK_OP_COS, true); else LINK_DCDM(LINK_OP_COS, false); break;
    case LINK_OP_SIN: if (pair) LINK_DCDM(LINK_OP_SIN, true); else LINK_DCDM(LINK_OP_SIN, false); break;
    default: LINK_DCDM(LINK_OP_COSX, false); break;
  }
#undef LINK_DCDM
#undef LINK_DCDM2
}

// ---------------------------------------------------------------------------------------------
// r^3 box sum + per-voxel de-modulate + LayerNorm in one kernel (C = 64)
// ---------------------------------------------------------------------------------------------
// Counters on cfg2: k_dc_gather and k_dc_demod each move ~75 MB (the A table written, then gathered row by
// row per voxel) and run at memory speed.  Fused, the A rows live for one z-plane in LDS: the workgroup's 16
// groups (4 x 4 columns, dense.hip's plane ring unchanged) put their normalised neighbour sums into an 8 KB
// LDS image, and the plane's voxels -- about 32, whatever their distribution over the 16 cells -- are dealt
// out to the 16 groups as PAIRS: pair p = voxels 2p, 2p+1 of the plane's concatenated slot lists.  A voxel
// finds its cell by a 16-lane ballot against the row-scanned counts (lane li of every DPP row holds cell
// li), its record and A row come from LDS (the inline slot records of the plane's cells travel with the
// plane by LDS-DMA).  Every LDS access is inline asm (dense_gather.h explains why); out rows leave through
// buffer stores, whose unknown number the plane ring's counted wait tolerates by construction.

template <int OP, int R>
struct dc_k2_cfg {
  static constexpr int C = 64, LPR = 16, NG = 16;
  static constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  using G = dc_gather_cfg<C, P, R>;
  static constexpr int RB = P * C * 4;
  static constexpr int REC_OFF = G::PLANE_BYTES + G::CNT_BYTES;
  static constexpr int REC_BYTES = NG * DC_INL * 16;
  static constexpr int BUF_BYTES = REC_OFF + REC_BYTES;
  static constexpr int ABUF_OFF = 3 * BUF_BYTES;
  static constexpr int NCNT_OFF = ABUF_OFF + NG * RB;
  static constexpr int LDS_BYTES = NCNT_OFF + NG * 4;
  // producer / consumer form: two A images (ABUF_OFF, 2 * NG * RB), two count images, a ring of 4 record images
  // (plane-ring slots without the record image: 2 workgroups of 79 KB per CU)
  static constexpr int SPLIT_PLANE = G::NPC * 16;      // no padding pass: surplus DMA lanes re-load pass 0's pieces
  static constexpr int SPLIT_BUF_BYTES = SPLIT_PLANE + G::CNT_BYTES;
  static constexpr int SPLIT_ABUF_OFF = 3 * SPLIT_BUF_BYTES;
  static constexpr int SPLIT_NCNT_OFF = SPLIT_ABUF_OFF + 2 * NG * RB;
  static constexpr int SPLIT_REC_OFF = SPLIT_NCNT_OFF + 2 * NG * 4;
  static constexpr int SPLIT_LDS_BYTES = SPLIT_REC_OFF + 4 * REC_BYTES;
  static constexpr bool SPLIT_FITS = 2 * SPLIT_LDS_BYTES <= 160 * 1024;      // two workgroups per CU (not cos_x: 3-part rows)
  static constexpr int NI = G::PASSES + 2;            // DMA instructions per plane and wave
  static_assert(G::NG == NG && G::TX * G::TY == NG, "16 columns");
};

__device__ __forceinline__ void lds_rd2_b128(uint32_t a0, uint32_t a1, v4f_t &x0, v4f_t &x1) {
  asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(x0), "=&v"(x1) : "v"(a0), "v"(a1) : "memory");
}
// six b128 reads in flight, ONE wait (the pair's two records and four A-row pieces: three round trips -> one)
__device__ __forceinline__ void lds_rd6_b128(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4, uint32_t a5,
                                             v4f_t &x0, v4f_t &x1, v4f_t &x2, v4f_t &x3, v4f_t &x4, v4f_t &x5) {
  asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\tds_read_b128 %2, %8\n\tds_read_b128 %3, %9\n\t"
               "ds_read_b128 %4, %10\n\tds_read_b128 %5, %11\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5)
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5)
               : "memory");
}
__device__ __forceinline__ int lds_rd_b32(uint32_t a) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wr_b128(uint32_t a, float4 v) {
  const v4f_t x = {v.x, v.y, v.z, v.w};
  asm volatile("ds_write_b128 %0, %1" ::"v"(a), "v"(x) : "memory");
}
__device__ __forceinline__ void lds_wr_b32(uint32_t a, int v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory");
}

template <int OP, int R, bool PAIR, bool DIV>
__global__ void __launch_bounds__(256, 2) k_dc_gather_demod(
    const float *__restrict__ S_, const int32_t *__restrict__ cell_n, const int4 *__restrict__ slots,
    const float *__restrict__ fin, const float *__restrict__ w_pos, const float *__restrict__ alpha,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, int cg, float coord_div, float eps, int64_t n,
    link_dc_grid_t g, int txn, int tyn, int zsplit, int nwg, void *__restrict__ out, int single) {
  using K2 = dc_k2_cfg<OP, R>;
  using K = typename K2::G;
  constexpr int C = 64, P = K2::P, LPR = 16, TY = K::TY, TX = K::TX, HY = K::HY, HLO = K::HLO;
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
  uint32_t src_off[K::PASSES];
#pragma unroll
  for (int i = 0; i < K::PASSES; i++) {
    int pid = i * 256 + tid;
    if (pid >= K::NPC) pid = K::NPC - 1;
    const int col = pid / K::RP, pcs = pid % K::RP;
    src_off[i] = col_cell0(col / HY, col % HY) * (uint32_t)RB + (uint32_t)pcs * 16u;
  }
  uint32_t cnt_cell0;
  {
    int e = wave * 64 + lane;
    if (e >= K::NCOL) e = K::NCOL - 1;
    cnt_cell0 = col_cell0(e / HY, e % HY);
  }
  // inline slot records of the 16 interior cells of an output plane: wave w, lane l < 16 -> piece w*16 + l
  uint32_t rec_cell0;
  int rec_k;
  {
    const int piece = wave * 16 + (lane & 15);
    const int col = piece >> 2;
    rec_k = piece & 3;
    rec_cell0 = col_cell0(col / TY + HLO, col % TY + HLO);
  }
  const char *Sb = reinterpret_cast<const char *>(S_);
  auto issue = [&](int plane) {
    int pz = pz0 + plane;
    pz = pz < PDz - 1 ? pz : PDz - 1;
    char *buf = lds + (plane % 3) * K2::BUF_BYTES;
#pragma unroll
    for (int i = 0; i < K::PASSES; i++) {
      const char *src = Sb + (size_t)src_off[i] + (size_t)pz * RB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(buf + (i * 256 + wave * 64) * 16), 16, 0, 0);
    }
    const int32_t *csrc = cell_n + cnt_cell0 + pz;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)csrc,
                                     (__attribute__((address_space(3))) void *)(buf + K::PLANE_BYTES + wave * 256), 4, 0, 0);
    int po = pz0 + plane - (R - 1) + HLO;             // output plane closed by this plane
    po = po < 0 ? 0 : (po < PDz - 1 ? po : PDz - 1);
    const int4 *rsrc = slots + ((size_t)(rec_cell0 + po) * DC_INL + rec_k);
    if (lane < 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)rsrc,
                                       (__attribute__((address_space(3))) void *)(buf + K2::REC_OFF + wave * 256), 16, 0, 0);
  };
  // ---- this group's column ----
  const int grp = tid >> 4, li = tid & 15;
  const int ix = grp / TY, iy = grp % TY;
  const bool col_ok = (x0 + ix < Dx) && (y0 + iy < Dy);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)lds;
  const uint32_t row_lane = (uint32_t)((ix * HY + iy) * RB + li * 16);
  const uint32_t cnt_lane = (uint32_t)((ix * HY + iy) * 4);
  const uint32_t abuf = lds_base + K2::ABUF_OFF, ncnt = lds_base + K2::NCNT_OFF;
  const int rowbase = lane & ~15;                      // first lane of this group's DPP row
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
  issue(0);
  if (nplanes > 1) issue(1);
  for (int i = 0; i < nplanes; i++) {
    if (i + 1 < nplanes) wait_vmcnt<K2::NI>(); else wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    if (i + 2 < nplanes) issue(i + 2);
    const uint32_t bufa = lds_base + (uint32_t)((i % 3) * K2::BUF_BYTES);
    float4 cur[P];
    float cc = 0.f;
#pragma unroll
    for (int pp = 0; pp < P; pp++) cur[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const uint32_t ra = bufa + row_lane, ca = bufa + (uint32_t)K::PLANE_BYTES + cnt_lane;
      dc_read_dx<C, P, R, 0>(ra, ca, cur, cc);
      dc_read_dx<C, P, R, 1>(ra, ca, cur, cc);
      if (R == 3) dc_read_dx<C, P, R, R == 3 ? 2 : 1>(ra, ca, cur, cc);
    }
    const int n_here = lds_rd_b32(bufa + (uint32_t)K::PLANE_BYTES + cnt_lane + (uint32_t)((HLO * HY + HLO) * 4));
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
      // ---- A rows + counts of the plane's 16 cells -> LDS ----
#pragma unroll
      for (int pp = 0; pp < P; pp++)
        lds_wr_b128(abuf + (uint32_t)(grp * RB + pp * C * 4 + li * 16),
                    make_float4(a[pp].x * inv, a[pp].y * inv, a[pp].z * inv, a[pp].w * inv));
      const int n_cell = (R == 3) ? n_prev : n_prev;   // the plane that closed is the previous one for both R
      if (li == 0) lds_wr_b32(ncnt + (uint32_t)(grp * 4), col_ok ? n_cell : 0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      // ---- deal the plane's voxels out as pairs ----
      const int nli = lds_rd_b32(ncnt + (uint32_t)(li * 4));          // lane li of every row: cell li
      int incl = nli;
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
      const int Tv = __builtin_amdgcn_readlane(incl, 15);          // every DPP row holds the same 16 counts: lane 15 has the total
      const int npair = single ? Tv : (Tv + 1) >> 1;
      const uint32_t recb = lds_base + (uint32_t)(((i % 3) * K2::BUF_BYTES) + K2::REC_OFF);
      for (int p = grp; p < npair; p += 16) {
        const int vA = single ? p : 2 * p, vB = (!single && 2 * p + 1 < Tv) ? 2 * p + 1 : vA;
        const bool hasB = !single && 2 * p + 1 < Tv;
        const unsigned long long mA = __ballot(incl <= vA), mB = __ballot(incl <= vB);
        const int cA = __popc((unsigned)(mA >> rowbase) & 0xFFFFu), cB = __popc((unsigned)(mB >> rowbase) & 0xFFFFu);
        const int eA = cA ? __shfl(incl, rowbase + cA - 1, 64) : 0, eB = cB ? __shfl(incl, rowbase + cB - 1, 64) : 0;
        const int kA = vA - eA, kB = vB - eB;
        v4f_t qa, qb;
        lds_rd2_b128(recb + (uint32_t)((cA * DC_INL + (kA < DC_INL ? kA : 0)) * 16),
                     recb + (uint32_t)((cB * DC_INL + (kB < DC_INL ? kB : 0)) * 16), qa, qb);
        int4 recA = make_int4(__float_as_int(qa.x), __float_as_int(qa.y), __float_as_int(qa.z), __float_as_int(qa.w));
        int4 recB = make_int4(__float_as_int(qb.x), __float_as_int(qb.y), __float_as_int(qb.z), __float_as_int(qb.w));
        if (kA >= DC_INL || kB >= DC_INL) {             // overflow records: rare, ordinary loads
          const int pcA = ((b * PDx + x0 + cA / TY + 1) * PDy + y0 + cA % TY + 1) * PDz + po;
          const int pcB = ((b * PDx + x0 + cB / TY + 1) * PDy + y0 + cB % TY + 1) * PDz + po;
          if (kA >= DC_INL) recA = slots[dc_slot(g, pcA, kA)];
          if (kB >= DC_INL) recB = slots[dc_slot(g, pcB, kB)];
        }
        v4f_t A0v, A1v, B0v, B1v, A2v = {0.f, 0.f, 0.f, 0.f}, B2v = {0.f, 0.f, 0.f, 0.f};
        lds_rd2_b128(abuf + (uint32_t)(cA * RB + li * 16), abuf + (uint32_t)(cA * RB + C * 4 + li * 16), A0v, A1v);
        lds_rd2_b128(abuf + (uint32_t)(cB * RB + li * 16), abuf + (uint32_t)(cB * RB + C * 4 + li * 16), B0v, B1v);
        float4 fx = make_float4(0.f, 0.f, 0.f, 0.f), fy = fx;
        if (OP == LINK_OP_COSX) {
          lds_rd2_b128(abuf + (uint32_t)(cA * RB + 2 * C * 4 + li * 16), abuf + (uint32_t)(cB * RB + 2 * C * 4 + li * 16), A2v, B2v);
          fx = *reinterpret_cast<const float4 *>(&fin[(int64_t)recA.w * C + ch0]);
          fy = *reinterpret_cast<const float4 *>(&fin[(int64_t)recB.w * C + ch0]);
        }
        // ---- theta / sincos / de-modulate / LayerNorm / store (k_dc_demod's body) ----
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
        if (PAIR) {
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
          const float A0 = A0v[e], A1 = A1v[e], B0 = B0v[e], B1 = B1v[e];
          if (OP == LINK_OP_SIN) {                                                   // linkunet.py:148
            nvA[e] = __fsub_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
            nvB[e] = __fsub_rn(__fmul_rn(B0, csB[e]), __fmul_rn(B1, snB[e]));
          } else {                                                                   // :162
            nvA[e] = __fadd_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
            nvB[e] = __fadd_rn(__fmul_rn(B0, csB[e]), __fmul_rn(B1, snB[e]));
          }
          if (OP == LINK_OP_COSX) {                                                  // :176
            nvA[e] = __fadd_rn(nvA[e], __fsub_rn(A2v[e], link_mul_rn(fxa[e], thA[e])));
            nvB[e] = __fadd_rn(nvB[e], __fsub_rn(B2v[e], link_mul_rn(fya[e], thB[e])));
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
  }
}

#include "dense_gather_own_impl.h"

#include "dense_gather_quad_impl.h"

template <int OP, int R, bool PAIR, bool DIV>
__global__ void __launch_bounds__(512, 4) k_dc_gather_demod_split(
    const float *__restrict__ S_, const int32_t *__restrict__ cell_n, const int4 *__restrict__ slots,
    const float *__restrict__ fin, const float *__restrict__ w_pos, const float *__restrict__ alpha,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, int cg, float coord_div, float eps, int64_t n,
    link_dc_grid_t g, int txn, int tyn, int zsplit, int nwg, void *__restrict__ out, int single,
    unsigned long long *__restrict__ dbg) {
  using K2 = dc_k2_cfg<OP, R>;
  using K = typename K2::G;
  constexpr int C = 64, P = K2::P, LPR = 16, TY = K::TY, TX = K::TX, HY = K::HY, HLO = K::HLO;
  constexpr int RB = P * C * 4;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  // optional per-wave timing (tools/k2prof.py): s_memtime ticks spent waiting for the plane DMA, in the barrier, in the box
  // sums and in the pair loop
  unsigned long long tq0 = dbg ? __builtin_amdgcn_s_memtime() : 0, tq_dma = 0, tq_bar = 0, tq_box = 0, tq_pairs = 0;
  int tq_iters = 0;
  // Producer / consumer form: waves 0-3 run the plane ring and the box sums exactly as k_dc_gather_demod does and
  // leave the A rows + counts of output plane j in LDS buffer j & 1; waves 4-7 deal plane j's voxels out as pairs
  // and finish them ONE STEP LATER, while the producers already sum plane j+1.  One barrier per step instead of
  // two, the two halves of a step overlap instead of adding up, and the producers issue no stores, so their
  // counted DMA waits are exact.  The inline slot records get their own ring of 4 (a plane-ring slot is reused
  // while the consumers still read the records that travelled with it).
  const bool producer = threadIdx.x < 256;
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane((threadIdx.x & 255) >> 6);
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
  uint32_t src_off[K::PASSES];
#pragma unroll
  for (int i = 0; i < K::PASSES; i++) {
    int pid = i * 256 + tid;
    if (pid >= K::NPC) pid = K::NPC - 1;
    const int col = pid / K::RP, pcs = pid % K::RP;
    src_off[i] = col_cell0(col / HY, col % HY) * (uint32_t)RB + (uint32_t)pcs * 16u;
  }
  uint32_t cnt_cell0;
  {
    int e = wave * 64 + lane;
    if (e >= K::NCOL) e = K::NCOL - 1;
    cnt_cell0 = col_cell0(e / HY, e % HY);
  }
  // inline slot records of the 16 interior cells of an output plane: wave w, lane l < 16 -> piece w*16 + l
  uint32_t rec_cell0;
  int rec_k;
  {
    const int piece = wave * 16 + (lane & 15);
    const int col = piece >> 2;
    rec_k = piece & 3;
    rec_cell0 = col_cell0(col / TY + HLO, col % TY + HLO);
  }
  const char *Sb = reinterpret_cast<const char *>(S_);
  auto issue = [&](int plane) {
    int pz = pz0 + plane;
    pz = pz < PDz - 1 ? pz : PDz - 1;
    char *buf = lds + (plane % 3) * K2::SPLIT_BUF_BYTES;
#pragma unroll
    for (int i = 0; i < K::PASSES; i++) {
      // a wave whose 64 pieces of the last pass all lie beyond the plane repeats its pass-0 load (same data to the
      // same place): the instruction count per wave stays NI and the ring slots need no padding
      const bool surplus = (i * 256 + wave * 64) >= K::NPC;            // wave-uniform
      const int ii = surplus ? 0 : i;
      const char *src = Sb + (size_t)(surplus ? src_off[0] : src_off[i]) + (size_t)pz * RB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(buf + (ii * 256 + wave * 64) * 16), 16, 0, 0);
    }
    const int32_t *csrc = cell_n + cnt_cell0 + pz;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)csrc,
                                     (__attribute__((address_space(3))) void *)(buf + K2::SPLIT_PLANE + wave * 256), 4, 0, 0);
    int po = pz0 + plane - (R - 1) + HLO;             // output plane closed by this plane
    po = po < 0 ? 0 : (po < PDz - 1 ? po : PDz - 1);
    const int4 *rsrc = slots + ((size_t)(rec_cell0 + po) * DC_INL + rec_k);
    if (lane < 16)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)rsrc,
                                       (__attribute__((address_space(3))) void *)(lds + K2::SPLIT_REC_OFF + (plane & 3) * K2::REC_BYTES + wave * 256), 16, 0, 0);
  };
  // ---- this group's column ----
  const int grp = tid >> 4, li = tid & 15;
  const int ix = grp / TY, iy = grp % TY;
  const bool col_ok = (x0 + ix < Dx) && (y0 + iy < Dy);
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)lds;
  const uint32_t row_lane = (uint32_t)((ix * HY + iy) * RB + li * 16);
  const uint32_t cnt_lane = (uint32_t)((ix * HY + iy) * 4);
  const uint32_t abuf0 = lds_base + K2::SPLIT_ABUF_OFF, ncnt0 = lds_base + K2::SPLIT_NCNT_OFF;
  const int rowbase = lane & ~15;                      // first lane of this group's DPP row
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
  // pairs pstart, pstart + 32, ... of output plane jp (A rows / counts in LDS image jp & 1, records in ring slot jp & 3).
  // The consumer waves take pairs 0..15 (+32k), the producer waves -- once their box sums of the next plane are in LDS --
  // pairs 16..31 (+32k): about half of the planes hold more than 32 voxels, and their second round of pairs used to
  // double the step.
  auto pairs_of = [&](int jp, int pstart) {
      if (DC_K2_ABL & 4) return;
      const uint32_t abuf = abuf0 + (uint32_t)((jp & 1) * K2::NG * RB), ncnt = ncnt0 + (uint32_t)((jp & 1) * K2::NG * 4);
      const int po = pz0 + jp - (R - 1) + HLO;
      // ---- deal the plane's voxels out as pairs ----
      const int nli = lds_rd_b32(ncnt + (uint32_t)(li * 4));          // lane li of every row: cell li
      int incl = nli;
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
      incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
      const int Tv = __builtin_amdgcn_readlane(incl, 15);          // every DPP row holds the same 16 counts: lane 15 has the total
      const int npair = single ? Tv : (Tv + 1) >> 1;
      const uint32_t recb = lds_base + (uint32_t)(K2::SPLIT_REC_OFF + (jp & 3) * K2::REC_BYTES);
      for (int p = pstart; p < npair; p += DC_K2_SECOND_ROUND ? 32 : 16) {
        if (dbg) tq_iters++;
        const int vA = single ? p : 2 * p, vB = (!single && 2 * p + 1 < Tv) ? 2 * p + 1 : vA;
        const bool hasB = !single && 2 * p + 1 < Tv;
        const unsigned long long mA = __ballot(incl <= vA), mB = __ballot(incl <= vB);
        const int cA = __popc((unsigned)(mA >> rowbase) & 0xFFFFu), cB = __popc((unsigned)(mB >> rowbase) & 0xFFFFu);
        const int eA = cA ? __shfl(incl, rowbase + cA - 1, 64) : 0, eB = cB ? __shfl(incl, rowbase + cB - 1, 64) : 0;
        const int kA = vA - eA, kB = vB - eB;
        v4f_t qa, qb;
        v4f_t A0v, A1v, B0v, B1v, A2v = {0.f, 0.f, 0.f, 0.f}, B2v = {0.f, 0.f, 0.f, 0.f};
        lds_rd6_b128(recb + (uint32_t)((cA * DC_INL + (kA < DC_INL ? kA : 0)) * 16),
                     recb + (uint32_t)((cB * DC_INL + (kB < DC_INL ? kB : 0)) * 16),
                     abuf + (uint32_t)(cA * RB + li * 16), abuf + (uint32_t)(cA * RB + C * 4 + li * 16),
                     abuf + (uint32_t)(cB * RB + li * 16), abuf + (uint32_t)(cB * RB + C * 4 + li * 16), qa, qb, A0v, A1v, B0v, B1v);
        int4 recA = make_int4(__float_as_int(qa.x), __float_as_int(qa.y), __float_as_int(qa.z), __float_as_int(qa.w));
        int4 recB = make_int4(__float_as_int(qb.x), __float_as_int(qb.y), __float_as_int(qb.z), __float_as_int(qb.w));
        if (kA >= DC_INL || kB >= DC_INL) {             // overflow records: rare, ordinary loads
          const int pcA = ((b * PDx + x0 + cA / TY + 1) * PDy + y0 + cA % TY + 1) * PDz + po;
          const int pcB = ((b * PDx + x0 + cB / TY + 1) * PDy + y0 + cB % TY + 1) * PDz + po;
          if (kA >= DC_INL) recA = slots[dc_slot(g, pcA, kA)];
          if (kB >= DC_INL) recB = slots[dc_slot(g, pcB, kB)];
        }
        float4 fx = make_float4(0.f, 0.f, 0.f, 0.f), fy = fx;
        if (OP == LINK_OP_COSX) {
          lds_rd2_b128(abuf + (uint32_t)(cA * RB + 2 * C * 4 + li * 16), abuf + (uint32_t)(cB * RB + 2 * C * 4 + li * 16), A2v, B2v);
          fx = *reinterpret_cast<const float4 *>(&fin[(int64_t)recA.w * C + ch0]);
          fy = *reinterpret_cast<const float4 *>(&fin[(int64_t)recB.w * C + ch0]);
        }
        // ---- theta / sincos / de-modulate / LayerNorm / store (k_dc_demod's body) ----
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
            if (DC_K2_ABL & 1) { snA[e] = thA[e]; csA[e] = 1.0f - thA[e]; snB[e] = thB[e]; csB[e] = 1.0f - thB[e]; continue; }
            sincos_small(thA[e], snA[e], csA[e]);
            if (PAIR) { snB[e] = snA[e]; csB[e] = csA[e]; } else sincos_small(thB[e], snB[e], csB[e]);
          }
        }
        if (PAIR) {
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
          const float A0 = A0v[e], A1 = A1v[e], B0 = B0v[e], B1 = B1v[e];
          if (OP == LINK_OP_SIN) {                                                   // linkunet.py:148
            nvA[e] = __fsub_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
            nvB[e] = __fsub_rn(__fmul_rn(B0, csB[e]), __fmul_rn(B1, snB[e]));
          } else {                                                                   // :162
            nvA[e] = __fadd_rn(__fmul_rn(A0, csA[e]), __fmul_rn(A1, snA[e]));
            nvB[e] = __fadd_rn(__fmul_rn(B0, csB[e]), __fmul_rn(B1, snB[e]));
          }
          if (OP == LINK_OP_COSX) {                                                  // :176
            nvA[e] = __fadd_rn(nvA[e], __fsub_rn(A2v[e], link_mul_rn(fxa[e], thA[e])));
            nvB[e] = __fadd_rn(nvB[e], __fsub_rn(B2v[e], link_mul_rn(fya[e], thB[e])));
          }
          sA += nvA[e]; sB += nvB[e];
        }
        if (!(DC_K2_ABL & 2)) { sA = grp_sum<LPR>(sA); sB = grp_sum<LPR>(sB); }
        const float meanA = sA * (1.0f / C), meanB = sB * (1.0f / C);
        float qA = 0.f, qB = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float dA = nvA[e] - meanA, dB = nvB[e] - meanB;
          qA += dA * dA; qB += dB * dB;
        }
        if (!(DC_K2_ABL & 2)) { qA = grp_sum<LPR>(qA); qB = grp_sum<LPR>(qB); }
        const float rsA = __builtin_amdgcn_rsqf(qA * (1.0f / C) + eps), rsB = __builtin_amdgcn_rsqf(qB * (1.0f / C) + eps);
        float4 oa, ob;
        oa.x = (nvA[0] - meanA) * rsA * gw[0] + gb[0]; oa.y = (nvA[1] - meanA) * rsA * gw[1] + gb[1];
        oa.z = (nvA[2] - meanA) * rsA * gw[2] + gb[2]; oa.w = (nvA[3] - meanA) * rsA * gw[3] + gb[3];
        ob.x = (nvB[0] - meanB) * rsB * gw[0] + gb[0]; ob.y = (nvB[1] - meanB) * rsB * gw[1] + gb[1];
        ob.z = (nvB[2] - meanB) * rsB * gw[2] + gb[2]; ob.w = (nvB[3] - meanB) * rsB * gw[3] + gb[3];
        io_st4(r_out, (uint32_t)recA.w * (uint32_t)C + (uint32_t)ch0, !(DC_K2_ABL & 8) || oa.x == 123.456f, oa);
        io_st4(r_out, (uint32_t)recB.w * (uint32_t)C + (uint32_t)ch0, (!(DC_K2_ABL & 8) || ob.x == 123.456f) && hasB, ob);
      }
  };
  if (producer) {
    issue(0);
    if (nplanes > 1) issue(1);
  }
  for (int i = 0; i <= nplanes; i++) {
    unsigned long long tqa = dbg ? __builtin_amdgcn_s_memtime() : 0;
    if (producer) {
      if (i + 1 < nplanes) wait_vmcnt<K2::NI>(); else wait_vmcnt<0>();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the A rows of the previous step are in LDS
    }
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_dma += tqb - tqa; tqa = tqb; }
    asm volatile("s_barrier" ::: "memory");
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_bar += tqb - tqa; tqa = tqb; }
    if (producer && i >= nplanes) {
      if (DC_K2_SECOND_ROUND && i - 1 >= R - 1) pairs_of(i - 1, grp + 16);
      if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_pairs += tqb - tqa; }
      continue;
    }
    if (producer && i + 2 < nplanes) issue(i + 2);
    const uint32_t bufa = lds_base + (uint32_t)((i % 3) * K2::SPLIT_BUF_BYTES);
    const int j = producer ? i : i - 1;                      // the output-plane step this half works on
    const uint32_t abuf = abuf0 + (uint32_t)((j & 1) * K2::NG * RB), ncnt = ncnt0 + (uint32_t)((j & 1) * K2::NG * 4);
    if (!producer) {
      if (j >= R - 1 && j < nplanes) pairs_of(j, grp);
      if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_pairs += tqb - tqa; }
      continue;
    }
    float4 cur[P];
    float cc = 0.f;
    int n_here = 0;
#pragma unroll
    for (int pp = 0; pp < P; pp++) cur[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const uint32_t ra = bufa + row_lane, ca = bufa + (uint32_t)K2::SPLIT_PLANE + cnt_lane;
      if constexpr (P == 2 && R == 3 && DC_K2_PIPE_READS) {
        dc_read_plane_p2r3<C>(ra, ca, cur, cc);       // one pipelined LDS request instead of three dependent round trips
      } else {
        dc_read_dx<C, P, R, 0>(ra, ca, cur, cc);
        dc_read_dx<C, P, R, 1>(ra, ca, cur, cc);
        if (R == 3) dc_read_dx<C, P, R, R == 3 ? 2 : 1>(ra, ca, cur, cc);
      }
    }
    n_here = lds_rd_b32(bufa + (uint32_t)K2::SPLIT_PLANE + cnt_lane + (uint32_t)((HLO * HY + HLO) * 4));
    if (j >= R - 1) {
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
      // ---- A rows + counts of the plane's 16 cells -> LDS ----
#pragma unroll
      for (int pp = 0; pp < P; pp++)
        lds_wr_b128(abuf + (uint32_t)(grp * RB + pp * C * 4 + li * 16),
                    make_float4(a[pp].x * inv, a[pp].y * inv, a[pp].z * inv, a[pp].w * inv));
      if (li == 0) lds_wr_b32(ncnt + (uint32_t)(grp * 4), col_ok ? n_prev : 0);   // the plane that closed is the previous one for both R
    }
#pragma unroll
    for (int pp = 0; pp < P; pp++) { r0[pp] = r1[pp]; r1[pp] = cur[pp]; }
    c0 = c1; c1 = cc;
    n_prev = n_here;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_box += tqb - tqa; tqa = tqb; }
    if (DC_K2_SECOND_ROUND && i - 1 >= R - 1) pairs_of(i - 1, grp + 16);
    if (dbg) { const unsigned long long tqb = __builtin_amdgcn_s_memtime(); tq_pairs += tqb - tqa; }
  }
  if (dbg && lane == 0) {
    unsigned long long *d = dbg + ((size_t)L * 8 + (threadIdx.x >> 6)) * 8;
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    d[0] = te - tq0; d[1] = tq_dma; d[2] = tq_bar; d[3] = tq_box; d[4] = tq_pairs; d[5] = tq_iters; d[6] = nplanes; d[7] = producer ? 1 : 2;
  }
}


template <int OP, int R>
static int launch_k2(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     hipStream_t st) {
  using K2 = dc_k2_cfg<OP, R>;
  using K = typename K2::G;
  const int txn = (g.dim[0] + K::TX - 1) / K::TX, tyn = (g.dim[1] + K::TY - 1) / K::TY;
  int zsplit = b->tune.k2_zsplit;
  int k2_pad = b->tune.k2_lds_pad;
  k2_pad = k2_pad < 0 ? 0 : (k2_pad > 4096 ? 4096 : k2_pad);
  const int k2_single = (b->tune.k2_form & 2) ? 1 : 0;
  if (zsplit <= 0) {
    const int64_t tiles = (int64_t)txn * tyn * g.dim[3];
    zsplit = (int)(512 / tiles);
    if (zsplit < 1) zsplit = 1;
  }
  if (zsplit > g.dim[2]) zsplit = g.dim[2];
  const int64_t nwg = (int64_t)txn * tyn * g.dim[3] * zsplit;
  const int64_t grid = (nwg + 7) / 8 * 8;
  const bool two_part = d.op == LINK_OP_COS || d.op == LINK_OP_SIN;
  const bool pair = d.c == 2 * d.cg && two_part;
  const bool div = d.coord_div != 1.0f;
#define LINK_K2S(PP, DD)                                                                                              \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_gather_demod_split<OP, R, PP, DD>),                \
                              hipFuncAttributeMaxDynamicSharedMemorySize, K2::SPLIT_LDS_BYTES + k2_pad);        \
    hipLaunchKernelGGL((k_dc_gather_demod_split<OP, R, PP, DD>), dim3((unsigned)grid), dim3(512), K2::SPLIT_LDS_BYTES + k2_pad, st, \
                       b->S, b->cell_n, reinterpret_cast<const int4 *>(b->slots), b->fin, b->w_pos, b->alpha, b->ln_w, \
                       b->ln_b, d.cg, d.coord_div, d.eps, n, g, txn, tyn, zsplit, (int)nwg, b->out, k2_single,       \
                       reinterpret_cast<unsigned long long *>(b->tune.k2_dbg));                                       \
  } while (0)
  // k2_form 0 = by measurement (tools/k2ab.py, cfg2-sized frames): the producer / consumer form where two of its workgroups fit
  // a CU (two-part rows: 29.9 us, own-cell 29.9, single-role 30.4); three-part rows (cos_x) r = 3: own-cell 38.2 us against
  // 53.5 for the single-role kernel (81 KB plane ring: one workgroup per CU); r = 2: single-role 32.9, own-cell 34.4
  const bool own = (b->tune.k2_form & 4) || (!(b->tune.k2_form & 1) && !K2::SPLIT_FITS && R == 3);
  if (own) {                                           // own-cell form: 5 waves, 2-slot ring fed by a dedicated DMA wave
    using KO = dc_k2o_cfg<OP, R>;
#define LINK_K2O(PP, DD)                                                                                              \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_gather_demod_own<OP, R, PP, DD>),                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, KO::LDS_BYTES + k2_pad);                   \
    hipLaunchKernelGGL((k_dc_gather_demod_own<OP, R, PP, DD>), dim3((unsigned)grid), dim3(320), KO::LDS_BYTES + k2_pad, st, b->S, \
                       b->cell_n, reinterpret_cast<const int4 *>(b->slots), b->fin, b->w_pos, b->alpha, b->ln_w,      \
                       b->ln_b, d.cg, d.coord_div, d.eps, n, g, txn, tyn, zsplit, (int)nwg, b->out,                   \
                       reinterpret_cast<unsigned long long *>(b->tune.k2_dbg));                                       \
  } while (0)
    if (pair) { if (div) LINK_K2O(true, true); else LINK_K2O(true, false); }
    else { if (div) LINK_K2O(false, true); else LINK_K2O(false, false); }
#undef LINK_K2O
    return check_launch("link_dc_gather_demod");
  }
  if constexpr (dc_k2q_cfg<OP, R>::FITS) {             // two-part rows, theta shared by channels j / j + 32: quad consumers (bit 3: round-2 pair form)
    if (pair && !b->alpha && !(b->tune.k2_form & 9) && !k2_single) {
      using KQ = dc_k2q_cfg<OP, R>;
#define LINK_K2Q(DD)                                                                                                  \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_gather_demod_quad<OP, R, DD>),                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, KQ::LDS_BYTES + k2_pad);                   \
    hipLaunchKernelGGL((k_dc_gather_demod_quad<OP, R, DD>), dim3((unsigned)grid), dim3(KQ::THREADS), KQ::LDS_BYTES + k2_pad, st, b->S, \
                       b->cell_n, reinterpret_cast<const int4 *>(b->slots), b->w_pos, b->alpha, b->ln_w, b->ln_b, d.cg, \
                       d.coord_div, d.eps, n, g, txn, tyn, zsplit, (int)nwg, b->out,                                   \
                       reinterpret_cast<unsigned long long *>(b->tune.k2_dbg));                                       \
  } while (0)
      if (div) LINK_K2Q(true); else LINK_K2Q(false);
#undef LINK_K2Q
      return check_launch("link_dc_gather_demod");
    }
  }
  if (!(b->tune.k2_form & 1) && K2::SPLIT_FITS) {
    if (pair) { if (div) LINK_K2S(true, true); else LINK_K2S(true, false); }
    else { if (div) LINK_K2S(false, true); else LINK_K2S(false, false); }
    return check_launch("link_dc_gather_demod");
  }
#undef LINK_K2S
#define LINK_K2(PP, DD)                                                                                               \
  do {                                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_gather_demod<OP, R, PP, DD>),                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, K2::LDS_BYTES);                             \
    hipLaunchKernelGGL((k_dc_gather_demod<OP, R, PP, DD>), dim3((unsigned)grid), dim3(256), K2::LDS_BYTES, st, b->S,  \
                       b->cell_n, reinterpret_cast<const int4 *>(b->slots), b->fin, b->w_pos, b->alpha, b->ln_w,      \
                       b->ln_b, d.cg, d.coord_div, d.eps, n, g, txn, tyn, zsplit, (int)nwg, b->out, k2_single);     \
  } while (0)
  if (pair) { if (div) LINK_K2(true, true); else LINK_K2(true, false); }
  else { if (div) LINK_K2(false, true); else LINK_K2(false, false); }
#undef LINK_K2
  return check_launch("link_dc_gather_demod");
}

#include "dense_gather_sparse_impl.h"

int run_premix_modsum(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                      bool warm, hipStream_t st) {
  if (b->tune.k1_form == 2) {                          // matrix-core sums form (dense_fused_mm_impl.h): C = 32 / 64
    if (d.c == 64) return dispatch_k1m_op<64>(b, g, d, n, warm, st);
    if (d.c == 32) return dispatch_k1m_op<32>(b, g, d, n, warm, st);
  }
  switch (d.c) {
    case 16: return dispatch_k1_op<16>(b, g, d, n, warm, st);
    case 32: return dispatch_k1_op<32>(b, g, d, n, warm, st);
    default: return dispatch_k1_op<64>(b, g, d, n, warm, st);
  }
}

int run_premix_modsum_sparse(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                             bool warm, const int32_t *occ, hipStream_t st) {
  if (b->tune.k1_form == 2 && d.op != LINK_OP_COSX) {  // matrix-core sums form on request (measured slower here: 78 against 50 us on
                                                       // S-kitti stage 1; its cos_x instantiation spills): the cell-range form is the default
    if (d.c == 64) return dispatch_k1ms<64>(b, g, d, n, warm, occ, st);
    if (d.c == 32) return dispatch_k1ms<32>(b, g, d, n, warm, occ, st);
  }
  switch (d.c) {
    case 16: return dispatch_k1s_op<16>(b, g, d, n, warm, occ, st);
    case 32: return dispatch_k1s_op<32>(b, g, d, n, warm, occ, st);
    default: return dispatch_k1s_op<64>(b, g, d, n, warm, occ, st);
  }
}

int run_demod(const float *A, const float *fin, const int32_t *coords, const int32_t *vcell, const float *w_pos,
              const float *alpha, const float *ln_w, const float *ln_b, const link_elk_desc_t &d,
              const link_dc_grid_t &g, int64_t n, void *out, hipStream_t st) {
  switch (d.c) {
    case 16: launch_dc_demod<4>(d, n, g.vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
    case 32: launch_dc_demod<8>(d, n, g.vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
    case 64: launch_dc_demod<16>(d, n, g.vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
    default: launch_dc_demod<32>(d, n, g.vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
  }
  return check_launch("link_dc_demod");
}

int run_gather_demod(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     hipStream_t st) {
  if (d.r == 3) {
    switch (d.op) {
      case LINK_OP_COS: return launch_k2<LINK_OP_COS, 3>(b, g, d, n, st);
      case LINK_OP_SIN: return launch_k2<LINK_OP_SIN, 3>(b, g, d, n, st);
      default: return launch_k2<LINK_OP_COSX, 3>(b, g, d, n, st);
    }
  }
  switch (d.op) {
    case LINK_OP_COS: return launch_k2<LINK_OP_COS, 2>(b, g, d, n, st);
    case LINK_OP_SIN: return launch_k2<LINK_OP_SIN, 2>(b, g, d, n, st);
    default: return launch_k2<LINK_OP_COSX, 2>(b, g, d, n, st);
  }
}

#include "dense_step3_impl.h"

}  // namespace DC_IO_NS
