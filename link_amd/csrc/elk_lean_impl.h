// link_amd/csrc/elk_lean_impl.h -- lean form of R_core with the index REBUILT every call (round 4; include/link_amd.h
// section C, link_elk_core_lean_forward): three launches, none of them proportional to the grid, each a short chain of
// dependent memory round trips -- made for the LiDAR stage frames of linkunet.py:345-363 / scn.py:586-607 (a few thousand to a
// few ten thousand voxels in a block grid that is 1-20 % occupied), where the reference rebuilds its hash / unique / neighbour
// maps on every call and the six-launch path (four index kernels + the two tile kernels) spends most of its time in launch
// boundaries and dependency chains, not in bytes.
//
//   k_lean_insert_premix   one 16-voxel tile per wave IN INPUT ORDER (coalesced rows, no index needed): pre_mix on the matrix
//                          cores + LayerNorm + theta + sincos + modulate -> X rows [F cos | F sin (| F theta)] in a scratch
//                          matrix; at the same time the voxel's block cell gets it: rank = cnt[cell]++, list[cell][rank] = id,
//                          and the voxel whose rank is a multiple of 32 appends the work item (cell, rank / 32) to one of 16
//                          item lists (one atomic per workgroup and list counter: same-address atomics serialise).  Extra
//                          workgroups behind the frame's own clear the PREVIOUS frame's counters through its item lists
//                          (cnt is double-buffered).
//   k_lean_sums            a wave per item = one chunk of <= 32 voxels of one cell, taken in ascending voxel id (the ranks came
//                          from atomics, so the wave ranks the cell's ids by counting): S[cell][chunk] = sum of their X rows,
//                          lane groups over the voxels, combined in a fixed order -- bitwise reproducible; the chunk's
//                          records (x, y, z, id) in id order go to rec2.
//   k_lean_gather          a wave per item: the first chunk row of each of the r^3 neighbour cells is requested together with
//                          their counts (address arithmetic only; a row counts iff its cell's cnt > 0), further chunks of
//                          bigger neighbours after the counts; sum / summed count, then the chunk's voxels de-modulated +
//                          LayerNorm'ed by the lane groups.
// Dependent memory round trips per launch: 2-3 (coords + rows -> atomic -> stores; item -> count + ids -> X rows -> stores;
// item -> records + counts + rows -> stores).
// The block table is addressed by CELL (no numbering, no scan, no sort of the frame): rows are touched only where voxels
// land.  cos_x needs no fin matrix: its de-modulation term fin * theta is the third part of the voxel's own X row.
// Compiled per feature I/O type (DC_IO / DC_IO_NS) like the tile form.
#pragma once
#include "tile_common.h"

#ifdef LEAN_DBG           /* profiling builds only: phase ticks of launch 1 summed into hdr[16 + k] (hdr then has 64 words) */
#define LEAN_TICK(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define LEAN_WAITALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory")
#else
#define LEAN_TICK(v)
#define LEAN_WAITALL()
#endif

namespace DC_IO_NS {
using namespace link;

constexpr int LEAN_CH = 32;          // voxels per chunk (work item; 64 measured slower: the per-item voxel loop is serial)
constexpr int LEAN_KMAX = 352;       // slot capacity of a cell (7^3 = 343 rounded up to chunks)
constexpr int LEAN_SEGS = 16;        // item lists (their counters on separate 64-byte lines: same-address atomics serialise)
constexpr int LEAN_IW = 4;           // waves per workgroup of launches 2 / 3, a wave per item

struct lean_args {
  const void *feats;
  const int4 *coords;
  const float *w_pre, *pre_ln_w, *pre_ln_b, *w_pos, *alpha, *ln_w, *ln_b;
  link_grid_t g;
  int cg;
  float coord_div, eps;
  int n, k, kch, build, nwg, idx_cap, idx_cap_prev, cshift;
  int xl_stride, xl_off;              // where launch 3 finds fin * theta of voxel id (cos_x): X + id * xl_stride + xl_off
  int64_t seg_cap;
  uint32_t *cnt, *cnt_prev;
  int32_t *list, *occ;
  const int32_t *occ_prev;
  uint32_t *ctrl, *ctrl_prev;
  int4 *rec2;
  float *X, *S;
  int32_t *hdr;
  void *out;
};

// Items are appended to list sg = (workgroup of launch 1) % 16 with one atomic per workgroup; wave j of launches 2 / 3 takes
// entry j / 16 of list j % 16 -- the entry and the list's count are requested together (no prefix over the counts, no
// dependent fetch), a wave whose entry lies beyond the count has nothing to do.

// Lanes with `act` and the same `cell` form a group (LiDAR frames keep the voxels of a block close in memory: a tile's 16 voxels
// fall into a handful of cells): the group makes ONE atomic on the cell's counter.  A 7^3 block holds up to 343 voxels and a
// coarse stage has a few hundred cells, so per-voxel atomics pile up on a few counters -- 14 us of a 20 us launch on the
// 10k-voxel C = 128 stage (same-address atomics serialise in the L2; for the same reason a counter sits on its own line when
// the grid is small: cshift).  Returns the group's leader lane, this lane's position in the group and the group's size.
__device__ __forceinline__ void lean_groups(bool act, int cell, int lane, int &leader, int &off, int &size) {
  unsigned long long todo = __ballot(act);
  leader = lane; off = 0; size = act ? 1 : 0;
  while (todo) {                                         // wave-uniform
    const int l0 = __builtin_ctzll(todo);
    const int c0 = __shfl(cell, l0, 64);
    const unsigned long long m = __ballot(act && cell == c0);
    if (act && cell == c0) { leader = l0; off = __popcll(m & ((1ull << lane) - 1ull)); size = __popcll(m); }
    todo &= ~m;
  }
}

// ---------------------------------------------------------------------------------------------
// launch 1: X rows of the voxels in input order + slot insert + clean-up of the previous frame
// ---------------------------------------------------------------------------------------------
template <int C, int OP, int NB>
__global__ void __launch_bounds__(256, (C <= 64 ? 3 : 1)) k_lean_insert_premix(const lean_args a) {
  constexpr int T = C / 16, P = op_parts<OP>::value, W = P * C;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ int s_new[4], s_base;
  float *ln_lds = reinterpret_cast<float *>(smem_raw + dc_wimg<C>::WIMG_BYTES);
  float *pw_lds = ln_lds + 2 * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  // workgroups behind the frame's own: the previous indexed frame's counters go back to zero through its item lists
  // (nothing is proportional to the grid)
  if ((int)blockIdx.x >= a.nwg) {
    const int q = ((int)blockIdx.x - a.nwg) * 256 + tid;
    const int sg = q & (LEAN_SEGS - 1), idx = q >> 4;
    if (idx < a.idx_cap_prev) {
      const int c = (int)a.ctrl_prev[sg * 16];
      const int it = a.occ_prev[(int64_t)sg * a.seg_cap + idx];
      if (idx < c && (it & 15) == 0) a.cnt_prev[(int64_t)(it >> 4) << a.cshift] = 0u;
    }
    return;
  }
  const int i = blockIdx.x * 64 + wave * 16 + li;
  const bool valid = i < a.n;
  const int ic = valid ? i : a.n - 1;
  const int4 rec = a.coords[ic];
  const __amdgpu_buffer_rsrc_t r_feats = dc_rsrc(a.feats, (uint32_t)((int64_t)a.n * C * IO_BYTES));
  float4 ff[T];
  {
    const uint32_t ro = ((uint32_t)ic * (uint32_t)C + (uint32_t)(4 * gq)) * (uint32_t)IO_BYTES;
    ff[0] = io_ldb4<0>(r_feats, ro);
    if constexpr (T > 1) ff[1] = io_ldb4<16>(r_feats, ro);
    if constexpr (T > 2) { ff[2] = io_ldb4<32>(r_feats, ro); ff[3] = io_ldb4<48>(r_feats, ro); }
    if constexpr (T > 4) { ff[4] = io_ldb4<64>(r_feats, ro); ff[5] = io_ldb4<80>(r_feats, ro); ff[6] = io_ldb4<96>(r_feats, ro); ff[7] = io_ldb4<112>(r_feats, ro); }
  }
  bool w_big = dc_stage_weights<C, 256, LINK_TILE_EXACT(OP)>(smem_raw, a.w_pre, a.pre_ln_w, a.pre_ln_b, a.w_pos, a.alpha, a.cg, tid);
  // ---- slot insert: the lane group gq == 0 of each wave speaks for the tile's 16 voxels; the atomic is on its way while
  // the tile is worked on ----
  const bool ins = a.build && valid && gq == 0;
  int cell = -1, base = 0, g_lead, g_off, g_size;
  if (ins) {
    cell = cell_of(a.g, floordiv(rec.x, a.g.s), floordiv(rec.y, a.g.s), floordiv(rec.z, a.g.s), rec.w);
    if (cell < 0) atomicOr(&a.hdr[LINK_HDR_STATUS_ACC], 1);
  }
  lean_groups(ins && cell >= 0, cell, lane, g_lead, g_off, g_size);
  if (ins && cell >= 0 && lane == g_lead) base = (int)atomicAdd(&a.cnt[(int64_t)cell << a.cshift], (unsigned)g_size);
  w_big = __syncthreads_or(w_big) != 0 || (LINK_COSX_EXACT && OP == LINK_OP_COSX);   // cos_x: exact contraction (elk_common.h)
  const unsigned short *wh = reinterpret_cast<const unsigned short *>(smem_raw);
  floatx4 ac[T];
  dc_premix_tile<C, LINK_TILE_EXACT(OP)>(wh, a.w_pre, w_big, li, gq, ff, ac);
  float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
  if (a.coord_div != 1.0f) { x = x / a.coord_div; y = y / a.coord_div; z = z / a.coord_div; }
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
    if (__builtin_expect(__any(big), 0)) {
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
  const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + a.eps);
  float *xrow = a.X + (int64_t)ic * W + 4 * gq;
#pragma unroll
  for (int tp = 0; tp < T; tp++) {
    const int tb = tp % NB;
    const float4 lw = *reinterpret_cast<const float4 *>(&ln_lds[16 * tp + 4 * gq]);
    const float4 lb = *reinterpret_cast<const float4 *>(&ln_lds[C + 16 * tp + 4 * gq]);
    float th1[4], sn1[4], cs1[4];
    if constexpr (NB == T) trig(tp, th1, sn1, cs1);
    const float fv[4] = {(ac[tp][0] - mean) * rstd * lw.x + lb.x, (ac[tp][1] - mean) * rstd * lw.y + lb.y,
                         (ac[tp][2] - mean) * rstd * lw.z + lb.z, (ac[tp][3] - mean) * rstd * lw.w + lb.w};
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
      if (valid) *reinterpret_cast<float4 *>(xrow + pp * C + 16 * tp) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    }
  }
  if (!a.build) return;                                  // kernel-uniform
  // ---- the ranks are back: slot lists, and the work items (one per started chunk of 32) ----
  const int rank = __shfl(base, g_lead, 64) + g_off;
  const bool full = cell >= 0 && rank >= a.k;
  if (full) atomicOr(&a.hdr[LINK_HDR_STATUS_ACC], 2);
  const bool keep = cell >= 0 && !full;
  if (keep) a.list[(int64_t)cell * a.k + rank] = i;
  const bool item = keep && (rank & (LEAN_CH - 1)) == 0;
  const unsigned long long im = __ballot(item);
  if (lane == 0) s_new[wave] = __popcll(im);
  __syncthreads();
  const int n0 = s_new[0], n1 = s_new[1], n2 = s_new[2], n3 = s_new[3];
  const int sg = blockIdx.x % LEAN_SEGS;
  if (tid == 0) s_base = (n0 + n1 + n2 + n3) ? (int)atomicAdd(&a.ctrl[sg * 16], (unsigned)(n0 + n1 + n2 + n3)) : 0;
  __syncthreads();
  if (item) {
    const int before = (wave > 0 ? n0 : 0) + (wave > 1 ? n1 : 0) + (wave > 2 ? n2 : 0) + __popcll(im & ((1ull << lane) - 1ull));
    a.occ[(int64_t)sg * a.seg_cap + s_base + before] = cell * 16 + rank / LEAN_CH;
  }
}

// ---------------------------------------------------------------------------------------------
// launch 1, channel-split form (C = 64 / 128): a workgroup takes ONE 16-voxel tile and its four waves split the OUTPUT channels
// of pre_mix -- wave w owns channels [w C/4, (w + 1) C/4).  Each wave fetches its own C/4 x C slice of W straight into the
// matrix-core A layout (no LDS image of W, no staging barrier) and issues 1/4 of the tile's matrix instructions of the
// tile-per-wave form above at a quarter of the operand traffic; the LayerNorm statistics cross the waves through 128 bytes of
// LDS.  At C = 128 the tile-per-wave form spends ~12 of its 17 us staging a 64 KB W image per workgroup and issuing 192 matrix
// instructions per wave; small frames have nothing to amortise that over.
// ---------------------------------------------------------------------------------------------
template <int C, int OP>
__global__ void __launch_bounds__(256) k_lean_insert_premix_cs(const lean_args a) {
  constexpr int T = C / 16, TB = T / 4, P = op_parts<OP>::value, W = P * C;
  static_assert(C == 64 || C == 128, "channel-split form: 16-channel blocks per wave");
  __shared__ float s_part[2][4][16];
  __shared__ __attribute__((aligned(16))) float s_par[6 * C];      // LayerNorm weight | bias | theta w0 | w1 | w2 | alpha, per channel
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  if ((int)blockIdx.x >= a.nwg) {                        // clean-up workgroups: as in the tile-per-wave form
    const int q = ((int)blockIdx.x - a.nwg) * 256 + tid;
    const int sg = q & (LEAN_SEGS - 1), idx = q >> 4;
    if (idx < a.idx_cap_prev) {
      const int c = (int)a.ctrl_prev[sg * 16];
      const int it = a.occ_prev[(int64_t)sg * a.seg_cap + idx];
      if (idx < c && (it & 15) == 0) a.cnt_prev[(int64_t)(it >> 4) << a.cshift] = 0u;
    }
    return;
  }
  LEAN_TICK(tk0);
  const int i = blockIdx.x * 16 + li;
  const bool valid = i < a.n;
  const int ic = valid ? i : a.n - 1;
  const int4 rec = a.coords[ic];
  const __amdgpu_buffer_rsrc_t r_feats = dc_rsrc(a.feats, (uint32_t)((int64_t)a.n * C * IO_BYTES));
  float4 ff[T];
  {
    const uint32_t ro = ((uint32_t)ic * (uint32_t)C + (uint32_t)(4 * gq)) * (uint32_t)IO_BYTES;
    ff[0] = io_ldb4<0>(r_feats, ro); ff[1] = io_ldb4<16>(r_feats, ro); ff[2] = io_ldb4<32>(r_feats, ro); ff[3] = io_ldb4<48>(r_feats, ro);
    if constexpr (T > 4) { ff[4] = io_ldb4<64>(r_feats, ro); ff[5] = io_ldb4<80>(r_feats, ro); ff[6] = io_ldb4<96>(r_feats, ro); ff[7] = io_ldb4<112>(r_feats, ro); }
  }
  // this wave's slice of W in the A layout: lane (li, gq) holds W[16 tq + li][16 tt + 4 gq .. + 3]
  float4 wv[TB][T];
#pragma unroll
  for (int u = 0; u < TB; u++)
#pragma unroll
    for (int tt = 0; tt < T; tt++)
      wv[u][tt] = *reinterpret_cast<const float4 *>(&a.w_pre[(16 * (wave * TB + u) + li) * C + 16 * tt + 4 * gq]);
  // parameters per channel (one thread per channel)
  if (tid < C) {
    const int tc = tid % a.cg;
    s_par[tid] = a.pre_ln_w[tid]; s_par[C + tid] = a.pre_ln_b[tid];
    s_par[2 * C + tid] = a.w_pos[3 * tc + 0]; s_par[3 * C + tid] = a.w_pos[3 * tc + 1]; s_par[4 * C + tid] = a.w_pos[3 * tc + 2];
    s_par[5 * C + tid] = a.alpha ? a.alpha[tc] : 1.0f;
  }
  // slot insert by wave 0 (its lane group gq == 0 speaks for the 16 voxels)
  const bool ins = a.build && valid && gq == 0 && wave == 0;
  int cell = -1, base = 0, g_lead = 0, g_off = 0, g_size = 0;
  if (ins) {
    cell = cell_of(a.g, floordiv(rec.x, a.g.s), floordiv(rec.y, a.g.s), floordiv(rec.z, a.g.s), rec.w);
    if (cell < 0) atomicOr(&a.hdr[LINK_HDR_STATUS_ACC], 1);
  }
  if (wave == 0) {                                       // wave-uniform
    lean_groups(ins && cell >= 0, cell, lane, g_lead, g_off, g_size);
    if (ins && cell >= 0 && lane == g_lead) base = (int)atomicAdd(&a.cnt[(int64_t)cell << a.cshift], (unsigned)g_size);
  }
  LEAN_WAITALL();
  LEAN_TICK(tk1);
  // ---- contraction: fp16 hi | lo split of both operands, exact products, fp32 accumulation (as dc_premix_tile) ----
  floatx4 ac[TB];
#pragma unroll
  for (int u = 0; u < TB; u++) ac[u] = (floatx4){0.f, 0.f, 0.f, 0.f};
  uint2 bh[T], bl[T];
  float mx = 0.f;
#pragma unroll
  for (int tt = 0; tt < T; tt++) {
    dc_split4(ff[tt], bh[tt], bl[tt]);
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(ff[tt].x), fabsf(ff[tt].y)), fmaxf(fabsf(ff[tt].z), fabsf(ff[tt].w))));
#pragma unroll
    for (int u = 0; u < TB; u++)
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(wv[u][tt].x), fabsf(wv[u][tt].y)), fmaxf(fabsf(wv[u][tt].z), fabsf(wv[u][tt].w))));
  }
  if (__builtin_expect(!(LINK_COSX_EXACT && OP == LINK_OP_COSX) && !__any(!(mx < 32768.0f)), 1)) {
#pragma unroll
    for (int u = 0; u < TB; u++)
#pragma unroll
      for (int tt = 0; tt < T; tt += 2) {
        uint2 ah0, al0, ah1, al1;
        dc_split4(wv[u][tt], ah0, al0);
        dc_split4(wv[u][tt + 1], ah1, al1);
        ac[u] = dc_mfma_f16x2(al0, al1, bh[tt], bh[tt + 1], ac[u]);
        if constexpr (IO != 1) ac[u] = dc_mfma_f16x2(ah0, ah1, bl[tt], bl[tt + 1], ac[u]);
        ac[u] = dc_mfma_f16x2(ah0, ah1, bh[tt], bh[tt + 1], ac[u]);
      }
  } else {                                               // a value outside the fp16 split's range: the fp32 instruction
#pragma unroll
    for (int u = 0; u < TB; u++)
#pragma unroll
      for (int tt = 0; tt < T; tt++) {
        ac[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][tt].x, ff[tt].x, ac[u], 0, 0, 0);
        ac[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][tt].y, ff[tt].y, ac[u], 0, 0, 0);
        ac[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][tt].z, ff[tt].z, ac[u], 0, 0, 0);
        ac[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[u][tt].w, ff[tt].w, ac[u], 0, 0, 0);
      }
  }
#ifdef LEAN_DBG
  asm volatile("s_nop 0" ::"v"(ac[0][0]), "v"(ac[TB - 1][3]));
#endif
  LEAN_TICK(tk2);
  // ---- LayerNorm over the voxel's C channels: in-lane, the 4 lane groups, the 4 waves ----
  float s = 0.f;
#pragma unroll
  for (int u = 0; u < TB; u++) s += (ac[u][0] + ac[u][1]) + (ac[u][2] + ac[u][3]);
  s = dc_sum_groups(s);
  if (gq == 0) s_part[0][wave][li] = s;
  __syncthreads();
  const float mean = ((s_part[0][0][li] + s_part[0][1][li]) + (s_part[0][2][li] + s_part[0][3][li])) * (1.0f / C);
  float qq = 0.f;
#pragma unroll
  for (int u = 0; u < TB; u++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float d = ac[u][r] - mean;
      qq += d * d;
    }
  qq = dc_sum_groups(qq);
  if (gq == 0) s_part[1][wave][li] = qq;
  __syncthreads();
  const float rstd = 1.0f / sqrtf(((s_part[1][0][li] + s_part[1][1][li]) + (s_part[1][2][li] + s_part[1][3][li])) * (1.0f / C) + a.eps);
#ifdef LEAN_DBG
  asm volatile("s_nop 0" ::"v"(rstd));
#endif
  LEAN_TICK(tk3);
  float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
  if (a.coord_div != 1.0f) { x = x / a.coord_div; y = y / a.coord_div; z = z / a.coord_div; }
  float *xrow = a.X + (int64_t)ic * W;
#pragma unroll
  for (int u = 0; u < TB; u++) {
    const int cb = 16 * (wave * TB + u) + 4 * gq;        // this lane's four channels of the block
    const float4 lw = *reinterpret_cast<const float4 *>(&s_par[cb]), lb = *reinterpret_cast<const float4 *>(&s_par[C + cb]);
    const float4 q0 = *reinterpret_cast<const float4 *>(&s_par[2 * C + cb]), q1 = *reinterpret_cast<const float4 *>(&s_par[3 * C + cb]);
    const float4 q2 = *reinterpret_cast<const float4 *>(&s_par[4 * C + cb]), qa = *reinterpret_cast<const float4 *>(&s_par[5 * C + cb]);
    float th[4], sn[4], cs[4];
    th[0] = theta_of(x, y, z, q0.x, q1.x, q2.x, qa.x); th[1] = theta_of(x, y, z, q0.y, q1.y, q2.y, qa.y);
    th[2] = theta_of(x, y, z, q0.z, q1.z, q2.z, qa.z); th[3] = theta_of(x, y, z, q0.w, q1.w, q2.w, qa.w);
    bool big = false;
#pragma unroll
    for (int r = 0; r < 4; r++) big |= !(fabsf(th[r]) < 32768.0f);
    if (__builtin_expect(__any(big), 0)) {
#pragma unroll
      for (int r = 0; r < 4; r++) sincos_nocall(th[r], sn[r], cs[r]);
    } else {
#pragma unroll
      for (int r = 0; r < 4; r++) sincos_small(th[r], sn[r], cs[r]);
    }
    const float fv[4] = {(ac[u][0] - mean) * rstd * lw.x + lb.x, (ac[u][1] - mean) * rstd * lw.y + lb.y,
                         (ac[u][2] - mean) * rstd * lw.z + lb.z, (ac[u][3] - mean) * rstd * lw.w + lb.w};
#pragma unroll
    for (int pp = 0; pp < P; pp++) {
      float pv[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        if (pp == 2) pv[r] = fv[r] * th[r];
        else if ((pp == 0) == (OP == LINK_OP_SIN)) pv[r] = fv[r] * sn[r];
        else pv[r] = fv[r] * cs[r];
      }
      if (valid) *reinterpret_cast<float4 *>(xrow + pp * C + cb) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    }
  }
  LEAN_WAITALL();
  LEAN_TICK(tk4);
#ifdef LEAN_DBG
  if (tid == 0) {
    atomicAdd(&a.hdr[16], (int)((tk1 - tk0) >> 4)); atomicAdd(&a.hdr[17], (int)((tk2 - tk1) >> 4)); atomicAdd(&a.hdr[18], (int)((tk3 - tk2) >> 4));
    atomicAdd(&a.hdr[19], (int)((tk4 - tk3) >> 4)); atomicAdd(&a.hdr[20], 1);
  }
#endif
  if (!a.build || wave != 0) return;                     // wave-uniform
  // ---- the ranks are back: slot lists and work items, all inside wave 0 ----
  const int rank = __shfl(base, g_lead, 64) + g_off;
  const bool full = cell >= 0 && rank >= a.k;
  if (full) atomicOr(&a.hdr[LINK_HDR_STATUS_ACC], 2);
  const bool keep = cell >= 0 && !full;
  if (keep) a.list[(int64_t)cell * a.k + rank] = i;
  const bool item = keep && (rank & (LEAN_CH - 1)) == 0;
  const unsigned long long im = __ballot(item);
  if (im == 0) return;
  const int sg = blockIdx.x % LEAN_SEGS;
  int ibase = 0;
  if (lane == 0) ibase = (int)atomicAdd(&a.ctrl[sg * 16], (unsigned)__popcll(im));
  ibase = __builtin_amdgcn_readfirstlane(ibase);
  if (item) a.occ[(int64_t)sg * a.seg_cap + ibase + __popcll(im & ((1ull << lane) - 1ull))] = cell * 16 + rank / LEAN_CH;
}

// ---------------------------------------------------------------------------------------------
// The form WITHOUT the scratch matrix X (C <= 64; selected by flag -- elk_lean_dispatch.h says what it measured): launch 1 is the slot insert alone, launch 2 gathers
// the chunk's feature rows in id order and runs pre_mix + LayerNorm + theta + modulate on them itself, summing the tile in the
// accumulator layout (a DPP row holds the tile's 16 voxels) -- the n x P*C floats of X are neither written nor read (45 MB each
// way on the 59k-voxel cos_x stage); cos_x leaves fin * theta of each voxel in an n x C matrix for launch 3.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_lean_insert(const lean_args a) {
  const int tid = threadIdx.x, lane = tid & 63;
  if ((int)blockIdx.x >= a.nwg) {                        // clean-up workgroups (a.nwg counts 256-voxel workgroups here)
    const int q = ((int)blockIdx.x - a.nwg) * 256 + tid;
    const int sg = q & (LEAN_SEGS - 1), idx = q >> 4;
    if (idx < a.idx_cap_prev) {
      const int c = (int)a.ctrl_prev[sg * 16];
      const int it = a.occ_prev[(int64_t)sg * a.seg_cap + idx];
      if (idx < c && (it & 15) == 0) a.cnt_prev[(int64_t)(it >> 4) << a.cshift] = 0u;
    }
    return;
  }
  const int i = blockIdx.x * 256 + tid;
  const bool valid = i < a.n;
  int cell = -1;
  if (valid) {
    const int4 rec = a.coords[i];
    cell = cell_of(a.g, floordiv(rec.x, a.g.s), floordiv(rec.y, a.g.s), floordiv(rec.z, a.g.s), rec.w);
    if (cell < 0) atomicOr(&a.hdr[LINK_HDR_STATUS_ACC], 1);
  }
  int g_lead, g_off, g_size, base = 0;
  lean_groups(cell >= 0, cell, lane, g_lead, g_off, g_size);
  if (cell >= 0 && lane == g_lead) base = (int)atomicAdd(&a.cnt[(int64_t)cell << a.cshift], (unsigned)g_size);
  const int rank = __shfl(base, g_lead, 64) + g_off;
  const bool full = cell >= 0 && rank >= a.k;
  if (full) atomicOr(&a.hdr[LINK_HDR_STATUS_ACC], 2);
  const bool keep = cell >= 0 && !full;
  if (keep) a.list[(int64_t)cell * a.k + rank] = i;
  const bool item = keep && (rank & (LEAN_CH - 1)) == 0;
  const unsigned long long im = __ballot(item);
  if (im == 0) return;                                   // wave-uniform
  const int sg = (blockIdx.x * 4 + (tid >> 6)) % LEAN_SEGS;      // a list per wave here (64 voxels, as the other forms' workgroups)
  int ibase = 0;
  if (lane == 0) ibase = (int)atomicAdd(&a.ctrl[sg * 16], (unsigned)__popcll(im));
  ibase = __builtin_amdgcn_readfirstlane(ibase);
  if (item) a.occ[(int64_t)sg * a.seg_cap + ibase + __popcll(im & ((1ull << lane) - 1ull))] = cell * 16 + rank / LEAN_CH;
}

template <int C, int OP, int NB>
__global__ void __launch_bounds__(256, (C <= 32 ? 3 : 2)) k_lean_sums_pm(const lean_args a) {
  constexpr int T = C / 16, P = op_parts<OP>::value, W = P * C;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __shared__ __attribute__((aligned(16))) int ids_lds[4][LEAN_KMAX + 4];
  __shared__ int sorted_lds[4][LEAN_CH];
  float *ln_lds = reinterpret_cast<float *>(smem_raw + dc_wimg<C>::WIMG_BYTES);
  float *pw_lds = ln_lds + 2 * C;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  if (a.build && blockIdx.x == 0 && tid < LEAN_SEGS) a.ctrl_prev[tid * 16] = 0u;      // the next frame appends from zero
  int *ids = ids_lds[wave], *sorted = sorted_lds[wave];
  const int kl = a.k < 64 ? a.k : 64;
  const int j0 = blockIdx.x * 4 + wave;
  const int sg = j0 & (LEAN_SEGS - 1);
  const int nitem = (int)a.ctrl[sg * 16];
  int idx = j0 >> 4;
  int it_n = idx < a.idx_cap ? a.occ[(int64_t)sg * a.seg_cap + idx] : 0;
  const int step = (int)(gridDim.x * 4) >> 4;
  bool w_big = dc_stage_weights<C, 256, LINK_TILE_EXACT(OP)>(smem_raw, a.w_pre, a.pre_ln_w, a.pre_ln_b, a.w_pos, a.alpha, a.cg, tid);
  w_big = __syncthreads_or(w_big) != 0 || (LINK_COSX_EXACT && OP == LINK_OP_COSX);   // before any wave leaves; cos_x: exact contraction
  const unsigned short *wh = reinterpret_cast<const unsigned short *>(smem_raw);
  const __amdgpu_buffer_rsrc_t r_feats = dc_rsrc(a.feats, (uint32_t)((int64_t)a.n * C * IO_BYTES));
  for (; idx < nitem; idx += step) {
    const int it = it_n;
    if (idx + step < nitem) it_n = a.occ[(int64_t)sg * a.seg_cap + idx + step];
    const int64_t slot = (int64_t)sg * a.seg_cap + idx;
    const int cell = it >> 4, chunk = it & 15;
    const int32_t *lst = a.list + (int64_t)cell * a.k;
    const int spec = lane < kl ? lst[lane] : INT_MAX;
    const int cn = (int)a.cnt[(int64_t)cell << a.cshift];
    const int nc = cn < a.k ? cn : a.k;
    ids[lane] = lane < nc ? spec : INT_MAX;
    for (int l = 64 + lane; l < nc + 4; l += 64) ids[l] = l < nc ? lst[l] : INT_MAX;
    if (nc < 64 && lane < 4) ids[64 + lane] = INT_MAX;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int lo = chunk * LEAN_CH;
    for (int l0 = 0; l0 < nc; l0 += 64) {
      const int l = l0 + lane;
      const int my = l < nc ? ids[l] : INT_MAX;
      int rk = 0;
      for (int q = 0; q < nc; q += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(&ids[q]);
        rk += (v.x < my) + (v.y < my) + (v.z < my) + (v.w < my);
      }
      if (l < nc && rk >= lo && rk < lo + LEAN_CH) sorted[rk - lo] = my;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nch = nc - lo < LEAN_CH ? nc - lo : LEAN_CH;
    if (lane < nch) {                                    // the chunk's records in id order for launch 3
      const int id = sorted[lane];
      int4 myrec = a.coords[id];
      myrec.w = id;
      a.rec2[slot * LEAN_CH + lane] = myrec;
    }
    float4 acc[P][T];
#pragma unroll
    for (int pp = 0; pp < P; pp++)
#pragma unroll
      for (int tp = 0; tp < T; tp++) acc[pp][tp] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = 0; t0 < nch; t0 += 16) {
      const int m = t0 + li;
      const bool valid = m < nch;
      const int id = sorted[valid ? m : 0];
      const int4 rec = a.coords[id];
      float4 ff[T];
      {
        const uint32_t ro = ((uint32_t)id * (uint32_t)C + (uint32_t)(4 * gq)) * (uint32_t)IO_BYTES;
        ff[0] = io_ldb4<0>(r_feats, ro);
        if constexpr (T > 1) ff[1] = io_ldb4<16>(r_feats, ro);
        if constexpr (T > 2) { ff[2] = io_ldb4<32>(r_feats, ro); ff[3] = io_ldb4<48>(r_feats, ro); }
        if constexpr (T > 4) { ff[4] = io_ldb4<64>(r_feats, ro); ff[5] = io_ldb4<80>(r_feats, ro); ff[6] = io_ldb4<96>(r_feats, ro); ff[7] = io_ldb4<112>(r_feats, ro); }
      }
      floatx4 ac[T];
      dc_premix_tile<C, LINK_TILE_EXACT(OP)>(wh, a.w_pre, w_big, li, gq, ff, ac);
      float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
      if (a.coord_div != 1.0f) { x = x / a.coord_div; y = y / a.coord_div; z = z / a.coord_div; }
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
        if (__builtin_expect(__any(big), 0)) {
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
      const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + a.eps);
#pragma unroll
      for (int tp = 0; tp < T; tp++) {
        const int tb = tp % NB;
        const float4 lw = *reinterpret_cast<const float4 *>(&ln_lds[16 * tp + 4 * gq]);
        const float4 lb = *reinterpret_cast<const float4 *>(&ln_lds[C + 16 * tp + 4 * gq]);
        float th1[4], sn1[4], cs1[4];
        if constexpr (NB == T) trig(tp, th1, sn1, cs1);
        const float fv[4] = {(ac[tp][0] - mean) * rstd * lw.x + lb.x, (ac[tp][1] - mean) * rstd * lw.y + lb.y,
                             (ac[tp][2] - mean) * rstd * lw.z + lb.z, (ac[tp][3] - mean) * rstd * lw.w + lb.w};
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
          if (pp == 2 && valid)                          // cos_x: fin * theta of this voxel, for launch 3 (linkunet.py:176)
            *reinterpret_cast<float4 *>(a.X + (int64_t)id * C + 16 * tp + 4 * gq) = make_float4(pv[0], pv[1], pv[2], pv[3]);
          // the tile's 16 voxels sit in one DPP row: all-reduce along it (lanes beyond the chunk add zero)
          acc[pp][tp].x += grp_sum<16>(valid ? pv[0] : 0.f); acc[pp][tp].y += grp_sum<16>(valid ? pv[1] : 0.f);
          acc[pp][tp].z += grp_sum<16>(valid ? pv[2] : 0.f); acc[pp][tp].w += grp_sum<16>(valid ? pv[3] : 0.f);
        }
      }
    }
    if (li == 0) {
      float *srow = a.S + ((int64_t)cell * a.kch + chunk) * W + 4 * gq;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
#pragma unroll
        for (int tp = 0; tp < T; tp++) *reinterpret_cast<float4 *>(srow + pp * C + 16 * tp) = acc[pp][tp];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// launch 2: chunk sums
// ---------------------------------------------------------------------------------------------
template <int C, int P>
__global__ void __launch_bounds__(64 * LEAN_IW) k_lean_sums(const lean_args a) {
  constexpr int W = P * C, LPR = C / 4, G = 64 / LPR;
  __shared__ __attribute__((aligned(16))) int ids_lds[LEAN_IW][LEAN_KMAX + 4];
  __shared__ int sorted_lds[LEAN_IW][LEAN_CH];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & (LPR - 1), grp = lane / LPR;
  if (a.build && blockIdx.x == 0 && tid < LEAN_SEGS) a.ctrl_prev[tid * 16] = 0u;      // the next frame appends from zero
  int *ids = ids_lds[wave], *sorted = sorted_lds[wave];
  const int kl = a.k < 64 ? a.k : 64;
  const int j0 = blockIdx.x * LEAN_IW + wave;
  const int sg = j0 & (LEAN_SEGS - 1);                   // the stride below is a multiple of 16: a wave stays with its list
  const int nitem = (int)a.ctrl[sg * 16];
  int idx = j0 >> 4;
  int it_n = idx < a.idx_cap ? a.occ[(int64_t)sg * a.seg_cap + idx] : 0;      // garbage beyond nitem: never used
  const int step = (int)(gridDim.x * LEAN_IW) >> 4;
  for (; idx < nitem; idx += step) {
    const int it = it_n;
    if (idx + step < nitem) it_n = a.occ[(int64_t)sg * a.seg_cap + idx + step];
    const int64_t slot = (int64_t)sg * a.seg_cap + idx;
    const int cell = it >> 4, chunk = it & 15;
    const int32_t *lst = a.list + (int64_t)cell * a.k;
    const int spec = lane < kl ? lst[lane] : INT_MAX;    // the first 64 ids without waiting for the count
    const int cn = (int)a.cnt[(int64_t)cell << a.cshift];
    const int nc = cn < a.k ? cn : a.k;
    ids[lane] = lane < nc ? spec : INT_MAX;
    for (int l = 64 + lane; l < nc + 4; l += 64) ids[l] = l < nc ? lst[l] : INT_MAX;    // padded to whole int4 pieces
    if (nc < 64 && lane < 4) ids[64 + lane] = INT_MAX;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    // rank of every id among the cell's ids (unique: voxel ids) by counting; the chunk's 32 go to their places
    const int lo = chunk * LEAN_CH;
    for (int l0 = 0; l0 < nc; l0 += 64) {
      const int l = l0 + lane;
      const int my = l < nc ? ids[l] : INT_MAX;
      int rk = 0;
      for (int q = 0; q < nc; q += 4) {
        const int4 v = *reinterpret_cast<const int4 *>(&ids[q]);
        rk += (v.x < my) + (v.y < my) + (v.z < my) + (v.w < my);
      }
      if (l < nc && rk >= lo && rk < lo + LEAN_CH) sorted[rk - lo] = my;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int nch = nc - lo < LEAN_CH ? nc - lo : LEAN_CH;
    // the chunk's records in id order for launch 3 (no dependent coordinate fetch there)
    int4 myrec = make_int4(0, 0, 0, 0);
    if (lane < nch) {
      const int id = sorted[lane];
      myrec = a.coords[id];
      myrec.w = id;
    }
    float4 acc[P];
#pragma unroll
    for (int pp = 0; pp < P; pp++) acc[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = 4;                                 // rows in flight per lane group
    for (int m0 = 0; m0 < nch; m0 += U * G) {
      float4 v[U][P];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const int m = m0 + u * G + grp;
        const int id = sorted[m < nch ? m : 0];
        const float *row = a.X + (int64_t)id * W + 4 * li;
#pragma unroll
        for (int pp = 0; pp < P; pp++) v[u][pp] = *reinterpret_cast<const float4 *>(row + pp * C);
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const bool ok = m0 + u * G + grp < nch;
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          acc[pp].x += ok ? v[u][pp].x : 0.f; acc[pp].y += ok ? v[u][pp].y : 0.f;
          acc[pp].z += ok ? v[u][pp].z : 0.f; acc[pp].w += ok ? v[u][pp].w : 0.f;
        }
      }
    }
    if (lane < nch) a.rec2[slot * LEAN_CH + lane] = myrec;
#pragma unroll
    for (int pp = 0; pp < P; pp++) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        acc[pp].x += __shfl_xor(acc[pp].x, o, 64); acc[pp].y += __shfl_xor(acc[pp].y, o, 64);
        acc[pp].z += __shfl_xor(acc[pp].z, o, 64); acc[pp].w += __shfl_xor(acc[pp].w, o, 64);
      }
      if (grp == 0) *reinterpret_cast<float4 *>(a.S + ((int64_t)cell * a.kch + chunk) * W + pp * C + 4 * li) = acc[pp];
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// launch 3: neighbour sum + de-modulate + LayerNorm
// ---------------------------------------------------------------------------------------------
template <int C, int OP, int R>
#ifndef LEAN_GATHER_WAVES
#define LEAN_GATHER_WAVES 1          /* minimum waves per SIMD the gather kernel is compiled for (registers <= 512 / that) */
#endif
__global__ void __launch_bounds__(64 * LEAN_IW, LEAN_GATHER_WAVES) k_lean_gather(const lean_args a) {
  constexpr int P = op_parts<OP>::value, W = P * C, LPR = C / 4, G = 64 / LPR, R3 = R * R * R;
  constexpr int LO = -((R + 1) / 2) + 1;               // nn/utils/kernel.py:21
  constexpr int MAXROWS = R3 * ((LEAN_KMAX + LEAN_CH - 1) / LEAN_CH);
  constexpr int NT = (R3 + G - 1) / G;                 // neighbours per lane group
  constexpr int UBMAX = P == 3 ? 4 : 6;                // first-chunk rows in flight per lane group (9 / 14: -1 us on the 10k-30k-voxel
                                                       // detection stages, +2-5 us on every frame above 50k voxels: registers)
  constexpr int UB = NT < UBMAX ? NT : UBMAX;
  __shared__ int rows_lds[LEAN_IW][MAXROWS];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & (LPR - 1), grp = lane / LPR;
  if (a.build && blockIdx.x == 0 && tid == 0) {          // publish the status bits the insert collected
    a.hdr[LINK_HDR_STATUS] = a.hdr[LINK_HDR_STATUS_ACC];
    a.hdr[LINK_HDR_STATUS_ACC] = 0;
    a.hdr[LINK_HDR_NVALID] = a.n;
  }
  LEAN_TICK(tg0);
  const int j0 = blockIdx.x * LEAN_IW + wave;
  const int sg = j0 & (LEAN_SEGS - 1);
  const int nitem = (int)a.ctrl[sg * 16];
  int idx = j0 >> 4;
  int it_n = idx < a.idx_cap ? a.occ[(int64_t)sg * a.seg_cap + idx] : 0;
  const int step = (int)(gridDim.x * LEAN_IW) >> 4;
  const int ch0 = 4 * li;
  // The grid is sized by the voxel count (the item count lives on the device), so most waves find no item -- 50-90 % on the
  // LiDAR stage frames -- and what a wave does before it knows is paid by all of them.  One thread per channel fetches the
  // channel's six theta / LayerNorm values (6 loads per workgroup instead of 18 gathers per wave), a workgroup without an item
  // leaves at the first barrier, the others pass the values through LDS.  (Measured: no change in the launch's duration -- with
  // the per-voxel loop removed it drops from 18.4 to 10.8 us on the 150k-voxel stage, with only the loop's stores removed it
  // stays at 17.5: what the loop costs is its registers -- 108-134 per lane, four waves per SIMD, so 8k items run as two rounds
  // of a ~8 us dependency chain -- not its arithmetic, its stores or these fetches.)
  __shared__ __attribute__((aligned(16))) float par_lds[6 * C];        // ln weight | ln bias | w0 | w1 | w2 | alpha, by channel
  float pv_[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 1.0f};
  if (tid < C) {
    const int tc = tid % a.cg;
    pv_[0] = a.ln_w[tid]; pv_[1] = a.ln_b[tid];
    pv_[2] = a.w_pos[3 * tc + 0]; pv_[3] = a.w_pos[3 * tc + 1]; pv_[4] = a.w_pos[3 * tc + 2];
    if (a.alpha) pv_[5] = a.alpha[tc];
  }
  if (!__syncthreads_or(idx < nitem)) return;
  if (tid < C) {
#pragma unroll
    for (int q = 0; q < 6; q++) par_lds[q * C + tid] = pv_[q];
  }
  __syncthreads();
  const float4 gw = *reinterpret_cast<const float4 *>(&par_lds[ch0]), gb = *reinterpret_cast<const float4 *>(&par_lds[C + ch0]);
  const float4 p0 = *reinterpret_cast<const float4 *>(&par_lds[2 * C + ch0]), p1 = *reinterpret_cast<const float4 *>(&par_lds[3 * C + ch0]);
  const float4 p2 = *reinterpret_cast<const float4 *>(&par_lds[4 * C + ch0]), p3 = *reinterpret_cast<const float4 *>(&par_lds[5 * C + ch0]);
  const float w0[4] = {p0.x, p0.y, p0.z, p0.w}, w1[4] = {p1.x, p1.y, p1.z, p1.w};
  const float w2[4] = {p2.x, p2.y, p2.z, p2.w}, al[4] = {p3.x, p3.y, p3.z, p3.w};
  int *rows = rows_lds[wave];
  const int d0 = a.g.dim[0], d1 = a.g.dim[1], d2 = a.g.dim[2], d3 = a.g.dim[3];
  for (; idx < nitem; idx += step) {
    const int it = it_n;
    if (idx + step < nitem) it_n = a.occ[(int64_t)sg * a.seg_cap + idx + step];
    LEAN_WAITALL();
    LEAN_TICK(tg1);
    const int64_t slot = (int64_t)sg * a.seg_cap + idx;
    const int cell = it >> 4, chunk = it & 15;
    const int lo = chunk * LEAN_CH;
    // everything the item needs is addressed from (cell, slot): the chunk's records, the counts of the r^3 neighbour cells
    // and -- without waiting for those counts -- the first chunk row of each of them
    const int4 myrec = a.rec2[slot * LEAN_CH + (lane & (LEAN_CH - 1))];
    const int cn_own = (int)a.cnt[(int64_t)cell << a.cshift];
    const int ub = cell % d3, t1 = cell / d3;
    const int uz = t1 % d2, t2 = t1 / d2;
    const int uy = t2 % d1, ux = t2 / d1;
    auto nb_cell = [&](int t) -> int {
      const int ox = LO + t % R, oy = LO + (t / R) % R, oz = LO + t / (R * R);
      const unsigned vx = (unsigned)(ux + ox), vy = (unsigned)(uy + oy), vz = (unsigned)(uz + oz);
      return (t < R3 && vx < (unsigned)d0 && vy < (unsigned)d1 && vz < (unsigned)d2) ? (((int)vx * d1 + (int)vy) * d2 + (int)vz) * d3 + ub : -1;
    };
    const int nbc = nb_cell(lane);
    int cn = nbc >= 0 ? (int)a.cnt[(int64_t)nbc << a.cshift] : 0;
    float4 acc[P];
#pragma unroll
    for (int pp = 0; pp < P; pp++) acc[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    cn = cn < a.k ? cn : a.k;
#pragma unroll 1
    for (int b0 = 0; b0 < NT; b0 += UB) {
      float4 v[UB][P];
      int tt[UB];
#pragma unroll
      for (int u = 0; u < UB; u++) {
        tt[u] = (b0 + u) * G + grp;
        const int c_ = (b0 + u < NT) ? nb_cell(tt[u]) : -1;
        const float *row = a.S + (int64_t)(c_ >= 0 ? c_ : cell) * a.kch * W + ch0;
#pragma unroll
        for (int pp = 0; pp < P; pp++) v[u][pp] = *reinterpret_cast<const float4 *>(row + pp * C);
      }
#pragma unroll
      for (int u = 0; u < UB; u++) {
        const int cn_t = __shfl(cn, tt[u] < R3 ? tt[u] : 0, 64);
        const bool ok = (b0 + u < NT) && tt[u] < R3 && cn_t > 0;
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          acc[pp].x += ok ? v[u][pp].x : 0.f; acc[pp].y += ok ? v[u][pp].y : 0.f;
          acc[pp].z += ok ? v[u][pp].z : 0.f; acc[pp].w += ok ? v[u][pp].w : 0.f;
        }
      }
    }
    LEAN_WAITALL();
    LEAN_TICK(tg2);
    // neighbours of more than 32 voxels: their further chunk rows, in (neighbour, chunk) order
    const int nrow = (cn + LEAN_CH - 1) / LEAN_CH;
    const int nx = nrow > 1 ? nrow - 1 : 0;
    int pre = nx, den_i = cn;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const int u = __shfl_up(pre, o, 64);
      if (lane >= o) pre += u;
      den_i += __shfl_xor(den_i, o, 64);
    }
    const int NR = __shfl(pre, 31, 64);
    const float den = (float)__shfl(den_i, 0, 64);
    if (NR > 0) {                                        // wave-uniform
      for (int c = 0; c < nx; c++) rows[pre - nx + c] = nbc * a.kch + 1 + c;
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      constexpr int U = 4;
      for (int m0 = 0; m0 < NR; m0 += U * G) {
        float4 v[U][P];
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int m = m0 + u * G + grp;
          const float *row = a.S + (int64_t)rows[m < NR ? m : 0] * W + ch0;
#pragma unroll
          for (int pp = 0; pp < P; pp++) v[u][pp] = *reinterpret_cast<const float4 *>(row + pp * C);
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
          const bool ok = m0 + u * G + grp < NR;
#pragma unroll
          for (int pp = 0; pp < P; pp++) {
            acc[pp].x += ok ? v[u][pp].x : 0.f; acc[pp].y += ok ? v[u][pp].y : 0.f;
            acc[pp].z += ok ? v[u][pp].z : 0.f; acc[pp].w += ok ? v[u][pp].w : 0.f;
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    LEAN_WAITALL();
    LEAN_TICK(tg3);
    const int nc = cn_own < a.k ? cn_own : a.k;
    const int nch = nc - lo < LEAN_CH ? nc - lo : LEAN_CH;
    float A[P][4];
#pragma unroll
    for (int pp = 0; pp < P; pp++) {
#pragma unroll
      for (int o = LPR; o < 64; o <<= 1) {
        acc[pp].x += __shfl_xor(acc[pp].x, o, 64); acc[pp].y += __shfl_xor(acc[pp].y, o, 64);
        acc[pp].z += __shfl_xor(acc[pp].z, o, 64); acc[pp].w += __shfl_xor(acc[pp].w, o, 64);
      }
      A[pp][0] = acc[pp].x / den; A[pp][1] = acc[pp].y / den; A[pp][2] = acc[pp].z / den; A[pp][3] = acc[pp].w / den;   // utils.py:80
    }
#ifdef LEAN_DBG
    asm volatile("s_nop 0" ::"v"(A[0][0]), "v"(A[P - 1][3]));
#endif
    LEAN_TICK(tg4);
    // ---- the chunk's voxels, G at a time: the record of voxel m sits in lane m ----
    const int steps = (nch + G - 1) / G;
    float4 xl_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if (OP == LINK_OP_COSX) {
      const int id0 = __shfl(myrec.w, grp < nch ? grp : 0, 64);
      xl_n = *reinterpret_cast<const float4 *>(a.X + (int64_t)id0 * a.xl_stride + a.xl_off + ch0);
    }
    LEAN_WAITALL();
    LEAN_TICK(tg4b);
    for (int mi = 0; mi < steps; mi++) {
      const int m = mi * G + grp;
      const bool ok = m < nch;
      const int src = ok ? m : 0;
      const int rx = __shfl(myrec.x, src, 64), ry = __shfl(myrec.y, src, 64), rz = __shfl(myrec.z, src, 64);
      const int id = __shfl(myrec.w, src, 64);
      const float4 xl = xl_n;
      if (OP == LINK_OP_COSX && mi + 1 < steps) {
        const int idn = __shfl(myrec.w, m + G < nch ? m + G : 0, 64);
        xl_n = *reinterpret_cast<const float4 *>(a.X + (int64_t)idn * a.xl_stride + a.xl_off + ch0);
      }
      float x = (float)rx, y = (float)ry, z = (float)rz;
      if (a.coord_div != 1.0f) { x = x / a.coord_div; y = y / a.coord_div; z = z / a.coord_div; }
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
      const float xlv[4] = {xl.x, xl.y, xl.z, xl.w};
      float nvv[4], s = 0.f;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float va;
        if (OP == LINK_OP_SIN) va = __fsub_rn(__fmul_rn(A[0][q], cs[q]), __fmul_rn(A[1][q], sn[q]));
        else va = __fadd_rn(__fmul_rn(A[0][q], cs[q]), __fmul_rn(A[1][q], sn[q]));
        if (OP == LINK_OP_COSX) va = __fadd_rn(va, __fsub_rn(A[P - 1][q], xlv[q]));     // linkunet.py:176; X's third part is fin * theta
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
      const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + a.eps);
      if (ok) {
        const float4 o = make_float4((nvv[0] - mean) * rstd * gw.x + gb.x, (nvv[1] - mean) * rstd * gw.y + gb.y,
                                     (nvv[2] - mean) * rstd * gw.z + gb.z, (nvv[3] - mean) * rstd * gw.w + gb.w);
        io_st4_ptr(a.out, (int64_t)id * C + ch0, o);
      }
    }
#ifdef LEAN_DBG
    {
      LEAN_TICK(tg4c);
      LEAN_WAITALL();
      LEAN_TICK(tg5);
      if (lane == 0) {
        atomicAdd(&a.hdr[24], (int)((tg1 - tg0) >> 8)); atomicAdd(&a.hdr[25], (int)((tg2 - tg1) >> 8)); atomicAdd(&a.hdr[26], (int)((tg3 - tg2) >> 8));
        atomicAdd(&a.hdr[27], (int)((tg4 - tg3) >> 8)); atomicAdd(&a.hdr[28], (int)((tg4b - tg4) >> 8)); atomicAdd(&a.hdr[30], (int)((tg4c - tg4b) >> 8));
        atomicAdd(&a.hdr[31], (int)((tg5 - tg4c) >> 8)); atomicAdd(&a.hdr[32], steps); atomicAdd(&a.hdr[29], 1);
      }
    }
#endif
  }
}

}  // namespace DC_IO_NS
