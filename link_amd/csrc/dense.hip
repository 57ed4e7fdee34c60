// link_amd/csrc/dense.hip -- section E of include/link_amd.h: R_core of ELKBlock.forward
// (segmentation/core/models/semantic_kitti/linkunet.py:124-185; detection/det3d/models/utils/
// ts_elk.py:144-230) on the DENSE-CELL layout, for frames whose block grid is mostly occupied.
//
// What the reference does per call (utils.py:45-52,65-82): sphash, torch.unique, two hash-table builds,
// spcount, an atomic scatter-add, a 27-way gather.  Here the block table is indexed by grid cell:
//   k_dc_premix_insert   streaming: fin = LayerNorm(F Wpre^T) on f32 MFMA with the next tile's rows in
//                        flight, and -- in the shadow of the MFMAs -- one atomic per voxel that appends
//                        (x,y,z,id) to its cell's slot list.  That IS the index build.
//   k_dc_modsum          a 16-lane group per cell: slot list -> ascending voxel id (sorting network) ->
//                        theta / sincos / modulate / sum in registers -> ONE row store.  Every interior
//                        cell is written (empty ones as zeros), so the table never needs clearing.
//   k_dc_gather          r^3 box filter over the padded grid.  A workgroup owns a tile of columns and
//                        marches z; each z-plane of the haloed tile is brought in ONCE by LDS-DMA
//                        (global_load_lds_dwordx4, 3-deep ring, counted vmcnt, one raw barrier per plane),
//                        the xy sum is read from LDS, the z sum is a register ring.  2.25 row fetches per
//                        cell instead of 27; all addresses are arithmetic, no lookups.
// The per-voxel de-modulation + LayerNorm is section C's k_voxel_demod_ln_g fed with vrec/vcell.
#include "dense_gather.h"

using namespace link;

// launch geometry of the unfused stages (fixed: they are the reference points the fused kernels are tested against)
static constexpr int DC_PREMIX_WGS = 512, DC_MODSUM_WGS = 768;

extern "C" int64_t link_dc_grid_from(const link_grid_t *grid, int32_t k, link_dc_grid_t *out) {
  if (!grid || !out || grid->s <= 0) return -1;
  out->s = grid->s;
  int64_t vp = 1;
  for (int a = 0; a < 4; a++) {
    out->lo[a] = grid->lo[a];
    out->dim[a] = grid->dim[a];
    if (grid->dim[a] <= 0) return -1;
    if (a < 3) {
      out->pdim[a] = grid->dim[a] + 2;
      vp *= out->pdim[a];
    } else {
      vp *= grid->dim[a];
    }
    if (vp >= (1LL << 30)) return -1;
  }
  if (k <= 0) {
    int64_t s3 = (int64_t)grid->s * grid->s * grid->s;
    if (s3 > (1 << 20)) return -1;
    k = (int32_t)s3;
  }
  if (vp * (int64_t)k >= (1LL << 31)) return -1;
  out->k = k;
  out->vp = vp;
  return vp;
}

// ---------------------------------------------------------------------------------------------
// pre_mix + LayerNorm + slot insert
// ---------------------------------------------------------------------------------------------
// MFMA schedule of k_premix_ln_tlp (elk.hip): D = W * F^T on v_mfma_f32_16x16x4_f32, a lane ends up with
// 16 channels of one voxel.  Here a wave loops over tiles with the NEXT tile's feature rows already in
// flight, consecutive MFMAs go to different accumulators, and lanes 0..15 append their voxel to its cell's
// slot list: the atomic of tile t goes out before the MFMAs of t and its result is stored at the top of
// t+1, so nothing ever waits for a store or an atomic to come back.
template <int C, bool INSERT>
__global__ void __launch_bounds__(256) k_dc_premix_insert(
    const float *__restrict__ feats, const int4 *__restrict__ coords, const float *__restrict__ w_pre,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, int64_t n, float eps, link_dc_grid_t g,
    float *__restrict__ fin, uint32_t *__restrict__ cnt, int4 *__restrict__ slots, int4 *__restrict__ vrec,
    int32_t *__restrict__ vcell, int32_t *__restrict__ hdr) {
  constexpr int T = C / 16;
  constexpr int LDW = C + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  float *ln_lds = w_lds + C * LDW;                   // [ln_w | ln_b]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, gq = lane >> 4;
  {                                                  // stage W: all loads first, one wait
    // all loads first, ONE wait, then the LDS writes -- no predicate around the writes (hipcc turns a
    // predicated write into load / wait / write per iteration: four dependent round trips at C = 64)
    constexpr int NF4 = C * C / 4;                     // float4 pieces of W
    constexpr int NV = (NF4 + 255) / 256;
    float4 wv[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int e = (i * 256 + tid) * 4;
      wv[i] = *reinterpret_cast<const float4 *>(&w_pre[(NF4 % 256 == 0 || e < C * C) ? e : 0]);
    }
#pragma unroll
    for (int i = 0; i < NV; i++) {
      int e = (i * 256 + tid) * 4;
      if (NF4 % 256 != 0 && e >= C * C) e = 0;         // C = 16: surplus lanes rewrite piece 0 with its own value
      const int r = e / C, col = e - r * C;
      *reinterpret_cast<float4 *>(&w_lds[r * LDW + col]) = (NF4 % 256 == 0 || (i * 256 + tid) * 4 < C * C) ? wv[i] : *reinterpret_cast<const float4 *>(&w_pre[0]);
    }
    if (tid < C) ln_lds[tid] = ln_w[tid];
    else if (tid < 2 * C) ln_lds[tid] = ln_b[tid - C];
  }
  if (INSERT && blockIdx.x == 0 && tid == 0) hdr[LINK_HDR_NVALID] = (int32_t)n;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * C * 4));
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_vrec = dc_rsrc(vrec, (uint32_t)(n * 16));
  const __amdgpu_buffer_rsrc_t r_vcell = dc_rsrc(vcell, (uint32_t)(n * 4));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const int64_t tiles = (n + 15) / 16;
  const int64_t stride = (int64_t)gridDim.x * 4;
  int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile >= tiles) return;
  float4 f[T];
  int4 rc = make_int4(0, 0, 0, 0);
  {
    const int64_t v = tile * 16 + li;
    const int64_t vl = v < n ? v : n - 1;
#pragma unroll
    for (int t = 0; t < T; t++) f[t] = *reinterpret_cast<const float4 *>(&feats[vl * C + 16 * t + 4 * gq]);
    if (INSERT) rc = coords[vl];
  }
  // deferred insert of the previous tile (its atomic has been in flight for a whole iteration)
  uint32_t p_voff = DC_OOB;                           // byte offset of the previous tile's vrec entry, OOB = none
  int p_cell = 0, p_rank = 0;
  int4 p_rec = make_int4(0, 0, 0, 0);
  for (; tile < tiles; tile += stride) {
    const int64_t v = tile * 16 + li;
    const bool ok = v < n;
    float4 fn[T];
    int4 rn = make_int4(0, 0, 0, 0);
    {
      const int64_t vn = (tile + stride) * 16 + li;
      const int64_t vl = vn < n ? vn : n - 1;
#pragma unroll
      for (int t = 0; t < T; t++) fn[t] = *reinterpret_cast<const float4 *>(&feats[vl * C + 16 * t + 4 * gq]);
      if (INSERT) rn = coords[vl];
    }
    if (INSERT) {
      {                                               // finish the previous tile's insert
        const bool full = p_rank >= g.k;
        const bool keep = p_voff != DC_OOB && p_cell != 0 && !full;
        if (p_voff != DC_OOB && p_cell != 0 && full) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 2);
        st16i(r_slots, keep ? dc_slot(g, p_cell, p_rank) * 16u : DC_OOB, p_rec);
        st16i(r_vrec, p_voff, p_rec);
        st4i(r_vcell, p_voff == DC_OOB ? DC_OOB : (p_voff >> 2), keep ? p_cell : 0);
      }
      const unsigned ux = (unsigned)(floordiv(rc.x, g.s) - g.lo[0]), uy = (unsigned)(floordiv(rc.y, g.s) - g.lo[1]);
      const unsigned uz = (unsigned)(floordiv(rc.z, g.s) - g.lo[2]), ub = (unsigned)(rc.w - g.lo[3]);
      const bool inside = ux < (unsigned)g.dim[0] && uy < (unsigned)g.dim[1] && uz < (unsigned)g.dim[2] &&
                          ub < (unsigned)g.dim[3];
      const bool mine = ok && gq == 0;
      if (mine && !inside) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 1);
      p_cell = (mine && inside) ? dc_cell(g, (int)ux, (int)uy, (int)uz, (int)ub) : 0;   // cell 0 is a border cell
      p_rank = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_cnt, p_cell ? (uint32_t)p_cell * 4u : DC_OOB, 0, 0);
      p_voff = mine ? (uint32_t)v * 16u : DC_OOB;
      p_rec = make_int4(rc.x, rc.y, rc.z, (int)v);
    }
    floatx4 acc[T];
#pragma unroll
    for (int tp = 0; tp < T; tp++) acc[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; t++) {
      float4 a[T];
#pragma unroll
      for (int tp = 0; tp < T; tp++)
        a[tp] = *reinterpret_cast<const float4 *>(&w_lds[(16 * tp + li) * LDW + 16 * t + 4 * gq]);
#pragma unroll
      for (int tp = 0; tp < T; tp++) acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].x, f[t].x, acc[tp], 0, 0, 0);
#pragma unroll
      for (int tp = 0; tp < T; tp++) acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].y, f[t].y, acc[tp], 0, 0, 0);
#pragma unroll
      for (int tp = 0; tp < T; tp++) acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].z, f[t].z, acc[tp], 0, 0, 0);
#pragma unroll
      for (int tp = 0; tp < T; tp++) acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].w, f[t].w, acc[tp], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int tp = 0; tp < T; tp++) s += (acc[tp][0] + acc[tp][1]) + (acc[tp][2] + acc[tp][3]);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int tp = 0; tp < T; tp++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float d = acc[tp][r] - mean;
        q += d * d;
      }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    const float rstd = 1.0f / sqrtf(q * (1.0f / C) + eps);
    const uint32_t rowb = ok ? (uint32_t)v * (uint32_t)(C * 4) : DC_OOB;
#pragma unroll
    for (int tp = 0; tp < T; tp++) {
      const float4 lw = *reinterpret_cast<const float4 *>(&ln_lds[16 * tp + 4 * gq]);
      const float4 lb = *reinterpret_cast<const float4 *>(&ln_lds[C + 16 * tp + 4 * gq]);
      float4 o;
      o.x = (acc[tp][0] - mean) * rstd * lw.x + lb.x;
      o.y = (acc[tp][1] - mean) * rstd * lw.y + lb.y;
      o.z = (acc[tp][2] - mean) * rstd * lw.z + lb.z;
      o.w = (acc[tp][3] - mean) * rstd * lw.w + lb.w;
      st16(r_fin, ok ? rowb + (uint32_t)((16 * tp + 4 * gq) * 4) : DC_OOB, o);
    }
#pragma unroll
    for (int t = 0; t < T; t++) f[t] = fn[t];
    rc = rn;
  }
  if (INSERT) {                                       // the last tile's insert
    const bool full = p_rank >= g.k;
    const bool keep = p_voff != DC_OOB && p_cell != 0 && !full;
    if (p_voff != DC_OOB && p_cell != 0 && full) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 2);
    st16i(r_slots, keep ? dc_slot(g, p_cell, p_rank) * 16u : DC_OOB, p_rec);
    st16i(r_vrec, p_voff, p_rec);
    st4i(r_vcell, p_voff == DC_OOB ? DC_OOB : (p_voff >> 2), keep ? p_cell : 0);
  }
}

template <int C>
static int launch_dc_premix(const float *feats, const int32_t *coords, const float *w_pre, const float *ln_w,
                            const float *ln_b, int64_t n, float eps, const link_dc_grid_t &g, bool insert,
                            float *fin, uint32_t *cnt, int32_t *slots, int32_t *vrec, int32_t *vcell, int32_t *hdr,
                            hipStream_t st) {
  const size_t lds = ((size_t)C * (C + 4) + 2 * C) * sizeof(float);
  const int64_t tiles = (n + 15) / 16;
  // equal tiles per wave: waves = ceil(tiles / tiles_per_wave) with tiles_per_wave from the workgroup cap
  int64_t cap = (int64_t)DC_PREMIX_WGS * 4;
  int64_t tpw = (tiles + cap - 1) / cap;
  int64_t waves = (tiles + tpw - 1) / tpw;
  int64_t wgs = (waves + 3) / 4;
  if (lds > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_premix_insert<C, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_premix_insert<C, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  if (insert)
    hipLaunchKernelGGL((k_dc_premix_insert<C, true>), dim3((unsigned)wgs), dim3(256), lds, st, feats,
                       reinterpret_cast<const int4 *>(coords), w_pre, ln_w, ln_b, n, eps, g, fin, cnt,
                       reinterpret_cast<int4 *>(slots), reinterpret_cast<int4 *>(vrec), vcell, hdr);
  else
    hipLaunchKernelGGL((k_dc_premix_insert<C, false>), dim3((unsigned)wgs), dim3(256), lds, st, feats,
                       reinterpret_cast<const int4 *>(coords), w_pre, ln_w, ln_b, n, eps, g, fin, cnt,
                       reinterpret_cast<int4 *>(slots), reinterpret_cast<int4 *>(vrec), vcell, hdr);
  return check_launch("link_dc_premix_insert");
}


extern "C" int link_dc_premix_insert(const float *feats, const int32_t *coords, const float *w_pre,
                                     const float *ln_w, const float *ln_b, int64_t n, int32_t c, float eps,
                                     const link_dc_grid_t *g, int32_t insert, float *fin, uint32_t *cnt,
                                     int32_t *slots, int32_t *vrec, int32_t *vcell, int32_t *hdr, void *stream) {
  if (n < 0 || !g || !dc_width_ok(c)) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!feats || !w_pre || !ln_w || !ln_b || !fin) return LINK_ERR_ARG;
  if (n * (int64_t)c * 4 >= (1LL << 32) || g->vp * (int64_t)g->k * 16 >= (1LL << 32)) return LINK_ERR_ARG;
  if (g->k < DC_INL) return LINK_ERR_ARG;
  if (insert && (!coords || !cnt || !slots || !vrec || !vcell || !hdr)) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (c) {
    case 16: return launch_dc_premix<16>(feats, coords, w_pre, ln_w, ln_b, n, eps, *g, insert != 0, fin, cnt, slots, vrec, vcell, hdr, st);
    case 32: return launch_dc_premix<32>(feats, coords, w_pre, ln_w, ln_b, n, eps, *g, insert != 0, fin, cnt, slots, vrec, vcell, hdr, st);
    case 64: return launch_dc_premix<64>(feats, coords, w_pre, ln_w, ln_b, n, eps, *g, insert != 0, fin, cnt, slots, vrec, vcell, hdr, st);
    default: return launch_dc_premix<128>(feats, coords, w_pre, ln_w, ln_b, n, eps, *g, insert != 0, fin, cnt, slots, vrec, vcell, hdr, st);
  }
}

// ---------------------------------------------------------------------------------------------
// modulate + per-cell sum
// ---------------------------------------------------------------------------------------------
// lane `src` (0..LPR-1) of the caller's group
template <int LPR>
__device__ __forceinline__ int grp_bcast(int v, int src) {
  return __shfl(v, ((threadIdx.x & 63) & ~(LPR - 1)) + src, 64);
}

// One pair-step of the modulate-and-accumulate.  `own` is the record this lane evaluates sincos for: PAIR
// (channels j and j + C/2 share theta): the low half of the group takes voxel A, the high half voxel B, and
// the halves swap results; otherwise every lane evaluates A (and then B).
template <int LPR, int OP, bool PAIR>
__device__ __forceinline__ void dc_mod_step(int ax, int ay, int az, int bx, int by, int bz, const float4 &fA,
                                            const float4 &fB, bool hasA, bool hasB, bool hi, float coord_div,
                                            const float (&w0)[4], const float (&w1)[4], const float (&w2)[4],
                                            const float (&al)[4], float (&a0)[4], float (&a1)[4], float (&a2)[4]) {
  const float fvA[4] = {hasA ? fA.x : 0.f, hasA ? fA.y : 0.f, hasA ? fA.z : 0.f, hasA ? fA.w : 0.f};
  const float fvB[4] = {hasB ? fB.x : 0.f, hasB ? fB.y : 0.f, hasB ? fB.z : 0.f, hasB ? fB.w : 0.f};
  if (PAIR) {
    const bool swapped = hi && hasB;
    float x = (float)(swapped ? bx : ax), y = (float)(swapped ? by : ay), z = (float)(swapped ? bz : az);
    if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
      float sn, cs;
      sincos_nocall(th, sn, cs);
      const float so = partner<LPR>(sn), co = partner<LPR>(cs);
      const float snA = swapped ? so : sn, csA = swapped ? co : cs;
      const float snB = hi ? sn : so, csB = hi ? cs : co;
      mod_accum<OP>(a0[e], a1[e], a2[e], fvA[e], snA, csA, th);
      mod_accum<OP>(a0[e], a1[e], a2[e], fvB[e], snB, csB, 0.f);
    }
  } else {
    float x = (float)ax, y = (float)ay, z = (float)az;
    if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
      float sn, cs;
      sincos_nocall(th, sn, cs);
      mod_accum<OP>(a0[e], a1[e], a2[e], fvA[e], sn, cs, th);
    }
    x = (float)bx; y = (float)by; z = (float)bz;
    if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
      float sn, cs;
      sincos_nocall(th, sn, cs);
      mod_accum<OP>(a0[e], a1[e], a2[e], fvB[e], sn, cs, th);
    }
  }
}

// A group of LPR lanes (row of C floats = LPR x float4) per cell; each group owns a run of `run`
// consecutive interior cells (z fastest) and walks it as a three-stage software pipeline:
//   meta(c+2)  count + the cell's 4 inline records (lane k of the group loads record k: one 64-byte access)
//   rows(c+1)  records ordered by voxel id across lanes 0..3 (rank by quad rotations + one ds_permute per
//              field), then the four feature rows requested
//   sum(c)     two pair-steps of theta / sincos / modulate / accumulate, ONE row store (zeros for an
//              empty cell: the table never needs clearing), count published, counter reset
// so a cell never waits on a round trip it issued itself.  Cells with more than 4 voxels (5 % at 2
// voxels per cell) finish in a second loop: all their records ordered across the group's lanes, then
// batches of four rows.  More than LPR voxels in a cell: selection by ascending id, one voxel at a time.
template <int LPR, int OP, bool PAIR>
__global__ void __launch_bounds__(256) k_dc_modsum(
    const float *__restrict__ fin, const int4 *__restrict__ slots, uint32_t *__restrict__ cnt,
    int32_t *__restrict__ cell_n, const float *__restrict__ w_pos, const float *__restrict__ alpha, int c,
    int cg, float coord_div, link_dc_grid_t g, int run, bool warm, float *__restrict__ S_,
    int32_t *__restrict__ hdr) {
  constexpr int P = op_parts<OP>::value;
  constexpr int G = 64 / LPR;
  static_assert(LPR >= DC_INL, "a group must hold the inline records");
  const int lane = threadIdx.x & 63;
  const int li = lane & (LPR - 1);
  const int ch0 = 4 * li;
  const bool hi = PAIR && (li >= LPR / 2);
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int Vi = Dx * Dy * Dz * g.dim[3];
  if (blockIdx.x == 0 && threadIdx.x == 0 && !warm) {     // publish the step's status word
    hdr[LINK_HDR_STATUS] = hdr[LINK_HDR_STATUS_ACC];
    hdr[LINK_HDR_STATUS_ACC] = 0;
  }
  const int gi = (blockIdx.x * 4 + (threadIdx.x >> 6)) * G + lane / LPR;
  const int q0 = gi * run;
  const int q1 = (q0 + run < Vi) ? q0 + run : Vi;
  if (q0 >= q1) return;
  float w0[4], w1[4], w2[4], al[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int tc = (ch0 + e) % cg;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
  }
  const int rs = P * c;
  const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, (uint32_t)((g.vp + 1) * rs * 4));
  const __amdgpu_buffer_rsrc_t r_n = dc_rsrc(cell_n, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const uint32_t *__restrict__ csrc = warm ? reinterpret_cast<const uint32_t *>(cell_n) : cnt;
  const int lk = li < DC_INL ? li : DC_INL - 1;        // inline record this lane loads
  // cell walker (z fastest)
  int z = q0 % Dz, t = q0 / Dz;
  int y = t % Dy;
  t /= Dy;
  int x = t % Dx, b = t / Dx;
  auto advance = [&]() {
    if (++z == Dz) { z = 0; if (++y == Dy) { y = 0; if (++x == Dx) { x = 0; ++b; } } }
  };
  auto order4 = [&](int4 &rec, int cntc) {
    // records of lanes 0..3 -> ascending id over lanes 0..3 (lanes >= cntc hold INT_MAX)
    const int key = (li < DC_INL && li < cntc) ? rec.w : INT_MAX;
    const int k1 = __builtin_amdgcn_update_dpp(0, key, 0x39, 0xF, 0xF, true);   // quad_perm [1,2,3,0]
    const int k2 = __builtin_amdgcn_update_dpp(0, key, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    const int k3 = __builtin_amdgcn_update_dpp(0, key, 0x93, 0xF, 0xF, true);   // quad_perm [3,0,1,2]
    // rank = number of strictly smaller keys; equal keys (INT_MAX padding) ordered by lane
    const int q = li & 3;
    int rank = (k1 < key || (k1 == key && ((q + 1) & 3) < q)) + (k2 < key || (k2 == key && ((q + 2) & 3) < q)) +
               (k3 < key || (k3 == key && ((q + 3) & 3) < q));
    const int dst = ((lane & ~3) + rank) << 2;          // only quads 0 of each group matter
    rec.x = __builtin_amdgcn_ds_permute(dst, rec.x);
    rec.y = __builtin_amdgcn_ds_permute(dst, rec.y);
    rec.z = __builtin_amdgcn_ds_permute(dst, rec.z);
    rec.w = __builtin_amdgcn_ds_permute(dst, key);
  };
  // ---- prologue: meta(0) -> rows(0); meta(1) ----
  int pc0 = dc_cell(g, x, y, z, b);                    // cell being summed
  int n0 = (int)csrc[pc0];
  int4 rec0 = slots[(int64_t)pc0 * DC_INL + lk];
  order4(rec0, n0);
  int idv[4], cx[4], cy[4], cz[4];
  float4 f0[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    idv[k] = grp_bcast<LPR>(rec0.w, k);
    cx[k] = grp_bcast<LPR>(rec0.x, k); cy[k] = grp_bcast<LPR>(rec0.y, k); cz[k] = grp_bcast<LPR>(rec0.z, k);
    f0[k] = *reinterpret_cast<const float4 *>(&fin[(int64_t)(k < n0 ? idv[k] : 0) * c + ch0]);
  }
  if (q0 + 1 < q1) advance();
  int pc1 = dc_cell(g, x, y, z, b);
  int n1 = (int)csrc[pc1];
  int4 rec1 = slots[(int64_t)pc1 * DC_INL + lk];
  for (int q = q0; q < q1; q++) {
    // ---- stage rows(q+1): order + request rows; stage meta(q+2) ----
    order4(rec1, n1);
    int idn[4], nx[4], ny[4], nz[4];
    float4 f1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      idn[k] = grp_bcast<LPR>(rec1.w, k);
      nx[k] = grp_bcast<LPR>(rec1.x, k); ny[k] = grp_bcast<LPR>(rec1.y, k); nz[k] = grp_bcast<LPR>(rec1.z, k);
      f1[k] = *reinterpret_cast<const float4 *>(&fin[(int64_t)(k < n1 ? idn[k] : 0) * c + ch0]);
    }
    const int pc_next = pc1, n_next = n1;
    if (q + 2 < q1) advance();
    const int pc2 = dc_cell(g, x, y, z, b);
    const int n2 = (int)csrc[pc2];
    const int4 rec2 = slots[(int64_t)pc2 * DC_INL + lk];
    // ---- stage sum(q) ----
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
    const int nc = n0 < g.k ? n0 : g.k;
    if (nc <= DC_INL) {
      if (nc > 0)
        dc_mod_step<LPR, OP, PAIR>(cx[0], cy[0], cz[0], cx[1], cy[1], cz[1], f0[0], f0[1], true, nc > 1, hi, coord_div,
                                   w0, w1, w2, al, a0, a1, a2);
      if (nc > 2)
        dc_mod_step<LPR, OP, PAIR>(cx[2], cy[2], cz[2], cx[3], cy[3], cz[3], f0[2], f0[3], true, nc > 3, hi, coord_div,
                                   w0, w1, w2, al, a0, a1, a2);
    } else if (nc <= LPR) {
      // all records of the cell, one per lane, ordered by id across the group's lanes
      int4 r = slots[dc_slot(g, pc0, li < nc ? li : 0)];
      const int key = li < nc ? r.w : INT_MAX;
      int rank = 0;
#pragma unroll
      for (int o = 1; o < LPR; o++) {
        const int other = grp_bcast<LPR>(key, (li + o) & (LPR - 1));
        rank += (other < key) || (other == key && ((li + o) & (LPR - 1)) < li);
      }
      const int dst = ((lane & ~(LPR - 1)) + rank) << 2;
      r.x = __builtin_amdgcn_ds_permute(dst, r.x);
      r.y = __builtin_amdgcn_ds_permute(dst, r.y);
      r.z = __builtin_amdgcn_ds_permute(dst, r.z);
      r.w = __builtin_amdgcn_ds_permute(dst, key);
      for (int base = 0; base < nc; base += 4) {
        int bx[4], by[4], bz[4];
        float4 fb[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int src = base + k < nc ? base + k : base;
          const int id = grp_bcast<LPR>(r.w, src);
          bx[k] = grp_bcast<LPR>(r.x, src); by[k] = grp_bcast<LPR>(r.y, src); bz[k] = grp_bcast<LPR>(r.z, src);
          fb[k] = *reinterpret_cast<const float4 *>(&fin[(int64_t)id * c + ch0]);
        }
        dc_mod_step<LPR, OP, PAIR>(bx[0], by[0], bz[0], bx[1], by[1], bz[1], fb[0], fb[1], true, base + 1 < nc, hi,
                                   coord_div, w0, w1, w2, al, a0, a1, a2);
        if (base + 2 < nc)
          dc_mod_step<LPR, OP, PAIR>(bx[2], by[2], bz[2], bx[3], by[3], bz[3], fb[2], fb[3], true, base + 3 < nc, hi,
                                     coord_div, w0, w1, w2, al, a0, a1, a2);
      }
    } else {
      // selection by ascending voxel id: per step the group scans the cell's records LPR at a time
      int last = -1;
      for (int tstep = 0; tstep < nc; tstep++) {
        int best = INT_MAX, bestk = 0;
        for (int base = 0; base < nc; base += LPR) {
          const int k = base + li;
          const int id = k < nc ? slots[dc_slot(g, pc0, k)].w : INT_MAX;
          const int cand = id > last ? id : INT_MAX;
          int m = cand;
#pragma unroll
          for (int o = LPR / 2; o > 0; o >>= 1) { const int w = __shfl_xor(m, o, 64); m = w < m ? w : m; }
          if (m < best) {
            best = m;
            int kk = cand == m ? k : INT_MAX;
#pragma unroll
            for (int o = LPR / 2; o > 0; o >>= 1) { const int w = __shfl_xor(kk, o, 64); kk = w < kk ? w : kk; }
            bestk = kk;
          }
        }
        last = best;
        const int4 rec = slots[dc_slot(g, pc0, bestk)];
        const float4 fr = *reinterpret_cast<const float4 *>(&fin[(int64_t)rec.w * c + ch0]);
        dc_mod_step<LPR, OP, false>(rec.x, rec.y, rec.z, rec.x, rec.y, rec.z, fr, fr, true, false, false, coord_div,
                                    w0, w1, w2, al, a0, a1, a2);
      }
    }
    const uint32_t rowb = (uint32_t)pc0 * (uint32_t)rs * 4u;
    st16(r_S, rowb + (uint32_t)ch0 * 4u, make_float4(a0[0], a0[1], a0[2], a0[3]));
    st16(r_S, rowb + (uint32_t)(c + ch0) * 4u, make_float4(a1[0], a1[1], a1[2], a1[3]));
    if (P == 3) st16(r_S, rowb + (uint32_t)(2 * c + ch0) * 4u, make_float4(a2[0], a2[1], a2[2], a2[3]));
    const uint32_t coff = (li == 0 && !warm) ? (uint32_t)pc0 * 4u : DC_OOB;
    st4i(r_n, coff, nc);
    st4i(r_cnt, coff, 0);
    // ---- rotate the pipeline ----
    pc0 = pc_next; n0 = n_next;
#pragma unroll
    for (int k = 0; k < 4; k++) { cx[k] = nx[k]; cy[k] = ny[k]; cz[k] = nz[k]; f0[k] = f1[k]; }
    pc1 = pc2; n1 = n2; rec1 = rec2;
  }
}

template <int LPR>
static void launch_dc_modsum(const link_elk_desc_t &d, const link_dc_grid_t &g, hipStream_t st, const float *fin,
                             const int32_t *slots, uint32_t *cnt, int32_t *cell_n, const float *w_pos,
                             const float *alpha, bool warm, float *S_, int32_t *hdr) {
  constexpr int G = 64 / LPR;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  int64_t wgs = DC_MODSUM_WGS;
  int64_t groups = wgs * 4 * G;
  int run = (int)((vi + groups - 1) / groups);
  if (run < 1) run = 1;
  wgs = (vi + (int64_t)run * 4 * G - 1) / ((int64_t)run * 4 * G);
  const bool two_part = d.op == LINK_OP_COS || d.op == LINK_OP_SIN;
  const bool pair = LPR >= 2 && d.c == 2 * d.cg && d.c == 4 * LPR && two_part;
  const int4 *sl = reinterpret_cast<const int4 *>(slots);
#define LINK_DCMS(OPP, PP)                                                                                          \
  hipLaunchKernelGGL((k_dc_modsum<LPR, OPP, PP>), dim3((unsigned)wgs), dim3(256), 0, st, fin, sl, cnt, cell_n, w_pos, \
                     alpha, d.c, d.cg, d.coord_div, g, run, warm, S_, hdr)
  switch (d.op) {
    case LINK_OP_COS: if (pair) LINK_DCMS(LINK_OP_COS, true); else LINK_DCMS(LINK_OP_COS, false); break;
    case LINK_OP_SIN: if (pair) LINK_DCMS(LINK_OP_SIN, true); else LINK_DCMS(LINK_OP_SIN, false); break;
    default: LINK_DCMS(LINK_OP_COSX, false); break;
  }
#undef LINK_DCMS
}

static int dc_desc_ok(const link_elk_desc_t *d, const link_dc_grid_t *g) {
  if (!d || !g) return LINK_ERR_ARG;
  if (d->op < 0 || d->op > 2 || !dc_width_ok(d->c) || d->cg <= 0 || d->c % d->cg != 0) return LINK_ERR_ARG;
  if (d->r != 2 && d->r != 3) return LINK_ERR_ARG;
  if (g->k < DC_INL) return LINK_ERR_ARG;
  const int parts = d->op == LINK_OP_COSX ? 3 : 2;
  if ((g->vp + 1) * (int64_t)parts * d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;   // 32-bit row offsets
  if (g->vp * (int64_t)g->k * 16 >= (1LL << 32)) return LINK_ERR_ARG;
  return LINK_OK;
}

extern "C" int link_dc_modsum(const float *fin, const int32_t *slots, uint32_t *cnt, int32_t *cell_n,
                              const float *w_pos, const float *alpha, const link_elk_desc_t *desc,
                              const link_dc_grid_t *g, int32_t warm, float *S_, int32_t *hdr, void *stream) {
  if (dc_desc_ok(desc, g) != LINK_OK) return LINK_ERR_ARG;
  if (!fin || !slots || !cnt || !cell_n || !w_pos || !S_ || !hdr) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (desc->c) {
    case 16: launch_dc_modsum<4>(*desc, *g, st, fin, slots, cnt, cell_n, w_pos, alpha, warm != 0, S_, hdr); break;
    case 32: launch_dc_modsum<8>(*desc, *g, st, fin, slots, cnt, cell_n, w_pos, alpha, warm != 0, S_, hdr); break;
    case 64: launch_dc_modsum<16>(*desc, *g, st, fin, slots, cnt, cell_n, w_pos, alpha, warm != 0, S_, hdr); break;
    default: launch_dc_modsum<32>(*desc, *g, st, fin, slots, cnt, cell_n, w_pos, alpha, warm != 0, S_, hdr); break;
  }
  return check_launch("link_dc_modsum");
}

// ---------------------------------------------------------------------------------------------
// r^3 box sum over the padded grid (LDS-DMA plane ring)
// ---------------------------------------------------------------------------------------------
// Geometry (C = 64, P = 2: rows of 512 B): a workgroup of 256 threads = 16 groups owns TX x TY = 4 x 4
// columns and a z-segment.  Per z-plane it needs the haloed (TX+R-1) x (TY+R-1) = 36 rows = 18 KB, brought
// in by 5 global_load_lds_dwordx4 per wave (64 lanes x 16 B = two whole rows per instruction, LDS image
// lane-linear) plus one global_load_lds_dword for the 36 cell counts.  Three plane buffers: while plane i
// is summed, planes i+1 and i+2 are in flight.  `s_waitcnt vmcnt(NI)` (NI = DMA instructions per plane and
// wave) before the barrier of plane i is exact enough for any number of interleaved output stores: the
// NI newest operations can only be loads of plane i+1 and stores younger than them, never loads of plane i.
// All LDS lives in ONE extern array and every load is an LDS-DMA, so hipcc inserts no vmcnt(0) of its own.
template <int C, int P, int R>
__global__ void __launch_bounds__(256) k_dc_gather(const float *__restrict__ S_, const int32_t *__restrict__ cell_n,
                                                   link_dc_grid_t g, int txn, int tyn, int zsplit, int nwg,
                                                   float *__restrict__ A) {
  using K = dc_gather_cfg<C, P, R>;
  constexpr int LPR = K::LPR, TX = K::TX, TY = K::TY, HY = K::HY, HLO = K::HLO;
  constexpr int RB = P * C * 4;                     // row bytes
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware logical id: workgroup w runs on XCD w % 8 (observed), give every XCD a contiguous range of
  // tiles so that tiles sharing halo rows share an L2
  const int per = (nwg + 7) >> 3;
  const int L = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
  if (L >= nwg) return;
  int t = L;
  const int zseg = t % zsplit; t /= zsplit;
  const int ty = t % tyn; t /= tyn;
  const int tx = t % txn;
  const int b = t / txn;
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int PDy = g.pdim[1], PDz = g.pdim[2];
  const int x0 = tx * TX, y0 = ty * TY;
  const int zs = (int)(((long long)Dz * zseg) / zsplit), ze = (int)(((long long)Dz * (zseg + 1)) / zsplit);
  if (zs >= ze) return;
  const int nplanes = (ze - zs) + R - 1;
  const int pz0 = zs + 1 - HLO;                      // first padded plane loaded
  // ---- per-lane DMA source offsets (loop-invariant part) ----
  uint32_t src_off[K::PASSES];
#pragma unroll
  for (int i = 0; i < K::PASSES; i++) {
    int pid = i * 256 + tid;
    if (pid >= K::NPC) pid = K::NPC - 1;             // padding lanes re-load the last piece (lands in padding)
    const int col = pid / K::RP, pcs = pid % K::RP;
    const int hx = col / HY, hy = col % HY;
    int px = x0 + 1 - HLO + hx, py = y0 + 1 - HLO + hy;
    px = px < g.pdim[0] - 1 ? px : g.pdim[0] - 1;
    py = py < g.pdim[1] - 1 ? py : g.pdim[1] - 1;
    const int cell0 = ((b * g.pdim[0] + px) * PDy + py) * PDz;
    src_off[i] = (uint32_t)cell0 * (uint32_t)RB + (uint32_t)pcs * 16u;
  }
  uint32_t cnt_cell0;
  {
    int e = wave * 64 + lane;                         // entry e of the plane's count image (linear by column)
    if (e >= K::NCOL) e = K::NCOL - 1;               // padding lanes / waves re-load the last entry
    const int hx = e / HY, hy = e % HY;
    int px = x0 + 1 - HLO + hx, py = y0 + 1 - HLO + hy;
    px = px < g.pdim[0] - 1 ? px : g.pdim[0] - 1;
    py = py < g.pdim[1] - 1 ? py : g.pdim[1] - 1;
    cnt_cell0 = (uint32_t)(((b * g.pdim[0] + px) * PDy + py) * PDz);
  }
  const char *Sb = reinterpret_cast<const char *>(S_);
  auto issue = [&](int plane) {
    int pz = pz0 + plane;
    pz = pz < PDz - 1 ? pz : PDz - 1;
    char *buf = lds + (plane % 3) * K::BUF_BYTES;
#pragma unroll
    for (int i = 0; i < K::PASSES; i++) {
      const char *src = Sb + (size_t)src_off[i] + (size_t)pz * RB;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)(buf + (i * 256 + wave * 64) * 16),
                                       16, 0, 0);
    }
    const int32_t *csrc = cell_n + cnt_cell0 + pz;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)csrc,
                                     (__attribute__((address_space(3))) void *)(buf + K::PLANE_BYTES + wave * 256),
                                     4, 0, 0);
  };
  // ---- this group's column ----
  const int grp = tid / LPR, li = tid % LPR;
  const int ix = grp / TY, iy = grp % TY;
  const bool col_ok = (x0 + ix < Dx) && (y0 + iy < Dy);
  const int ocell0 = ((b * g.pdim[0] + x0 + ix + 1) * PDy + y0 + iy + 1) * PDz;
  const __amdgpu_buffer_rsrc_t r_A = dc_rsrc(A, (uint32_t)((g.vp + 1) * RB));
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)lds;
  const uint32_t row_lane = (uint32_t)((ix * HY + iy) * RB + li * 16);
  const uint32_t cnt_lane = (uint32_t)((ix * HY + iy) * 4);
  float4 r0[P], r1[P];
  float c0 = 0.f, c1 = 0.f;
#pragma unroll
  for (int pp = 0; pp < P; pp++) r0[pp] = r1[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
  issue(0);
  if (nplanes > 1) issue(1);
  for (int i = 0; i < nplanes; i++) {
    if (i + 1 < nplanes) wait_vmcnt<K::NI>(); else wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");
    if (i + 2 < nplanes) issue(i + 2);
    float4 cur[P];
    float cc = 0.f;
#pragma unroll
    for (int pp = 0; pp < P; pp++) cur[pp] = make_float4(0.f, 0.f, 0.f, 0.f);
    {
      const uint32_t ra = lds_base + (uint32_t)((i % 3) * K::BUF_BYTES) + row_lane;
      const uint32_t ca = lds_base + (uint32_t)((i % 3) * K::BUF_BYTES + K::PLANE_BYTES) + cnt_lane;
      dc_read_dx<C, P, R, 0>(ra, ca, cur, cc);
      dc_read_dx<C, P, R, 1>(ra, ca, cur, cc);
      if (R == 3) dc_read_dx<C, P, R, R == 3 ? 2 : 1>(ra, ca, cur, cc);
    }
    // plane i is padded z = pz0 + i; with it the window of output plane po = pz0 + i - (R - 1) + HLO closes
    if (i >= R - 1) {
      const int po = pz0 + i - (R - 1) + HLO;
      float4 sum[P];
      float den;
      if (R == 3) {
        den = (c0 + c1) + cc;
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          sum[pp].x = (r0[pp].x + r1[pp].x) + cur[pp].x; sum[pp].y = (r0[pp].y + r1[pp].y) + cur[pp].y;
          sum[pp].z = (r0[pp].z + r1[pp].z) + cur[pp].z; sum[pp].w = (r0[pp].w + r1[pp].w) + cur[pp].w;
        }
      } else {
        den = c1 + cc;
#pragma unroll
        for (int pp = 0; pp < P; pp++) {
          sum[pp].x = r1[pp].x + cur[pp].x; sum[pp].y = r1[pp].y + cur[pp].y;
          sum[pp].z = r1[pp].z + cur[pp].z; sum[pp].w = r1[pp].w + cur[pp].w;
        }
      }
      const float inv = den > 0.f ? 1.0f / den : 0.f;
      const uint32_t rowb = (uint32_t)(ocell0 + po) * (uint32_t)RB;
#pragma unroll
      for (int pp = 0; pp < P; pp++)
        st16(r_A, col_ok ? rowb + (uint32_t)(pp * C * 4 + li * 16) : DC_OOB,
             make_float4(sum[pp].x * inv, sum[pp].y * inv, sum[pp].z * inv, sum[pp].w * inv));
    }
#pragma unroll
    for (int pp = 0; pp < P; pp++) { r0[pp] = r1[pp]; r1[pp] = cur[pp]; }
    c0 = c1; c1 = cc;
  }
}

template <int C, int P, int R>
static int launch_dc_gather(const link_dc_grid_t &g, hipStream_t st, const float *S_, const int32_t *cell_n, float *A) {
  using K = dc_gather_cfg<C, P, R>;
  const int txn = (g.dim[0] + K::TX - 1) / K::TX, tyn = (g.dim[1] + K::TY - 1) / K::TY;
  int zsplit = 0;                                      // auto
  if (zsplit <= 0) {                                  // aim at ~2 workgroups per CU
    const int64_t tiles = (int64_t)txn * tyn * g.dim[3];
    zsplit = (int)(512 / tiles);                      // <= 2 workgroups per CU: one resident round
    if (zsplit < 1) zsplit = 1;
  }
  if (zsplit > g.dim[2]) zsplit = g.dim[2];
  const int64_t nwg = (int64_t)txn * tyn * g.dim[3] * zsplit;
  const int64_t grid = (nwg + 7) / 8 * 8;
  if (K::LDS_BYTES > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_gather<C, P, R>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  hipLaunchKernelGGL((k_dc_gather<C, P, R>), dim3((unsigned)grid), dim3(256), K::LDS_BYTES, st, S_, cell_n, g, txn,
                     tyn, zsplit, (int)nwg, A);
  return check_launch("link_dc_gather");
}

template <int C>
static int dispatch_dc_gather(const link_elk_desc_t &d, const link_dc_grid_t &g, hipStream_t st, const float *S_,
                              const int32_t *cell_n, float *A) {
  const bool p3 = d.op == LINK_OP_COSX;
  if (d.r == 3) return p3 ? launch_dc_gather<C, 3, 3>(g, st, S_, cell_n, A) : launch_dc_gather<C, 2, 3>(g, st, S_, cell_n, A);
  return p3 ? launch_dc_gather<C, 3, 2>(g, st, S_, cell_n, A) : launch_dc_gather<C, 2, 2>(g, st, S_, cell_n, A);
}

extern "C" int link_dc_gather(const float *S_, const int32_t *cell_n, const link_elk_desc_t *desc,
                              const link_dc_grid_t *g, float *A, void *stream) {
  if (dc_desc_ok(desc, g) != LINK_OK || !S_ || !cell_n || !A) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (desc->c) {
    case 16: return dispatch_dc_gather<16>(*desc, *g, st, S_, cell_n, A);
    case 32: return dispatch_dc_gather<32>(*desc, *g, st, S_, cell_n, A);
    case 64: return dispatch_dc_gather<64>(*desc, *g, st, S_, cell_n, A);
    default: return dispatch_dc_gather<128>(*desc, *g, st, S_, cell_n, A);
  }
}

// ---------------------------------------------------------------------------------------------
// one-call R_core on the dense-cell path
// ---------------------------------------------------------------------------------------------
extern "C" int link_dc_index(const int32_t *coords, int64_t n, const link_dc_grid_t *g, uint32_t *cnt, int32_t *slots,
                             int32_t *vcell, int32_t *hdr, void *stream);
extern "C" int link_dc_premix_modsum(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                     int64_t n, int32_t warm, void *stream);
extern "C" int link_dc_gather_demod(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                    int64_t n, void *stream);
extern "C" int link_dc_demod(const float *A, const float *fin, const int32_t *coords, const int32_t *vcell,
                             const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                             const link_elk_desc_t *d, const link_dc_grid_t *g, int64_t n, void *out, int32_t io_dtype,
                             void *stream);

namespace link {
int dc_index_ids_stats(const link_dc_buffers_t *, const link_dc_grid_t *, int64_t, int32_t *, hipStream_t);
int dc_index_stats_run(const link_dc_buffers_t *, const link_dc_grid_t *, int64_t, int32_t *, hipStream_t);
}  // namespace link

// The slot insert of a step (the form b->tune picks) + the frame's occupancy on this grid: stats i32[16][16], zeroed by the
// caller; slot k holds partial (voxels inside the grid, occupied cells, fullest cell's count) in [k][0..2] -- sum, sum, max.  A caller that likes what it reads runs
// link_elk_core_dense_forward with build_index = 2 (the insert is already there); one that does not zero-fills cnt and hdr.
extern "C" int link_dc_index_probe(const link_dc_buffers_t *b, const link_dc_grid_t *g, int64_t n, int32_t *stats, void *stream) {
  if (!b || !g || !stats || n < 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  return b->tune.k1_form == 1 ? dc_index_ids_stats(b, g, n, stats, S(stream)) : dc_index_stats_run(b, g, n, stats, S(stream));
}

extern "C" int link_elk_core_dense_forward(const link_dc_buffers_t *b, const link_dc_grid_t *g,
                                           const link_elk_desc_t *desc, int64_t n, int32_t build_index,
                                           void *stream) {
  if (!b || dc_desc_ok(desc, g) != LINK_OK || n < 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  int rc;
  const int mode = b->tune.mode ? b->tune.mode : 7;   // bit0: fused pre_mix+modsum kernel, bit1: dense-cell demod kernel, bit2: fused gather+demod (C = 64), bit3: ... at the other widths too
  const bool fused = (mode & 1) && desc->c <= 64 && g->k <= 352;
  if (b->io_dtype != LINK_IO_F32 && (!fused || !(mode & 2))) return LINK_ERR_ARG;   // half rows: fused kernels only
  if (build_index == 2 && !fused) return LINK_ERR_ARG;                              // the probe feeds the fused kernels only
  if (fused) {
    // index -> fused pre_mix + modulate + per-cell sum -> box gather -> per-voxel de-modulate
    if (build_index == 1) {                            // 2: link_dc_index_probe inserted this frame already
      rc = b->tune.k1_form == 1 ? link_dc_index_ids(b->coords, n, g, b->cnt, b->sid, b->vcell, b->hdr, stream)
                                : link_dc_index(b->coords, n, g, b->cnt, b->slots, b->vcell, b->hdr, stream);
      if (rc != LINK_OK) return rc;
    }
    rc = link_dc_premix_modsum(b, g, desc, n, build_index ? 0 : 1, stream);
    if (rc != LINK_OK) return rc;
  } else {
    rc = link_dc_premix_insert(reinterpret_cast<const float *>(b->feats), b->coords, b->w_pre, b->pre_ln_w, b->pre_ln_b, n, desc->c, desc->eps, g,
                               build_index, b->fin, b->cnt, b->slots, b->vrec, b->vcell, b->hdr, stream);
    if (rc != LINK_OK) return rc;
    rc = link_dc_modsum(b->fin, b->slots, b->cnt, b->cell_n, b->w_pos, b->alpha, desc, g, build_index ? 0 : 1, b->S,
                        b->hdr, stream);
    if (rc != LINK_OK) return rc;
  }
  // box sum + de-modulate fused, the A table never exists: C = 64 (producer / consumer forms).  The other widths take the two
  // kernels below: a fused form for them (a wave per 16 cells, round 3) measured slower at every size (C = 16: 22.6 against
  // 19.7 us at 10k voxels; C = 128: 108 against 50 us at 30k -- DESIGN.md section 5c) and was removed in round 4
  if ((mode & 4) && desc->c == 64)
    return link_dc_gather_demod(b, g, desc, n, stream);
  rc = link_dc_gather(b->S, b->cell_n, desc, g, b->A, stream);
  if (rc != LINK_OK) return rc;
  if (mode & 2)
    return link_dc_demod(b->A, b->fin, b->coords, b->vcell, b->w_pos, b->alpha, b->ln_w, b->ln_b, desc, g, n, b->out,
                         b->io_dtype, stream);
  if (fused && build_index) return LINK_ERR_ARG;       // section C's kernel needs vrec, which only pre_mix+insert writes
  return link_voxel_demod_ln(b->A, b->fin, b->vrec, b->vcell, b->w_pos, b->alpha, b->ln_w, b->ln_b, b->hdr, desc, n,
                             reinterpret_cast<float *>(b->out), stream);
}
