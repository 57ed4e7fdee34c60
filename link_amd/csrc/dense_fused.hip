// link_amd/csrc/dense_fused.hip -- the fused kernels of the dense-cell layout (include/link_amd.h section E).
//
// Why fuse: counters on cfg2 (profiles/r02_*) show pre_mix bound by the f32 MFMA pipe with the VALU idle, and
// the per-cell modulate+sum bound by VALU issue with the MFMA pipe idle -- and `fin` making a 51 MB round trip
// between them.  k_dc_premix_modsum does both on the same 16-voxel tile: MFMA of one wave runs beside the
// sincos/modulate VALU work of the other wave on its SIMD, `fin` never leaves registers, and the per-cell
// sums are formed from an LDS image of the tile.
//
//   k_dc_index            coords -> cell, rank = cnt[cell]++, slots[cell][rank] = (x,y,z,id); vcell[i] = cell
//   k_dc_premix_modsum    a wave owns a range of cells: counts -> wave prefix scan -> flat, id-ordered voxel
//                         list in LDS -> tiles of 16 voxels: gather F rows, LayerNorm(F Wpre^T) on MFMA, theta /
//                         sincos / modulate in the MFMA layout (a lane holds 16 channels of ONE voxel: no lane
//                         is ever idle, whatever the cell sizes), X tile -> LDS -> per-cell sums -> S rows
//   k_dc_demod            per voxel pair, original order, persistent + software-pipelined: A[cell] row,
//                         de-modulate, LayerNorm, store
#include "dense_common.h"

using namespace link;

static int g_k1_wgs = 512;
static int g_demod_wgs = 1024;
static int g_index_wgs = 0;

extern "C" int link_dc_set_tuning2(int key, int value) {
  if (value < 0) return LINK_ERR_ARG;
  switch (key) {
    case 0: g_k1_wgs = value > 0 ? value : 512; break;
    case 1: g_demod_wgs = value > 0 ? value : 1024; break;
    case 2: g_index_wgs = value; break;
    default: return LINK_ERR_ARG;
  }
  return LINK_OK;
}

// ---------------------------------------------------------------------------------------------
// index: slot insert
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_dc_index(const int4 *__restrict__ coords, int64_t n, link_dc_grid_t g,
                                                  uint32_t *__restrict__ cnt, int4 *__restrict__ slots,
                                                  int32_t *__restrict__ vcell, int32_t *__restrict__ hdr) {
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  if (blockIdx.x == 0 && threadIdx.x == 0) hdr[LINK_HDR_NVALID] = (int32_t)n;
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < n; v += (int64_t)gridDim.x * 256) {
    const int4 rc = coords[v];
    const unsigned ux = (unsigned)(floordiv(rc.x, g.s) - g.lo[0]), uy = (unsigned)(floordiv(rc.y, g.s) - g.lo[1]);
    const unsigned uz = (unsigned)(floordiv(rc.z, g.s) - g.lo[2]), ub = (unsigned)(rc.w - g.lo[3]);
    const bool inside = ux < (unsigned)g.dim[0] && uy < (unsigned)g.dim[1] && uz < (unsigned)g.dim[2] &&
                        ub < (unsigned)g.dim[3];
    if (!inside) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 1);
    const int pcell = inside ? dc_cell(g, (int)ux, (int)uy, (int)uz, (int)ub) : 0;
    const int rank = __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_cnt, pcell ? (uint32_t)pcell * 4u : DC_OOB, 0, 0);
    const bool full = pcell != 0 && rank >= g.k;
    if (full) atomicOr(&hdr[LINK_HDR_STATUS_ACC], 2);
    const bool keep = pcell != 0 && !full;
    st16i(r_slots, keep ? dc_slot(g, pcell, rank) * 16u : DC_OOB, make_int4(rc.x, rc.y, rc.z, (int)v));
    vcell[v] = keep ? pcell : 0;
  }
}

extern "C" int link_dc_index(const int32_t *coords, int64_t n, const link_dc_grid_t *g, uint32_t *cnt,
                             int32_t *slots, int32_t *vcell, int32_t *hdr, void *stream) {
  if (n < 0 || !g) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!coords || !cnt || !slots || !vcell || !hdr) return LINK_ERR_ARG;
  if (g->k < DC_INL || g->vp * (int64_t)g->k * 16 >= (1LL << 32) || n >= (1LL << 29)) return LINK_ERR_ARG;
  int64_t wgs = g_index_wgs > 0 ? g_index_wgs : (n + 255) / 256;
  if (wgs > 4096) wgs = 4096;
  hipLaunchKernelGGL(k_dc_index, dim3((unsigned)wgs), dim3(256), 0, S(stream), reinterpret_cast<const int4 *>(coords), n,
                     *g, cnt, reinterpret_cast<int4 *>(slots), vcell, hdr);
  return check_launch("link_dc_index");
}

// ---------------------------------------------------------------------------------------------
// pre_mix + LayerNorm + modulate + per-cell sum
// ---------------------------------------------------------------------------------------------
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
  static constexpr int LCAP = 384;                     // records of one cell range kept in LDS (>= 7^3)
  static constexpr int W_BYTES = (C * LDW + 2 * C) * 4;
  static constexpr int LIST_OFF = 0;
  static constexpr int CEND_OFF = LCAP * 16;
  static constexpr int PCL_OFF = CEND_OFF + 64 * 4;
  static constexpr int X_OFF = PCL_OFF + 64 * 4;
  static constexpr int WAVE_BYTES = X_OFF + 16 * XROW;
  static constexpr int LDS_BYTES = W_BYTES + 4 * WAVE_BYTES;
};

// NB = number of distinct 16-channel theta blocks of a voxel: channel ch uses theta[ch % cg]; when cg is a
// multiple of 16 the MFMA channel block tp (channels 16 tp + 4 g + r of lane group g) uses theta block
// tp % (cg/16), so a lane evaluates 4*NB sincos per voxel instead of 4*T; otherwise NB = T.
template <int C, int OP, int NB>
__global__ void __launch_bounds__(256) k_dc_premix_modsum(
    const float *__restrict__ feats, const int4 *__restrict__ slots, uint32_t *__restrict__ cnt,
    int32_t *__restrict__ cell_n, const float *__restrict__ w_pre, const float *__restrict__ ln_w,
    const float *__restrict__ ln_b, const float *__restrict__ w_pos, const float *__restrict__ alpha, int cg,
    float coord_div, float eps, int64_t n, link_dc_grid_t g, int cpw, bool warm, float *__restrict__ S_,
    float *__restrict__ fin, int32_t *__restrict__ hdr) {
  using K = dc_k1_cfg<C, OP>;
  constexpr int T = K::T, P = K::P, LDW = K::LDW;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *w_lds = reinterpret_cast<float *>(smem_raw);
  float *ln_lds = w_lds + C * LDW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, gq = lane >> 4;
  char *wbase = smem_raw + K::W_BYTES + wave * K::WAVE_BYTES;
  int4 *list = reinterpret_cast<int4 *>(wbase + K::LIST_OFF);
  int *cend = reinterpret_cast<int *>(wbase + K::CEND_OFF);
  int *pcl = reinterpret_cast<int *>(wbase + K::PCL_OFF);
  char *xbuf = wbase + K::X_OFF;
  {                                                    // stage W and the LayerNorm parameters
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
  if (blockIdx.x == 0 && tid == 0 && !warm) {          // publish the step's status word
    hdr[LINK_HDR_STATUS] = hdr[LINK_HDR_STATUS_ACC];
    hdr[LINK_HDR_STATUS_ACC] = 0;
  }
  __syncthreads();
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int Vi = Dx * Dy * Dz * g.dim[3];
  const int wid = blockIdx.x * 4 + wave;
  const int c_begin = wid * cpw;
  const int c_end = (c_begin + cpw < Vi) ? c_begin + cpw : Vi;
  if (c_begin >= c_end) return;
  const __amdgpu_buffer_rsrc_t r_S = dc_rsrc(S_, (uint32_t)((g.vp + 1) * K::RB));
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * C * 4));     // written for cos_x only
  const __amdgpu_buffer_rsrc_t r_n = dc_rsrc(cell_n, (uint32_t)(g.vp * 4));
  const __amdgpu_buffer_rsrc_t r_cnt = dc_rsrc(cnt, (uint32_t)(g.vp * 4));
  const uint32_t *__restrict__ csrc = warm ? reinterpret_cast<const uint32_t *>(cell_n) : cnt;
  // theta weights of this lane's channels
  float w0[NB][4], w1[NB][4], w2[NB][4], al[NB][4];
#pragma unroll
  for (int tb = 0; tb < NB; tb++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int tc = (16 * tb + 4 * gq + r) % cg;
      w0[tb][r] = w_pos[3 * tc + 0]; w1[tb][r] = w_pos[3 * tc + 1]; w2[tb][r] = w_pos[3 * tc + 2];
      al[tb][r] = alpha ? alpha[tc] : 1.0f;
    }
  const int rg = lane / K::RGS, rl = lane % K::RGS;
  const bool ract = rl < K::RGL;

  for (int chunk = c_begin; chunk < c_end;) {
    const int nrem = (c_end - chunk < 64) ? c_end - chunk : 64;
    // ---- cell lanes: count, padded cell id, inline records (all requested before anything is consumed) ----
    int pc = 0, nv = 0;
    {
      const int q = chunk + (lane < nrem ? lane : 0);
      const int z = q % Dz;
      int t = q / Dz;
      const int y = t % Dy;
      t /= Dy;
      pc = dc_cell(g, t % Dx, y, z, t / Dx);
    }
    const int4 r0 = slots[(int64_t)pc * DC_INL + 0], r1 = slots[(int64_t)pc * DC_INL + 1];
    const int4 r2 = slots[(int64_t)pc * DC_INL + 2], r3 = slots[(int64_t)pc * DC_INL + 3];
    nv = (int)csrc[pc];
    nv = nv < g.k ? nv : g.k;
    nv = nv < K::LCAP ? nv : K::LCAP;
    if (lane >= nrem) nv = 0;
    int incl = nv;                                      // inclusive prefix over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o, 64);
      if (lane >= o) incl += u;
    }
    const unsigned long long fit = __ballot(lane < nrem && incl <= K::LCAP);
    const int nfit = __builtin_amdgcn_readfirstlane(__popcll(fit));      // >= 1: a cell never exceeds LCAP
    const int Ttot = __builtin_amdgcn_readfirstlane(__shfl(incl, nfit - 1, 64));
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
        if (j < nv) list[excl + j] = r;
      }
      for (int k = DC_INL; k < nv; k++) {               // overflow records: insertion by id (rare)
        const int4 r = slots[dc_slot(g, pc, k)];
        int pos = k;
        while (pos > 0 && list[excl + pos - 1].w > r.w) {
          list[excl + pos] = list[excl + pos - 1];
          pos--;
        }
        list[excl + pos] = r;
      }
      cend[lane] = incl;
      pcl[lane] = pc;
    }
    {                                                   // publish the counts, reset the counters
      const uint32_t coff = (lane < nfit && !warm) ? (uint32_t)pc * 4u : DC_OOB;
      st4i(r_n, coff, nv);
      st4i(r_cnt, coff, 0);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- tiles of 16 voxels ----
    const int ntile = (Ttot + 15) >> 4;
    const int nloop = ntile > 0 ? ntile : 1;
    int jcur = rg;                                      // next cell of this row group
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int4 rec = make_int4(0, 0, 0, 0);
    float4 f[T];
    if (ntile > 0) {                                    // rows of tile 0
      rec = list[li < Ttot ? li : Ttot - 1];
#pragma unroll
      for (int tt = 0; tt < T; tt++)
        f[tt] = *reinterpret_cast<const float4 *>(&feats[(int64_t)rec.w * C + 16 * tt + 4 * gq]);
    }
    for (int t = 0; t < nloop; t++) {
      if (ntile > 0) {
        const int slot = 16 * t + li;
        // rows of tile t+1 requested before tile t is multiplied (clamped: the last tile re-reads itself)
        const int nslot = (t + 1 < ntile) ? slot + 16 : slot;
        const int4 recn = list[nslot < Ttot ? nslot : Ttot - 1];
        float4 fn[T];
#pragma unroll
        for (int tt = 0; tt < T; tt++)
          fn[tt] = *reinterpret_cast<const float4 *>(&feats[(int64_t)recn.w * C + 16 * tt + 4 * gq]);
        floatx4 ac[T];
#pragma unroll
        for (int tp = 0; tp < T; tp++) ac[tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < T; tt++) {
          float4 a[T];
#pragma unroll
          for (int tp = 0; tp < T; tp++)
            a[tp] = *reinterpret_cast<const float4 *>(&w_lds[(16 * tp + li) * LDW + 16 * tt + 4 * gq]);
#pragma unroll
          for (int tp = 0; tp < T; tp++) ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].x, f[tt].x, ac[tp], 0, 0, 0);
#pragma unroll
          for (int tp = 0; tp < T; tp++) ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].y, f[tt].y, ac[tp], 0, 0, 0);
#pragma unroll
          for (int tp = 0; tp < T; tp++) ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].z, f[tt].z, ac[tp], 0, 0, 0);
#pragma unroll
          for (int tp = 0; tp < T; tp++) ac[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[tp].w, f[tt].w, ac[tp], 0, 0, 0);
        }
        // theta / sincos of this voxel for the lane's theta blocks (in the shadow of the MFMAs)
        float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
        if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
        float sn[NB][4], cs[NB][4], th[NB][4];
#pragma unroll
        for (int tb = 0; tb < NB; tb++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            th[tb][r] = theta_of(x, y, z, w0[tb][r], w1[tb][r], w2[tb][r], al[tb][r]);
            sincos_nocall(th[tb][r], sn[tb][r], cs[tb][r]);
          }
        // LayerNorm over the voxel's C channels: 16 in-lane values + the 4 lane groups
        float s = 0.f;
#pragma unroll
        for (int tp = 0; tp < T; tp++) s += (ac[tp][0] + ac[tp][1]) + (ac[tp][2] + ac[tp][3]);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mean = s * (1.0f / C);
        float qq = 0.f;
#pragma unroll
        for (int tp = 0; tp < T; tp++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float d = ac[tp][r] - mean;
            qq += d * d;
          }
        qq += __shfl_xor(qq, 16, 64);
        qq += __shfl_xor(qq, 32, 64);
        const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + eps);
#pragma unroll
        for (int tp = 0; tp < T; tp++) {
          const float4 lw = *reinterpret_cast<const float4 *>(&ln_lds[16 * tp + 4 * gq]);
          const float4 lb = *reinterpret_cast<const float4 *>(&ln_lds[C + 16 * tp + 4 * gq]);
          const float fv[4] = {(ac[tp][0] - mean) * rstd * lw.x + lb.x, (ac[tp][1] - mean) * rstd * lw.y + lb.y,
                               (ac[tp][2] - mean) * rstd * lw.z + lb.z, (ac[tp][3] - mean) * rstd * lw.w + lb.w};
          if (OP == LINK_OP_COSX)                       // the de-modulation of cos_x needs fin (linkunet.py:176)
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
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rec = recn;
#pragma unroll
        for (int tt = 0; tt < T; tt++) f[tt] = fn[tt];
      }
      // ---- per-cell sums of this tile (cells of the range, row groups side by side) ----
      const int tile_lo = 16 * t;
      const int tile_hi = (t == nloop - 1) ? INT_MAX : tile_lo + 16;
      while (jcur < nfit) {
        const int cs_ = jcur ? cend[jcur - 1] : 0, ce = cend[jcur];
        if (cs_ >= tile_hi) break;
        const int lo = cs_ > tile_lo ? cs_ : tile_lo, hi = ce < tile_hi ? ce : tile_hi;
        for (int sl = lo; sl < hi; sl++) {
          const float4 v = *reinterpret_cast<const float4 *>(xbuf + (sl - tile_lo) * K::XROW + (ract ? rl : 0) * 16);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (ce > tile_hi) break;                        // the cell continues in the next tile
        st16(r_S, ract ? (uint32_t)pcl[jcur] * (uint32_t)K::RB + (uint32_t)rl * 16u : DC_OOB, acc);
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        jcur += K::RG;
      }
      __builtin_amdgcn_wave_barrier();
    }
    chunk += nfit;
  }
}

template <int C, int OP, int NB>
static int launch_k1(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     bool warm, hipStream_t st) {
  using K = dc_k1_cfg<C, OP>;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  int64_t waves = (int64_t)g_k1_wgs * 4;
  int cpw = (int)((vi + waves - 1) / waves);
  if (cpw < 1) cpw = 1;
  const int64_t wgs = (vi + (int64_t)cpw * 4 - 1) / ((int64_t)cpw * 4);
  if (K::LDS_BYTES > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_premix_modsum<C, OP, NB>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  hipLaunchKernelGGL((k_dc_premix_modsum<C, OP, NB>), dim3((unsigned)wgs), dim3(256), K::LDS_BYTES, st, b->feats,
                     reinterpret_cast<const int4 *>(b->slots), b->cnt, b->cell_n, b->w_pre, b->pre_ln_w, b->pre_ln_b,
                     b->w_pos, b->alpha, d.cg, d.coord_div, d.eps, n, g, cpw, warm, b->S, b->fin, b->hdr);
  return check_launch("link_dc_premix_modsum");
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

extern "C" int link_dc_premix_modsum(const link_dc_buffers_t *b, const link_dc_grid_t *g, const link_elk_desc_t *d,
                                     int64_t n, int32_t warm, void *stream) {
  if (!b || !g || !d || n < 0) return LINK_ERR_ARG;
  if (d->c != 16 && d->c != 32 && d->c != 64) return LINK_ERR_ARG;
  if (d->op < 0 || d->op > 2 || d->cg <= 0 || d->c % d->cg != 0 || g->k < DC_INL) return LINK_ERR_ARG;
  if (g->k > 384) return LINK_ERR_ARG;                 // a cell's records must fit the wave's LDS list (LCAP)
  if (n * (int64_t)d->c * 4 >= (1LL << 32) || (d->op == LINK_OP_COSX && !b->fin)) return LINK_ERR_ARG;
  const int parts = d->op == LINK_OP_COSX ? 3 : 2;
  if ((g->vp + 1) * (int64_t)parts * d->c * 4 >= (1LL << 32) || g->vp * (int64_t)g->k * 16 >= (1LL << 32)) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!b->feats || !b->slots || !b->cnt || !b->cell_n || !b->w_pre || !b->pre_ln_w || !b->pre_ln_b || !b->w_pos ||
      !b->S || !b->hdr)
    return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (d->c) {
    case 16: return dispatch_k1_op<16>(b, *g, *d, n, warm != 0, st);
    case 32: return dispatch_k1_op<32>(b, *g, *d, n, warm != 0, st);
    default: return dispatch_k1_op<64>(b, *g, *d, n, warm != 0, st);
  }
}

// ---------------------------------------------------------------------------------------------
// per-voxel de-modulate + LayerNorm, original voxel order
// ---------------------------------------------------------------------------------------------
// A group of LPR lanes (row of C floats = LPR x float4) per voxel PAIR (2p, 2p+1); a group walks pairs p,
// p + #groups, ... as a three-stage pipeline -- meta(p+2): the two coordinate rows and cell ids; rows(p+1):
// the two A rows (buffer loads, 32-bit offsets); out(p): theta / sincos (shared between channels j and
// j + C/2 when PAIR) / de-modulate / LayerNorm / store.  All parameters live in registers for the whole
// kernel; invalid lanes store to an out-of-range offset.
template <int LPR, int OP, bool PAIR>
__global__ void __launch_bounds__(256) k_dc_demod(const float *__restrict__ A_, const float *__restrict__ fin,
                                                  const int4 *__restrict__ coords, const int32_t *__restrict__ vcell,
                                                  const float *__restrict__ w_pos, const float *__restrict__ alpha,
                                                  const float *__restrict__ ln_w, const float *__restrict__ ln_b, int c,
                                                  int cg, float coord_div, float eps, int64_t n, int64_t a_rows,
                                                  float *__restrict__ out) {
  constexpr int P = (OP == LINK_OP_COSX) ? 3 : 2;
  constexpr int G = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int li = lane & (LPR - 1);
  const int ch0 = 4 * li;
  const bool hi = PAIR && (li >= LPR / 2);
  const int ra = P * c * 4;                            // A row bytes
  const __amdgpu_buffer_rsrc_t r_A = dc_rsrc(A_, (uint32_t)(a_rows * ra));
  const __amdgpu_buffer_rsrc_t r_fin = dc_rsrc(fin, (uint32_t)(n * c * 4));
  const __amdgpu_buffer_rsrc_t r_out = dc_rsrc(out, (uint32_t)(n * c * 4));
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
    if (PAIR) {
      const bool swapped = hi && hasB;
      float x = (float)(swapped ? c0b.x : c0a.x), y = (float)(swapped ? c0b.y : c0a.y), z = (float)(swapped ? c0b.z : c0a.z);
      if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
        float sn, cs;
        sincos_nocall(th, sn, cs);
        const float so = partner<LPR>(sn), co = partner<LPR>(cs);
        const float snA = swapped ? so : sn, csA = swapped ? co : cs;
        const float snB = hi ? sn : so, csB = hi ? cs : co;
        const float A0 = __int_as_float(a0[0][e]), A1 = __int_as_float(a0[1][e]);
        const float B0 = __int_as_float(b0[0][e]), B1 = __int_as_float(b0[1][e]);
        if (OP == LINK_OP_SIN) {                                                 // linkunet.py:148
          nvA[e] = __fsub_rn(__fmul_rn(A0, csA), __fmul_rn(A1, snA));
          nvB[e] = __fsub_rn(__fmul_rn(B0, csB), __fmul_rn(B1, snB));
        } else {                                                                 // :162
          nvA[e] = __fadd_rn(__fmul_rn(A0, csA), __fmul_rn(A1, snA));
          nvB[e] = __fadd_rn(__fmul_rn(B0, csB), __fmul_rn(B1, snB));
        }
        sA += nvA[e]; sB += nvB[e];
      }
    } else {
#pragma unroll
      for (int hb = 0; hb < 2; hb++) {
        const int4 cc = hb ? c0b : c0a;
        float x = (float)cc.x, y = (float)cc.y, z = (float)cc.z;
        if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const float th = theta_of(x, y, z, w0[e], w1[e], w2[e], al[e]);
          float sn, cs;
          sincos_nocall(th, sn, cs);
          const float A0 = __int_as_float(hb ? b0[0][e] : a0[0][e]);
          const float A1 = __int_as_float(hb ? b0[1][e] : a0[1][e]);
          float v;
          if (OP == LINK_OP_SIN) v = __fsub_rn(__fmul_rn(A0, cs), __fmul_rn(A1, sn));
          else v = __fadd_rn(__fmul_rn(A0, cs), __fmul_rn(A1, sn));
          if (OP == LINK_OP_COSX) {                                              // :176
            const float A2 = __int_as_float(hb ? b0[P - 1][e] : a0[P - 1][e]);
            const float f = __int_as_float(hb ? fy[e] : fx[e]);
            v = __fadd_rn(v, __fsub_rn(A2, __fmul_rn(f, th)));
          }
          if (hb) { nvB[e] = v; sB += v; } else { nvA[e] = v; sA += v; }
        }
      }
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
    const float rsA = 1.0f / sqrtf(qA * inv_c + eps), rsB = 1.0f / sqrtf(qB * inv_c + eps);
    float4 oa, ob;
    oa.x = (nvA[0] - meanA) * rsA * gw[0] + gb[0]; oa.y = (nvA[1] - meanA) * rsA * gw[1] + gb[1];
    oa.z = (nvA[2] - meanA) * rsA * gw[2] + gb[2]; oa.w = (nvA[3] - meanA) * rsA * gw[3] + gb[3];
    ob.x = (nvB[0] - meanB) * rsB * gw[0] + gb[0]; ob.y = (nvB[1] - meanB) * rsB * gw[1] + gb[1];
    ob.z = (nvB[2] - meanB) * rsB * gw[2] + gb[2]; ob.w = (nvB[3] - meanB) * rsB * gw[3] + gb[3];
    const uint32_t offA = (uint32_t)(2 * p) * (uint32_t)(c * 4) + (uint32_t)(ch0 * 4);
    st16(r_out, offA, oa);
    st16(r_out, hasB ? offA + (uint32_t)(c * 4) : DC_OOB, ob);
    c0a = c1a; c0b = c1b; v0a = v1a; v0b = v1b;
    c1a = c2a; c1b = c2b; v1a = v2a; v1b = v2b;
#pragma unroll
    for (int pp = 0; pp < P; pp++) { a0[pp] = a1[pp]; b0[pp] = b1[pp]; }
  }
}

template <int LPR>
static void launch_dc_demod(const link_elk_desc_t &d, int64_t n, int64_t a_rows, hipStream_t st, const float *A,
                            const float *fin, const int32_t *coords, const int32_t *vcell, const float *w_pos,
                            const float *alpha, const float *ln_w, const float *ln_b, float *out) {
  constexpr int G = 64 / LPR;
  const int64_t npair = (n + 1) / 2;
  int64_t wgs = (npair + 4 * G - 1) / (4 * G);
  if (wgs > g_demod_wgs) wgs = g_demod_wgs;
  const bool two_part = d.op == LINK_OP_COS || d.op == LINK_OP_SIN;
  const bool pair = LPR >= 2 && d.c == 2 * d.cg && two_part;
  const int4 *co = reinterpret_cast<const int4 *>(coords);
#define LINK_DCDM(OPP, PP)                                                                                              \
  hipLaunchKernelGGL((k_dc_demod<LPR, OPP, PP>), dim3((unsigned)wgs), dim3(256), 0, st, A, fin, co, vcell, w_pos, alpha, \
                     ln_w, ln_b, d.c, d.cg, d.coord_div, d.eps, n, a_rows, out)
  switch (d.op) {
    case LINK_OP_COS: if (pair) LINK_DCDM(LINK_OP_COS, true); else LINK_DCDM(LINK_OP_COS, false); break;
    case LINK_OP_SIN: if (pair) LINK_DCDM(LINK_OP_SIN, true); else LINK_DCDM(LINK_OP_SIN, false); break;
    default: LINK_DCDM(LINK_OP_COSX, false); break;
  }
#undef LINK_DCDM
}

extern "C" int link_dc_demod(const float *A, const float *fin, const int32_t *coords, const int32_t *vcell,
                             const float *w_pos, const float *alpha, const float *ln_w, const float *ln_b,
                             const link_elk_desc_t *d, const link_dc_grid_t *g, int64_t n, float *out, void *stream) {
  if (!d || !g || n < 0 || !dc_width_ok(d->c) || d->op < 0 || d->op > 2 || d->cg <= 0 || d->c % d->cg != 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!A || !coords || !vcell || !w_pos || !ln_w || !ln_b || !out || (d->op == LINK_OP_COSX && !fin)) return LINK_ERR_ARG;
  const int parts = d->op == LINK_OP_COSX ? 3 : 2;
  if ((g->vp + 1) * (int64_t)parts * d->c * 4 >= (1LL << 32) || n * (int64_t)d->c * 4 >= (1LL << 32)) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  switch (d->c) {
    case 16: launch_dc_demod<4>(*d, n, g->vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
    case 32: launch_dc_demod<8>(*d, n, g->vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
    case 64: launch_dc_demod<16>(*d, n, g->vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
    default: launch_dc_demod<32>(*d, n, g->vp + 1, st, A, fin, coords, vcell, w_pos, alpha, ln_w, ln_b, out); break;
  }
  return check_launch("link_dc_demod");
}
