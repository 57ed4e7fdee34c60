// link_amd/csrc/dense_gather_cells_impl.h -- fused box sum + de-modulate + LayerNorm on the dense-cell layout for the widths the
// C = 64 kernels (dense_fused_impl.h: producer / consumer forms built around 16-lane rows) do not take: C = 16 / 32 / 128,
// every base op, r = 2 / 3 (include/link_amd.h section E, link_dc_gather_demod).  Included inside DC_IO_NS by
// dense_fused_impl.h.  With it a dense-cell step is 3 launches at every width the fused pre_mix kernel takes (C <= 64), and
// the A table of the two-kernel form (link_dc_gather + link_dc_demod) is never written.
//
// A wave takes 16 consecutive interior cells.  The occupied ones among them are its "runs" (as the distinct blocks of a tile
// in elk_tiles_impl.h): their r^3 neighbour rows are plain address arithmetic on the padded grid (border rows are zero, no
// validity test, no lookup), a lane group per run forms the normalised neighbour sum into LDS; then the cells' voxel
// records (slot lists, any order: every voxel's output is its own) are flattened 64 at a time through the prefix of the
// counts and dealt to the lane groups: theta, sincos, de-modulation in separate IEEE mul / add, LayerNorm inside the
// group, one row store.
#pragma once

template <int C, int OP, int R>
struct dc_k2c_cfg {
  static constexpr int LPR = C / 4, G = 64 / LPR, P = op_parts<OP>::value, RS = P * C, R3 = R * R * R;
  static constexpr int NC = 16, NW = 4;                // cells per wave, waves per workgroup
  static constexpr int A_BYTES = NC * RS * 4;
  static constexpr int NB_BYTES = (NC * R3 * 4 + 15) & ~15;
  static constexpr int REC_BYTES = 64 * 16 + 64 * 4;   // a pass's voxel records + their run index
  static constexpr int META_BYTES = (3 * NC + 1) * 4 + 12;   // run cell | run count | run prefix (+ total)
  static constexpr int WAVE_BYTES = (A_BYTES + NB_BYTES + REC_BYTES + META_BYTES + 15) & ~15;
  static constexpr int LDS_BYTES = NW * WAVE_BYTES;
};

template <int C, int OP, int R>
__global__ void __launch_bounds__(256) k_dc_gather_demod_cells(
    const float *__restrict__ S, const int32_t *__restrict__ cell_n, const int4 *__restrict__ slots,
    const float *__restrict__ fin, const float *__restrict__ w_pos, const float *__restrict__ alpha,
    const float *__restrict__ ln_w, const float *__restrict__ ln_b, int cg, float coord_div, float eps, int64_t n,
    link_dc_grid_t g, void *__restrict__ out) {
  using K = dc_k2c_cfg<C, OP, R>;
  constexpr int LPR = K::LPR, G = K::G, P = K::P, RS = K::RS, R2 = R * R, R3 = K::R3, NC = K::NC;
  constexpr int LO = -((R + 1) / 2) + 1;               // nn/utils/kernel.py:21
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & (LPR - 1), grp = lane / LPR;
  char *wbase = smem_raw + wave * K::WAVE_BYTES;
  float *A_lds = reinterpret_cast<float *>(wbase);
  int32_t *nb_lds = reinterpret_cast<int32_t *>(wbase + K::A_BYTES);
  int4 *rec_lds = reinterpret_cast<int4 *>(wbase + K::A_BYTES + K::NB_BYTES);
  int32_t *run_of = reinterpret_cast<int32_t *>(rec_lds + 64);
  int32_t *run_pc = run_of + 64, *run_cn = run_pc + NC, *run_pre = run_cn + NC;     // run_pre[nr] = total
  const int Dx = g.dim[0], Dy = g.dim[1], Dz = g.dim[2];
  const int64_t Vi = (int64_t)Dx * Dy * Dz * g.dim[3];
  const int64_t q0 = ((int64_t)blockIdx.x * K::NW + wave) * NC;
  if (q0 >= Vi) return;                                // wave-uniform; no workgroup barrier below
  // ---- the wave's cells: lane (mod 16) owns one; occupied cells become runs ----
  const int c16 = lane & 15;
  const int64_t q = q0 + c16;
  const bool act = q < Vi;
  int pc = 0, cn = 0;
  {
    const int64_t qq = act ? q : q0;
    const int z = (int)(qq % Dz);
    int64_t t = qq / Dz;
    const int y = (int)(t % Dy);
    t /= Dy;
    pc = dc_cell(g, (int)(t % Dx), y, z, (int)(t / Dx));
    const int c0 = cell_n[pc];
    cn = act ? (c0 < g.k ? c0 : g.k) : 0;
  }
  const unsigned occ = (unsigned)(__ballot(lane < 16 && cn > 0) & 0xFFFFull);
  const int nr = __popc(occ);
  if (nr == 0) return;
  int incl = (lane < 16) ? cn : 0;                     // inclusive prefix of the counts over lanes 0..15
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    const int u = __shfl_up(incl, o, 64);
    if (c16 >= o) incl += u;
  }
  const int total = __shfl(incl, 15, 64);
  if (lane < 16 && cn > 0) {
    const int rj = __popc(occ & ((1u << lane) - 1u));
    run_pc[rj] = pc;
    run_cn[rj] = cn;
    run_pre[rj] = incl - cn;
  }
  if (lane == 0) run_pre[nr] = total;
  // parameters of this lane's four channels
  const int ch0 = 4 * li;
  float w0[4], w1[4], w2[4], al[4];
#pragma unroll
  for (int e = 0; e < 4; e++) {
    const int tc = (ch0 + e) % cg;
    w0[e] = w_pos[3 * tc + 0]; w1[e] = w_pos[3 * tc + 1]; w2[e] = w_pos[3 * tc + 2];
    al[e] = alpha ? alpha[tc] : 1.0f;
  }
  const float4 gw = *reinterpret_cast<const float4 *>(&ln_w[ch0]), gb = *reinterpret_cast<const float4 *>(&ln_b[ch0]);
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- neighbour cells of every run: offsets on the padded grid ----
  const int PY = g.pdim[1], PZ = g.pdim[2];
  for (int e = lane; e < nr * R3; e += 64) {
    const int j = e / R3, k = e - j * R3;
    const int dz = k / R2, t = k - dz * R2;
    nb_lds[e] = run_pc[j] + ((LO + t % R) * PY + (LO + t / R)) * PZ + (LO + dz);
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- A rows: a group per run, ~28 16-byte pieces in flight per batch ----
  for (int j = grp; j < nr; j += G) {
    float acc[P][4], den = 0.f;
#pragma unroll
    for (int pp = 0; pp < P; pp++)
#pragma unroll
      for (int qv = 0; qv < 4; qv++) acc[pp][qv] = 0.f;
    constexpr int NBAT = (R3 * P + 27) / 28, BR = (R3 + NBAT - 1) / NBAT;
#pragma unroll
    for (int bt = 0; bt < NBAT; bt++) {
      float4 v[BR][P];
      int vc[BR];
#pragma unroll
      for (int t = 0; t < BR; t++) {
        if (bt * BR + t < R3) {
          const int nb = nb_lds[j * R3 + bt * BR + t];
          const float *row = S + (int64_t)nb * RS + ch0;
          vc[t] = cell_n[nb];
#pragma unroll
          for (int pp = 0; pp < P; pp++) v[t][pp] = *reinterpret_cast<const float4 *>(&row[pp * C]);
        }
      }
#pragma unroll
      for (int t = 0; t < BR; t++) {
        if (bt * BR + t < R3) {
          den += (float)(vc[t] < g.k ? vc[t] : g.k);
#pragma unroll
          for (int pp = 0; pp < P; pp++) {
            acc[pp][0] += v[t][pp].x; acc[pp][1] += v[t][pp].y; acc[pp][2] += v[t][pp].z; acc[pp][3] += v[t][pp].w;
          }
        }
      }
    }
#pragma unroll
    for (int pp = 0; pp < P; pp++)                       // utils.py:80: the neighbourhood mean
      *reinterpret_cast<float4 *>(&A_lds[j * RS + pp * C + ch0]) =
          make_float4(acc[pp][0] / den, acc[pp][1] / den, acc[pp][2] / den, acc[pp][3] / den);
  }
  __builtin_amdgcn_wave_barrier();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // ---- voxels: 64 flat positions per pass ----
  const __amdgpu_buffer_rsrc_t r_out = dc_rsrc(out, (uint32_t)(n * C * IO_BYTES));
  for (int base = 0; base < total; base += 64) {
    const int t = base + lane;
    {
      const int tc = t < total ? t : total - 1;
      int j = 0;
#pragma unroll
      for (int s2 = 1; s2 < NC; s2++) j += (s2 < nr && run_pre[s2] <= tc) ? 1 : 0;
      rec_lds[lane] = slots[dc_slot(g, run_pc[j], tc - run_pre[j])];
      run_of[lane] = j;
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const int npos = total - base < 64 ? total - base : 64;
    for (int l = grp; l < npos; l += G) {
      const int4 rec = rec_lds[l];
      const int ra = run_of[l];
      float4 Av[P];
#pragma unroll
      for (int pp = 0; pp < P; pp++) Av[pp] = *reinterpret_cast<const float4 *>(&A_lds[ra * RS + pp * C + ch0]);
      float4 f4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (OP == LINK_OP_COSX) f4 = *reinterpret_cast<const float4 *>(&fin[(int64_t)rec.w * C + ch0]);
      float x = (float)rec.x, y = (float)rec.y, z = (float)rec.z;
      if (coord_div != 1.0f) { x = x / coord_div; y = y / coord_div; z = z / coord_div; }
      float th[4], sn[4], cs[4];
      bool big = false;
#pragma unroll
      for (int qv = 0; qv < 4; qv++) {
        th[qv] = theta_of(x, y, z, w0[qv], w1[qv], w2[qv], al[qv]);
        big |= !(fabsf(th[qv]) < 32768.0f);
      }
      if (__builtin_expect(__any(big), 0)) {
#pragma unroll
        for (int qv = 0; qv < 4; qv++) sincos_nocall(th[qv], sn[qv], cs[qv]);
      } else {
#pragma unroll
        for (int qv = 0; qv < 4; qv++) sincos_small(th[qv], sn[qv], cs[qv]);
      }
      const float a0[4] = {Av[0].x, Av[0].y, Av[0].z, Av[0].w}, a1[4] = {Av[1].x, Av[1].y, Av[1].z, Av[1].w};
      const float a2[4] = {Av[P - 1].x, Av[P - 1].y, Av[P - 1].z, Av[P - 1].w};
      const float fv[4] = {f4.x, f4.y, f4.z, f4.w};
      float nvv[4], sm = 0.f;
#pragma unroll
      for (int qv = 0; qv < 4; qv++) {
        float va;
        if (OP == LINK_OP_SIN) va = __fsub_rn(__fmul_rn(a0[qv], cs[qv]), __fmul_rn(a1[qv], sn[qv]));
        else va = __fadd_rn(__fmul_rn(a0[qv], cs[qv]), __fmul_rn(a1[qv], sn[qv]));
        if (OP == LINK_OP_COSX) va = __fadd_rn(va, __fsub_rn(a2[qv], __fmul_rn(fv[qv], th[qv])));
        nvv[qv] = va;
        sm += va;
      }
      sm = grp_sum<LPR>(sm);
      const float mean = sm * (1.0f / C);
      float qq = 0.f;
#pragma unroll
      for (int qv = 0; qv < 4; qv++) {
        const float d = nvv[qv] - mean;
        qq += d * d;
      }
      qq = grp_sum<LPR>(qq);
      const float rstd = 1.0f / sqrtf(qq * (1.0f / C) + eps);
      io_st4(r_out, (uint32_t)rec.w * (uint32_t)C + (uint32_t)ch0, true,
             make_float4((nvv[0] - mean) * rstd * gw.x + gb.x, (nvv[1] - mean) * rstd * gw.y + gb.y,
                         (nvv[2] - mean) * rstd * gw.z + gb.z, (nvv[3] - mean) * rstd * gw.w + gb.w));
    }
    __builtin_amdgcn_wave_barrier();
  }
}

template <int C, int OP, int R>
static int launch_k2_cells(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n, hipStream_t st) {
  using K = dc_k2c_cfg<C, OP, R>;
  const int64_t vi = (int64_t)g.dim[0] * g.dim[1] * g.dim[2] * g.dim[3];
  const int64_t wgs = (vi + (int64_t)K::NC * K::NW - 1) / ((int64_t)K::NC * K::NW);
  if (K::LDS_BYTES > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_gather_demod_cells<C, OP, R>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  hipLaunchKernelGGL((k_dc_gather_demod_cells<C, OP, R>), dim3((unsigned)wgs), dim3(64 * K::NW), K::LDS_BYTES, st, b->S, b->cell_n,
                     reinterpret_cast<const int4 *>(b->slots), b->fin, b->w_pos, b->alpha, b->ln_w, b->ln_b, d.cg, d.coord_div,
                     d.eps, n, g, b->out);
  return check_launch("link_dc_gather_demod");
}

template <int C>
static int run_cells_c(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n, hipStream_t st) {
#define LINK_K2C(OPV) return d.r == 2 ? launch_k2_cells<C, OPV, 2>(b, g, d, n, st) : launch_k2_cells<C, OPV, 3>(b, g, d, n, st)
  switch (d.op) {
    case LINK_OP_COS: LINK_K2C(LINK_OP_COS);
    case LINK_OP_SIN: LINK_K2C(LINK_OP_SIN);
    default: LINK_K2C(LINK_OP_COSX);
  }
#undef LINK_K2C
}

int run_gather_demod_cells(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n, hipStream_t st) {
  switch (d.c) {
    case 16: return run_cells_c<16>(b, g, d, n, st);
    case 32: return run_cells_c<32>(b, g, d, n, st);
    default: return run_cells_c<128>(b, g, d, n, st);
  }
}

