// link_amd/csrc/dense_gather_sparse_impl.h -- gather + de-modulate kernel of the SPARSE-CELL layout (round 4; included inside
// DC_IO_NS by dense_fused_impl.h; C ABI: link_elk_core_sparse_forward, include/link_amd.h section E).
//
// The layout.  LiDAR-shaped frames occupy 1-2 % of their block grid (S-kitti stage 1: 13 845 of 982 464 cells), so the
// dense-cell kernels -- which stream every cell of the grid -- do not fit them, and the general layout pays a four-launch
// index (count -> scan over the whole grid -> place -> segment sort: ~25 us, profiles/r03_v9_lidar_stages.jsonl) to number
// the occupied blocks.  The sparse-cell layout keeps the dense-cell ADDRESSING (a block's table row, counter and slot list
// sit at its padded grid cell; neighbours by arithmetic; memory is not what a sparse grid costs on a 288 GB part -- only the
// occupied rows are ever touched) and makes the ITERATION sparse: the slot insert marks, per voxel id, the cell that voxel
// was the first of (`occ[i]`, k_dc_index_sparse), and both kernels walk ranges of voxel ids and take the cells marked there.
// No scan, no sort, no block numbering, no list compaction, no atomics besides the insert's one per voxel:
//   k_dc_index_sparse                 coords -> cell, rank = cnt[cell]++, slots[cell][rank]; occ[i] = cell if rank == 0;
//                                     cell_n of the PREVIOUS frame's cells back to zero (its occ array)
//   k_dc_premix_modsum<.., SPARSE>    dense_fused_impl.h: the cells of a wave's 64 voxel ids -> id-ordered records -> MFMA
//                                     tiles -> S[cell], cell_n[cell], sorted records back to slots[cell]
//   k_dc_gather_demod_sparse (here)   the cells of a wave's 64 voxel ids -> their records (already id-ordered) into LDS as a
//                                     block-major position list -> per block the r^3 neighbour rows (present iff cell_n > 0)
//                                     -> normalised sums in LDS -> per voxel theta / sincos / de-modulate / LayerNorm -> out
// Three launches per step with the index rebuilt (reference: two hash-table builds + torch.unique + scatter kernels per
// call, utils.py:44-84, query_cuda.cu:9-58).  Results do not depend on which voxel of a cell came first.
// Cells of up to 64 voxels (the host picks the layout by the slot capacity): a cell is one wave's serial work in the
// pre_mix kernel, blocks of hundreds of voxels stay on the general layout's tile form, which splits them over waves.
#pragma once

template <int C, int OP, int R>
struct dc_gs_cfg {
  static constexpr int LPR = C / 4;                    // lanes of one feature row (16 bytes each)
  static constexpr int G = 64 / LPR;                   // lane groups of a wave
  static constexpr int P = op_parts<OP>::value;
  static constexpr int RS = P * C;                     // floats of one table row
  static constexpr int R3 = R * R * R;
  static constexpr int LCAP = DC_SP_LCAP;              // records of one chunk of cells in LDS (>= the largest cell: 64)
  static constexpr int WP = 4 * G < 64 ? 4 * G : 64;   // positions per pass: 4 voxel steps per lane group
  static constexpr int STEPS = WP / G;
  static constexpr int RMAX = WP / 2 < 8 ? WP / 2 : 8; // blocks whose neighbour sums sit in LDS at a time
  static constexpr int NW = 4;
  static constexpr int LIST_OFF = 0;
  static constexpr int SCELL_OFF = LCAP * 16;
  static constexpr int SSEG_OFF = SCELL_OFF + LCAP * 4;
  static constexpr int A_OFF = SSEG_OFF + LCAP * 4;
  static constexpr int NB_OFF = A_OFF + RMAX * RS * 4;
  static constexpr int CN_OFF = NB_OFF + ((RMAX * R3 * 4 + 15) & ~15);
  static constexpr int RUN_OFF = CN_OFF + ((RMAX * R3 * 4 + 15) & ~15);
  static constexpr int WAVE_BYTES = RUN_OFF + ((RMAX * 4 + 15) & ~15);
  static constexpr int LDS_BYTES = NW * WAVE_BYTES;
};

template <int C, int OP, int R>
__global__ void __launch_bounds__(256) k_dc_gather_demod_sparse(
    const float *__restrict__ S, const int32_t *__restrict__ cell_n, const int4 *__restrict__ slots,
    const int32_t *__restrict__ occ, const float *__restrict__ fin, const float *__restrict__ w_pos,
    const float *__restrict__ alpha, const float *__restrict__ ln_w, const float *__restrict__ ln_b, int cg, float coord_div,
    float eps, int64_t n, link_dc_grid_t g, int ipw, void *__restrict__ out) {
  using K = dc_gs_cfg<C, OP, R>;
  constexpr int LPR = K::LPR, G = K::G, P = K::P, RS = K::RS, R2 = R * R, R3 = K::R3, WP = K::WP;
  constexpr int LO = -((R + 1) / 2) + 1;               // nn/utils/kernel.py:21: r = 2 -> {0, 1}, r = 3 -> {-1, 0, 1}
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & (LPR - 1), grp = lane / LPR;
  char *wbase = smem_raw + wave * K::WAVE_BYTES;
  int4 *list = reinterpret_cast<int4 *>(wbase + K::LIST_OFF);
  int32_t *scell = reinterpret_cast<int32_t *>(wbase + K::SCELL_OFF);
  int32_t *sseg = reinterpret_cast<int32_t *>(wbase + K::SSEG_OFF);
  const __amdgpu_buffer_rsrc_t r_slots = dc_rsrc(slots, (uint32_t)((int64_t)g.vp * g.k * 16));
  float *A_lds = reinterpret_cast<float *>(wbase + K::A_OFF);
  int32_t *nb_lds = reinterpret_cast<int32_t *>(wbase + K::NB_OFF);
  int32_t *cn_lds = reinterpret_cast<int32_t *>(wbase + K::CN_OFF);
  int32_t *run_cell = reinterpret_cast<int32_t *>(wbase + K::RUN_OFF);
  const int64_t c_begin = ((int64_t)blockIdx.x * K::NW + wave) * ipw;       // ipw <= 64 voxel ids per wave
  if (c_begin >= n) return;                            // wave-uniform; nothing below is a workgroup barrier
  const int c_end = (int)(c_begin + ipw < n ? c_begin + ipw : n);
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
  const int sx = g.pdim[1] * g.pdim[2], sy = g.pdim[2];           // cell strides of the padded grid (z fastest)

  for (int chunk = (int)c_begin; chunk < c_end;) {
    const int nrem = c_end - chunk;                    // <= 64
    // ---- cell lanes: the cell this voxel id was first in (0: none), its count, its inline records ----
    const int pc = lane < nrem ? occ[chunk + lane] : 0;
    int nv = pc ? cell_n[pc] : 0;
    nv = nv < g.k ? nv : g.k;
    nv = nv < K::LCAP ? nv : K::LCAP;
    int incl = nv;                                      // inclusive prefix over the wave (DPP row scan + row totals)
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
    {
      const int t0 = __builtin_amdgcn_readlane(incl, 15), t1 = __builtin_amdgcn_readlane(incl, 31), t2 = __builtin_amdgcn_readlane(incl, 47);
      const int row = lane >> 4;
      incl += row == 0 ? 0 : (row == 1 ? t0 : (row == 2 ? t0 + t1 : t0 + t1 + t2));
    }
    const unsigned long long fit = __ballot(lane < nrem && incl <= K::LCAP);
    const int nfit = __builtin_amdgcn_readfirstlane(__popcll(fit));      // >= 1: a cell never exceeds LCAP
    const int Ttot = __builtin_amdgcn_readlane(incl, nfit - 1);
    // the records are in id order already (the pre_mix kernel wrote them back): all of them in one round trip
    dc_sparse_fetch<false>(g, r_slots, lane, lane < nfit, pc, nv, incl - nv, Ttot, list, scell, sseg, nullptr);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    // ---- passes of WP positions of the block-major list ----
    for (int p0 = 0; p0 < Ttot; p0 += WP) {
      const int npos = Ttot - p0 < WP ? Ttot - p0 : WP;
      const bool valid = lane < npos;
      const int blk = scell[p0 + (valid ? lane : npos - 1)];
      const int prev = __shfl_up(blk, 1, 64);
      const bool head = valid && (lane == 0 || blk != prev);
      const unsigned long long hm = __ballot(head);
      const int nruns = __popcll(hm);
      const int my_run = __popcll(hm & ((2ull << lane) - 1ull)) - 1;      // run of this lane's position
      const int4 *rec_lds = list + p0;
      // cos_x: the fin rows of this group's positions are requested now (HBM / L2), they land while the sums are formed
      float4 fpre[OP == LINK_OP_COSX ? K::STEPS : 1];
      if (OP == LINK_OP_COSX) {
#pragma unroll
        for (int i = 0; i < K::STEPS; i++) {
          const int l = grp + i * G;
          const int id = rec_lds[l < npos ? l : npos - 1].w;
          fpre[i] = *reinterpret_cast<const float4 *>(&fin[(int64_t)id * C + ch0]);
        }
      }
      for (int rb = 0; rb < nruns; rb += K::RMAX) {
        const int nr = nruns - rb < K::RMAX ? nruns - rb : K::RMAX;
        if (head && my_run >= rb && my_run < rb + nr) run_cell[my_run - rb] = blk;
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // neighbour cells of the pass's blocks by arithmetic on the padded grid (the halo keeps them inside the table); a
        // neighbour is present iff its published count is non-zero, an absent one points at cell 0 (never written: zero row)
        for (int e_ = lane; e_ < nr * R3; e_ += 64) {
          const int j = e_ / R3, k = e_ - j * R3;
          const int dz = k / R2, t = k - dz * R2;
          const int nbc = run_cell[j] + (LO + t % R) * sx + (LO + t / R) * sy + (LO + dz);
          const int cn = cell_n[nbc];
          nb_lds[e_] = cn > 0 ? nbc : 0;
          cn_lds[e_] = cn > 0 ? cn : 0;
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        // ---- A rows: a lane group per block ----
        for (int j = grp; j < nr; j += G) {
          float acc[P][4], den = 0.f;
#pragma unroll
          for (int pp = 0; pp < P; pp++)
#pragma unroll
            for (int q = 0; q < 4; q++) acc[pp][q] = 0.f;
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
                vd[t] = (float)cn_lds[j * R3 + bt * BR + t];
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
          for (int pp = 0; pp < P; pp++)                 // utils.py:80: the neighbourhood mean
            *reinterpret_cast<float4 *>(&A_lds[j * RS + pp * C + ch0]) =
                make_float4(acc[pp][0] / den, acc[pp][1] / den, acc[pp][2] / den, acc[pp][3] / den);
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- voxels of the pass: position grp, grp + G, ... ----
#pragma unroll
        for (int i = 0; i < K::STEPS; i++) {
          const int l = grp + i * G;
          const int4 rec = rec_lds[l < npos ? l : npos - 1];
          const float4 f4 = OP == LINK_OP_COSX ? fpre[i] : make_float4(0.f, 0.f, 0.f, 0.f);
          const int rl = __popcll(hm & ((2ull << l) - 1ull)) - 1;
          const bool ok = l < npos && rl >= rb && rl < rb + nr;
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
            if (OP == LINK_OP_SIN) va = __fsub_rn(__fmul_rn(a0[q], cs[q]), __fmul_rn(a1[q], sn[q]));      // linkunet.py:148
            else va = __fadd_rn(__fmul_rn(a0[q], cs[q]), __fmul_rn(a1[q], sn[q]));                         // :162
            if (OP == LINK_OP_COSX) va = __fadd_rn(va, __fsub_rn(a2[q], link_mul_rn(fv[q], th[q])));         // :176
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
      }
    }
    __builtin_amdgcn_wave_barrier();
    chunk += nfit;
  }
}

template <int C, int OP, int R>
static int launch_gs(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     const int32_t *occ, hipStream_t st) {
  using K = dc_gs_cfg<C, OP, R>;
  const int ipw = dc_sparse_ids_per_wave(b, n);
  const int64_t wgs = (n + (int64_t)ipw * K::NW - 1) / ((int64_t)ipw * K::NW);
  if (K::LDS_BYTES > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_dc_gather_demod_sparse<C, OP, R>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, K::LDS_BYTES);
  hipLaunchKernelGGL((k_dc_gather_demod_sparse<C, OP, R>), dim3((unsigned)wgs), dim3(256), K::LDS_BYTES, st, b->S, b->cell_n,
                     reinterpret_cast<const int4 *>(b->slots), occ, b->fin, b->w_pos, b->alpha, b->ln_w, b->ln_b, d.cg,
                     d.coord_div, d.eps, n, g, ipw, b->out);
  return check_launch("link_dc_gather_demod(sparse)");
}

template <int C>
static int dispatch_gs(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                       const int32_t *occ, hipStream_t st) {
#define LINK_GS(OPV, RV) return launch_gs<C, OPV, RV>(b, g, d, n, occ, st)
  if (d.r == 3) {
    switch (d.op) {
      case LINK_OP_COS: LINK_GS(LINK_OP_COS, 3);
      case LINK_OP_SIN: LINK_GS(LINK_OP_SIN, 3);
      default: LINK_GS(LINK_OP_COSX, 3);
    }
  }
  switch (d.op) {
    case LINK_OP_COS: LINK_GS(LINK_OP_COS, 2);
    case LINK_OP_SIN: LINK_GS(LINK_OP_SIN, 2);
    default: LINK_GS(LINK_OP_COSX, 2);
  }
#undef LINK_GS
}

int run_gather_demod_sparse(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                            const int32_t *occ, hipStream_t st) {
  switch (d.c) {
    case 16: return dispatch_gs<16>(b, g, d, n, occ, st);
    case 32: return dispatch_gs<32>(b, g, d, n, occ, st);
    default: return dispatch_gs<64>(b, g, d, n, occ, st);
  }
}
