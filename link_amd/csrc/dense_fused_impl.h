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

#include "dense_k1_impl.h"

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

template <int C, int OP, int NB>
static int launch_k1(const link_dc_buffers_t *b, const link_dc_grid_t &g, const link_elk_desc_t &d, int64_t n,
                     bool warm, hipStream_t st) {
  // (the software-pipelined tile variant -- PIPE, 253 registers -- was re-measured in round 6 in the timed geometry, where its 256
  // registers do fit beside two gather waves: 35.7 against 34.2 us / frame; instantiations removed, docs/experiments.md section 6)
  return launch_k1p<C, OP, NB, false>(b, g, d, n, warm, st);
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

#include "dense_k2_cfg.h"

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

}  // namespace DC_IO_NS
