// link_amd/csrc/conv.hip -- row N1 of SURVEY.md section 8f: the stride-1 submanifold sparse convolution
// that ELKBlock.local_mix runs (spnn.Conv3d(inc, inc, kernel_size=3), linkunet.py:109,125), for gfx950.
//
// The reference (torchsparse-u/torchsparse/backend/convolution/convolution_cuda.cu:53-165) loops over the
// 27 kernel offsets and, per offset, gathers the participating input rows into a buffer, calls cuBLAS,
// and scatter-adds the result: 27 x (gather kernel + GEMM + scatter kernel), every intermediate
// through HBM.  Here it is ONE output-stationary kernel: a wave owns NT tiles of 16 output voxels and
// keeps their [16 x C] accumulators in registers across the whole offset loop; per offset the
// workgroup stages W_k (transposed, padded) in LDS once for all its 4*NT tiles, each lane gathers the
// neighbour row of its voxel straight into the MFMA B-operand layout (one dwordx4 per 16-channel
// tile), and v_mfma_f32_16x16x4_f32 accumulates D[co][voxel] += W_k^T[co][ci] * F[nbr][ci] (exact f32).
// Tiles in which no voxel has the offset's neighbour skip the MFMAs.  Output is written once.
//
// out[v, co] = sum_k sum_ci feats[nbr[v, k], ci] * w[k, ci, co]          (nbr < 0: absent)
#include "common.h"

using namespace link;

typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int C, int NT>
__global__ void __launch_bounds__(256) k_subm_conv_mfma(const float *__restrict__ feats,
                                                        const int32_t *__restrict__ nbr,
                                                        const float *__restrict__ w, int64_t n, int kvol,
                                                        float *__restrict__ out) {
  constexpr int T = C / 16;
  constexpr int LDW = C + 4;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *wt_lds = reinterpret_cast<float *>(smem_raw);          // W_k^T [co][ci], row stride LDW
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int64_t tiles = (n + 15) / 16;
  const int64_t tiles_per_pass = (int64_t)gridDim.x * 4 * NT;
  for (int64_t base = 0; base < tiles; base += tiles_per_pass) {
    const int64_t tile0 = base + ((int64_t)blockIdx.x * 4 + wave) * NT;
    floatx4 acc[NT][T];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
      for (int tp = 0; tp < T; tp++) acc[j][tp] = (floatx4){0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < kvol; k++) {
      __syncthreads();                              // every wave is done with the previous W_k
      const float *wk = w + (int64_t)k * C * C;     // [ci][co]
      for (int e = tid * 4; e < C * C; e += 256 * 4) {
        const int ci = e / C, co = e - ci * C;
        const float4 w4 = *reinterpret_cast<const float4 *>(&wk[e]);
        wt_lds[(co + 0) * LDW + ci] = w4.x; wt_lds[(co + 1) * LDW + ci] = w4.y;
        wt_lds[(co + 2) * LDW + ci] = w4.z; wt_lds[(co + 3) * LDW + ci] = w4.w;
      }
      __syncthreads();
      int id[NT];
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const int64_t v = (tile0 + j) * 16 + li;
        id[j] = (v < n) ? nbr[v * kvol + k] : -1;
      }
      float4 f[NT][T];
#pragma unroll
      for (int j = 0; j < NT; j++) {                // all gathers of the step back to back
        const int64_t row = (id[j] >= 0) ? id[j] : 0;
#pragma unroll
        for (int t = 0; t < T; t++) f[j][t] = *reinterpret_cast<const float4 *>(&feats[row * C + 16 * t + 4 * g]);
      }
#pragma unroll
      for (int j = 0; j < NT; j++) {
        const bool has = id[j] >= 0;
        if (!__any(has)) continue;                  // wave-uniform: no voxel of the tile has neighbour k
#pragma unroll
        for (int t = 0; t < T; t++) {
          const float fx = has ? f[j][t].x : 0.f, fy = has ? f[j][t].y : 0.f;
          const float fz = has ? f[j][t].z : 0.f, fw = has ? f[j][t].w : 0.f;
#pragma unroll
          for (int tp = 0; tp < T; tp++) {
            const float4 a = *reinterpret_cast<const float4 *>(&wt_lds[(16 * tp + li) * LDW + 16 * t + 4 * g]);
            acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, fx, acc[j][tp], 0, 0, 0);
            acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, fy, acc[j][tp], 0, 0, 0);
            acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, fz, acc[j][tp], 0, 0, 0);
            acc[j][tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, fw, acc[j][tp], 0, 0, 0);
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NT; j++) {
      const int64_t v = (tile0 + j) * 16 + li;
      if (v < n) {
#pragma unroll
        for (int tp = 0; tp < T; tp++)
          *reinterpret_cast<float4 *>(&out[v * C + 16 * tp + 4 * g]) =
              make_float4(acc[j][tp][0], acc[j][tp][1], acc[j][tp][2], acc[j][tp][3]);
      }
    }
  }
}

// any Cin / Cout <= 256: one wave per output voxel, lanes = output channels (no MFMA; small widths)
template <int CPL>
__global__ void __launch_bounds__(256) k_subm_conv_generic(const float *__restrict__ feats,
                                                           const int32_t *__restrict__ nbr,
                                                           const float *__restrict__ w, int64_t n, int cin,
                                                           int cout, int kvol, float *__restrict__ out) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (v >= n) return;
  float acc[CPL];
#pragma unroll
  for (int q = 0; q < CPL; q++) acc[q] = 0.f;
  for (int k = 0; k < kvol; k++) {
    const int id = nbr[v * kvol + k];
    if (id < 0) continue;                            // wave-uniform
    const float *row = feats + (int64_t)id * cin;
    const float *wk = w + (int64_t)k * cin * cout;
    for (int ci = 0; ci < cin; ci++) {
      const float fv = row[ci];
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        const int co = lane + 64 * q;
        if (co < cout) acc[q] = fmaf(fv, wk[(int64_t)ci * cout + co], acc[q]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < CPL; q++) {
    const int co = lane + 64 * q;
    if (co < cout) out[v * cout + co] = acc[q];
  }
}

static int g_conv_wgs = 512;   // 2 workgroups per CU: one stages W_k while the other runs MFMAs

template <int C>
static int launch_conv_mfma(const float *feats, const int32_t *nbr, const float *w, int64_t n, int kvol,
                            float *out, hipStream_t st) {
  constexpr int NT = 4;
  const size_t lds = (size_t)C * (C + 4) * sizeof(float);
  if (lds > 64 * 1024) {
    static bool done = false;
    if (!done) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_subm_conv_mfma<C, NT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      done = true;
    }
  }
  const int64_t tiles = (n + 15) / 16;
  int64_t wgs = (tiles + 4 * NT - 1) / (4 * NT);
  if (wgs > g_conv_wgs) wgs = g_conv_wgs;
  hipLaunchKernelGGL((k_subm_conv_mfma<C, NT>), dim3((unsigned)wgs), dim3(256), lds, st, feats, nbr, w, n, kvol, out);
  return check_launch("link_subm_conv_forward");
}

extern "C" int link_subm_conv_forward(const float *feats, const int32_t *nbr, const float *w, int64_t n,
                                      int32_t cin, int32_t cout, int32_t kvol, float *out, void *stream) {
  if (n < 0 || cin <= 0 || cout <= 0 || cin > 256 || cout > 256 || kvol <= 0) return LINK_ERR_ARG;
  if (n == 0) return LINK_OK;
  if (!feats || !nbr || !w || !out) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (cin == cout && (cin & 15) == 0 && cin <= 128) {
    switch (cin) {
      case 16: return launch_conv_mfma<16>(feats, nbr, w, n, kvol, out, st);
      case 32: return launch_conv_mfma<32>(feats, nbr, w, n, kvol, out, st);
      case 48: return launch_conv_mfma<48>(feats, nbr, w, n, kvol, out, st);
      case 64: return launch_conv_mfma<64>(feats, nbr, w, n, kvol, out, st);
      case 80: return launch_conv_mfma<80>(feats, nbr, w, n, kvol, out, st);
      case 96: return launch_conv_mfma<96>(feats, nbr, w, n, kvol, out, st);
      case 112: return launch_conv_mfma<112>(feats, nbr, w, n, kvol, out, st);
      default: return launch_conv_mfma<128>(feats, nbr, w, n, kvol, out, st);
    }
  }
  dim3 grid(blocks_for(n * 64, 256)), block(256);
  const int cpl = (cout + 63) / 64;
  if (cpl == 1) hipLaunchKernelGGL(k_subm_conv_generic<1>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out);
  else if (cpl == 2) hipLaunchKernelGGL(k_subm_conv_generic<2>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out);
  else if (cpl == 3) hipLaunchKernelGGL(k_subm_conv_generic<3>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out);
  else hipLaunchKernelGGL(k_subm_conv_generic<4>, grid, block, 0, st, feats, nbr, w, n, (int)cin, (int)cout, (int)kvol, out);
  return check_launch("link_subm_conv_forward");
}
