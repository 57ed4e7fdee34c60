// link_amd/csrc/bn.hip -- batch statistics of a [N, C] feature matrix for the training form of the row-wise
// BatchNorm the reference networks put after every convolution (torchsparse/nn/modules/norm.py:10-13: nn.BatchNorm1d
// applied to the feature rows; linkunet.py:18-92).  torch's channels-last statistics kernels move a 113k x 64 matrix
// at < 1 TB/s (36 us forward, 34 us backward-reduce per layer on the cfg3 encoder: 18 % of its training step); these
// two are plain column reductions at memory speed:
//   k_col_moments<false>   per workgroup: sum x, sum x^2 of its rows, per channel        (forward statistics)
//   k_col_moments<true>    per workgroup: sum g, sum g * xhat,  xhat = (x - mean) invstd  (backward reduction)
//   k_bn_finalize_*        sums the workgroup partials in a fixed order (deterministic), forms mean / invstd and
//                          updates the running statistics exactly as nn.BatchNorm1d does (biased variance for the
//                          normalisation, unbiased for running_var)
// Accumulation is double throughout (E[x^2] - mean^2 is then safe); a thread owns 4 consecutive channels (one 16-byte
// load per row), the threads of a workgroup cover 256 / (C/4) rows per pass, 4 passes in flight.
#include "common.h"

using namespace link;

// RELU (backward only): the gradient arriving is that of relu(y), y = (x - mean) * scale + shift -- rows are masked by y > 0,
// recomputed from x with the forward's own scale / shift (one multiply-add per element; no second [N, C] matrix is kept).
template <bool BWD, bool RELU = false>
__global__ void __launch_bounds__(256) k_col_moments(const float *__restrict__ x, const float *__restrict__ g,
                                                     const float *__restrict__ mean, const float *__restrict__ invstd,
                                                     int64_t n, int c, double *__restrict__ partial,
                                                     const float *__restrict__ scale = nullptr, const float *__restrict__ shift = nullptr) {
  __shared__ double red[256][8];
  const int tid = threadIdx.x;
  const int cq = c >> 2, rpp = 256 / cq;
  const int r = tid / cq, q = tid - r * cq;
  const bool act = r < rpp;
  double s0[4] = {0., 0., 0., 0.}, s1[4] = {0., 0., 0., 0.};
  float4 m = make_float4(0.f, 0.f, 0.f, 0.f), is = make_float4(1.f, 1.f, 1.f, 1.f);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (BWD && act) {
    m = *reinterpret_cast<const float4 *>(mean + 4 * q);
    is = *reinterpret_cast<const float4 *>(invstd + 4 * q);
    if (RELU) { sc = *reinterpret_cast<const float4 *>(scale + 4 * q); sh = *reinterpret_cast<const float4 *>(shift + 4 * q); }
  }
  if (act) {
    const int64_t stride = (int64_t)gridDim.x * rpp;
    auto add = [&](const float4 &v, const float4 &g_) {
      float4 gv = g_;
      if (BWD && RELU) {                                     // gradient of relu(y): zero where y <= 0
        gv.x = fmaf(v.x - m.x, sc.x, sh.x) > 0.f ? gv.x : 0.f; gv.y = fmaf(v.y - m.y, sc.y, sh.y) > 0.f ? gv.y : 0.f;
        gv.z = fmaf(v.z - m.z, sc.z, sh.z) > 0.f ? gv.z : 0.f; gv.w = fmaf(v.w - m.w, sc.w, sh.w) > 0.f ? gv.w : 0.f;
      }
      if (BWD) {
        s0[0] += gv.x; s0[1] += gv.y; s0[2] += gv.z; s0[3] += gv.w;
        s1[0] += (double)(gv.x * ((v.x - m.x) * is.x)); s1[1] += (double)(gv.y * ((v.y - m.y) * is.y));
        s1[2] += (double)(gv.z * ((v.z - m.z) * is.z)); s1[3] += (double)(gv.w * ((v.w - m.w) * is.w));
      } else {
        s0[0] += v.x; s0[1] += v.y; s0[2] += v.z; s0[3] += v.w;
        s1[0] += (double)v.x * v.x; s1[1] += (double)v.y * v.y; s1[2] += (double)v.z * v.z; s1[3] += (double)v.w * v.w;
      }
    };
    int64_t row = (int64_t)blockIdx.x * rpp + r;
    for (; row + 3 * stride < n; row += 4 * stride) {          // four independent rows in flight
      float4 v[4], gv[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        v[j] = *reinterpret_cast<const float4 *>(x + (row + j * stride) * c + 4 * q);
        gv[j] = BWD ? *reinterpret_cast<const float4 *>(g + (row + j * stride) * c + 4 * q) : v[j];
      }
#pragma unroll
      for (int j = 0; j < 4; j++) add(v[j], gv[j]);
    }
    for (; row < n; row += stride) {
      const float4 v = *reinterpret_cast<const float4 *>(x + row * c + 4 * q);
      const float4 gv = BWD ? *reinterpret_cast<const float4 *>(g + row * c + 4 * q) : v;
      add(v, gv);
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) { red[tid][j] = s0[j]; red[tid][4 + j] = s1[j]; }
  __syncthreads();
  if (act && r == 0) {                                        // fixed order over the workgroup's row slots
    double t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = red[q][j];
    for (int rr = 1; rr < rpp; rr++)
#pragma unroll
      for (int j = 0; j < 8; j++) t[j] += red[rr * cq + q][j];
    double *p = partial + (int64_t)blockIdx.x * 2 * c;
#pragma unroll
    for (int j = 0; j < 4; j++) { p[4 * q + j] = t[j]; p[c + 4 * q + j] = t[4 + j]; }
  }
}

// Finalisation: a workgroup owns 16 channels; 16 threads per channel sum every 16th workgroup partial (independent
// loads), their 16 subtotals are added in a fixed order -- deterministic, and ~50 loads deep instead of ~2000.
__device__ __forceinline__ void bn_gather_partials(const double *__restrict__ partial, int wgs, int c, int ch, int wl,
                                                   double (&red)[2][16][17], int cl, double &s, double &sq) {
  double a = 0., b = 0.;
  if (ch < c)
    for (int w = wl; w < wgs; w += 16) { a += partial[(int64_t)w * 2 * c + ch]; b += partial[(int64_t)w * 2 * c + c + ch]; }
  red[0][wl][cl] = a;
  red[1][wl][cl] = b;
  __syncthreads();
  s = 0.; sq = 0.;
  if (wl == 0)
    for (int k = 0; k < 16; k++) { s += red[0][k][cl]; sq += red[1][k][cl]; }
}

__global__ void __launch_bounds__(256) k_bn_finalize_forward(const double *__restrict__ partial, int wgs, int64_t n, int c,
                                                             float eps, float momentum, float *__restrict__ mean,
                                                             float *__restrict__ invstd, float *__restrict__ running_mean,
                                                             float *__restrict__ running_var, const float *__restrict__ weight,
                                                             const float *__restrict__ bias, float *__restrict__ scale,
                                                             float *__restrict__ shift) {
  __shared__ double red[2][16][17];
  const int cl = threadIdx.x & 15, wl = threadIdx.x >> 4, ch = blockIdx.x * 16 + cl;
  double s, sq;
  bn_gather_partials(partial, wgs, c, ch, wl, red, cl, s, sq);
  if (wl != 0 || ch >= c) return;
  const double mu = s / (double)n;
  double var = sq / (double)n - mu * mu;
  if (var < 0.) var = 0.;
  const float mf = (float)mu, isf = (float)(1.0 / sqrt(var + (double)eps));
  mean[ch] = mf;
  invstd[ch] = isf;
  if (scale) {                                               // y = (x - mean) * scale + shift (centred first: no cancellation)
    scale[ch] = (weight ? weight[ch] : 1.0f) * isf;
    shift[ch] = bias ? bias[ch] : 0.0f;
  }
  if (running_mean) running_mean[ch] = (float)((1.0 - (double)momentum) * (double)running_mean[ch] + (double)momentum * mu);
  if (running_var) {
    const double unb = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
    running_var[ch] = (float)((1.0 - (double)momentum) * (double)running_var[ch] + (double)momentum * unb);
  }
}

__global__ void __launch_bounds__(256) k_bn_finalize_backward(const double *__restrict__ partial, int wgs, int c,
                                                              float *__restrict__ sum_g, float *__restrict__ sum_gx,
                                                              const float *__restrict__ weight, const float *__restrict__ mean,
                                                              const float *__restrict__ invstd, int64_t n,
                                                              float *__restrict__ coef) {
  __shared__ double red[2][16][17];
  const int cl = threadIdx.x & 15, wl = threadIdx.x >> 4, ch = blockIdx.x * 16 + cl;
  double s, sx;
  bn_gather_partials(partial, wgs, c, ch, wl, red, cl, s, sx);
  if (wl != 0 || ch >= c) return;
  sum_g[ch] = (float)s;
  sum_gx[ch] = (float)sx;
  if (coef) {                                                // grad_x = a * g + bq * (x - mean) + cq  (coef = a | bq | cq)
    const double a = (double)(weight ? weight[ch] : 1.0f) * (double)invstd[ch];
    coef[ch] = (float)a;
    coef[c + ch] = (float)(-a * (double)invstd[ch] * sx / (double)n);
    coef[2 * c + ch] = (float)(-a * s / (double)n);
  }
}

// y = (x - mean) * scale + shift (centred first: no cancellation), optionally relu -- the normalisation itself, one pass.
// BWD: grad_x = a * g' + bq * (x - mean) + cq with g' = g masked by y > 0 when the forward ended in relu (coef = a | bq | cq of
// k_bn_finalize_backward, formed from the same masked gradient).  A thread owns 4 consecutive channels of a row.
template <bool BWD, bool RELU>
__global__ void __launch_bounds__(256) k_bn_apply(const float *__restrict__ x, const float *__restrict__ g, const float *__restrict__ mean,
                                                  const float *__restrict__ scale, const float *__restrict__ shift,
                                                  const float *__restrict__ coef, int64_t n, int c, float *__restrict__ out) {
  const int cq = c >> 2;
  const int64_t total = n * cq;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int q = (int)(e % cq);
    const float4 v = *reinterpret_cast<const float4 *>(x + 4 * e);
    const float4 m = *reinterpret_cast<const float4 *>(mean + 4 * q);
    float4 o;
    if (!BWD) {
      const float4 sc = *reinterpret_cast<const float4 *>(scale + 4 * q), sh = *reinterpret_cast<const float4 *>(shift + 4 * q);
      o.x = fmaf(v.x - m.x, sc.x, sh.x); o.y = fmaf(v.y - m.y, sc.y, sh.y); o.z = fmaf(v.z - m.z, sc.z, sh.z); o.w = fmaf(v.w - m.w, sc.w, sh.w);
      if (RELU) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    } else {
      float4 gv = *reinterpret_cast<const float4 *>(g + 4 * e);
      if (RELU) {
        const float4 sc = *reinterpret_cast<const float4 *>(scale + 4 * q), sh = *reinterpret_cast<const float4 *>(shift + 4 * q);
        gv.x = fmaf(v.x - m.x, sc.x, sh.x) > 0.f ? gv.x : 0.f; gv.y = fmaf(v.y - m.y, sc.y, sh.y) > 0.f ? gv.y : 0.f;
        gv.z = fmaf(v.z - m.z, sc.z, sh.z) > 0.f ? gv.z : 0.f; gv.w = fmaf(v.w - m.w, sc.w, sh.w) > 0.f ? gv.w : 0.f;
      }
      const float4 a = *reinterpret_cast<const float4 *>(coef + 4 * q), bq = *reinterpret_cast<const float4 *>(coef + c + 4 * q);
      const float4 cq_ = *reinterpret_cast<const float4 *>(coef + 2 * c + 4 * q);
      o.x = fmaf(v.x - m.x, bq.x, fmaf(gv.x, a.x, cq_.x)); o.y = fmaf(v.y - m.y, bq.y, fmaf(gv.y, a.y, cq_.y));
      o.z = fmaf(v.z - m.z, bq.z, fmaf(gv.z, a.z, cq_.z)); o.w = fmaf(v.w - m.w, bq.w, fmaf(gv.w, a.w, cq_.w));
    }
    *reinterpret_cast<float4 *>(out + 4 * e) = o;
  }
}

static bool bn_width_ok(int32_t c) { return c >= 4 && c <= 1024 && (c & 3) == 0; }

extern "C" int32_t link_bn_partial_workgroups(int64_t n, int32_t c) {
  if (n <= 0 || !bn_width_ok(c)) return 0;
  const int rpp = 256 / (c / 4);
  int64_t wgs = (n + (int64_t)rpp * 8 - 1) / ((int64_t)rpp * 8);     // >= 8 rows per thread slot
  if (wgs > 512) wgs = 512;
  if (wgs < 1) wgs = 1;
  return (int32_t)wgs;
}

extern "C" int link_bn_forward_stats(const float *x, int64_t n, int32_t c, float eps, float momentum, double *partial,
                                     float *mean, float *invstd, float *running_mean, float *running_var,
                                     const float *weight, const float *bias, float *scale, float *shift, void *stream) {
  if (n < 1 || !bn_width_ok(c) || n * (int64_t)c >= (1LL << 40)) return LINK_ERR_ARG;
  if (!x || !partial || !mean || !invstd || (scale == nullptr) != (shift == nullptr)) return LINK_ERR_ARG;
  const int wgs = link_bn_partial_workgroups(n, c);
  hipStream_t st = S(stream);
  hipLaunchKernelGGL(k_col_moments<false>, dim3(wgs), dim3(256), 0, st, x, nullptr, nullptr, nullptr, n, (int)c, partial);
  hipLaunchKernelGGL(k_bn_finalize_forward, dim3((c + 15) / 16), dim3(256), 0, st, partial, wgs, n, (int)c, eps, momentum, mean,
                     invstd, running_mean, running_var, weight, bias, scale, shift);
  return check_launch("link_bn_forward_stats");
}

extern "C" int link_bn_backward_reduce(const float *g, const float *x, const float *mean, const float *invstd, int64_t n,
                                       int32_t c, double *partial, float *sum_g, float *sum_gx, const float *weight,
                                       float *coef, void *stream) {
  if (n < 1 || !bn_width_ok(c) || n * (int64_t)c >= (1LL << 40)) return LINK_ERR_ARG;
  if (!g || !x || !mean || !invstd || !partial || !sum_g || !sum_gx) return LINK_ERR_ARG;
  const int wgs = link_bn_partial_workgroups(n, c);
  hipStream_t st = S(stream);
  hipLaunchKernelGGL(k_col_moments<true>, dim3(wgs), dim3(256), 0, st, x, g, mean, invstd, n, (int)c, partial);
  hipLaunchKernelGGL(k_bn_finalize_backward, dim3((c + 15) / 16), dim3(256), 0, st, partial, wgs, (int)c, sum_g, sum_gx, weight, mean, invstd, n, coef);
  return check_launch("link_bn_backward_reduce");
}

static unsigned bn_apply_grid(int64_t n, int32_t c) {
  const int64_t wgs = (n * (c / 4) + 256 * 4 - 1) / (256 * 4);       // ~4 pieces per thread
  return (unsigned)(wgs < 1 ? 1 : (wgs > 4096 ? 4096 : wgs));
}

extern "C" int link_bn_apply_forward(const float *x, const float *mean, const float *scale, const float *shift, int64_t n, int32_t c,
                                     int32_t relu, float *y, void *stream) {
  if (n < 1 || !bn_width_ok(c) || n * (int64_t)c >= (1LL << 40) || !x || !mean || !scale || !shift || !y) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (relu) hipLaunchKernelGGL((k_bn_apply<false, true>), dim3(bn_apply_grid(n, c)), dim3(256), 0, st, x, nullptr, mean, scale, shift, nullptr, n, (int)c, y);
  else hipLaunchKernelGGL((k_bn_apply<false, false>), dim3(bn_apply_grid(n, c)), dim3(256), 0, st, x, nullptr, mean, scale, shift, nullptr, n, (int)c, y);
  return check_launch("link_bn_apply_forward");
}

extern "C" int link_bn_backward_reduce_relu(const float *g, const float *x, const float *mean, const float *invstd, const float *scale,
                                            const float *shift, int64_t n, int32_t c, double *partial, float *sum_g, float *sum_gx,
                                            const float *weight, float *coef, void *stream) {
  if (n < 1 || !bn_width_ok(c) || n * (int64_t)c >= (1LL << 40)) return LINK_ERR_ARG;
  if (!g || !x || !mean || !invstd || !scale || !shift || !partial || !sum_g || !sum_gx) return LINK_ERR_ARG;
  const int wgs = link_bn_partial_workgroups(n, c);
  hipStream_t st = S(stream);
  hipLaunchKernelGGL((k_col_moments<true, true>), dim3(wgs), dim3(256), 0, st, x, g, mean, invstd, n, (int)c, partial, scale, shift);
  hipLaunchKernelGGL(k_bn_finalize_backward, dim3((c + 15) / 16), dim3(256), 0, st, partial, wgs, (int)c, sum_g, sum_gx, weight, mean, invstd, n, coef);
  return check_launch("link_bn_backward_reduce_relu");
}

extern "C" int link_bn_apply_backward(const float *g, const float *x, const float *mean, const float *coef, const float *scale,
                                      const float *shift, int64_t n, int32_t c, float *gx, void *stream) {
  if (n < 1 || !bn_width_ok(c) || n * (int64_t)c >= (1LL << 40) || !g || !x || !mean || !coef || !gx) return LINK_ERR_ARG;
  if ((scale == nullptr) != (shift == nullptr)) return LINK_ERR_ARG;
  hipStream_t st = S(stream);
  if (scale) hipLaunchKernelGGL((k_bn_apply<true, true>), dim3(bn_apply_grid(n, c)), dim3(256), 0, st, x, g, mean, scale, shift, coef, n, (int)c, gx);
  else hipLaunchKernelGGL((k_bn_apply<true, false>), dim3(bn_apply_grid(n, c)), dim3(256), 0, st, x, g, mean, nullptr, nullptr, coef, n, (int)c, gx);
  return check_launch("link_bn_apply_backward");
}
