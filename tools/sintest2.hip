// tools/sintest2.hip -- accuracy and issue cost of the hardware-trig sincos (elk_common.h: sincos_hw) against the
// polynomial path (sincos_small), over theta in [-R, R]:   hipcc --offload-arch=gfx950 -O3 -I link_amd/csrc -I include tools/sintest2.hip -o /tmp/sintest2 && /tmp/sintest2
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "elk_common.h"
using namespace link;
template <int MODE>
__global__ void k(const float* x, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float sn, cs;
    if (MODE == 0) sincos_small(x[i], sn, cs); else sincos_hw(x[i], sn, cs);
    s[i] = sn; c[i] = cs;
  }
}
template <int MODE>
__global__ void spin(float* out, int iters) {           // dependent-free issue cost: 8 independent chains per lane
  float a[8], acc = 0.f;
  for (int j = 0; j < 8; j++) a[j] = threadIdx.x * 0.37f + j;
  for (int it = 0; it < iters; it++)
    for (int j = 0; j < 8; j++) {
      float sn, cs;
      if (MODE == 0) sincos_small(a[j], sn, cs); else sincos_hw(a[j], sn, cs);
      acc += sn * cs; a[j] += 0.61f;
    }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  const int n = 1 << 22;
  float *hx = (float*)malloc(n * 4), *hs = (float*)malloc(n * 4), *hc = (float*)malloc(n * 4);
  float *dx, *ds, *dc;
  hipMalloc(&dx, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  for (float R : {8.f, 200.f, 3000.f, 30000.f}) {
    srand(1);
    for (int i = 0; i < n; i++) hx[i] = (float)((rand() / (double)RAND_MAX * 2.0 - 1.0) * R);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; mode++) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
      else hipLaunchKernelGGL(k<1>, dim3(n / 256), dim3(256), 0, 0, dx, ds, dc, n);
      hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, n * 4, hipMemcpyDeviceToHost);
      double es = 0, ec = 0;
      for (int i = 0; i < n; i++) {
        es = fmax(es, fabs(hs[i] - sin((double)hx[i])));
        ec = fmax(ec, fabs(hc[i] - cos((double)hx[i])));
      }
      printf("|theta| <= %7.0f  %s: max abs err sin %.3e  cos %.3e\n", R, mode ? "hw  " : "poly", es, ec);
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 2; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(spin<0>, dim3(1024), dim3(256), 0, 0, ds, 2000);
      else hipLaunchKernelGGL(spin<1>, dim3(1024), dim3(256), 0, 0, ds, 2000);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.3f ms for 1024 x 256 lanes x 16000 sincos  (%.2f ns per wave-sincos per SIMD)\n", mode ? "hw  " : "poly", ms,
                      ms * 1e6 / (16000.0 * 1024 * 4 / (256 * 4)));
    }
  }
  return 0;
}
