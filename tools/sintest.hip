// accuracy of v_sin_f32 / v_cos_f32 (input in revolutions) on a reduced argument r in [-pi/4, pi/4]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const float* r, float* s, float* c, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    float x = r[i] * 0.15915494309189535f;
    s[i] = __builtin_amdgcn_sinf(x);
    c[i] = __builtin_amdgcn_cosf(x);
  }
}
int main() {
  const int n = 1 << 22;
  float *hr = (float*)malloc(n * 4), *hs = (float*)malloc(n * 4), *hc = (float*)malloc(n * 4);
  for (int i = 0; i < n; i++) hr[i] = (float)((i + 0.5) / n * 2.0 - 1.0) * 0.78539816339f;
  float *dr, *ds, *dc;
  hipMalloc(&dr, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
  hipMemcpy(dr, hr, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dr, ds, dc, n);
  hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, dc, n * 4, hipMemcpyDeviceToHost);
  double es = 0, ec = 0;
  for (int i = 0; i < n; i++) {
    es = fmax(es, fabs(hs[i] - sin((double)hr[i])));
    ec = fmax(ec, fabs(hc[i] - cos((double)hr[i])));
  }
  printf("v_sin_f32 max abs err %.3e   v_cos_f32 max abs err %.3e  (fp32 eps 5.96e-8)\n", es, ec);
  return 0;
}
