// tools/pkbench.hip -- is v_pk_fma_f32 issued at the rate of v_fma_f32 on gfx950?  8 independent chains per lane,
// 4 waves per SIMD resident, fixed trip count; prints ns per (wave-instruction) for both forms.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
  f2 x[8];
  for (int j = 0; j < 8; j++) x[j] = (f2)(threadIdx.x + j, threadIdx.x - j);
  const f2 av = (f2)(a, a * 1.5f), bv = (f2)(b, b * 0.5f);
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (PK) x[j] = __builtin_elementwise_fma(x[j], av, bv);
      else { x[j].x = __builtin_fmaf(x[j].x, av.x, bv.x); asm volatile("" : "+v"(x[j].x)); }
    }
  }
  float s = 0;
  for (int j = 0; j < 8; j++) s += x[j].x + x[j].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *d; hipMalloc(&d, 1024 * 1024 * 4 * 4);
  const int iters = 20000, wgs = 256 * 4;      // 4 WGs of 4 waves per CU -> 4 waves per SIMD
  for (int rep = 0; rep < 2; rep++)
    for (int pk = 0; pk < 2; pk++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (pk) hipLaunchKernelGGL(k<true>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
      else hipLaunchKernelGGL(k<false>, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = 4.0 * iters * 8;          // wave-instructions issued on one SIMD
      printf("%s: %.3f ms, %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms,
             ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    }
  return 0;
}
