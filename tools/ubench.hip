// tools/ubench.hip -- calibration microbenchmarks for gfx950 (not part of the product).
// hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench && ./ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float floatx4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void __launch_bounds__(256) k_mfma(float *out, int iters) {
  floatx4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
  for (int i = 0; i < iters; i++) {
    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
}
__global__ void __launch_bounds__(256) k_fma(float *out, int iters) {
  float a = threadIdx.x, b = 1.0001f, c0 = 0, c1 = 1, c2 = 2, c3 = 3, c4 = 4, c5 = 5, c6 = 6, c7 = 7;
  for (int i = 0; i < iters; i++) {
    c0 = fmaf(a, b, c0); c1 = fmaf(a, b, c1); c2 = fmaf(a, b, c2); c3 = fmaf(a, b, c3);
    c4 = fmaf(a, b, c4); c5 = fmaf(a, b, c5); c6 = fmaf(a, b, c6); c7 = fmaf(a, b, c7);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
}
__global__ void __launch_bounds__(256) k_read(const float4 *in, float *out, size_t n4) {
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = in[i]; acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_write(float4 *out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    out[i] = make_float4(1, 2, 3, 4);
}
__global__ void __launch_bounds__(256) k_copy(const float4 *in, float4 *out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    out[i] = in[i];
}
// random 256-byte row gather: rows permuted
__global__ void __launch_bounds__(256) k_rowgather(const float4 *in, const int *perm, float4 *out, int rows) {
  int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
  if (r < rows) out[(size_t)r * 16 + l] = in[(size_t)perm[r] * 16 + l];
}

// the same rows read only (summed into a register) / written only (to permuted positions): the two access patterns of the fused
// R_core kernels on S-uniform frames -- voxel rows gathered by id, output rows stored by id (round 5)
__global__ void __launch_bounds__(256) k_rowgather_ro(const float4 *in, const int *perm, float *out, int rows, int reps) {
  int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
  float acc = 0;
  if (r < rows)
    for (int k = 0; k < reps; k++) { const float4 v = in[((size_t)perm[r] + (size_t)k * rows) * 16 + l]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_rowscatter(float4 *out, const int *perm, int rows, int reps) {
  int r = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, l = threadIdx.x & 15;
  if (r < rows)
    for (int k = 0; k < reps; k++) out[((size_t)perm[r] + (size_t)k * rows) * 16 + l] = make_float4(1, 2, 3, (float)k);
}

template <typename F> float timeit(F f, int reps = 20) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / reps;
}
int main() {
  float *buf; size_t bytes = 512ull << 20; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  float *buf2; CK(hipMalloc(&buf2, bytes));
  printf("empty kernel back-to-back (256 WGs): %.2f us each\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0); }, 200));
  printf("empty kernel back-to-back (2048 WGs): %.2f us each\n", timeit([&] { hipLaunchKernelGGL(k_empty, dim3(2048), dim3(256), 0, 0); }, 200));
  for (int wgs : {256, 512, 1024}) {
    int iters = 4000;
    float us = timeit([&] { hipLaunchKernelGGL(k_mfma, dim3(wgs), dim3(256), 0, 0, buf, iters); }, 5);
    double flops = (double)wgs * 4 * iters * 4 * 2048.0;
    printf("mfma f32 16x16x4: %d WGs: %.1f us -> %.1f TFLOP/s ; cycles/MFMA/SIMD at 2.4GHz = %.1f\n", wgs, us, flops / us / 1e6,
           us * 2400.0 / ((double)wgs * 4 / 1024.0 * iters * 4));
  }
  {
    int iters = 20000, wgs = 2048;
    float us = timeit([&] { hipLaunchKernelGGL(k_fma, dim3(wgs), dim3(256), 0, 0, buf, iters); }, 5);
    printf("valu fma: %.1f us -> %.1f TFLOP/s\n", us, (double)wgs * 256 * iters * 8 * 2.0 / us / 1e6);
  }
  for (size_t mb : {6, 25, 51, 102, 400}) {
    size_t n4 = mb * 1000000ull / 16;
    for (int wgs : {512, 2048}) {
      float r = timeit([&] { hipLaunchKernelGGL(k_read, dim3(wgs), dim3(256), 0, 0, (const float4 *)buf, buf2, n4); });
      float w = timeit([&] { hipLaunchKernelGGL(k_write, dim3(wgs), dim3(256), 0, 0, (float4 *)buf, n4); });
      float c = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(wgs), dim3(256), 0, 0, (const float4 *)buf, (float4 *)buf2, n4); });
      printf("%4zu MB, %4d WGs: read %.2f us (%.2f TB/s)  write %.2f us (%.2f TB/s)  copy %.2f us (%.2f TB/s moved)\n", mb, wgs, r,
             mb / r, w, mb / w, c, 2.0 * mb / c);
    }
  }
  {
    int rows = 100000; std::vector<int> h(rows); for (int i = 0; i < rows; i++) h[i] = (int)(((long long)i * 7919) % rows);
    int *perm; CK(hipMalloc(&perm, rows * 4)); CK(hipMemcpy(perm, h.data(), rows * 4, hipMemcpyHostToDevice));
    float us = timeit([&] { hipLaunchKernelGGL(k_rowgather, dim3((rows * 16 + 255) / 256), dim3(256), 0, 0, (const float4 *)buf, perm, (float4 *)buf2, rows); });
    printf("row gather 100k x 256 B (25.6 MB in + out): %.2f us\n", us);
    // a pseudo-random permutation (the multiplicative one above keeps neighbours 7919 rows apart: DRAM pages still line up)
    unsigned long long x = 88172645463325252ull;
    for (int i = rows - 1; i > 0; i--) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; int j = (int)(x % (unsigned)(i + 1)); int t = h[i]; h[i] = h[j]; h[j] = t; }
    CK(hipMemcpy(perm, h.data(), rows * 4, hipMemcpyHostToDevice));
    for (int reps : {1, 8}) {
      const double mb = 25.6 * reps;
      float g = timeit([&] { hipLaunchKernelGGL(k_rowgather_ro, dim3((rows * 16 + 255) / 256), dim3(256), 0, 0, (const float4 *)buf, perm, buf2, rows, reps); });
      float w = timeit([&] { hipLaunchKernelGGL(k_rowscatter, dim3((rows * 16 + 255) / 256), dim3(256), 0, 0, (float4 *)buf, perm, rows, reps); });
      printf("random 256-B rows, %d x 100k (%.1f MB): read-only gather %.2f us (%.2f TB/s)  scatter store %.2f us (%.2f TB/s)\n", reps, mb, g, mb / g, w, mb / w);
    }
  }
  return 0;
}
