// tools/ubench2.hip -- access-pattern calibration: [N,64] fp32 rows read/written by a wave as
//  (A) contiguous: lane l -> 16 B at byte 16*l of a 1 KB chunk (4 rows per instruction)
//  (B) MFMA-operand pattern: lane (li=l&15, g=l>>4) -> row li, 16 B at column 16t+4g (16 rows x 64 B per instruction)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k_readA(const float4 *in, float *out, long tiles) {
  float acc = 0; int lane = threadIdx.x & 63; long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6), W = (long)gridDim.x * 4;
  for (long t = w; t < tiles; t += W) { const float4 *p = in + t * 256;
    for (int k = 0; k < 4; k++) { float4 v = p[k * 64 + lane]; acc += v.x + v.y + v.z + v.w; } }
  if (acc == 1.2345f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_readB(const float4 *in, float *out, long tiles) {
  float acc = 0; int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4; long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6), W = (long)gridDim.x * 4;
  for (long t = w; t < tiles; t += W) { const float4 *p = in + t * 256 + li * 16 + g;
    for (int k = 0; k < 4; k++) { float4 v = p[k * 4]; acc += v.x + v.y + v.z + v.w; } }
  if (acc == 1.2345f) out[0] = acc;
}
__global__ void __launch_bounds__(256) k_writeA(float4 *out, long tiles) {
  int lane = threadIdx.x & 63; long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6), W = (long)gridDim.x * 4;
  for (long t = w; t < tiles; t += W) { float4 *p = out + t * 256; for (int k = 0; k < 4; k++) p[k * 64 + lane] = make_float4(1, 2, 3, 4); }
}
__global__ void __launch_bounds__(256) k_writeB(float4 *out, long tiles) {
  int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4; long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6), W = (long)gridDim.x * 4;
  for (long t = w; t < tiles; t += W) { float4 *p = out + t * 256 + li * 16 + g; for (int k = 0; k < 4; k++) p[k * 4] = make_float4(1, 2, 3, 4); }
}
template <typename F> float timeit(F f, int reps = 30) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); f(); f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms * 1e3f / reps; }
int main() {
  float *a, *b; size_t bytes = 64ull << 20; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 0, bytes);
  long tiles = 6250;
  for (int wgs : {256, 512, 1024, 1563}) {
    float rA = timeit([&] { hipLaunchKernelGGL(k_readA, dim3(wgs), dim3(256), 0, 0, (const float4 *)a, b, tiles); });
    float rB = timeit([&] { hipLaunchKernelGGL(k_readB, dim3(wgs), dim3(256), 0, 0, (const float4 *)a, b, tiles); });
    float wA = timeit([&] { hipLaunchKernelGGL(k_writeA, dim3(wgs), dim3(256), 0, 0, (float4 *)b, tiles); });
    float wB = timeit([&] { hipLaunchKernelGGL(k_writeB, dim3(wgs), dim3(256), 0, 0, (float4 *)b, tiles); });
    printf("wgs %4d: read contiguous %.2f us | read MFMA-pattern %.2f us | write contiguous %.2f us | write MFMA-pattern %.2f us (25.6 MB each)\n", wgs, rA, rB, wA, wB);
  }
  return 0;
}
