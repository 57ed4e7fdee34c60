// tools/queue_probe.hip -- which HIP streams get hardware queues of their own?  (round 6: the three persistent kernels of a batch call
// must run side by side; a kernel queued behind another in one hardware queue starts when that one ends.)
//   hipcc --offload-arch=gfx950 -O2 tools/queue_probe.hip -o tools/bin/queue_probe && tools/bin/queue_probe
// `pre` busy application streams are created and used first (as a host program's own streams would be); then three streams are
// created the way the case says and a small spin kernel (64 workgroups, no LDS, 200 us) is launched on each, in order.  Side by side
// = all three start within a few us; "behind" = a start ~200 / ~400 us late.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void spin(unsigned long long *stamp, int ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) *stamp = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(16);
}

int main() {
  unsigned long long *st;
  CK(hipMalloc(&st, 64 * 8));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  const char *names[] = {"3 x default priority", "3 x high priority", "high, high, low", "high, normal, low", "3 x low priority"};
  for (int pre : {0, 2, 4, 6}) {
    std::vector<hipStream_t> app(pre);
    for (auto &s : app) { CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, st + 32, 100); }
    CK(hipDeviceSynchronize());
    for (int mode = 0; mode < 5; mode++) {
      hipStream_t s[3];
      const int pr[5][3] = {{0, 0, 0}, {hi, hi, hi}, {hi, hi, lo}, {hi, 0, lo}, {lo, lo, lo}};
      for (int i = 0; i < 3; i++) CK(hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, pr[mode][i]));
      for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(st, 0, 64 * 8));
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 3; i++) hipLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s[i], st + i, 20000);
        CK(hipDeviceSynchronize());
        unsigned long long h[3];
        CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
        if (rep == 1)
          printf("app streams %d | %-22s | starts: 0, %+.1f, %+.1f us\n", pre, names[mode], ((double)h[1] - (double)h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0);
      }
      for (int i = 0; i < 3; i++) CK(hipStreamDestroy(s[i]));
    }
    for (auto &s : app) CK(hipStreamDestroy(s));
  }
  // ONE stream, hipExtLaunchKernelGGL with hipExtAnyOrderLaunch (the AQL barrier bit cleared): do consecutive kernels of a stream overlap?
  {
    hipStream_t s1;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(st, 0, 64 * 8));
      CK(hipDeviceSynchronize());
      for (int i = 0; i < 3; i++) hipExtLaunchKernelGGL(spin, dim3(64), dim3(64), 0, s1, nullptr, nullptr, 1 /* hipExtAnyOrderLaunch */, st + i, 20000);
      CK(hipDeviceSynchronize());
      unsigned long long h[3];
      CK(hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost));
      if (rep == 1) printf("one stream, any-order launches | starts: 0, %+.1f, %+.1f us\n", ((double)h[1] - (double)h[0]) / 100.0, ((double)h[2] - (double)h[0]) / 100.0);
    }
  }
  return 0;
}
