// tools/coresidency_probe.hip -- do two kernels launched on two HIP streams really run side by side on a CU?  (round 6: the persistent
// batch kernels of csrc/dense_batch.hip need one K1-role and one K2-role workgroup resident on every CU at the same time.)
//   hipcc --offload-arch=gfx950 -O2 tools/coresidency_probe.hip -o tools/bin/coresidency_probe && tools/bin/coresidency_probe
// Kernel A: 256 workgroups x 256 threads, `ldsA` bytes of LDS, >= 208 vector registers, spins `spin_us`.  Kernel B: 256 workgroups x
// 512 threads, `ldsB` bytes, >= 128 registers, spins 20 us.  Every workgroup stamps the 100 MHz clock at entry.  Printed per case:
// when B's first / median / last workgroup became resident relative to A's first entry -- "side by side" means within a few us, "behind"
// means after A's spin.  Cases: how the two streams were created (flags / priorities), the LDS sizes, a third small kernel in between.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256, 2) kA(unsigned long long *stamps, int spin_ticks) {
  extern __shared__ char lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("v_mov_b32 v207, 0" ::: "v207");        // 208 registers
  if (threadIdx.x == 0) { lds[0] = 1; stamps[blockIdx.x] = t0; }
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void __launch_bounds__(512, 4) kB(unsigned long long *stamps, int spin_ticks) {
  extern __shared__ char lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("v_mov_b32 v127, 0" ::: "v127");        // 128 registers
  if (threadIdx.x == 0) { lds[0] = 1; stamps[blockIdx.x] = t0; }
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(16);
}
__global__ void __launch_bounds__(256) kC(unsigned long long *stamps, int spin_ticks) {   // the insert's shape: no LDS, 40 registers
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("v_mov_b32 v39, 0" ::: "v39");
  if (threadIdx.x == 0) stamps[blockIdx.x] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(16);
}

static void report(const char *name, const std::vector<unsigned long long> &a, const std::vector<unsigned long long> &b) {
  const unsigned long long a0 = *std::min_element(a.begin(), a.end());
  std::vector<double> e;
  for (auto x : b) e.push_back(((double)x - (double)a0) / 100.0);
  std::sort(e.begin(), e.end());
  std::vector<double> ea;
  for (auto x : a) ea.push_back(((double)x - (double)a0) / 100.0);
  std::sort(ea.begin(), ea.end());
  printf("%-58s A resident 0 / %.1f / %.1f us | B resident %.1f / %.1f / %.1f us (first / median / last)\n", name, ea[ea.size() / 2], ea.back(), e[0],
         e[e.size() / 2], e.back());
}

int main(int argc, char **argv) {
  const int spinA = 30000;                               // 300 us
  const int ldsA = argc > 1 ? atoi(argv[1]) : 80896, ldsB = argc > 2 ? atoi(argv[2]) : 82384;
  unsigned long long *sa, *sb, *sc;
  CK(hipMalloc(&sa, 8 * 1024)); CK(hipMalloc(&sb, 8 * 1024)); CK(hipMalloc(&sc, 8 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kA), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&kB), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("stream priority range: least %d greatest %d; ldsA %d ldsB %d\n", lo, hi, ldsA, ldsB);
  struct Case { const char *name; int mode; bool third; };
  const Case cases[] = {{"two default-flag streams", 0, false}, {"two non-blocking streams", 1, false}, {"two high-priority streams", 2, false},
                        {"high + low priority", 3, false}, {"two non-blocking streams + insert-shaped kernel on a third", 1, true},
                        {"two high-priority + insert-shaped on a low-priority third", 2, true}};
  for (const Case &c : cases) {
    hipStream_t s1, s2, s3;
    if (c.mode == 0) { CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2)); CK(hipStreamCreate(&s3)); }
    else if (c.mode == 1) { CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking)); }
    else if (c.mode == 2) { CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, lo)); }
    else { CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, hi)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, lo)); }
    for (int rep = 0; rep < 2; rep++) {
      CK(hipMemset(sa, 0, 8 * 1024)); CK(hipMemset(sb, 0, 8 * 1024)); CK(hipMemset(sc, 0, 8 * 1024));
      CK(hipDeviceSynchronize());
      if (c.third) hipLaunchKernelGGL(kC, dim3(256), dim3(256), 0, s3, sc, 10000);
      hipLaunchKernelGGL(kA, dim3(256), dim3(256), ldsA, s1, sa, spinA);
      hipLaunchKernelGGL(kB, dim3(256), dim3(512), ldsB, s2, sb, 2000);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> a(256), b(256);
      CK(hipMemcpy(a.data(), sa, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), sb, 256 * 8, hipMemcpyDeviceToHost));
      if (rep == 1) report(c.name, a, b);
    }
    CK(hipStreamDestroy(s1)); CK(hipStreamDestroy(s2)); CK(hipStreamDestroy(s3));
  }
  // LDS sweep on two non-blocking streams: which (A, B) sizes share a CU
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int pairs[][2] = {{80896, 80304}, {80896, 81920}, {81152, 81872}, {81152, 81920}, {81152, 82048}, {81408, 82048}, {80896, 82176}, {80896, 82384}, {82944, 80304}, {82944, 80896}, {83200, 80384}, {81920, 81920}, {40000, 40000}};
  for (auto &pr : pairs) {
    CK(hipMemset(sa, 0, 8 * 1024)); CK(hipMemset(sb, 0, 8 * 1024));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kA, dim3(256), dim3(256), pr[0], s1, sa, spinA);
    hipLaunchKernelGGL(kB, dim3(256), dim3(512), pr[1], s2, sb, 2000);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> a(256), b(256);
    CK(hipMemcpy(a.data(), sa, 256 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), sb, 256 * 8, hipMemcpyDeviceToHost));
    char nm[96];
    snprintf(nm, sizeof nm, "LDS A %d + B %d (dynamic bytes; A has no static, B none)", pr[0], pr[1]);
    report(nm, a, b);
  }
  return 0;
}
