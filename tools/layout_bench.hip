// tools/layout_bench.hip -- where the layout step of the pair plan (column scans + finish kernel) spends its time (s_memtime at its phase
// boundaries; -DPP_DBG build of conv_pairs.hip included as a whole).   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPP_DBG
//   -Iinclude -Ilink_amd/csrc tools/layout_bench.hip -o tools/bin/layout_bench
#include "../link_amd/csrc/conv_pairs.hip"
#include <cstdio>
namespace link { void set_error(const char *what, hipError_t e) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); } }
#include <vector>
int main() {
  const int64_t n = 100000; const int kvol = 27;
  const int64_t nwg = (n + 255) / 256;
  int64_t offs[11];
  // the arena layout of csrc/block.hip (link_pair_plan_arena), restated here to keep this tool self-contained
  const int64_t cap_pairs = n * (kvol - 1), gran_cap = (cap_pairs + 127 * kvol + 127) / 128;
  const int64_t sizes[10] = {nwg * (kvol + 1), n, kvol + nwg * kvol + kvol + 1, nwg, gran_cap, 8, n + 1, gran_cap * 128, gran_cap * 128, cap_pairs};
  offs[0] = 0; for (int i = 0; i < 10; i++) offs[i + 1] = offs[i] + ((sizes[i] + 3) & ~3LL);
  int32_t *arena; hipMalloc(&arena, offs[10] * 4);
  std::vector<int32_t> nbr_h(n * kvol, -1);
  for (int64_t i = 0; i < n; i++) { nbr_h[i * kvol + 13] = (int)i; if (i % 7 == 0 && i + 1 < n) nbr_h[i * kvol + 14] = (int)i + 1; if (i % 7 == 1) nbr_h[i * kvol + 12] = (int)i - 1; }
  int32_t *nbr; hipMalloc(&nbr, n * kvol * 4); hipMemcpy(nbr, nbr_h.data(), n * kvol * 4, hipMemcpyHostToDevice);
  int32_t *meta = arena + offs[2];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 5; it++) {
    link_pair_plan_count(nbr, n, kvol, arena + offs[0], arena + offs[1], nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    pair_plan_layout_run(arena + offs[0], nwg, kvol, 1, gran_cap, meta, meta + kvol, meta + kvol + nwg * kvol, arena + offs[4], arena + offs[5],
                         arena + offs[3], arena + offs[7], arena + offs[8], arena + offs[6] + n, nullptr, "layout");
    hipEventRecord(e1, nullptr);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t[8]; hipMemcpyFromSymbol(t, HIP_SYMBOL(g_pp_dbg), sizeof(t));
    int32_t hdr[8]; hipMemcpy(hdr, arena + offs[5], 32, hipMemcpyDeviceToHost);
    printf("colscan + finish %.1f us (events); finish ticks: totals + wg_ext %llu, serial %llu, tails %llu, wg_k %llu  | pairs %d rows %d gran %d\n", ms * 1e3, t[1] - t[0], t[2] - t[1],
           t[3] - t[2], t[4] - t[3], hdr[0], hdr[1], hdr[2]);
  }
  return 0;
}
