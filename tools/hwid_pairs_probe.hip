// Diagnostic: where do the wavefronts of 256-thread workgroups land?  Prints, for a launch with two workgroups per CU (80 KB of LDS
// each), the SIMD of every wave index and the hardware workgroup slots (TG_ID) of the workgroups sharing a CU.
// build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/hwid_probe tools/hwid_pairs_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
#include <algorithm>
__global__ void __launch_bounds__(256, 2) probe(unsigned *out, int spin) {
  extern __shared__ double lds[];
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID, offset 0, size 32
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));      // HW_REG_XCC_ID
  long long t0 = __builtin_amdgcn_s_memtime();
  double v = threadIdx.x;
  for (int i = 0; i < spin; i++) { v = v * 1.0000001 + 0.5; lds[threadIdx.x] = v; }      // keep the workgroup resident for a while
  if ((threadIdx.x & 63) == 0) {
    unsigned *o = out + 4 * (blockIdx.x * 4 + (threadIdx.x >> 6));
    o[0] = hw; o[1] = xcc; o[2] = (unsigned)(t0 >> 8); o[3] = (unsigned)v;
  }
}
int main() {
  const int nwg = 1024;
  unsigned *d; hipMalloc(&d, nwg * 16 * sizeof(unsigned));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  probe<<<nwg, 256, 80 * 1024>>>(d, 20000);
  std::vector<unsigned> h(nwg * 16);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  int simd_hist[4][4] = {};
  std::map<unsigned, std::vector<std::pair<unsigned, int>>> per_cu;      // cu key -> (start time, tg id | simd of wave 0 << 4)
  int distinct = 0;
  for (int g = 0; g < nwg; g++) {
    unsigned seen = 0;
    for (int w = 0; w < 4; w++) {
      unsigned hw = h[4 * (g * 4 + w)], simd = (hw >> 4) & 3;
      simd_hist[w][simd]++; seen |= 1u << simd;
    }
    distinct += seen == 15;
    unsigned hw = h[16 * g], xcc = h[16 * g + 1] & 15;
    unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
    per_cu[key].push_back({h[16 * g + 2], (int)(((hw >> 16) & 15) | (((hw >> 4) & 3) << 4))});
    if (g < 8) printf("wg %d: hw_id %08x xcc %u simd of waves %u %u %u %u tg_id %u cu %u sh %u se %u\n", g, hw, xcc, (h[16 * g] >> 4) & 3, (h[16 * g + 4] >> 4) & 3,
                      (h[16 * g + 8] >> 4) & 3, (h[16 * g + 12] >> 4) & 3, (hw >> 16) & 15, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7);
  }
  printf("workgroups whose four waves sit on four distinct SIMDs: %d of %d\n", distinct, nwg);
  for (int w = 0; w < 4; w++) printf("wave %d on SIMD 0..3: %d %d %d %d\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
  printf("CUs seen: %zu\n", per_cu.size());
  int tg_hist[16] = {}; for (auto &kv : per_cu) for (auto &p : kv.second) tg_hist[p.second & 15]++;
  printf("tg_id histogram:"); for (int i = 0; i < 16; i++) printf(" %d", tg_hist[i]); printf("\n");
  // the two workgroups that start together on a CU: how far apart (in SIMDs) are their waves 0?
  int dist_hist[4] = {};
  for (auto &kv : per_cu) {
    auto v = kv.second; std::sort(v.begin(), v.end());
    for (size_t i = 0; i + 1 < v.size(); i += 2) dist_hist[((v[i].second >> 4) - (v[i + 1].second >> 4)) & 3]++;
  }
  printf("co-resident pairs, (SIMD of wave 0 of A - of B) mod 4: %d %d %d %d\n", dist_hist[0], dist_hist[1], dist_hist[2], dist_hist[3]);
  int shown = 0;
  for (auto &kv : per_cu) { if (shown++ >= 4) break; printf("cu %06x:", kv.first); for (auto &p : kv.second) printf(" (t=%u tg=%d simd0=%d)", p.first, p.second & 15, p.second >> 4); printf("\n"); }
  return 0;
}
