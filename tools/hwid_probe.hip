// Diagnostic: where the hardware puts the 4 waves of two co-resident 80 KiB workgroups (SIMD / CU / XCC of every wave).
//   hipcc --offload-arch=gfx950 -O2 tools/hwid_probe.hip -o gpurun_out/hwid_probe && gpurun_out/hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
extern "C" __global__ void __launch_bounds__(256, 2) probe(unsigned *out, int spin) {
  extern __shared__ double lds[];
  unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508);
  long long t0 = __builtin_amdgcn_s_memtime();
  lds[threadIdx.x] = 1.0;
  while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if ((threadIdx.x & 63) == 0) {
    int w = threadIdx.x >> 6;
    out[(blockIdx.x * 4 + w) * 2] = hw; out[(blockIdx.x * 4 + w) * 2 + 1] = xcc;
  }
}
int main() {
  int N = 512;
  unsigned *d; hipMalloc(&d, N * 8 * sizeof(unsigned));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
  hipLaunchKernelGGL(probe, dim3(N), dim3(256), 80 * 1024, 0, d, 2000000);
  std::vector<unsigned> h(N * 8);
  hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> cu;      // (xcc, se, sh, cu) -> workgroups
  int hist[4][4] = {};
  for (int b = 0; b < N; b++) {
    unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 0xf;
    unsigned key = (xcc << 16) | (hw & 0xff00);
    cu[key].push_back(b);
    for (int w = 0; w < 4; w++) hist[w][(h[(b * 4 + w) * 2] >> 4) & 3]++;
  }
  printf("CUs used: %zu\nwave -> SIMD histogram (rows: wave 0..3, cols: SIMD 0..3)\n", cu.size());
  for (int w = 0; w < 4; w++) printf("  %d %d %d %d\n", hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  int same = 0, pairs = 0, shown = 0;
  for (auto &kv : cu) {
    if (kv.second.size() == 2) {
      pairs++;
      int a = kv.second[0], b = kv.second[1];
      int sa = (h[a * 8] >> 4) & 3, sb = (h[b * 8] >> 4) & 3;
      if (sa == sb) same++;
      if (shown++ < 12) {
        printf("  cu %05x: wg %d simds", kv.first, a); for (int w = 0; w < 4; w++) printf(" %u", (h[(a * 4 + w) * 2] >> 4) & 3);
        printf(" | wg %d simds", b); for (int w = 0; w < 4; w++) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
        printf("\n");
      }
    } else if (shown++ < 12) printf("  cu %05x holds %zu workgroups\n", kv.first, kv.second.size());
  }
  printf("CUs with two workgroups: %d; wave 0 of both on the same SIMD: %d\n", pairs, same);
  return 0;
}
