// LDS latency calibration on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#define R 2048
template <int MODE>
__global__ void __launch_bounds__(256) k(double *out, int stride) {
  __shared__ int idx[4096];
  __shared__ double val[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) { idx[i] = (i * stride + 1) & 4095; val[i] = i * 0.5; }
  __syncthreads();
  if (threadIdx.x >= 64 && MODE < 10) return;
  int p = threadIdx.x; double acc = 0;
  long long t0 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x < 64) {
  for (int r = 0; r < R; r++) {
    if (MODE == 0 || MODE == 10) { p = idx[p]; }                               // dependent b32 chain
    if (MODE == 1) { p = idx[p]; acc += val[p]; }                // 2-level chain: index -> value (value not on the chain)
    if (MODE == 2) { acc += val[(p + r) & 4095]; }               // independent b64 loads, accumulate (dependent add)
    if (MODE == 3) { p = (int)val[p & 4095] & 4095; }            // dependent b64 chain with cvt
  }
  } else {
    // MODE 10: three other waves hammer LDS with polling reads
    volatile int *f = idx;
    for (int r = 0; r < R * 8; r++) { if (f[4095] == -1) break; }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = (double)(t1 - t0) / R;
  if (p == -5 || acc == -1.0) out[1] = p + acc;
}
template <int MODE> void run(const char *name, double *d, int stride) {
  double h[2];
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, stride);
  hipDeviceSynchronize(); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf("%-60s %.1f ticks per iteration\n", name, h[0]);
}
int main() {
  double *d; hipMalloc(&d, 64);
  run<0>("dependent ds_read_b32 chain (1 wave/CU)", d, 33);
  run<10>("dependent ds_read_b32 chain + 3 polling waves", d, 33);
  run<1>("idx chain + value load (2 loads / iter)", d, 33);
  run<2>("independent b64 load + dependent add", d, 33);
  run<3>("dependent ds_read_b64 + cvt chain", d, 33);
  return 0;
}
