// instruction timing calibration on gfx950: one wave per CU, s_memtime ticks + wall clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define R 4096
__device__ __forceinline__ double rl(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
template <int MODE>
__global__ void __launch_bounds__(64) k(double *out, double a, double b) {
  double x0 = threadIdx.x * 1e-3 + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  long long t0 = __builtin_amdgcn_s_memtime();
  long long w0 = wall_clock64();
  for (int r = 0; r < R / 8; r++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (MODE == 0) { x0 = __builtin_fma(x0, b, a); }                                   // dependent DFMA chain
      if (MODE == 1) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); }     // 2 chains
      if (MODE == 2) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a); }
      if (MODE == 3) { x0 = __builtin_fmaf((float)x0, (float)b, (float)a); }              // dependent FP32 chain (with converts folded? no)
      if (MODE == 4) { x0 = __builtin_fma(x0, rl(x0, u), a); }                            // readlane -> fma dependent
      if (MODE == 5) { x0 += rl(x1, u) * b; }                                              // readlane of an independent value feeding a chain
      if (MODE == 6) { x0 = x0 * __builtin_amdgcn_rcp(x0 + b); }                           // rcp chain
      if (MODE == 7) { asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x0) : "v"(b), "v"(a)); }
      if (MODE == 8) { asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(b), "v"(a)); }
      if (MODE == 9) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(x0) : "v"(a)); }
      if (MODE == 10) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x0) : "v"(b)); }
      if (MODE == 11) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(((float*)&x0)[0]) : "v"((float)a)); }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long w1 = wall_clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(w1 - w0); }
  if (x0 + x1 + x2 + x3 == 12345.678) out[2] = x0;
}
template <int MODE> void run(const char *name, int per_iter, double *d) {
  double h[3];
  for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64), 0, 0, d, 0.5, 0.999);
  hipDeviceSynchronize();
  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("%-34s memtime ticks/instr %.2f   wall_clock(100MHz) ticks total %.0f => %.2f ns/instr\n", name, h[0] / (R * per_iter), h[1], h[1] * 10.0 / (R * per_iter));
}
int main() {
  double *d; hipMalloc(&d, 64);
  run<0>("dependent DFMA", 1, d); run<1>("2 indep DFMA chains", 2, d); run<2>("4 indep DFMA chains", 4, d);
  run<4>("readlane(x)->fma dependent (3 instr)", 3, d); run<5>("readlane indep + fma chain (3 instr)", 3, d); run<6>("rcp chain (3 instr)", 3, d);
  run<7>("asm dependent v_fma_f64", 1, d); run<8>("asm 2 indep v_fma_f64", 2, d); run<9>("asm dependent v_add_f64", 1, d); run<10>("asm dependent v_mul_f64", 1, d); run<11>("asm dependent v_add_f32", 1, d);
  return 0;
}
