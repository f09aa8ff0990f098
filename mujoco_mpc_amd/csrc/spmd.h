// spmd.h — execution-model shim for the rollout engine.
//
// Product build (hipcc, gfx950): one 64-lane wavefront owns one candidate rollout; per-candidate
// mjData-like state lives in LDS; `PFOR` is a lane-strided loop, `SYNC` a workgroup barrier (the
// workgroup IS one wave, so it only orders LDS traffic), reductions/scans use cross-lane shuffles.
//
// MJPC_EMU build (g++, tests only): NLANE = 1, the same source runs as plain sequential C++ so
// that the CPU test tier can exercise the kernel logic (indexing, formulas) before a GPU run.
// The emu library is never loaded by the product (mujoco_mpc_amd/capi.py loads libmjpc_hip.so only).
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef MJPC_EMU
#define DEV static inline
#define DEV_NOINLINE static
#define LANE 0
#define NLANE 1
#define SYNC() ((void)0)
DEV double mul_rn(double a, double b) { return a * b; }   // emu is built with -ffp-contract=off
DEV double add_rn(double a, double b) { return a + b; }
DEV double add_mul3_rn(double a, double b, double c, double d) { return a + (b * c) * d; }
DEV double wave_sum(double v) { return v; }
DEV double wave_min(double v) { return v; }
DEV int wave_sum_i(int v) { return v; }
DEV int wave_or_i(int v) { return v; }
DEV int wave_excl_scan(int v, int *total) { *total = v; return 0; }
#else
#include <hip/hip_runtime.h>
#define DEV static __device__ __forceinline__
#define DEV_NOINLINE static __device__ __noinline__
#define LANE ((int)threadIdx.x)
#define NLANE 64
#define SYNC() __syncthreads()
// individually rounded ops (no FMA contraction): used where results must be bit-identical to the CPU path
DEV double mul_rn(double a, double b) {
#pragma clang fp contract(off)
  return a * b;
}
DEV double add_rn(double a, double b) {
#pragma clang fp contract(off)
  return a + b;
}
// a + (b*c)*d with every operation rounded separately
DEV double add_mul3_rn(double a, double b, double c, double d) {
#pragma clang fp contract(off)
  double t = b * c;
  double u = t * d;
  return a + u;
}
DEV double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEV double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { double w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
  return v;
}
DEV int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
DEV int wave_or_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
  return v;
}
// exclusive prefix sum over the 64 lanes (lane order), total returned to every lane
DEV int wave_excl_scan(int v, int *total) {
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if (LANE >= o) x += y; }
  *total = __shfl(x, 63, 64);
  return x - v;
}
#endif

#define PFOR(i, n) for (int i = LANE; i < (n); i += NLANE)
