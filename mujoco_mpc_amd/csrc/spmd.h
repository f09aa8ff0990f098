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
// the emulation runs both wave roles of a candidate one after the other in a single thread
#define XBAR() ((void)0)
#define ROLE0 1
#define ROLE1 1
#define ROLEH 0
DEV void flag_set(int *p, int v) { *p = v; }
DEV int flag_wait(int *p, int v) { return *p == v; }
DEV int flag_wait_ge(int *p, int v) { return *p >= v; }
DEV double mul_rn(double a, double b) { return a * b; }   // emu is built with -ffp-contract=off
DEV double add_rn(double a, double b) { return a + b; }
DEV double add_mul3_rn(double a, double b, double c, double d) { return a + (b * c) * d; }
DEV double wave_sum(double v) { return v; }
DEV void wave_sum3(double &a, double &b, double &c) {}
DEV void wave_sum4(double &a, double &b, double &c, double &d) {}
DEV double wave_min(double v) { return v; }
DEV int wave_sum_i(int v) { return v; }
DEV int wave_or_i(int v) { return v; }
DEV int wave_excl_scan(int v, int *total) { *total = v; return 0; }
DEV int wave_flag_scan(int flag, int *total) { *total = flag ? 1 : 0; return 0; }
DEV int wave_any(int flag) { return flag != 0; }
#else
#include <hip/hip_runtime.h>
#define DEV static __device__ __forceinline__
#define DEV_NOINLINE static __device__ __noinline__
// A candidate is owned by MJPC_WAVES (1 or 2) wavefronts of one workgroup on different SIMDs of a CU:
//   role 0 (wave 0): the serial critical path (kinematics -> collision/constraints -> Newton solver -> integration);
//   role 1 (last wave): work that only hangs off that path (inertia + factor M + smooth dynamics while role 0 builds
//   the constraints; residual / cost / trajectory record while role 0 solves);
//   helpers (waves 1 .. MJPC_WAVES-2): share the data-parallel parts of every Newton iteration with role 0 (solver.h).
// SYNC() orders LDS traffic inside ONE wave (DS operations of a wave execute in order; only the compiler must not
// reorder them), XBAR() is the workgroup barrier between the roles.
#ifndef MJPC_WAVES
#define MJPC_WAVES 4
#endif
#define LANE ((int)(threadIdx.x & 63))
#define NLANE 64
#define SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#if MJPC_WAVES > 1
#define XBAR() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); } while (0)
#define WAVE_ID() (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))
#define ROLE0 (WAVE_ID() == 0)
#define ROLE1 (WAVE_ID() == MJPC_WAVES - 1)
#define ROLEH (MJPC_WAVES >= 3 && WAVE_ID() >= 1 && WAVE_ID() < MJPC_WAVES - 1)     // helper k = WAVE_ID() - 1
#else
#define XBAR() SYNC()
#define WAVE_ID() 0
#define ROLE0 1
#define ROLE1 1
#define ROLEH 0
#endif
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
// butterfly inside each row of 16 lanes with DPP moves (VALU rate), then the 4 row sums through readlane
DEV double dpp_xchg(double v, const int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(0, lo, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0xB1, 0xF, 0xF, true); break;   // quad_perm [1,0,3,2]
    case 1: lo = __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xF, 0xF, true); break;   // quad_perm [2,3,0,1]
    case 2: lo = __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xF, 0xF, true); break; // row_half_mirror
    default: lo = __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xF, 0xF, true); break; // row_mirror
  }
  return __hiloint2double(hi, lo);
}
// the four row sums (identical in every lane of a row after the butterflies) -> wave total: row_bcast:15 adds row 0 into
// row 1 and row 2 into row 3, row_bcast:31 adds row 1 into rows 2 / 3; lane 63 then holds (r3 + r2) + (r1 + r0), the same
// value as (r0 + r1) + (r2 + r3)
DEV double row_total(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  double t = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false), __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false));
  v += t;
  lo = __double2loint(v); hi = __double2hiint(v);
  t = __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false), __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false));
  v += t;
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
DEV double wave_sum(double v) {
  v += dpp_xchg(v, 0);
  v += dpp_xchg(v, 1);
  v += dpp_xchg(v, 2);
  v += dpp_xchg(v, 3);
  // every lane of a row now holds that row's sum (the butterflies are symmetric => identical bits in all lanes)
  return row_total(v);
}
// several reductions in lock-step: the butterfly steps of independent sums interleave, which hides the DPP / add latency of
// each chain behind the others (one after the other they are a serial chain of ~30 dependent instructions each)
DEV void wave_sum3(double &a, double &b, double &c) {
#pragma unroll
  for (int s = 0; s < 4; s++) {
    double ta = dpp_xchg(a, s), tb = dpp_xchg(b, s), tc = dpp_xchg(c, s);
    a += ta; b += tb; c += tc;
  }
  a = row_total(a); b = row_total(b); c = row_total(c);
}
DEV void wave_sum4(double &a, double &b, double &c, double &d) {
#pragma unroll
  for (int s = 0; s < 4; s++) {
    double ta = dpp_xchg(a, s), tb = dpp_xchg(b, s), tc = dpp_xchg(c, s), td = dpp_xchg(d, s);
    a += ta; b += tb; c += tc; d += td;
  }
  a = row_total(a); b = row_total(b); c = row_total(c); d = row_total(d);
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
DEV int wave_any(int flag) { return __builtin_amdgcn_ballot_w64(flag != 0) != 0; }
// point-to-point hand-shake between two waves of the workgroup through a sequence number in LDS: the setter publishes all
// its earlier LDS writes (release), the waiter spins (bounded: a protocol bug must not hang the GPU) and then acquires
DEV void flag_set(int *p, int v) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  if (LANE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
DEV int flag_wait(int *p, int v) {
  int ok = 0;
  for (int n = 0; n < (1 << 21); n++) {
    int cur = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (cur == v) { ok = 1; break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ok;
}
// the same for a monotonically increasing sequence number: returns once *p >= v (a waiter that was late for v still gets through)
DEV int flag_wait_ge(int *p, int v) {
  int ok = 0;
  for (int n = 0; n < (1 << 21); n++) {
    int cur = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    if (cur >= v) { ok = 1; break; }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return ok;
}
// exclusive prefix sum over the 64 lanes (lane order), total returned to every lane: Hillis-Steele inside each
// row of 16 with DPP row_shr (VALU rate, zero fill), then row_bcast:15 / row_bcast:31 carry the row totals
DEV int wave_excl_scan(int v, int *total) {
  int x = v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false);
  *total = __builtin_amdgcn_readlane(x, 63);
  return x - v;
}
// the same for 0/1 flags: ballot + popcount of the lower lanes
DEV int wave_flag_scan(int flag, int *total) {
  unsigned long long m = __builtin_amdgcn_ballot_w64(flag != 0);
  *total = __builtin_popcountll(m);
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
#endif

#if !defined(MJPC_EMU) && MJPC_WAVES >= 3
#define MJPC_HELPER 1
#define MJPC_NH (MJPC_WAVES - 2)
#else
#define MJPC_HELPER 0
#define MJPC_NH 0
#endif

#define PFOR(i, n) for (int i = LANE; i < (n); i += NLANE)
