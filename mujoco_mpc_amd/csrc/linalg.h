// linalg.h — small dense SPD factorisation / solves for one wavefront.
//
//  * generic (any n, also the 1-lane emulation build): A = L L^T, left-looking in LDS, one SYNC pair per column,
//    Linv[j] = 1/L[j][j];
//  * NVT > 0 (gfx950 only, n == NVT known at compile time): A = L^T D L with lane i keeping ROW i of the matrix in VGPRs;
//    pivots and rows are broadcast with v_readlane (no LDS traffic, no waits on the pivot chain); the substitutions keep x
//    in a VGPR per lane.  A factor/solve pair always uses the same form (the layouts in LDS differ).
// One wave per matrix is latency-bound, so the register form is several times faster than the LDS form.
#pragma once
#include "dmath.h"
#include "model.h"

struct Ctx;

#ifndef MJPC_EMU
DEV double readlane_d(double v, int lane) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
#endif

// ---- generic LDS versions -----------------------------------------------------------------
DEV void chol_factor_lds(double *A, double *Linv, double *tmp, int n, int nvp) {
  for (int j = 0; j < n; j++) {
    SYNC();
    PFOR(ii, n - j) {
      int i = j + ii;
      double s = A[i * nvp + j];
#pragma unroll 4
      for (int k = 0; k < j; k++) s -= A[i * nvp + k] * A[j * nvp + k];
      tmp[i] = s;
    }
    SYNC();
    double t = tmp[j];
    if (t < D_MINVAL) t = D_MINVAL;
    double dj = sqrt(t), inv = 1.0 / dj;
    PFOR(ii, n - j) {
      int i = j + ii;
      A[i * nvp + j] = (i == j) ? dj : tmp[i] * inv;
    }
    if (LANE == 0) Linv[j] = inv;
  }
  SYNC();
}
DEV void chol_solve_lds(const double *L, const double *Linv, double *x, int n, int nvp) {
  for (int i = 0; i < n; i++) {
    SYNC();
    double xi = x[i] * Linv[i];
    SYNC();
    PFOR(kk, n - i - 1) { int k = i + 1 + kk; x[k] -= L[k * nvp + i] * xi; }
    if (LANE == 0) x[i] = xi;
  }
  for (int i = n - 1; i >= 0; i--) {
    SYNC();
    double xi = x[i] * Linv[i];
    SYNC();
    PFOR(k, i) x[k] -= L[i * nvp + k] * xi;
    if (LANE == 0) x[i] = xi;
  }
  SYNC();
}

// ---- register versions (compile-time n) -----------------------------------------------------
// A = L^T D L (L unit lower triangular, pivots taken from the last dof up, like MuJoCo's mj_factorM order).
// Lane i keeps the full symmetric row i in VGPRs.  A pivot step broadcasts 1/d_k and row k with v_readlane
// (SGPR operands of the trailing FMAs): no LDS traffic and no waits on the pivot chain, which is
//   readlane(d_k) -> rcp + 2 Newton steps -> l = a[k]*r -> d_{k-1} update;
// the rank-1 update of the other columns is independent work the scheduler overlaps with that chain.
// Afterwards lane i holds  up[k] = L[k][i] (k > i, else 0)  and  lo[j] = L[i][j] (j < i, else 0),
// so both substitutions are "readlane + one FMA" per step.
#ifndef MJPC_EMU
template <int N>
struct LDLRegs { double lo[N], up[N], rinv; };

template <int N>
DEV void ldl_factor_regs(const double *A, int nvp, LDLRegs<N> &f) {
  static_assert(N >= 1 && N <= 64, "one matrix row per lane");
  const int i = LANE;
  const bool act = i < N;
  double a[N];
#pragma unroll
  for (int j = 0; j < N; j++) a[j] = act ? A[(j <= i) ? i * nvp + j : j * nvp + i] : 0.0;   // lower triangle is valid in LDS
  double dg = act ? A[i * nvp + i] : 1.0;
#pragma unroll
  for (int k = N - 1; k >= 1; k--) {
    double dk = readlane_d(dg, k);
    if (dk < D_MINVAL) dk = D_MINVAL;
    double rk = fast_rcp(dk);
    double hk = a[k];
    double l = (i < k) ? hk * rk : 0.0;
    dg -= l * hk;
    // A[i][j] -= L[k][i] * A[k][j]: broadcast row k in groups of 8 first, then the FMAs, so that the SGPR written by a
    // v_readlane is not consumed by the very next VALU instruction (that hazard costs an s_nop per update otherwise)
#pragma unroll
    for (int j0 = 0; j0 < k; j0 += 8) {
      double sj[8];
#pragma unroll
      for (int q = 0; q < 8; q++) if (j0 + q < k) sj[q] = readlane_d(a[j0 + q], k);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 8; q++) if (j0 + q < k) a[j0 + q] -= l * sj[q];
      __builtin_amdgcn_sched_barrier(0);
    }
    f.up[k] = l;
  }
  f.up[0] = 0.0;
  if (dg < D_MINVAL) dg = D_MINVAL;
  f.rinv = fast_rcp(dg);
#pragma unroll
  for (int j = 0; j < N; j++) f.lo[j] = (act && j < i) ? a[j] * f.rinv : 0.0;
}
template <int N>
DEV double ldl_solve_regs(const LDLRegs<N> &f, double xi) {
#pragma unroll
  for (int k = N - 1; k >= 1; k--) xi -= f.up[k] * readlane_d(xi, k);      // L^T u = b
  xi *= f.rinv;                                                            // D v = u
#pragma unroll
  for (int j = 0; j < N - 1; j++) xi -= f.lo[j] * readlane_d(xi, j);       // L x = v
  return xi;
}
// split form (factor kept in LDS between phases): L[i][j] (j < i) in the lower triangle of A, 1/d in Dinv
template <int N>
DEV void chol_factor_reg(double *A, double *Dinv, int nvp) {
  SYNC();
  LDLRegs<N> f;
  ldl_factor_regs<N>(A, nvp, f);
  const int i = LANE;
#pragma unroll
  for (int j = 0; j < N; j++) if (i < N && j < i) A[i * nvp + j] = f.lo[j];
  if (i < N) Dinv[i] = f.rinv;
  SYNC();
}
template <int N>
DEV void chol_solve_reg(const double *L, const double *Dinv, double *x, int nvp) {
  SYNC();
  const int i = LANE;
  const bool act = i < N;
  LDLRegs<N> f;
  f.rinv = act ? Dinv[i] : 0.0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    f.lo[k] = (act && k < i) ? L[i * nvp + k] : 0.0;
    f.up[k] = (act && k > i) ? L[k * nvp + i] : 0.0;
  }
  double xi = ldl_solve_regs<N>(f, act ? x[i] : 0.0);
  if (act) x[i] = xi;
  SYNC();
}
// fused factor + solve (the factor never leaves the registers): Newton direction, implicit-damping solve
template <int N>
DEV void chol_factor_solve_reg(const double *A, double *x, int nvp) {
  SYNC();
  LDLRegs<N> f;
  ldl_factor_regs<N>(A, nvp, f);
  const int i = LANE;
  double xi = ldl_solve_regs<N>(f, i < N ? x[i] : 0.0);
  if (i < N) x[i] = xi;
  SYNC();
}
#endif

template <int NVT>
DEV void chol_factor(double *A, double *Linv, double *tmp, int n, int nvp) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_factor_reg<NVT>(A, Linv, NVP_OF(NVT)); return; }
#endif
  chol_factor_lds(A, Linv, tmp, n, nvp);
}
template <int NVT>
DEV void chol_solve(const double *L, const double *Linv, double *x, int n, int nvp) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_solve_reg<NVT>(L, Linv, x, NVP_OF(NVT)); return; }
#endif
  chol_solve_lds(L, Linv, x, n, nvp);
}
// A (lower triangle) is consumed; x <- A^-1 x.  The generic build leaves the LL^T factor in A / Linv.
template <int NVT>
DEV void chol_factor_solve(double *A, double *Linv, double *tmp, double *x, int n, int nvp) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_factor_solve_reg<NVT>(A, x, NVP_OF(NVT)); return; }
#endif
  chol_factor_lds(A, Linv, tmp, n, nvp);
  chol_solve_lds(A, Linv, x, n, nvp);
}
