// linalg.h — small dense SPD factorisation / solves for one wavefront.
//
// Two implementations of A = L L^T (lower triangle, row stride nvp, Linv[j] = 1/L[j][j]):
//  * generic (any n, also the 1-lane emulation build): left-looking in LDS, one SYNC pair per column;
//  * NVT > 0 (gfx950 only, n == NVT known at compile time): lane i keeps ROW i of the matrix in VGPRs,
//    pivots and the L[k][j] factors are broadcast with v_readlane (no LDS traffic, no barriers inside the
//    factorisation); triangular solves keep x in a VGPR per lane and prefetch their L row/column once.
// A single wave per candidate is latency-bound, so the register form is ~8x faster than the LDS form.
#pragma once
#include "dmath.h"

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
#ifndef MJPC_EMU
template <int N>
DEV void chol_factor_reg(double *A, double *Linv, double *colbuf, int nvp) {
  static_assert(N >= 1 && N <= 64, "one matrix row per lane");
  SYNC();
  const int i = LANE;
  const bool act = i < N;
  double a[N];
#pragma unroll
  for (int k = 0; k < N; k++) a[k] = (act && k <= i) ? A[i * nvp + k] : 0.0;
#pragma unroll
  for (int j = 0; j < N; j++) {
    double ajj = readlane_d(a[j], j);
    if (ajj < D_MINVAL) ajj = D_MINVAL;
    double inv = fast_rsqrt(ajj), dj = ajj * inv;       // pivot without IEEE sqrt + divide on the column chain
    double lij = (i == j) ? dj : a[j] * inv;
    a[j] = lij;
    if (j + 1 < N) {
      // column j of L goes through an LDS line; every lane then reads L[k][j] as a broadcast with an
      // immediate offset (2 instructions per trailing update instead of 2 readlanes + FMA)
      if (act) colbuf[i] = lij;
      SYNC();
#pragma unroll
      for (int k = j + 1; k < N; k++) a[k] -= lij * colbuf[k];   // A[i][k] -= L[i][j] * L[k][j]
      SYNC();
    }
    if (i == 0) Linv[j] = inv;
  }
#pragma unroll
  for (int k = 0; k < N; k++) if (act && k <= i) A[i * nvp + k] = a[k];
  SYNC();
}
template <int N>
DEV void chol_solve_reg(const double *L, const double *Linv, double *x, int nvp) {
  SYNC();
  const int i = LANE;
  const bool act = i < N;
  double xi = act ? x[i] : 0.0;
  // unit-diagonal form A = L' D L'^T, L'[i][j] = L[i][j] / L[j][j], D = diag(L[j][j]^2): the substitution
  // chains are readlane + one FMA per step; the scalings are done off the chain
  double row[N], col[N];
  double mydinv = act ? Linv[i] : 0.0;
#pragma unroll
  for (int k = 0; k < N; k++) {
    row[k] = (act && k < i) ? L[i * nvp + k] * Linv[k] : 0.0;     // L'[i][k], forward substitution
    col[k] = (act && k > i) ? L[k * nvp + i] * mydinv : 0.0;      // L'[k][i], backward substitution
  }
#pragma unroll
  for (int j = 0; j < N; j++) xi -= row[j] * readlane_d(xi, j);    // row[j] == 0 for j >= i
  xi *= mydinv * mydinv;
#pragma unroll
  for (int j = N - 1; j >= 0; j--) xi -= col[j] * readlane_d(xi, j);   // col[j] == 0 for j <= i
  if (act) x[i] = xi;
  SYNC();
}
#endif

template <int NVT>
DEV void chol_factor(double *A, double *Linv, double *tmp, int n, int nvp) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_factor_reg<NVT>(A, Linv, tmp, NVT | 1); return; }
#endif
  chol_factor_lds(A, Linv, tmp, n, nvp);
}
template <int NVT>
DEV void chol_solve(const double *L, const double *Linv, double *x, int n, int nvp) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_solve_reg<NVT>(L, Linv, x, NVT | 1); return; }
#endif
  chol_solve_lds(L, Linv, x, n, nvp);
}
