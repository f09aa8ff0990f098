// linalg.h — small dense SPD factorisation / solves for one wavefront.
//
//  * generic (any n, also the 1-lane emulation build): A = L L^T, left-looking in LDS, one SYNC pair per column,
//    Linv[j] = 1/L[j][j];
//  * NVT > 0 (gfx950 only, n == NVT known at compile time): A = L^T D L with lane i keeping ROW i of the matrix in VGPRs;
//    pivots and rows are broadcast with v_readlane (no LDS traffic, no waits on the pivot chain); the substitutions keep x
//    in a VGPR per lane.  A factor/solve pair always uses the same form (the layouts in LDS differ).
// One wave per matrix is latency-bound, so the register form is several times faster than the LDS form.
#pragma once
#include <type_traits>
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
// the matrix to factor may arrive as a sum: A plus up to three partial matrices of the same layout (the worker waves' partial
// Hessians, solver_reg.h); they are added while the rows are loaded
struct LDLExtra { const double *x[3]; int n; const double *row; };      // row != nullptr: lane i already holds the full symmetric row i of A in registers (A itself is not read)

template <int I, int E, class F>
DEV void static_for(F &&f) {
  if constexpr (I < E) { f(std::integral_constant<int, I>{}); static_for<I + 1, E>(f); }
}
// every index below is a compile-time constant (static_for, not `#pragma unroll`: with N = 33 the nested pragma loops exceeded
// the unroller's budget, the row array stayed in scratch memory and the kernel wrote gigabytes of spills per launch)
// row i of the symmetric matrix (only its lower triangle is valid in LDS) and its diagonal entry, partial matrices added
template <int N>
DEV void ldl_load_row(const double *A, int nvp, const LDLExtra *ex, double *a, double &dg) {
  const int i = LANE;
  const bool act = i < N;
  if (ex && ex->row) {
    double d0 = 1.0;
    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = act ? ex->row[j] : 0.0; d0 = (j == i) ? ex->row[j] : d0; });
    dg = act ? d0 : 1.0;
  } else {
    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = act ? A[(j <= i) ? i * nvp + j : j * nvp + i] : 0.0; });
    dg = act ? A[i * nvp + i] : 1.0;
  }
  if (ex && ex->n > 0) {
    const int nx = ex->n;
    double t[3][N], td[3];
#pragma unroll
    for (int w = 0; w < 3; w++) {
      const double *X = ex->x[w < nx ? w : 0];
      static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; t[w][j] = X[(j <= i && act) ? i * nvp + j : (act ? j * nvp + i : 0)]; });
      td[w] = X[act ? i * nvp + i : 0];
      if (w + 1 >= nx) break;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int w = 0; w < 3; w++) {
      if (w >= nx) break;
      static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] += act ? t[w][j] : 0.0; });
      dg += act ? td[w] : 0.0;
    }
  }
}
template <int N>
DEV void ldl_factor_regs(const double *A, int nvp, LDLRegs<N> &f, const LDLExtra *ex = nullptr) {
  static_assert(N >= 1 && N <= 64, "one matrix row per lane");
  const int i = LANE;
  const bool act = i < N;
  double a[N];
  double dg;
  ldl_load_row<N>(A, nvp, ex, a, dg);
  static_for<1, N>([&](auto kc) {
    constexpr int k = N - decltype(kc)::value;                                 // N-1 ... 1
    double rk = readlane_d(fast_rcp(dg < D_MINVAL ? D_MINVAL : dg), k);      // reciprocal in the vector domain, then broadcast
    double hk = a[k];
    double l = (i < k) ? hk * rk : 0.0;
    dg -= l * hk;
    // A[i][j] -= L[k][i] * A[k][j]: broadcast row k in groups of 8 first, then the FMAs, so that the SGPR written by a
    // v_readlane is not consumed by the very next VALU instruction (that hazard costs an s_nop per update otherwise)
    static_for<0, (k + 7) / 8>([&](auto gc) {
      constexpr int j0 = decltype(gc)::value * 8;
      double sj[8];
      static_for<0, 8>([&](auto qc) { constexpr int q = decltype(qc)::value; if constexpr (j0 + q < k) sj[q] = readlane_d(a[j0 + q], k); });
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 8>([&](auto qc) { constexpr int q = decltype(qc)::value; if constexpr (j0 + q < k) a[j0 + q] -= l * sj[q]; });
      __builtin_amdgcn_sched_barrier(0);
    });
    f.up[k] = l;
  });
  f.up[0] = 0.0;
  if (dg < D_MINVAL) dg = D_MINVAL;
  f.rinv = fast_rcp(dg);
  static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; f.lo[j] = (act && j < i) ? a[j] * f.rinv : 0.0; });
}
template <int N>
DEV double ldl_solve_regs(const LDLRegs<N> &f, double xi) {
  static_for<1, N>([&](auto kc) { constexpr int k = N - decltype(kc)::value; xi -= f.up[k] * readlane_d(xi, k); });      // L^T u = b
  xi *= f.rinv;                                                                                                          // D v = u
  static_for<0, N - 1>([&](auto jc) { constexpr int j = decltype(jc)::value; xi -= f.lo[j] * readlane_d(xi, j); });      // L x = v
  return xi;
}
// ---- tree-structured elimination (DofTree<N>::known, DevModel::tree_ok) ----------------------------------------------
// Same L^T D L, same per-entry arithmetic, but (a) the pivots of one tree level (same height above the leaves: never
// related to each other) are processed together, so their reciprocal chains and trailing updates are independent work
// the scheduler interleaves, and (b) only the ancestors of a pivot are updated: the other columns of M's pattern are
// structural zeros.  Cross-branch contacts break the pattern of H; the caller then asks for the dense order.
template <int N> constexpr int tree_depth(int k) { int d = 0; for (int a = DofTree<N>::parent(k); a >= 0; a = DofTree<N>::parent(a)) d++; return d; }
template <int N> constexpr int tree_nth_anc(int k, int n) { int a = DofTree<N>::parent(k); for (int q = 0; q < n; q++) a = DofTree<N>::parent(a); return a; }
template <int N> constexpr int tree_height(int k) {
  int h = 0;
  for (int c = k + 1; c < N; c++) if (DofTree<N>::parent(c) == k) { int hc = tree_height<N>(c) + 1; if (hc > h) h = hc; }
  return h;
}
template <int N> constexpr int tree_nlevel() { int h = 0; for (int k = 0; k < N; k++) { int hk = tree_height<N>(k); if (hk > h) h = hk; } return h + 1; }   // forests: the tallest root
// (pivot, ancestor) pairs of one level, pivots in descending order
template <int N> constexpr int tree_level_npair(int H) { int n = 0; for (int k = N - 1; k >= 1; k--) if (tree_height<N>(k) == H) n += tree_depth<N>(k); return n; }
template <int N> constexpr int tree_level_pair(int H, int e, bool pivot) {
  for (int k = N - 1; k >= 1; k--) if (tree_height<N>(k) == H) {
    int d = tree_depth<N>(k);
    if (e < d) return pivot ? k : tree_nth_anc<N>(k, e);
    e -= d;
  }
  return 0;
}
#ifndef MJPC_TREE_CHUNK
#define MJPC_TREE_CHUNK 8
#endif
#ifdef MJPC_TREE_NOBAR
#define TREE_BAR() ((void)0)
#else
#define TREE_BAR() __builtin_amdgcn_sched_barrier(0)
#endif
template <int N>
DEV void ldl_factor_tree(const double *A, int nvp, LDLRegs<N> &f, const LDLExtra *ex = nullptr) {
  static_assert(DofTree<N>::known && N <= 64, "one matrix row per lane");
  const int i = LANE;
  const bool act = i < N;
  double a[N];
  double dg;
  ldl_load_row<N>(A, nvp, ex, a, dg);
  static_for<0, tree_nlevel<N>()>([&](auto hc) {
    constexpr int H = decltype(hc)::value;
    // 1/d in the vector domain (every lane inverts its own diagonal: one reciprocal chain per LEVEL, no scalar clamp
    // round trip), then the pivots' reciprocals are broadcast
    const double rv = fast_rcp(dg < D_MINVAL ? D_MINVAL : dg);
    static_for<1, N>([&](auto kc) {
      constexpr int k = N - decltype(kc)::value;
      if constexpr (tree_height<N>(k) == H) f.up[k] = readlane_d(rv, k);
    });
    static_for<1, N>([&](auto kc) {                       // l = L[k][i] on the ancestors of k, diagonal update
      constexpr int k = N - decltype(kc)::value;
      if constexpr (tree_height<N>(k) == H) {
        double hk = a[k];
        double l = (i < k) ? hk * f.up[k] : 0.0;
        dg -= l * hk;
        f.up[k] = l;
      }
    });
    constexpr int NP = tree_level_npair<N>(H);            // A[i][j] -= L[k][i] * A[k][j], j ancestor of k
    static_for<0, (NP + MJPC_TREE_CHUNK - 1) / MJPC_TREE_CHUNK>([&](auto gc) {
      constexpr int e0 = decltype(gc)::value * MJPC_TREE_CHUNK;
      double sj[MJPC_TREE_CHUNK];
      static_for<0, MJPC_TREE_CHUNK>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (e0 + q < NP) {
          constexpr int j = tree_level_pair<N>(H, e0 + q, false), k = tree_level_pair<N>(H, e0 + q, true);
          sj[q] = readlane_d(a[j], k);
        }
      });
      TREE_BAR();
      static_for<0, MJPC_TREE_CHUNK>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (e0 + q < NP) {
          constexpr int j = tree_level_pair<N>(H, e0 + q, false), k = tree_level_pair<N>(H, e0 + q, true);
          a[j] -= f.up[k] * sj[q];
        }
      });
      TREE_BAR();
    });
  });
  f.up[0] = 0.0;
  if (dg < D_MINVAL) dg = D_MINVAL;
  f.rinv = fast_rcp(dg);
#pragma unroll
  for (int j = 0; j < N; j++) f.lo[j] = (act && j < i) ? a[j] * f.rinv : 0.0;
}
template <int N>
DEV double ldl_solve_tree(const LDLRegs<N> &f, double xi) {
  static_for<0, tree_nlevel<N>()>([&](auto hc) {          // L^T u = b, leaves first
    constexpr int H = decltype(hc)::value;
    double s[N];
    static_for<1, N>([&](auto kc) { constexpr int k = N - decltype(kc)::value; if constexpr (tree_height<N>(k) == H) s[k] = readlane_d(xi, k); });
    static_for<1, N>([&](auto kc) { constexpr int k = N - decltype(kc)::value; if constexpr (tree_height<N>(k) == H) xi -= f.up[k] * s[k]; });
  });
  xi *= f.rinv;                                           // D v = u
  static_for<1, tree_nlevel<N>()>([&](auto hc) {          // L x = v, root first (leaves have no descendants)
    constexpr int H = tree_nlevel<N>() - decltype(hc)::value;
    double s[N];
    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (tree_height<N>(j) == H) s[j] = readlane_d(xi, j); });
    static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; if constexpr (tree_height<N>(j) == H) xi -= f.lo[j] * s[j]; });
  });
  return xi;
}
template <int N>
DEV void ldl_factor_any(const double *A, int nvp, LDLRegs<N> &f, int tree, const LDLExtra *ex = nullptr) {
  if constexpr (DofTree<N>::known) { if (tree) { ldl_factor_tree<N>(A, nvp, f, ex); return; } }
  ldl_factor_regs<N>(A, nvp, f, ex);
}
template <int N>
DEV double ldl_solve_any(const LDLRegs<N> &f, double xi, int tree) {
  if constexpr (DofTree<N>::known) { if (tree) return ldl_solve_tree<N>(f, xi); }
  return ldl_solve_regs<N>(f, xi);
}
// split form (factor kept in LDS between phases): L[i][j] (j < i) in the lower triangle of A, 1/d in Dinv
template <int N>
DEV void chol_factor_reg(double *A, double *Dinv, int nvp, int tree) {
  SYNC();
  LDLRegs<N> f;
  ldl_factor_any<N>(A, nvp, f, tree);
  const int i = LANE;
  static_for<0, N>([&](auto jc) { constexpr int j = decltype(jc)::value; if (i < N && j < i) A[i * nvp + j] = f.lo[j]; });
  if (i < N) Dinv[i] = f.rinv;
  SYNC();
}
template <int N>
DEV void chol_solve_reg(const double *L, const double *Dinv, double *x, int nvp, int tree) {
  SYNC();
  const int i = LANE;
  const bool act = i < N;
  LDLRegs<N> f;
  f.rinv = act ? Dinv[i] : 0.0;
  static_for<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    f.lo[k] = (act && k < i) ? L[i * nvp + k] : 0.0;
    f.up[k] = (act && k > i) ? L[k * nvp + i] : 0.0;
  });
  double xi = ldl_solve_any<N>(f, act ? x[i] : 0.0, tree);
  if (act) x[i] = xi;
  SYNC();
}
// fused factor + solve (the factor never leaves the registers): Newton direction, implicit-damping solve
template <int N>
DEV void chol_factor_solve_reg(const double *A, double *x, int nvp, int tree, const LDLExtra *ex = nullptr) {
  SYNC();
  LDLRegs<N> f;
  ldl_factor_any<N>(A, nvp, f, tree, ex);
  const int i = LANE;
  double xi = ldl_solve_any<N>(f, i < N ? x[i] : 0.0, tree);
  if (i < N) x[i] = xi;
  SYNC();
}
#endif

// `tree`: the matrix has the sparsity pattern of M and the model's dof tree is DofTree<NVT> (DevModel::tree_ok)
template <int NVT>
DEV void chol_factor(double *A, double *Linv, double *tmp, int n, int nvp, int tree) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_factor_reg<NVT>(A, Linv, NVP_OF(NVT), tree); return; }
#endif
  chol_factor_lds(A, Linv, tmp, n, nvp);
}
template <int NVT>
DEV void chol_solve(const double *L, const double *Linv, double *x, int n, int nvp, int tree) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_solve_reg<NVT>(L, Linv, x, NVP_OF(NVT), tree); return; }
#endif
  chol_solve_lds(L, Linv, x, n, nvp);
}
// A (lower triangle) is consumed; x <- A^-1 x.  The generic build leaves the LL^T factor in A / Linv.
template <int NVT>
DEV void chol_factor_solve(double *A, double *Linv, double *tmp, double *x, int n, int nvp, int tree) {
#ifndef MJPC_EMU
  if constexpr (NVT > 0) { chol_factor_solve_reg<NVT>(A, x, NVP_OF(NVT), tree); return; }
#endif
  chol_factor_lds(A, Linv, tmp, n, nvp);
  chol_solve_lds(A, Linv, x, n, nvp);
}
