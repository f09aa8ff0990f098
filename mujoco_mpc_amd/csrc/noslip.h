// noslip.h — mj_solNoSlip (MuJoCo engine_solver.c) on the owner wave, after the Newton solve of a model with noslip_iterations > 0:
// the friction-loss rows and the friction dimensions of the contacts are re-solved in the dual WITHOUT the regulariser R by
// projected Gauss-Seidel over efc_force (limits, equalities and the contacts' normal forces stay as the Newton solve left them),
// then qacc = qacc_smooth + M^-1 J^T force.  (part of core.h)
//
// MuJoCo works on the dense dual matrix AR = J M^-1 J^T + diag(R).  Here the rows B_r = M^-1 J_r^T are kept instead (same size
// as J) together with w = M^-1 J^T force, one entry per lane: the residual of a block is J_blk w + b_blk, a force change d updates
// w += sum_k d_k B_k, the small diagonal blocks J_blk B_blk^T are formed once per step.  Same sweeps, same block updates and
// the same stopping rule as the CPU restatement the tests check it against.
#pragma once
#define NS_BT(c) (lds_base() + (c).K->L.noslip)

// entry i of constraint row r (a friction-loss row is the unit vector of its dof and has no stored row)
DEV double ns_jent(Ctx &c, int r, int i) { return r < c.M->nfric ? (c.efc_dof[r] == i ? 1.0 : 0.0) : c.efc_J[r * c.M->nvp + i]; }

// min 0.5 x'Ax + x'b  s.t.  sum (x_i / d_i)^2 <= r^2  (mju_QCQP2 / QCQP3 / QCQP): scaled to a ball, Newton on the multiplier.
// Uniform scalar code: every lane computes the same values.
DEV void ns_qcqp2(double *res, const double *Ain, const double *bin, const double *d, double r) {
  double b1 = bin[0] * d[0], b2 = bin[1] * d[1];
  double A11 = Ain[0] * d[0] * d[0], A22 = Ain[3] * d[1] * d[1], A12 = Ain[1] * d[0] * d[1];
  double la = 0, v1 = 0, v2 = 0;
  for (int iter = 0; iter < 20; iter++) {
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) { res[0] = 0; res[1] = 0; return; }
    double detinv = fast_rcp(det);
    double P11 = (A22 + la) * detinv, P22 = (A11 + la) * detinv, P12 = -A12 * detinv;
    v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
    double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
    double delta = d_div(-val, deriv);
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d[0]; res[1] = v2 * d[1];
}
DEV void ns_qcqp3(double *res, const double *Ain, const double *bin, const double *d, double r) {
  double b1 = bin[0] * d[0], b2 = bin[1] * d[1], b3 = bin[2] * d[2];
  double A11 = Ain[0] * d[0] * d[0], A22 = Ain[4] * d[1] * d[1], A33 = Ain[8] * d[2] * d[2];
  double A12 = Ain[1] * d[0] * d[1], A13 = Ain[2] * d[0] * d[2], A23 = Ain[5] * d[1] * d[2];
  double la = 0, v1 = 0, v2 = 0, v3 = 0;
  for (int iter = 0; iter < 20; iter++) {
    double P11 = (A22 + la) * (A33 + la) - A23 * A23, P22 = (A11 + la) * (A33 + la) - A13 * A13, P33 = (A11 + la) * (A22 + la) - A12 * A12;
    double P12 = A13 * A23 - A12 * (A33 + la), P13 = A12 * A23 - A13 * (A22 + la), P23 = A12 * A13 - A23 * (A11 + la);
    double det = (A11 + la) * P11 + A12 * P12 + A13 * P13;
    if (det < 1e-10) { res[0] = res[1] = res[2] = 0; return; }
    double detinv = fast_rcp(det);
    P11 *= detinv; P22 *= detinv; P33 *= detinv; P12 *= detinv; P13 *= detinv; P23 *= detinv;
    v1 = -P11 * b1 - P12 * b2 - P13 * b3; v2 = -P12 * b1 - P22 * b2 - P23 * b3; v3 = -P13 * b1 - P23 * b2 - P33 * b3;
    double val = v1 * v1 + v2 * v2 + v3 * v3 - r * r;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + P22 * v2 * v2 + P33 * v3 * v3) - 4 * (P12 * v1 * v2 + P13 * v1 * v3 + P23 * v2 * v3);
    double delta = d_div(-val, deriv);
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d[0]; res[1] = v2 * d[1]; res[2] = v3 * d[2];
}
// mju_cholFactor with a rank threshold / mju_cholSolve.  The size is a compile-time constant and every loop unrolls: the small
// matrices live in registers (with run-time sizes they sit in scratch memory and a pass costs milliseconds - measured)
template <int N>
DEV int ns_small_chol(double *A, double *dinv, double mindiag) {      // dinv[j] = 1 / L[j][j]: the substitutions multiply
  int rank = N;
#pragma unroll
  for (int j = 0; j < N; j++) {
    double t = A[j * N + j];
#pragma unroll
    for (int k = 0; k < j; k++) t -= A[j * N + k] * A[j * N + k];
    if (t < mindiag) { t = mindiag; rank--; }
    const double inv = fast_rsqrt(t);
    A[j * N + j] = t * inv; dinv[j] = inv;
#pragma unroll
    for (int i = j + 1; i < N; i++) {
      double s = A[i * N + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= A[i * N + k] * A[j * N + k];
      A[i * N + j] = s * inv;
    }
  }
  return rank;
}
template <int N>
DEV void ns_small_chol_solve(double *x, const double *L, const double *dinv, const double *b) {
#pragma unroll
  for (int i = 0; i < N; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i * N + k] * x[k];
    x[i] = s * dinv[i];
  }
#pragma unroll
  for (int i = N - 1; i >= 0; i--) {
    double s = x[i];
#pragma unroll
    for (int k = i + 1; k < N; k++) s -= L[k * N + i] * x[k];
    x[i] = s * dinv[i];
  }
}
// mju_QCQP for N > 3.  The reference factors A + la I afresh in each of its (up to 20) Newton iterations on the multiplier - on a
// pinch grasp it runs 17 of them per call, its absolute tolerance 1e-10 is rarely met earlier - and a 5 x 5 Cholesky with two
// solves is one dependent chain of ~300 operations.  Here only the first iteration (la = 0: the rank test, the "already inside
// the cone" exit) is done that way; then A = Q T Q' is reduced to tridiagonal form once (three Householder reflections) and
// every further iteration solves with T + la I in O(N): the same function of la, so the same iterates up to rounding, at an
// eighth of the chain.  (|v| = |Q'v|: the norm and the derivative are taken in the rotated frame, v = Q y at the end.)
template <int N>
DEV void ns_qcqpn(double *res, const double *Ain, const double *bin, const double *d, double r) {
  double A[N * N], L[N * N], b[N], v[N], tmp[N], dinv[N], la = 0;
#pragma unroll
  for (int i = 0; i < N; i++) {
    b[i] = bin[i] * d[i];
#pragma unroll
    for (int j = 0; j < N; j++) A[j + i * N] = Ain[j + i * N] * d[i] * d[j];
  }
  // ---- iteration 0, as the reference does it
#pragma unroll
  for (int i = 0; i < N * N; i++) L[i] = A[i];
  if (ns_small_chol<N>(L, dinv, 1e-10) < N) {
#pragma unroll
    for (int i = 0; i < N; i++) res[i] = 0;
    return;
  }
  ns_small_chol_solve<N>(v, L, dinv, b);
  double val = 0;
#pragma unroll
  for (int i = 0; i < N; i++) { v[i] = -v[i]; val += v[i] * v[i]; }
  val -= r * r;
  int go = !(val < 1e-10);
  if (go) {
    ns_small_chol_solve<N>(tmp, L, dinv, v);
    double deriv = 0;
#pragma unroll
    for (int i = 0; i < N; i++) deriv += v[i] * tmp[i];
    deriv *= -2;
    double delta = d_div(-val, deriv);
    if (delta < 1e-10) go = 0; else la = delta;
  }
  if (go) {
    // ---- A = Q T Q' (Householder, lower triangle), c = Q' b
    double hv[(N - 2) * N], hb[N - 2], c[N];
#pragma unroll
    for (int i = 0; i < N; i++) c[i] = b[i];
#pragma unroll
    for (int k = 0; k < N - 2; k++) {
      double x2 = 0;
#pragma unroll
      for (int i = k + 2; i < N; i++) x2 += A[i * N + k] * A[i * N + k];
      const double x0 = A[(k + 1) * N + k];
      const double nrm2 = x0 * x0 + x2;
      const double nrm = nrm2 * fast_rsqrt(nrm2);
      const double alpha = x0 > 0 ? -nrm : nrm;
      const int skip = !(x2 > 0);                 // already tridiagonal in this column
#pragma unroll
      for (int i = 0; i < N; i++) hv[k * N + i] = (i > k && !skip) ? A[i * N + k] : 0.0;
      if (!skip) hv[k * N + k + 1] = x0 - alpha;
      double vv = 0;
#pragma unroll
      for (int i = k + 1; i < N; i++) vv += hv[k * N + i] * hv[k * N + i];
      const double beta = skip ? 0.0 : 2.0 * fast_rcp(vv);
      hb[k] = beta;
      // p = beta A v, K = beta/2 v'p, w = p - K v, A -= v w' + w v'   (trailing block; symmetric storage kept full)
      double pw[N];
      double vp = 0;
#pragma unroll
      for (int i = k + 1; i < N; i++) {
        double t = 0;
#pragma unroll
        for (int j = k + 1; j < N; j++) t += A[i * N + j] * hv[k * N + j];
        pw[i] = beta * t; vp += hv[k * N + i] * pw[i];
      }
      const double K = 0.5 * beta * vp;
#pragma unroll
      for (int i = k + 1; i < N; i++) pw[i] -= K * hv[k * N + i];
#pragma unroll
      for (int i = k + 1; i < N; i++)
#pragma unroll
        for (int j = k + 1; j < N; j++) A[i * N + j] -= hv[k * N + i] * pw[j] + pw[i] * hv[k * N + j];
      if (!skip) {
        A[(k + 1) * N + k] = alpha; A[k * N + k + 1] = alpha;
#pragma unroll
        for (int i = k + 2; i < N; i++) { A[i * N + k] = 0; A[k * N + i] = 0; }
      }
      double vc = 0;
#pragma unroll
      for (int i = k + 1; i < N; i++) vc += hv[k * N + i] * c[i];
#pragma unroll
      for (int i = k + 1; i < N; i++) c[i] -= beta * vc * hv[k * N + i];
    }
    double td[N], te[N], y[N], z[N], rp[N], lf[N];
#pragma unroll
    for (int i = 0; i < N; i++) { td[i] = A[i * N + i]; te[i] = i + 1 < N ? A[(i + 1) * N + i] : 0.0; }
    // ---- iterations 1 .. 19 in the rotated frame
    for (int iter = 1; iter < 20; iter++) {
      // T + la I = L D L': pivots and their reciprocals
      double piv = td[0] + la;
      rp[0] = fast_rcp(piv); lf[0] = 0;
#pragma unroll
      for (int i = 1; i < N; i++) { lf[i] = te[i - 1] * rp[i - 1]; piv = td[i] + la - lf[i] * te[i - 1]; rp[i] = fast_rcp(piv); }
      // y = -(T + la)^-1 c
      y[0] = -c[0];
#pragma unroll
      for (int i = 1; i < N; i++) y[i] = -c[i] - lf[i] * y[i - 1];
      y[N - 1] *= rp[N - 1];
#pragma unroll
      for (int i = N - 2; i >= 0; i--) y[i] = (y[i] - te[i] * y[i + 1]) * rp[i];
      val = 0;
#pragma unroll
      for (int i = 0; i < N; i++) val += y[i] * y[i];
      val -= r * r;
      if (val < 1e-10) break;
      z[0] = y[0];
#pragma unroll
      for (int i = 1; i < N; i++) z[i] = y[i] - lf[i] * z[i - 1];
      z[N - 1] *= rp[N - 1];
#pragma unroll
      for (int i = N - 2; i >= 0; i--) z[i] = (z[i] - te[i] * z[i + 1]) * rp[i];
      double deriv = 0;
#pragma unroll
      for (int i = 0; i < N; i++) deriv += y[i] * z[i];
      deriv *= -2;
      double delta = d_div(-val, deriv);
      if (delta < 1e-10) break;
      la += delta;
    }
    // v = Q y: the reflections in reverse order
#pragma unroll
    for (int i = 0; i < N; i++) v[i] = y[i];
#pragma unroll
    for (int k = N - 3; k >= 0; k--) {
      double vy = 0;
#pragma unroll
      for (int i = k + 1; i < N; i++) vy += hv[k * N + i] * v[i];
#pragma unroll
      for (int i = k + 1; i < N; i++) v[i] -= hb[k] * vy * hv[k * N + i];
    }
  }
#pragma unroll
  for (int i = 0; i < N; i++) res[i] = v[i] * d[i];
}
// cost change of a block update; an update that raises the cost is taken back (costChange)
template <int DIM>
DEV double ns_cost_change(const double *A, double *force, const double *oldforce, const double *res) {
  double delta[DIM], change = 0;
#pragma unroll
  for (int j = 0; j < DIM; j++) delta[j] = force[j] - oldforce[j];
#pragma unroll
  for (int j = 0; j < DIM; j++) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < DIM; k++) t += A[j * DIM + k] * delta[k];
    change += 0.5 * delta[j] * t;
  }
#pragma unroll
  for (int j = 0; j < DIM; j++) change += delta[j] * res[j];
  if (change > 1e-10) {
#pragma unroll
    for (int j = 0; j < DIM; j++) force[j] = oldforce[j];
    change = 0;
  }
  return change;
}
// one elliptic contact of dimension DIM (rows r0 .. r0 + DIM - 1): residual J_blk w + b_blk, the friction forces from the QCQP over
// the cone of the (unchanged) normal force, w updated by the change.  Returns the block's cost change.
template <int DIM>
DEV double ns_elliptic_block(Ctx &c, int r0, const double *Acs_c, const double *cc, const double *Bt, const double *bb, double *w) {
  const int nv = c.M->nv, nvp = c.M->nvp;
  double *force = c.efc_force;
  double pk[6] = {0, 0, 0, 0, 0, 0};
  PFOR(i, nv) {
    const double wi = w[i];
#pragma unroll
    for (int k = 0; k < DIM; k++) pk[k] += c.efc_J[(r0 + k) * nvp + i] * wi;
  }
  wave_sum3(pk[0], pk[1], pk[2]);
  if (DIM > 3) wave_sum3(pk[3], pk[4], pk[5]);
  double res[DIM], old[DIM], f[DIM], Ac[DIM * DIM];
#pragma unroll
  for (int k = 0; k < DIM; k++) { res[k] = pk[k] + bb[r0 + k]; old[k] = force[r0 + k]; f[k] = old[k]; }
#pragma unroll
  for (int e = 0; e < DIM * DIM; e++) Ac[e] = Acs_c[e];
#pragma unroll
  for (int k = 0; k < DIM; k++) Ac[k * (DIM + 1)] = fmax(1e-10, Ac[k * (DIM + 1)]);
  if (old[0] < D_MINVAL) {
#pragma unroll
    for (int k = 1; k < DIM; k++) f[k] = 0;
  } else {
    double bc[DIM - 1], Af[(DIM - 1) * (DIM - 1)], v[DIM - 1], mu[DIM - 1];
#pragma unroll
    for (int k = 0; k < DIM - 1; k++) mu[k] = cc[CON_FRICTION + k];
#pragma unroll
    for (int j = 0; j < DIM - 1; j++) {
      bc[j] = res[j + 1];
#pragma unroll
      for (int k = 0; k < DIM - 1; k++) { Af[j * (DIM - 1) + k] = Ac[(j + 1) * DIM + (k + 1)]; bc[j] -= Ac[(j + 1) * DIM + (k + 1)] * old[k + 1]; }
    }
    if constexpr (DIM == 3) ns_qcqp2(v, Af, bc, mu, old[0]);
    else if constexpr (DIM == 4) ns_qcqp3(v, Af, bc, mu, old[0]);
    else ns_qcqpn<DIM - 1>(v, Af, bc, mu, old[0]);
#pragma unroll
    for (int j = 0; j < DIM - 1; j++) f[1 + j] = v[j];
  }
  const double change = ns_cost_change<DIM>(Ac, f, old, res);
  SYNC();
  if (LANE == 0) {
#pragma unroll
    for (int k = 1; k < DIM; k++) force[r0 + k] = f[k];
  }
  PFOR(i, nv) {
    double acc = w[i];
#pragma unroll
    for (int k = 1; k < DIM; k++) acc += (f[k] - old[k]) * Bt[(r0 + k) * nvp + i];
    w[i] = acc;
  }
  SYNC();
  return change;
}

// B_r = M^-1 J_r^T for every row, from the factor of M in qL / Linv
template <int NVT>
DEV void ns_minv_rows(Ctx &c, double *Bt) {
  const DevModel &M = *c.M;
  const int nv = M.nv, nvp = M.nvp, nefc = c.nefc;
#ifndef MJPC_EMU
  if constexpr (NVT > 0) {
    // the factor is loaded into registers once; each row then costs one pair of substitutions
    SYNC();
    const int i = LANE;
    const bool act = i < NVT;
    LDLRegs<NVT> f;
    f.rinv = act ? c.Linv[i] : 0.0;
    static_for<0, NVT>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      f.lo[k] = (act && k < i) ? c.qL[i * NVP_OF(NVT) + k] : 0.0;
      f.up[k] = (act && k > i) ? c.qL[k * NVP_OF(NVT) + i] : 0.0;
    });
    for (int r = 0; r < nefc; r++) {
      double xi = ldl_solve_any<NVT>(f, act ? ns_jent(c, r, i) : 0.0, M.tree_ok);
      if (act) Bt[r * nvp + i] = xi;
    }
    SYNC();
    return;
  }
#endif
  for (int r = 0; r < nefc; r++) {
    PFOR(i, nv) c.Mv[i] = ns_jent(c, r, i);
    SYNC();
    chol_solve<NVT>(c.qL, c.Linv, c.Mv, nv, nvp, M.tree_ok);
    SYNC();
    PFOR(i, nv) Bt[r * nvp + i] = c.Mv[i];
    SYNC();
  }
}

template <int NVT>
DEV void noslip_pass(Ctx &c) {
  const DevModel &M = *c.M;
  const int nv = M.nv, nvp = M.nvp, nefc = c.nefc, ncon = c.ncon;
  // the next step's warm start is the Newton solution (mj_fwdConstraint saves it before the noslip pass)
  PFOR(i, nv) c.qacc_ws[i] = c.qacc[i];
  SYNC();
  if (nefc == 0) return;
  double *Bt = NS_BT(c), *Acs = Bt + M.nefcmax * nvp, *bb = Acs + M.nconmax * 36, *dg = bb + M.nefcmax;
  double *force = c.efc_force;
  ns_minv_rows<NVT>(c, Bt);
  // b = J qacc_smooth - aref;  diagonal J_r B_r of the friction-loss rows
  PFOR(r, nefc) {
    double s = 0, t = 0;
    for (int i = 0; i < nv; i++) { double j = ns_jent(c, r, i); s += j * c.qacc_smooth[i]; t += j * Bt[r * nvp + i]; }
    bb[r] = s - c.efc_aref[r]; dg[r] = t;
  }
  // diagonal blocks of the contacts (elliptic: dim x dim; pyramidal: one 2 x 2 per pair of opposing edges), R not included
  for (int ci = 0; ci < ncon; ci++) {
    const int dim = c.con_i[ci * CONI_STRIDE], r0 = c.con_i[ci * CONI_STRIDE + 3], type = c.efc_type[r0];
    if (dim == 1) continue;
    if (type == CNSTR_CONTACT_ELLIPTIC) {
      PFOR(e, dim * dim) {
        int k = e / dim, l = e - k * dim;
        double s = 0;
        for (int i = 0; i < nv; i++) s += c.efc_J[(r0 + k) * nvp + i] * Bt[(r0 + l) * nvp + i];
        Acs[ci * 36 + e] = s;
      }
    } else {
      PFOR(e, 4 * (dim - 1)) {
        int pr = e / 4, k = (e & 3) >> 1, l = e & 1;
        double s = 0;
        for (int i = 0; i < nv; i++) s += c.efc_J[(r0 + 2 * pr + k) * nvp + i] * Bt[(r0 + 2 * pr + l) * nvp + i];
        Acs[ci * 36 + e] = s;
      }
    }
  }
  SYNC();
  // w = M^-1 J^T force (in LDS: the one-lane emulation build runs the same source)
  double *w = c.search;
  PFOR(i, nv) { double s = 0; for (int r = 0; r < nefc; r++) s += force[r] * Bt[r * nvp + i]; w[i] = s; }
  SYNC();
  const double scale = 1.0 / (M.meaninertia * (nv > 1 ? nv : 1));
  int iter = 0;
  while (iter < M.noslip_iterations) {
    double improvement = 0;
    if (iter == 0) {
      double p = 0;
      PFOR(r, nefc) p += 0.5 * force[r] * force[r] * c.efc_R[r];
      improvement = wave_sum(p);
    }
    // dry friction: dof rows, then tendon rows (MuJoCo's row order)
    for (int pass = 0; pass < 2; pass++) {
      const int want = pass == 0 ? CNSTR_FRICTION_DOF : CNSTR_FRICTION_TENDON;
      for (int r = 0; r < nefc; r++) {
        if (uniform_i(c.efc_type[r]) != want) continue;
        double p = 0;
        PFOR(i, nv) p += ns_jent(c, r, i) * w[i];
        const double res = wave_sum(p) + bb[r];
        const double old = force[r], arinv = fast_rcp(dg[r]), fl = c.efc_floss[r];
        double f = old - res * arinv;
        if (f < -fl) f = -fl; else if (f > fl) f = fl;
        const double delta = f - old;
        improvement -= 0.5 * delta * delta * dg[r] + delta * res;
        SYNC();
        if (LANE == 0) force[r] = f;
        PFOR(i, nv) w[i] += delta * Bt[r * nvp + i];
        SYNC();
      }
    }
    // contact friction
    for (int ci = 0; ci < ncon; ci++) {
      const int dim = uniform_i(c.con_i[ci * CONI_STRIDE]), r0 = uniform_i(c.con_i[ci * CONI_STRIDE + 3]), type = uniform_i(c.efc_type[r0]);
      if (dim == 1) continue;
      const double *cc = c.contact + ci * M.con_stride;
      if (type == CNSTR_CONTACT_PYRAMIDAL) {
        for (int pr = 0; pr < dim - 1; pr++) {
          const int j = r0 + 2 * pr;
          double p0 = 0, p1 = 0, pz = 0;
          PFOR(i, nv) { p0 += c.efc_J[j * nvp + i] * w[i]; p1 += c.efc_J[(j + 1) * nvp + i] * w[i]; }
          wave_sum3(p0, p1, pz);
          double res[2] = {p0 + bb[j], p1 + bb[j + 1]}, old[2] = {force[j], force[j + 1]}, f[2], Ac[4];
          for (int e = 0; e < 4; e++) Ac[e] = Acs[ci * 36 + 4 * pr + e];
          Ac[0] = fmax(1e-10, Ac[0]); Ac[3] = fmax(1e-10, Ac[3]);
          const double mid = 0.5 * (old[0] + old[1]);
          const double bc0 = res[0] - Ac[0] * old[0] - Ac[1] * old[1], bc1 = res[1] - Ac[2] * old[0] - Ac[3] * old[1];
          const double K1 = Ac[0] + Ac[3] - Ac[1] - Ac[2], K0 = mid * (Ac[0] - Ac[3]) + bc0 - bc1;
          if (K1 < D_MINVAL) f[0] = f[1] = mid;
          else {
            double y = d_div(-K0, K1);
            if (y < -mid) { f[0] = 0; f[1] = 2 * mid; }
            else if (y > mid) { f[0] = 2 * mid; f[1] = 0; }
            else { f[0] = mid + y; f[1] = mid - y; }
          }
          improvement -= ns_cost_change<2>(Ac, f, old, res);
          SYNC();
          if (LANE == 0) { force[j] = f[0]; force[j + 1] = f[1]; }
          PFOR(i, nv) w[i] += (f[0] - old[0]) * Bt[j * nvp + i] + (f[1] - old[1]) * Bt[(j + 1) * nvp + i];
          SYNC();
        }
      } else if (type == CNSTR_CONTACT_ELLIPTIC) {
        const double *Ab = Acs + ci * 36;
        if (dim == 3) improvement -= ns_elliptic_block<3>(c, r0, Ab, cc, Bt, bb, w);
        else if (dim == 4) improvement -= ns_elliptic_block<4>(c, r0, Ab, cc, Bt, bb, w);
        else improvement -= ns_elliptic_block<6>(c, r0, Ab, cc, Bt, bb, w);
      }
    }
    improvement *= scale;
    iter++;
    if (improvement < M.noslip_tolerance) break;
  }
  // dualFinish: qfrc_constraint = J^T force, qacc = qacc_smooth + M^-1 J^T force (= w)
  PFOR(i, nv) {
    double s = 0;
    for (int r = 0; r < nefc; r++) { double f = force[r]; if (f != 0) s += ns_jent(c, r, i) * f; }
    c.qfrc_constraint[i] = s;
    c.qacc[i] = c.qacc_smooth[i] + w[i];
  }
  SYNC();
}
