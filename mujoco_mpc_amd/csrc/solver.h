// solver.h — primal Newton solver for  min_a 1/2 (a-a0)^T M (a-a0) + sum_i s_i(J_i a - aref_i)
// (what MuJoCo's default solver computes inside mj_step, mjpc/trajectory.cc:158), one wavefront.
//
// Layout of the work over the 64 lanes:
//   * rows with a single +-1 Jacobian entry (friction loss, joint limits: [0, nsingle)) only touch the
//     Hessian diagonal; contact rows [nsingle, nefc) go through WJ = blockdiag(W) J so that the Hessian is
//     the plain contraction H = M + J^T (W J) over contact rows — branch-free, unrollable inner loops;
//   * the exact line search keeps each lane's rows / contact in VGPRs for all its evaluations; one
//     evaluation = ALU + three wave reductions;
//   * Ma and jar are updated incrementally along the search direction.
#pragma once
#include "linalg.h"

#ifdef MJPC_EMU
#define LS_RPL 192     // rows per lane (1 lane owns everything in the emulation build)
#define LS_CPL 64
#else
#define LS_RPL 3       // nefcmax <= 192
#define LS_CPL 1       // nconmax <= 64
#endif

// constraint cost at efc_jar; fills force/state (and cone Hessians); returns this lane's partial cost
DEV double constraint_update(Ctx &c, int hess) {
  double cost = 0;
  PFOR(i, c.nefc) {
    int type = c.efc_type[i];
    if (type == CNSTR_CONTACT_ELLIPTIC) continue;
    double D = c.efc_D[i], R = c.efc_R[i], x = c.efc_jar[i];
    if (type == CNSTR_FRICTION_DOF) {
      double f = c.efc_floss[i];
      if (x <= -R * f) { cost += -0.5 * R * f * f - f * x; c.efc_force[i] = f; c.efc_state[i] = STATE_LINEARNEG; }
      else if (x >= R * f) { cost += -0.5 * R * f * f + f * x; c.efc_force[i] = -f; c.efc_state[i] = STATE_LINEARPOS; }
      else { cost += 0.5 * D * x * x; c.efc_force[i] = -D * x; c.efc_state[i] = STATE_QUADRATIC; }
    } else {
      if (x >= 0) { c.efc_force[i] = 0; c.efc_state[i] = STATE_SATISFIED; }
      else { cost += 0.5 * D * x * x; c.efc_force[i] = -D * x; c.efc_state[i] = STATE_QUADRATIC; }
    }
    if (i < c.nsingle) {
      // rows with one +-1 Jacobian entry: fold J^T force and the Hessian diagonal per dof
      // (slot 0/1: friction loss, slot 2/3: joint limit; at most one active row of each kind per dof)
      int d = c.efc_dof[i], k = (type == CNSTR_FRICTION_DOF) ? 0 : 2;
      int nv = c.M->nv;
      c.sgl[k * nv + d] = c.efc_J[i * c.M->nvp + d] * c.efc_force[i];
      c.sgl[(k + 1) * nv + d] = (c.efc_state[i] == STATE_QUADRATIC) ? D : 0.0;
    }
  }
  PFOR(ci, c.ncon) {
    int dim = c.con_i[ci * CONI_STRIDE];
    int i = c.con_i[ci * CONI_STRIDE + 3];
    if (dim <= 1 || c.efc_type[i] != CNSTR_CONTACT_ELLIPTIC) continue;     // pyramidal edges are plain unilateral rows
    double *cc = c.contact + ci * c.M->con_stride;
    double mu = cc[CON_MU], U[6], fr[6];
    fr[0] = mu;
#pragma unroll
    for (int j = 1; j < 6; j++) fr[j] = j < dim ? cc[CON_FRICTION + j - 1] : 0;
    double T2 = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { U[j] = j < dim ? c.efc_jar[i + j] * fr[j] : 0; if (j > 0) T2 += U[j] * U[j]; }
    double N = U[0], T = sqrt(T2);
    int st;
    if (N >= mu * T || (T <= 0 && N >= 0)) {
#pragma unroll
      for (int j = 0; j < 6; j++) if (j < dim) c.efc_force[i + j] = 0;
      st = STATE_SATISFIED;
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
      for (int j = 0; j < 6; j++) if (j < dim) { double xj = c.efc_jar[i + j], Dj = c.efc_D[i + j]; cost += 0.5 * Dj * xj * xj; c.efc_force[i + j] = -Dj * xj; }
      st = STATE_QUADRATIC;
    } else {
      double Dm = c.efc_D[i] / (mu * mu * (1 + mu * mu));
      double NmT = N - mu * T;
      cost += 0.5 * Dm * NmT * NmT;
      double f0 = -Dm * NmT * mu;
      c.efc_force[i] = f0;
#pragma unroll
      for (int j = 1; j < 6; j++) if (j < dim) c.efc_force[i + j] = -f0 / T * U[j] * fr[j];
      st = STATE_CONE;
      if (hess) {
        // H = S d2s/dU2 S with S = diag(mu, friction), s = 1/2 Dm (N - mu T)^2
        double g[6];
        g[0] = 1;
#pragma unroll
        for (int j = 1; j < 6; j++) g[j] = -mu * U[j] / T;
        double iT = 1.0 / T, iT3 = iT * iT * iT;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
          for (int b = 0; b < 6; b++) {
            if (a < dim && b < dim) {
              double h = g[a] * g[b];
              if (a > 0 && b > 0) h += NmT * (-mu) * ((a == b ? iT : 0.0) - U[a] * U[b] * iT3);
              cc[CON_H + a * 6 + b] = Dm * h * fr[a] * fr[b];
            }
          }
      }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) if (j < dim) c.efc_state[i + j] = st;
  }
  return cost;
}

// full evaluation at qacc: Ma = M qacc, jar = J qacc - aref, force/state; returns total cost (uniform)
DEV double solver_eval(Ctx &c, const double *qacc, double *gauss_out) {
  const DevModel &M = *c.M;
  int nv = M.nv, nvp = M.nvp;
  double part = 0;
  PFOR(i, nv) {
    double s = 0;
#pragma unroll 6
    for (int j = 0; j < nv; j++) s += c.qM[i * nvp + j] * qacc[j];
    c.Ma[i] = s;
    part += 0.5 * (s - c.qfrc_smooth[i]) * (qacc[i] - c.qacc_smooth[i]);
  }
  PFOR(r, c.nefc) {
    double s = 0;
    if (r < c.nsingle) s = c.efc_J[r * nvp + c.efc_dof[r]] * qacc[c.efc_dof[r]];
    else {
#pragma unroll 6
      for (int j = 0; j < nv; j++) s += c.efc_J[r * nvp + j] * qacc[j];
    }
    c.efc_jar[r] = s - c.efc_aref[r];
  }
  SYNC();
  double gauss = wave_sum(part);
  double cc = wave_sum(constraint_update(c, 1));
  SYNC();
  if (gauss_out) *gauss_out = gauss;
  return gauss + cc;
}

// gradient, Hessian (lower triangle) and Newton direction Mgrad = H^-1 grad
template <int NVT>
DEV void newton_gradient(Ctx &c) {
  const DevModel &M = *c.M;
  const int nv = NVT > 0 ? NVT : M.nv, nvp = NVT > 0 ? (NVT | 1) : M.nvp;     // compile-time strides => immediate LDS offsets
  int nefc = c.nefc, ns = c.nsingle;
  int ncrow = nefc - ns;
  PROF(c, 13);
  // ordered compaction of the ACTIVE contact rows (quadratic or cone state): satisfied rows contribute nothing
  int nact = 0;
  for (int base = 0; base < ncrow; base += NLANE) {
    int rr = base + LANE;
    int flag = (rr < ncrow) && (c.efc_state[ns + rr] != STATE_SATISFIED);
    int tot, off = wave_excl_scan(flag, &tot);
    if (flag) c.active[nact + off] = ns + rr;
    nact += tot;
  }
  SYNC();
  // JA = J rows, WJ = blockdiag(W) J rows of the active set: one lane per row, all columns
  PFOR(a, nact) {
    int r = c.active[a];
    int st = c.efc_state[r];
    double *W = c.efc_WJ + a * nvp, *JA = c.efc_JA + a * nvp;
    const double *Jr = c.efc_J + r * nvp;
    c.efc_jv[a] = c.efc_force[r];                 // compact forces (jv is free between line searches)
    if (st == STATE_QUADRATIC) {
      double D = c.efc_D[r];
#pragma unroll 6
      for (int j = 0; j < nv; j++) { double v = Jr[j]; JA[j] = v; W[j] = D * v; }
    } else {
      int ci = c.efc_id[r];
      int dim = c.con_i[ci * CONI_STRIDE], r0 = c.con_i[ci * CONI_STRIDE + 3];
      const double *Hc = c.contact + ci * c.M->con_stride + CON_H + (r - r0) * 6;
      double hc[6];
#pragma unroll
      for (int b = 0; b < 6; b++) hc[b] = b < dim ? Hc[b] : 0.0;
      int rb[6];
#pragma unroll
      for (int b = 0; b < 6; b++) rb[b] = ((r0 + b < nefc) ? r0 + b : nefc - 1) * nvp;   // rows >= dim carry hc = 0
#pragma unroll 3
      for (int j = 0; j < nv; j++) {
        double w = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) w += hc[b] * c.efc_J[rb[b] + j];
        W[j] = w; JA[j] = Jr[j];
      }
    }
  }
  PROF(c, 19);
  SYNC();
  // gradient: Ma - qfrc_smooth - J^T force; diagonal Hessian terms of the single-entry rows
  PFOR(i, nv) {
    double g = c.Ma[i] - c.qfrc_smooth[i] - (c.sgl[i] + c.sgl[2 * nv + i]);
    double hd = c.sgl[nv + i] + c.sgl[3 * nv + i];
#pragma unroll 8
    for (int a = 0; a < nact; a++) g -= c.efc_JA[a * nvp + i] * c.efc_jv[a];
    c.grad[i] = g;
    c.Mgrad[i] = g;
    c.vtmp[i] = hd;
  }
  SYNC();
  PROF(c, 15);
  // H = M + diag + JA^T WJ on the lower triangle; without cross-branch contacts only M's sparsity pattern is non-zero
  int nent = c.cross ? nv * (nv + 1) / 2 : M.nmpair;
  PFOR(e, nent) {
    int i, j;
    if (c.cross) {
      i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while ((i + 1) * (i + 2) / 2 <= e) i++;
      while (i * (i + 1) / 2 > e) i--;
      j = e - i * (i + 1) / 2;
    } else { i = M.mpair_i[e]; j = M.mpair_j[e]; }
    double h = c.qM[i * nvp + j];
    if (i == j) h += c.vtmp[i];
    const double *Ji = c.efc_JA + i, *Wj = c.efc_WJ + j;
#pragma unroll 8
    for (int a = 0; a < nact; a++) h += Ji[a * nvp] * Wj[a * nvp];
    c.qH[i * nvp + j] = h;
  }
  // the in-place factor of the previous iteration filled the structural zeros: clear them again
  if (!c.cross) PFOR(e, M.nzpair) c.qH[M.zpair_i[e] * nvp + M.zpair_j[e]] = 0;
  PROF(c, 16);
  chol_factor_solve<NVT>(c.qH, c.Hinv, c.vtmp, c.Mgrad, nv, nvp);
  PROF(c, 17);
}

// ---- exact line search: phi(alpha) = Gauss(alpha) + sum_i s_i(jar + alpha*jv), data in registers
struct LSPoint { double cost, d1, d2; };
struct LSData {
  double D[LS_RPL], R[LS_RPL], F[LS_RPL], X[LS_RPL], V[LS_RPL];
  int T[LS_RPL];                        // 0 none, 1 friction, 2 unilateral
  double U0[LS_CPL][6], UV[LS_CPL][6], E[LS_CPL][6], mu[LS_CPL], Dm[LS_CPL];
  int dim[LS_CPL];
};

DEV void ls_load(Ctx &c, LSData &d) {
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    int r = LANE + NLANE * k;
    d.T[k] = 0; d.D[k] = 0; d.R[k] = 0; d.F[k] = 0; d.X[k] = 0; d.V[k] = 0;
    if (r < c.nefc) {
      int type = c.efc_type[r];
      if (type != CNSTR_CONTACT_ELLIPTIC) {
        d.T[k] = (type == CNSTR_FRICTION_DOF) ? 1 : 2;
        d.D[k] = c.efc_D[r]; d.R[k] = c.efc_R[r]; d.F[k] = c.efc_floss[r]; d.X[k] = c.efc_jar[r]; d.V[k] = c.efc_jv[r];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    int ci = LANE + NLANE * q;
    d.dim[q] = 0; d.mu[q] = 0; d.Dm[q] = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) { d.U0[q][j] = 0; d.UV[q][j] = 0; d.E[q][j] = 0; }
    if (ci < c.ncon) {
      int dim = c.con_i[ci * CONI_STRIDE];
      int i = c.con_i[ci * CONI_STRIDE + 3];
      if (dim > 1 && c.efc_type[i] == CNSTR_CONTACT_ELLIPTIC) {
        const double *cc = c.contact + ci * c.M->con_stride;
        double mu = cc[CON_MU];
        d.dim[q] = dim; d.mu[q] = mu;
        d.Dm[q] = c.efc_D[i] / (mu * mu * (1 + mu * mu));
#pragma unroll
        for (int j = 0; j < 6; j++) if (j < dim) {
          double fr = j == 0 ? mu : cc[CON_FRICTION + j - 1];
          d.U0[q][j] = c.efc_jar[i + j] * fr; d.UV[q][j] = c.efc_jv[i + j] * fr;
          d.E[q][j] = c.efc_D[i + j] / (fr * fr);     // D_j jar_j^2 = E_j U_j^2
        }
      }
    }
  }
}

DEV LSPoint ls_eval(const LSData &d, double q0, double q1, double q2, double a) {
  LSPoint p; p.cost = 0; p.d1 = 0; p.d2 = 0;
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    if (d.T[k] == 0) continue;
    double D = d.D[k], v = d.V[k], x = d.X[k] + a * v;
    if (d.T[k] == 1) {
      double f = d.F[k], Rf = d.R[k] * f;
      if (x <= -Rf) { p.cost += -0.5 * Rf * f - f * x; p.d1 += -f * v; }
      else if (x >= Rf) { p.cost += -0.5 * Rf * f + f * x; p.d1 += f * v; }
      else { p.cost += 0.5 * D * x * x; p.d1 += D * x * v; p.d2 += D * v * v; }
    } else if (x < 0) { p.cost += 0.5 * D * x * x; p.d1 += D * x * v; p.d2 += D * v * v; }
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    if (d.dim[q] <= 1) continue;
    double mu = d.mu[q], U[6];
    double T2 = 0, UV = 0, VV = 0;
#pragma unroll
    for (int j = 0; j < 6; j++) {
      U[j] = d.U0[q][j] + a * d.UV[q][j];
      if (j > 0) { T2 += U[j] * U[j]; UV += U[j] * d.UV[q][j]; VV += d.UV[q][j] * d.UV[q][j]; }
    }
    double iT = fast_rsqrt(T2);            // 1/T without an IEEE divide + sqrt on the critical path
    double N = U[0], T = T2 * iT;
    if (N >= mu * T || (T <= 0 && N >= 0)) {
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
      for (int j = 0; j < 6; j++) { double E = d.E[q][j], vj = d.UV[q][j]; p.cost += 0.5 * E * U[j] * U[j]; p.d1 += E * U[j] * vj; p.d2 += E * vj * vj; }
    } else {
      double Dm = d.Dm[q], NmT = N - mu * T;
      double T1 = UV * iT, T2d = (VV - T1 * T1) * iT;
      double g1 = d.UV[q][0] - mu * T1;
      p.cost += 0.5 * Dm * NmT * NmT; p.d1 += Dm * NmT * g1; p.d2 += Dm * (g1 * g1 - NmT * mu * T2d);
    }
  }
  p.cost = wave_sum(p.cost) + q0 + a * q1 + a * a * q2;
  p.d1 = wave_sum(p.d1) + q1 + 2 * a * q2;
  p.d2 = wave_sum(p.d2) + 2 * q2;
  return p;
}

// returns alpha; q1/q2 = Gauss quadratic coefficients along the direction (for the incremental update)
DEV double line_search(Ctx &c, double gauss, double *q1_out, double *q2_out) {
  const DevModel &M = *c.M;
  int nv = M.nv, nvp = M.nvp;
  double p_sn = 0, p_q1 = 0, p_q2 = 0;
  PFOR(i, nv) {
    double s = 0;
#pragma unroll 6
    for (int j = 0; j < nv; j++) s += c.qM[i * nvp + j] * c.search[j];
    c.Mv[i] = s;
    double si = c.search[i];
    p_sn += si * si; p_q1 += si * (c.Ma[i] - c.qfrc_smooth[i]); p_q2 += 0.5 * si * s;
  }
  PFOR(r, c.nefc) {
    double s = 0;
    if (r < c.nsingle) s = c.efc_J[r * nvp + c.efc_dof[r]] * c.search[c.efc_dof[r]];
    else {
#pragma unroll 6
      for (int j = 0; j < nv; j++) s += c.efc_J[r * nvp + j] * c.search[j];
    }
    c.efc_jv[r] = s;
  }
  SYNC();
  double snorm = sqrt(wave_sum(p_sn)), q1 = wave_sum(p_q1), q2 = wave_sum(p_q2);
  *q1_out = q1; *q2_out = q2;
  PROF(c, 20);
  double scale = 1.0 / (M.meaninertia * (nv > 1 ? nv : 1));
  if (snorm < D_MINVAL) return 0;
  double gtol = M.tolerance * M.ls_tolerance * snorm / scale;
  LSData d;
  ls_load(c, d);
  PROF(c, 21);
  LSPoint p0 = ls_eval(d, gauss, q1, q2, 0.0);
  if (!(p0.d2 > 0) || p0.d1 >= 0) return 0;
  // safeguarded Newton on phi'(alpha) (rtsafe): expand until phi' changes sign, then Newton steps that stay
  // inside the bracket and at least halve the previous step, else bisection; return the best point seen
  double lo = 0, hi = -1, a = -p0.d1 / p0.d2;
  double best_a = 0, best_cost = p0.cost, dxold = a, dx = a;
  for (int it = 0; it < M.ls_iterations; it++) {
    LSPoint p = ls_eval(d, gauss, q1, q2, a);
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
    if (LANE == 0) c.prof[23] += 1;
#endif
    if (p.cost < best_cost) { best_cost = p.cost; best_a = a; }
    if (fabs(p.d1) < gtol) break;
    if (p.d1 < 0) lo = a; else hi = a;
    double an;
    if (hi < 0) {
      an = (p.d2 > 0) ? a - p.d1 / p.d2 : 2 * a;
      if (!(an > a)) an = 2 * a;
      dxold = dx; dx = an - a;
    } else {
      double nw = (p.d2 > 0) ? a - p.d1 / p.d2 : lo - 1;
      int ok = (nw > lo) && (nw < hi) && (fabs(2 * p.d1) <= fabs(dxold * p.d2));
      dxold = dx;
      if (ok) { dx = fabs(nw - a); an = nw; }
      else { dx = 0.5 * (hi - lo); an = lo + dx; }
    }
    if (an == a) break;
    a = an;
  }
  PROF(c, 22);
  return best_a;
}

template <int NVT>
DEV void solve_constraints(Ctx &c) {
  const DevModel &M = *c.M;
  int nv = M.nv, nvp = M.nvp;
  c.solver_iter = 0;
  if (c.nefc == 0) {
    PFOR(i, nv) { c.qacc[i] = c.qacc_smooth[i]; c.qfrc_constraint[i] = 0; }
    SYNC();
    return;
  }
  PROF(c, 7);
  // warm start: the better of qacc_smooth and qacc_warmstart (evaluated last, so its force/state stay valid)
  double gauss, cost;
  double cost_sm = solver_eval(c, c.qacc_smooth, 0);
  double cost_ws = solver_eval(c, c.qacc_ws, &gauss);
  if (cost_ws > cost_sm) {
    PFOR(i, nv) c.qacc[i] = c.qacc_smooth[i];
    SYNC();
    cost = solver_eval(c, c.qacc, &gauss);
  } else {
    PFOR(i, nv) c.qacc[i] = c.qacc_ws[i];
    SYNC();
    cost = cost_ws;
  }
  PROF(c, 12);
  newton_gradient<NVT>(c);
  PFOR(i, nv) c.search[i] = -c.Mgrad[i];
  SYNC();
  double scale = 1.0 / (M.meaninertia * (nv > 1 ? nv : 1));
  for (int iter = 0; iter < M.iterations; iter++) {
    PROF(c, 13);
    double q1, q2;
    double alpha = line_search(c, gauss, &q1, &q2);
    PROF(c, 14);
    if (alpha == 0) break;
    PFOR(i, nv) { c.qacc[i] += alpha * c.search[i]; c.Ma[i] += alpha * c.Mv[i]; }
    PFOR(r, c.nefc) c.efc_jar[r] += alpha * c.efc_jv[r];
    SYNC();
    gauss = gauss + alpha * q1 + alpha * alpha * q2;
    double oldcost = cost;
    cost = gauss + wave_sum(constraint_update(c, 1));
    SYNC();
    PROF(c, 12);
    newton_gradient<NVT>(c);
    c.solver_iter++;
    double pg = 0;
    PFOR(i, nv) pg += c.grad[i] * c.grad[i];
    double gradient = scale * sqrt(wave_sum(pg));
    double improvement = scale * (oldcost - cost);
    if (improvement < M.tolerance || gradient < M.tolerance) break;
    PFOR(i, nv) c.search[i] = -c.Mgrad[i];
    SYNC();
  }
  if (LANE == 0) { c.misc[5] += c.solver_iter; if (c.ncon > c.misc[6]) c.misc[6] = c.ncon; if (c.nefc > c.misc[7]) c.misc[7] = c.nefc; }
  PFOR(i, nv) {
    double s = c.sgl[i] + c.sgl[2 * nv + i];
#pragma unroll 8
    for (int r = c.nsingle; r < c.nefc; r++) s += c.efc_J[r * nvp + i] * c.efc_force[r];
    c.qfrc_constraint[i] = s;
  }
  SYNC();
}
