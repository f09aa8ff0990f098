// solver.h — primal Newton solver for  min_a 1/2 (a-a0)^T M (a-a0) + sum_i s_i(J_i a - aref_i)
// (what MuJoCo's default solver computes inside mj_step, mjpc/trajectory.cc:158): one owner wavefront, optionally helped by
// the candidate's helper waves for the data-parallel parts of an iteration.
//
// Layout of the work over the 64 lanes of the owner wave:
//   * rows with a single +-1 Jacobian entry (friction loss, joint limits: [0, nsingle)) only touch the Hessian diagonal
//     (`sgl`); every active contact row becomes one scaled row  sum_b coef_b J[row_b]  (plus one negative row per contact in
//     the cone zone), so that  H = M + JH+^T JH+ - JH-^T JH-  is a plain contraction over M's sparsity pattern and the
//     gradient is its extra column (newton_fill / newton_entries);
//   * the exact line search keeps each lane's rows / contact in VGPRs for all its evaluations; one evaluation = branch-free
//     ALU + three wave reductions; the point alpha = 0 is analytic;
//   * Ma and jar are updated incrementally along the search direction; the factorisation of H never leaves the registers.
#pragma once
#include "linalg.h"

#ifdef MJPC_EMU
#define LS_RPL 192     // rows per lane (1 lane owns everything in the emulation build)
#define LS_CPL 64
#else
#define LS_RPL 3       // nefcmax <= 192
#define LS_CPL 1       // nconmax <= 64
#endif

// constraint cost at efc_jar; fills force/state (and the cone Hessian factors); returns this lane's partial cost
// WRITE = false: cost only, at the residuals `jar` (nothing shared is written: a helper wave prices qacc_smooth with it)
template <int DIMT, bool WRITE = true>
DEV double constraint_update(Ctx &c, int hess, const double *jar) {
  double cost = 0;
  const int nefc = c.nefc, ncon = c.ncon, nv = c.M->nv, nvp = c.M->nvp, stride = c.M->con_stride;
  const int nslot = (nefc + NLANE - 1) / NLANE, ncslot = (ncon + NLANE - 1) / NLANE;
  // All loads first (two dependent levels: row tables / contact header, then the entries they point to), then the
  // arithmetic, then the stores: the LDS latencies overlap instead of chaining through divergent branches.  Indices of
  // lanes without a row / contact are clamped and their results discarded by selects.
  int rtype[LS_RPL], rdof[LS_RPL];
  double rD[LS_RPL], rx[LS_RPL], rfl[LS_RPL], rR[LS_RPL], rJ[LS_RPL];
  int cdim[LS_CPL], crow[LS_CPL], ctype[LS_CPL];
  double cmu[LS_CPL], cfr[LS_CPL][DIMT], cX[LS_CPL][DIMT], cD[LS_CPL][DIMT];
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    if (k >= nslot) break;
    int i = LANE + NLANE * k, ic = i < nefc ? i : nefc - 1;
    rtype[k] = c.efc_type[ic]; rD[k] = c.efc_D[ic]; rx[k] = jar[ic]; rfl[k] = c.efc_floss[ic]; rR[k] = c.efc_R[ic];
    rdof[k] = c.efc_dof[ic];
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    if (q >= ncslot) break;
    int ci = LANE + NLANE * q, cic = ci < ncon ? ci : ncon - 1;
    cdim[q] = c.con_i[cic * CONI_STRIDE]; crow[q] = c.con_i[cic * CONI_STRIDE + 3];
  }
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    if (k >= nslot) break;
    int i = LANE + NLANE * k, ic = i < nefc ? i : nefc - 1;
    const int nfr = c.M->nfric;                     // friction-loss rows have no stored Jacobian row: their entry is 1
    rJ[k] = (WRITE && ic < c.nsingle) ? (ic < nfr ? 1.0 : c.efc_J[(ic < nfr ? nfr : ic) * nvp + rdof[k]]) : 0.0;
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    if (q >= ncslot) break;
    int ci = LANE + NLANE * q, cic = ci < ncon ? ci : ncon - 1;
    int i = crow[q];
    const double *cc = c.contact + cic * stride;
    ctype[q] = c.efc_type[i];
    cmu[q] = cc[CON_MU];
#pragma unroll
    for (int j = 0; j < DIMT; j++) {
      int rj = i + j < nefc ? i + j : nefc - 1;
      cfr[q][j] = j == 0 ? cmu[q] : cc[CON_FRICTION + j - 1];
      cX[q][j] = jar[rj]; cD[q][j] = c.efc_D[rj];
    }
  }
  // rows: branch-free cost (see the line search): xc = clamp(x, lo, hi), s = 1/2 D xc^2 + F (|x| - |xc|), force = -D xc
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    if (k >= nslot) break;
    int i = LANE + NLANE * k;
    int type = rtype[k];
    int act = i < nefc && type != CNSTR_CONTACT_ELLIPTIC;
    double D = rD[k], x = rx[k];
    int fric = type <= CNSTR_FRICTION_TENDON;
    double f = fric ? rfl[k] : 0.0, Rf = rR[k] * f;
    double lo = fric ? -Rf : -1e300, hi = fric ? Rf : 0.0;
    double xc = fmin(fmax(x, lo), hi);
    int inside = x > lo && x < hi;
    double s = 0.5 * D * xc * xc + f * (fabs(x) - fabs(xc));
    cost += act ? s : 0.0;
    if (WRITE && act) {
      double force = fric ? (inside ? -D * x : (x <= lo ? f : -f)) : -D * xc;
      c.efc_force[i] = force;
      c.efc_state[i] = inside ? STATE_QUADRATIC : (fric ? (x <= lo ? STATE_LINEARNEG : STATE_LINEARPOS) : STATE_SATISFIED);
      if (i < c.nsingle) {
        // rows with one +-1 Jacobian entry: fold J^T force and the Hessian diagonal per dof
        // (slot 0/1: friction loss, slot 2/3: joint limit; at most one active row of each kind per dof)
        int d = rdof[k], kk = fric ? 0 : 2;
        c.sgl[kk * nv + d] = rJ[k] * force;
        c.sgl[(kk + 1) * nv + d] = inside ? D : 0.0;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    if (q >= ncslot) break;
    int ci = LANE + NLANE * q;
    int dim = cdim[q], i = crow[q];
    if (ci >= ncon || dim <= 1 || ctype[q] != CNSTR_CONTACT_ELLIPTIC) continue;     // pyramidal edges are plain unilateral rows
    double *cc = c.contact + ci * stride;
    double mu = cmu[q], U[DIMT], fr[DIMT], X[DIMT], Dj[DIMT], F[DIMT];
    fr[0] = mu;
#pragma unroll
    for (int j = 1; j < DIMT; j++) fr[j] = j < dim ? cfr[q][j] : 0;
    double T2 = 0;
#pragma unroll
    for (int j = 0; j < DIMT; j++) {
      X[j] = j < dim ? cX[q][j] : 0; Dj[j] = j < dim ? cD[q][j] : 0;
      U[j] = X[j] * fr[j]; F[j] = 0;
      if (j > 0) T2 += U[j] * U[j];
    }
    double iT = fast_rsqrt(T2);
    double N = U[0], T = T2 * iT;
    int st;
    if (N >= mu * T || (T <= 0 && N >= 0)) {
      st = STATE_SATISFIED;
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
      for (int j = 0; j < DIMT; j++) { double dx = Dj[j] * X[j]; cost += 0.5 * dx * X[j]; F[j] = -dx; }
      st = STATE_QUADRATIC;
      if (WRITE && hess) {
        // the same record for the quadratic zone (solver_reg.h assembles every elliptic contact as P P^T - Q Q^T + diag(T^2) + w0 e0 e0^T
        // without looking at its zone): P = Q = 0, T_j = sqrt(D_j), w0 = D_0 in the slot of Q_0.  The scaled-row path never reads it here.
#pragma unroll
        for (int j = 0; j < DIMT; j++) if (j < dim) {
          cc[CON_H + j] = 0;
          cc[CON_H + 6 + j] = j == 0 ? Dj[0] : 0.0;
          cc[CON_H + 12 + j] = Dj[j] * fast_rsqrt(Dj[j]);
        }
      }
    } else {
      double Dm = Dj[0] * fast_rcp(mu * mu * (1 + mu * mu));
      double NmT = N - mu * T;
      cost += 0.5 * Dm * NmT * NmT;
      double f0 = -Dm * NmT * mu;
      F[0] = f0;
#pragma unroll
      for (int j = 1; j < DIMT; j++) F[j] = -f0 * iT * U[j] * fr[j];
      st = STATE_CONE;
      if (WRITE && hess) {
        // cone Hessian  S d2s/dU2 S  (S = diag(mu, friction), s = 1/2 Dm (N - mu T)^2)  in factored form:
        //   Dm p p^T + kap (diag(fr_t^2) - q q^T),  p_a = fr_a g_a,  q_t = fr_t U_t / T,  kap = -mu (N - mu T) Dm / T > 0
        // stored pre-scaled by sqrt(Dm) / sqrt(kap) so that the Hessian is a sum of +- outer products of combined rows
        double kap = -mu * NmT * Dm * iT;
        double sD = Dm * fast_rsqrt(Dm), sk = kap > 0 ? kap * fast_rsqrt(kap) : 0.0;
        cc[CON_H] = sD * fr[0];
        cc[CON_H + 6] = 0;
        cc[CON_H + 12] = -sD * NmT;                       // phi of the p row: J^T force of this contact = p * (-Dm (N - mu T))
#pragma unroll
        for (int j = 1; j < DIMT; j++) if (j < dim) {
          double u = U[j] * iT;
          cc[CON_H + j] = -sD * fr[j] * mu * u;
          cc[CON_H + 6 + j] = sk * fr[j] * u;
          cc[CON_H + 12 + j] = sk * fr[j];
        }
      }
    }
    if (WRITE) {
#pragma unroll
      for (int j = 0; j < DIMT; j++) if (j < dim) { c.efc_force[i + j] = F[j]; c.efc_state[i + j] = st; }
    }
  }
  return cost;
}
DEV double constraint_update_any(Ctx &c, int hess) {
  return c.M->maxdim <= 3 ? constraint_update<3>(c, hess, c.efc_jar) : constraint_update<6>(c, hess, c.efc_jar);
}

// y_i = M_i . x  (i < nv)  and  out_r = J_r . x  (r < nefc; single-entry rows use their one column).
// With a compile-time nv the vector x sits in registers (broadcast reads) and every row product is a fully unrolled
// chain of loads with immediate offsets: the LDS latency is paid once per row instead of once per element.
template <int NVT>
DEV void mat_rows_times(Ctx &c, const double *x, double *Mx, double *Jx) {
  const DevModel &M = *c.M;
  const int nv = NVT > 0 ? NVT : M.nv, nvp = NVT > 0 ? NVP_OF(NVT) : M.nvp;
  if constexpr (NVT > 0) {
    double xs[NVT];
#pragma unroll
    for (int j = 0; j < NVT; j++) xs[j] = x[j];
    PFOR(i, nv) {
      const double *row = c.qM + i * nvp;
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int j = 0; j + 1 < NVT; j += 2) { s0 += row[j] * xs[j]; s1 += row[j + 1] * xs[j + 1]; }
      if (NVT & 1) s0 += row[NVT - 1] * xs[NVT - 1];
      Mx[i] = s0 + s1;
    }
    // every row takes the dense product (a limit row holds its one entry, the rest are exact zeros); a friction-loss row has no
    // stored Jacobian: its lane multiplies the first stored row instead and the select picks x[dof]
    const int nfr = c.M->nfric;
    PFOR(r, c.nefc) {
      const double *row = c.efc_J + (r < nfr ? nfr : r) * nvp;
      int dof = c.efc_dof[r < nfr ? r : 0];
      double xd = x[r < nfr ? dof : 0];
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int j = 0; j + 1 < NVT; j += 2) { s0 += row[j] * xs[j]; s1 += row[j + 1] * xs[j + 1]; }
      if (NVT & 1) s0 += row[NVT - 1] * xs[NVT - 1];
      Jx[r] = r < nfr ? xd : s0 + s1;
    }
  } else {
    PFOR(i, nv) {
      double s = 0;
      for (int j = 0; j < nv; j++) s += c.qM[i * nvp + j] * x[j];
      Mx[i] = s;
    }
    PFOR(r, c.nefc) {
      double s = 0;
      if (r < M.nfric) s = x[c.efc_dof[r]];
      else
      if (r < c.nsingle) s = c.efc_J[r * nvp + c.efc_dof[r]] * x[c.efc_dof[r]];
      else for (int j = 0; j < nv; j++) s += c.efc_J[r * nvp + j] * x[j];
      Jx[r] = s;
    }
  }
}

// full evaluation at qacc: Ma = M qacc, jar = J qacc - aref, force/state; returns total cost (uniform)
template <int NVT>
DEV double solver_eval(Ctx &c, const double *qacc, double *gauss_out) {
  const DevModel &M = *c.M;
  int nv = M.nv;
  mat_rows_times<NVT>(c, qacc, c.Ma, c.efc_jar);
  double part = 0;
  PFOR(i, nv) part += 0.5 * (c.Ma[i] - c.qfrc_smooth[i]) * (qacc[i] - c.qacc_smooth[i]);
  PFOR(r, c.nefc) c.efc_jar[r] -= c.efc_aref[r];
  SYNC();
  double gauss = wave_sum(part);
  double cc = wave_sum(constraint_update_any(c, 1));
  SYNC();
  if (gauss_out) *gauss_out = gauss;
  return gauss + cc;
}

// gradient and Hessian (lower triangle); newton_direction() then solves Mgrad = H^-1 grad.
// Every active contact row contributes one scaled row  jh = sum_b coef_b J[row_b]  with  H += jh jh^T:
//   quadratic row r: sqrt(D_r) J_r;   cone contact: p-row (normal row's slot), sqrt(kap) fr_t J_t for its tangential rows,
// and each cone contact adds one NEGATIVE row q (H -= q q^T).  Column nv of a scaled row holds phi with
// J^T force = sum_rows jh * phi, so the gradient falls out of the same contraction as an extra "column".
// grad_only: the caller already knows it will stop (no cost improvement) and only needs grad = Ma - qfrc_smooth - J^T force.
// ordered lists: positives = active contact rows, negatives = normal rows of the contacts in the cone zone
DEV void newton_lists(Ctx &c, int *npos_out, int *nneg_out) {
  int nefc = c.nefc, ns = c.nsingle;
  int ncrow = nefc - ns;
  const int negbase = c.M->nefcmax;
  int npos = 0, nneg = 0;
  for (int base = 0; base < ncrow; base += NLANE) {
    int rr = base + LANE, r = ns + rr;
    int rc = rr < ncrow ? r : ns;
    int st = c.efc_state[rc], id = c.efc_id[rc];
    st = (rr < ncrow) ? st : STATE_SATISFIED;
    // a tendon friction row in one of its linear zones has a force but no curvature: it goes on BOTH lists (H += D J J^T - D J J^T)
    // and the gradient picks up J^T force from the phi column of the positive copy
    int lin = st == STATE_LINEARNEG || st == STATE_LINEARPOS;
    int fpos = st != STATE_SATISFIED, fneg = (st == STATE_CONE && EFC_CON_R0(id) == r) || lin;
    int tp, tn;
    int op = wave_flag_scan(fpos, &tp), on = wave_flag_scan(fneg, &tn);
    if (fpos) c.active[npos + op] = r;
    if (fneg) c.active[negbase + nneg + on] = r;
    npos += tp; nneg += tn;
  }
  SYNC();
  *npos_out = npos; *nneg_out = nneg;
}

// scaled rows JH[e][C0..C1) (+ the phi column when PHI): rows [0, npos8) positives zero-padded to a multiple of 8,
// rows [npos8, npos8 + nneg4) negatives padded to 4.  The column range lets two waves share the fill.
template <int NVT, int C0, int C1, int PHI, int DIMT>
DEV void newton_fill_d(Ctx &c, int npos, int nneg) {
  const DevModel &M = *c.M;
  const int nv = NVT > 0 ? NVT : M.nv, nvp = NVT > 0 ? NVP_OF(NVT) : M.nvp;
  const int negbase = M.nefcmax;
  const int npos8 = (npos + 7) & ~7, nneg4 = (nneg + 3) & ~3;
  int ntot = npos8 + nneg4;
  double *JH = c.efc_JA;
  for (int base = 0; base < ntot; base += NLANE) {
    int e = base + LANE;
    // branch-free descriptor: every lane issues the same loads (row, its contact, the contact's cone factors) and the four
    // cases (quadratic row / cone p-row / cone tangential row / negative q-row) are selected afterwards, so the dependent
    // LDS chain is walked once instead of once per divergent case
    double coef[DIMT]; int rowb[DIMT]; int nb = 0; double phi = 0;
    int neg = e >= npos8;
    int valid = neg ? (e - npos8 < nneg) : (e < npos);
    int r = valid ? (neg ? c.active[negbase + e - npos8] : c.active[e]) : c.nsingle;
    if (r >= c.nefc) r = c.nefc - 1;
    int st = c.efc_state[r], type = c.efc_type[r], id = c.efc_id[r];
    double D = c.efc_D[r], frc = c.efc_force[r];
    int is_con = type >= CNSTR_CONTACT_FRICTIONLESS;
    int ci = is_con ? EFC_CON_CI(id) : 0, dim = is_con ? EFC_CON_DIM(id) : 1, r0 = is_con ? EFC_CON_R0(id) : r;
    const double *cf = c.contact + ci * M.con_stride + (M.con_stride > CON_H ? CON_H : 0);
    double cp[DIMT], cq[DIMT], ct[DIMT];
#pragma unroll
    for (int b = 0; b < DIMT; b++) { cp[b] = cf[b]; cq[b] = cf[6 + b]; ct[b] = cf[12 + b]; }
    int cone = valid && st == STATE_CONE;
    int k = r - r0;
    double rs = fast_rsqrt(D), sd = D * rs;
    double tk = 0;
#pragma unroll
    for (int b = 1; b < DIMT; b++) tk = (k == b) ? ct[b] : tk;
    int caseP = cone && !neg && k == 0, caseT = cone && !neg && k > 0, caseQ = cone && neg, caseD = valid && !cone;
#pragma unroll
    for (int b = 0; b < DIMT; b++) {
      double cb = caseP ? (b < dim ? cp[b] : 0.0) : (caseQ ? ((b + 1 < dim && b + 1 < DIMT) ? cq[b + 1 < DIMT ? b + 1 : 0] : 0.0) : 0.0);
      int rb = caseP ? (b < dim ? r0 + b : r0) : (caseQ ? (b + 1 < dim ? r0 + b + 1 : r0) : r);
      if (b == 0) { cb = caseD ? sd : (caseT ? tk : cb); }
      coef[b] = cb; rowb[b] = rb * nvp;
    }
    nb = caseP ? dim : (caseQ ? dim - 1 : (valid ? 1 : 0));
    phi = caseD ? frc * rs : (caseP ? ct[0] : 0.0);      // sqrt(D) J * phi = J^T force (force = -D jar on a quadratic row)
    if constexpr (NVT > 0) {
      constexpr int NC = C1 - C0;
      double acc[NC > 0 ? NC : 1];
      // all the Jacobian rows of the combination are fetched before the first use (unused slots: coefficient 0 on a valid row)
      double Jv[DIMT][NC > 0 ? NC : 1];
#pragma unroll
      for (int b = 0; b < DIMT; b++) {
        const double *Jr = c.efc_J + rowb[b] + C0;
#pragma unroll
        for (int j = 0; j < NC; j++) Jv[b][j] = Jr[j];
      }
#pragma unroll
      for (int j = 0; j < NC; j++) acc[j] = coef[0] * Jv[0][j];
#pragma unroll
      for (int b = 1; b < DIMT; b++)
#pragma unroll
        for (int j = 0; j < NC; j++) acc[j] += coef[b] * Jv[b][j];
      if (e < ntot) {
        double *o = JH + e * nvp + C0;
#pragma unroll
        for (int j = 0; j < NC; j++) o[j] = valid ? acc[j] : 0.0;
        if (PHI) JH[e * nvp + NVT] = phi;
      }
    } else {
      if (e < ntot) {
        double *o = JH + e * nvp;
        for (int j = 0; j < nv; j++) {
          double a = 0;
#pragma unroll
          for (int b = 0; b < DIMT; b++) if (b < nb) a += coef[b] * c.efc_J[rowb[b] + j];
          o[j] = a;
        }
        o[nv] = phi;
      }
    }
  }
}

template <int NVT, int C0, int C1, int PHI>
DEV void newton_fill(Ctx &c, int npos, int nneg) {
  if (c.M->maxdim <= 3) newton_fill_d<NVT, C0, C1, PHI, 3>(c, npos, nneg);
  else newton_fill_d<NVT, C0, C1, PHI, 6>(c, npos, nneg);
}

// H = M + diag(single-entry rows) + JH+^T JH+ - JH-^T JH-  on the lower triangle (only M's sparsity pattern without
// cross-branch contacts), and grad = Ma - qfrc_smooth - J^T force  as the entries (i, nv).  Entries e = part*NLANE + LANE
// + g*NLANE*nparts belong to this wave; each lane carries G of them through the row loop together (G*16 independent LDS
// reads in flight per trip; the row counts are multiples of 8 / 4).
template <int NVT, int G>
DEV void newton_entries(Ctx &c, int npos, int nneg, int grad_only, int part, int nparts) {
  const DevModel &M = *c.M;
  const int nv = NVT > 0 ? NVT : M.nv, nvp = NVT > 0 ? NVP_OF(NVT) : M.nvp;
  const int npos8 = (npos + 7) & ~7, ntot = npos8 + ((nneg + 3) & ~3);
  const double *JH = c.efc_JA;
  int nh = c.cross ? nv * (nv + 1) / 2 : M.nhpair;
  int nent = nh + nv;
  const int stride = NLANE * nparts;
  for (int e0 = (grad_only ? nh : 0) + part * NLANE + LANE; e0 < nent; e0 += G * stride) {
    int ii[G], jj[G];
#pragma unroll
    for (int g = 0; g < G; g++) {
      int e = e0 + g * stride, i = 0, j = 0;
      if (e < nent) {
#ifdef MJPC_LEAN_LDS
        if (!c.cross) { i = MI(hpair_i)[e]; j = MI(hpair_j)[e]; }                 // entry table from HBM / L2 (no LDS copy in the dense tier)
#else
        if (!c.cross) { int pk = c.hpair[e]; i = pk & 255; j = pk >> 8; }
#endif
        else if (e >= nh) { i = e - nh; j = nv; }
        else {
          i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
          while ((i + 1) * (i + 2) / 2 <= e) i++;
          while (i * (i + 1) / 2 > e) i--;
          j = e - i * (i + 1) / 2;
        }
      }
      ii[g] = i; jj[g] = j;
    }
    double hp[G], hq[G], hn[G];
#pragma unroll
    for (int g = 0; g < G; g++) { hp[g] = 0; hq[g] = 0; hn[g] = 0; }
    for (int a = 0; a < npos8; a += 8) {
      double x[G][8], y[G][8];
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < 8; k++) { x[g][k] = JH[(a + k) * nvp + ii[g]]; y[g][k] = JH[(a + k) * nvp + jj[g]]; }
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < 8; k += 2) { hp[g] += x[g][k] * y[g][k]; hq[g] += x[g][k + 1] * y[g][k + 1]; }
    }
    for (int a = npos8; a < ntot; a += 4) {          // negative rows (their phi column is zero: gradient entries unaffected)
      double x[G][4], y[G][4];
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < 4; k++) { x[g][k] = JH[(a + k) * nvp + ii[g]]; y[g][k] = JH[(a + k) * nvp + jj[g]]; }
#pragma unroll
      for (int g = 0; g < G; g++)
#pragma unroll
        for (int k = 0; k < 4; k++) hn[g] += x[g][k] * y[g][k];
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
      int e = e0 + g * stride, i = ii[g], j = jj[g];
      if (e >= nent) continue;
      double h = hp[g] + hq[g];
      if (j == nv) {
        double gr = c.Ma[i] - c.qfrc_smooth[i] - (c.sgl[i] + c.sgl[2 * nv + i]) - h;
        c.grad[i] = gr;
        c.Mgrad[i] = gr;
      } else {
        h = c.qM[i * nvp + j] + (h - hn[g]);
        if (i == j) h += c.sgl[nv + i] + c.sgl[3 * nv + i];
        c.qH[i * nvp + j] = h;
      }
    }
  }
}

// ---- the solver's helper waves (MJPC_WAVES >= 3, compile-time nv): NH = MJPC_WAVES - 2 helpers share the scaled-row fill
// (column thirds / halves) and the Hessian / gradient entries with the owner wave.  Hand-shake through sequence numbers
// in LDS (misc[12..]):
//   owner:    lists -> publish npos/nneg -> post job seq -> fill its columns -> W0FILL=seq -> wait HFILL_k==seq (all k)
//             -> its entries -> wait HDONE_k==seq (all k)
//   helper k: wait job seq -> fill its columns -> HFILL_k=seq -> wait W0FILL==seq and the other helpers' HFILL -> its entries
//             -> HDONE_k=seq
#define HX_JOB 12
#define HX_KIND 13
#define HX_W0FILL 14
#define HX_NPOS 15
#define HX_NNEG 16
#define HX_HFILL 28      // + k (k < 8)
#define HX_HDONE 36      // + k
#ifndef MJPC_SPLIT_FILL
#define MJPC_SPLIT_FILL 1
#endif
#define HX_MFACT 22      // factor of M ready (helper 0 -> side wave), value t + 1

// column range of part p of NP for an NVT-wide row
#define FILL_C0(NVT, p, NP) ((NVT) * (p) / (NP))
#define FILL_C1(NVT, p, NP) ((NVT) * ((p) + 1) / (NP))

template <int NVT>
DEV void newton_gradient(Ctx &c, int grad_only) {
  const DevModel &M = *c.M;
  const int nv = NVT > 0 ? NVT : M.nv, nvp = NVT > 0 ? NVP_OF(NVT) : M.nvp;     // compile-time strides => immediate LDS offsets
  PROF(c, 13);
  int npos, nneg;
  newton_lists(c, &npos, &nneg);
  PROF(c, 15);
#if MJPC_HELPER
  if (NVT > 0 && !grad_only) {
    constexpr int NP = MJPC_NH + 1;
    int seq = ++c.hseq;
#if MJPC_SPLIT_FILL
    if (LANE == 0) { c.misc[HX_NPOS] = npos; c.misc[HX_NNEG] = nneg; c.misc[HX_KIND] = 1; }
    flag_set(c.misc + HX_JOB, seq);
    newton_fill<NVT, FILL_C0(NVT, 0, NP), FILL_C1(NVT, 0, NP), 0>(c, npos, nneg);
    flag_set(c.misc + HX_W0FILL, seq);
    PROF(c, 19);
    for (int k = 0; k < MJPC_NH; k++) if (!flag_wait(c.misc + HX_HFILL + k, seq)) c.warning |= WARN_SYNC;
    PROF(c, 18);
#else
    // the owner fills all the scaled rows, then posts the job: one hand-shake (entries done) per call instead of two
    newton_fill<NVT, 0, NVT, 1>(c, npos, nneg);
    if (LANE == 0) { c.misc[HX_NPOS] = npos; c.misc[HX_NNEG] = nneg; c.misc[HX_KIND] = 1; }
    flag_set(c.misc + HX_JOB, seq);
#endif
    newton_entries<NVT, (NP >= 3 ? 1 : 2)>(c, npos, nneg, 0, 0, NP);
    PROF(c, 9);
    for (int k = 0; k < MJPC_NH; k++) if (!flag_wait(c.misc + HX_HDONE + k, seq)) c.warning |= WARN_SYNC;
    PROF(c, 10);
  } else
#endif
  {
    newton_fill<NVT, 0, NVT, 1>(c, npos, nneg);
    PROF(c, 19);
    SYNC();
    newton_entries<NVT, 3>(c, npos, nneg, grad_only, 0, 1);
  }
  // structural zeros of the pattern: the register factorisation never writes qH, so they only need clearing after a
  // dense (cross-branch) build; the generic in-place LDS factor fills them every time
  if (c.cross) { if (LANE == 0) c.misc[9] = 1; }
  else if (NVT == 0 || uniform_i(c.misc[9])) {
    PFOR(e, M.nzpair) c.qH[MI(zpair_i)[e] * nvp + MI(zpair_j)[e]] = 0;
    if (LANE == 0) c.misc[9] = 0;
  }
  SYNC();
  PROF(c, 16);
}
// helper wave: total cost at qacc_smooth (Gauss term is exactly 0 there), same arithmetic as solver_eval; the residuals go
// to the scratch array efc_pos (consumed by make_impedance before the solve phase), nothing the owner uses is written
#define HX_CSM 23
template <int NVT>
DEV double cost_at_smooth(Ctx &c) {
  double *jar = c.efc_pos;
  mat_rows_times<NVT>(c, c.qacc_smooth, c.scr_b, jar);       // M x lands in a scratch nobody reads during the solve (recomputed every step)
  PFOR(r, c.nefc) jar[r] -= c.efc_aref[r];
  SYNC();
  double part = c.M->maxdim <= 3 ? constraint_update<3, false>(c, 0, jar) : constraint_update<6, false>(c, 0, jar);
  return 0.0 + wave_sum(part);
}

#ifndef MJPC_SOLVER_REG
#define MJPC_SOLVER_REG 1      // compile-time nv: the owner wave solves alone with the Hessian in registers (solver_reg.h); 0 = scaled-row tables shared with the helper waves
#endif
#if MJPC_HELPER
template <int NVT> DEV void worker_loop(Ctx &c, int W, int last);          // solver_reg.h
template <int NVT> DEV void ls_records_build(Ctx &c);
#define HX_LSREC 17      // line-search records of this step ready (helper 0 -> owner), value t + 1
template <int NVT, int K>
DEV void solver_helper_loop(Ctx &c, int seq) {
  if (K == MJPC_NH - 1 && c.nefc > 0) {            // the last helper prices the unconstrained acceleration for the warm-start choice
    double cs = cost_at_smooth<NVT>(c);
    if (LANE == 0) c.red[2] = cs;
    // fault injection for the test-suite (fault = 1): the helper of candidate 1 never reports this price in step 2
    const int mute_csm = MJPC_SOLVER_REG && c.K->fault == 1 && cand_index() == 1 && seq == 2 * 256;
    if (!mute_csm) flag_set(c.misc + HX_CSM, seq / 256 + 1);
  }
  if constexpr (NVT > 0 && MJPC_SOLVER_REG) {
    if (K == 0 && c.nefc > 0) { ls_records_build<NVT>(c); flag_set(c.misc + HX_LSREC, seq / 256 + 1); }     // the line search's per-step constants
    worker_loop<NVT>(c, K, seq);
  }
  if constexpr (NVT > 0 && !MJPC_SOLVER_REG) {
    constexpr int NP = MJPC_NH + 1;
    // fault injection for the test-suite (diagnostics knob fault_inject = sync, mjpc_hip_debug.h): helper 0 of candidate 1 never reports its fill in step 2
    const int mute = c.K->fault == 1 && K == 0 && cand_index() == 1 && seq == 2 * 256;
    for (;;) {
      seq++;
      if (!flag_wait(c.misc + HX_JOB, seq)) return;             // timed out: the owner reports the failure
      if (uniform_i(c.misc[HX_KIND]) == 0) return;
      int npos = uniform_i(c.misc[HX_NPOS]), nneg = uniform_i(c.misc[HX_NNEG]);
#if MJPC_SPLIT_FILL
      newton_fill<NVT, FILL_C0(NVT, K + 1, NP), FILL_C1(NVT, K + 1, NP), (K + 1 == NP - 1)>(c, npos, nneg);
      if (!mute) flag_set(c.misc + HX_HFILL + K, seq);
      if (!flag_wait(c.misc + HX_W0FILL, seq)) return;
      for (int k = 0; k < MJPC_NH; k++) if (k != K && !flag_wait(c.misc + HX_HFILL + k, seq)) return;
#endif
      newton_entries<NVT, (NP >= 3 ? 1 : 2)>(c, npos, nneg, 0, K + 1, NP);
      flag_set(c.misc + HX_HDONE + K, seq);
    }
  }
}
#endif
template <int NVT>
DEV void newton_direction(Ctx &c) {
  const int nv = NVT > 0 ? NVT : c.M->nv, nvp = NVT > 0 ? NVP_OF(NVT) : c.M->nvp;
  chol_factor_solve<NVT>(c.qH, c.Hinv, c.vtmp, c.Mgrad, nv, nvp, c.M->tree_ok && !c.cross);
  PROF(c, 17);
}
#ifndef MJPC_EMU
// the same with the Hessian given as qH + partial matrices (the worker waves' cone blocks): summed while the rows are loaded
template <int NVT>
DEV void newton_direction_sum(Ctx &c, const LDLExtra &ex) {
  chol_factor_solve_reg<NVT>(c.qH, c.Mgrad, NVP_OF(NVT), c.M->tree_ok && !c.cross, &ex);
  PROF(c, 17);
}
#endif

// ---- exact line search: phi(alpha) = Gauss(alpha) + sum_i s_i(jar + alpha*jv), data in registers
// Row costs in one branch-free form: with xc = clamp(x, lo, hi),
//   s(x) = 1/2 D xc^2 + F (|x| - |xc|),  s' = D xc,  s'' = D inside (lo, hi)
// (friction loss: lo/hi = -+R f, F = f;  unilateral rows: lo = -inf, hi = 0, F = 0;  unused slots: D = F = 0).
struct LSPoint { double cost, d1, d2; };
template <int DIMT>
struct LSData {
  double lo[LS_RPL], hi[LS_RPL], hD[LS_RPL], F[LS_RPL], X[LS_RPL], V[LS_RPL], DV[LS_RPL], DVV[LS_RPL];
  double U0[LS_CPL][DIMT], UV[LS_CPL][DIMT], E[LS_CPL][DIMT], mu[LS_CPL], Dm[LS_CPL], VV[LS_CPL];
  int on[LS_CPL];
  int nslot, ncslot;
};

template <int DIMT>
DEV void ls_load(Ctx &c, LSData<DIMT> &d) {
  const int nefc = c.nefc, ncon = c.ncon, last = c.nefc - 1;
  d.nslot = (nefc + NLANE - 1) / NLANE; d.ncslot = (ncon + NLANE - 1) / NLANE;
  // unconditional loads with clamped indices, selects afterwards (no LDS latency inside divergent branches)
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    d.lo[k] = -1; d.hi[k] = 1; d.hD[k] = 0; d.F[k] = 0; d.X[k] = 0; d.V[k] = 0; d.DV[k] = 0; d.DVV[k] = 0;
    if (k < d.nslot) {
      int r = LANE + NLANE * k, rc = r < nefc ? r : last;
      int type = c.efc_type[rc];
      double D = c.efc_D[rc], v = c.efc_jv[rc], x = c.efc_jar[rc], f = c.efc_floss[rc], Rf = c.efc_R[rc] * f;
      int quad = r < nefc && type != CNSTR_CONTACT_ELLIPTIC;
      int fric = quad && type <= CNSTR_FRICTION_TENDON;
      D = quad ? D : 0.0; v = quad ? v : 0.0;
      d.X[k] = quad ? x : 0.0; d.V[k] = v; d.hD[k] = 0.5 * D; d.DV[k] = D * v; d.DVV[k] = D * v * v;
      d.lo[k] = fric ? -Rf : (quad ? -1e300 : -1.0); d.hi[k] = fric ? Rf : (quad ? 0.0 : 1.0); d.F[k] = fric ? f : 0.0;
    }
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    d.on[q] = 0; d.mu[q] = 0; d.Dm[q] = 0; d.VV[q] = 0;
#pragma unroll
    for (int j = 0; j < DIMT; j++) { d.U0[q][j] = 0; d.UV[q][j] = 0; d.E[q][j] = 0; }
    if (q < d.ncslot) {
      int ci = LANE + NLANE * q, cic = ci < ncon ? ci : ncon - 1;
      int dim = c.con_i[cic * CONI_STRIDE];
      int i = c.con_i[cic * CONI_STRIDE + 3];
      const double *cc = c.contact + cic * c.M->con_stride;
      int type = c.efc_type[i];
      double mu = cc[CON_MU], fr[DIMT], Dj[DIMT], jv[DIMT], jr[DIMT];
#pragma unroll
      for (int j = 0; j < DIMT; j++) {
        int rj = i + j < nefc ? i + j : last;
        fr[j] = j == 0 ? mu : cc[CON_FRICTION + j - 1];
        Dj[j] = c.efc_D[rj]; jv[j] = c.efc_jv[rj]; jr[j] = c.efc_jar[rj];
      }
      int on = ci < ncon && dim > 1 && type == CNSTR_CONTACT_ELLIPTIC;
      d.on[q] = on; d.mu[q] = on ? mu : 0.0;
      d.Dm[q] = on ? Dj[0] * fast_rcp(mu * mu * (1 + mu * mu)) : 0.0;
      double vv = 0;
#pragma unroll
      for (int j = 0; j < DIMT; j++) {
        int use = on && j < dim;
        double uv = jv[j] * fr[j];
        d.U0[q][j] = use ? jr[j] * fr[j] : 0.0; d.UV[q][j] = use ? uv : 0.0;
        d.E[q][j] = use ? Dj[j] * fast_rcp(fr[j] * fr[j]) : 0.0;     // D_j jar_j^2 = E_j U_j^2
        if (j > 0) vv += use ? uv * uv : 0.0;
      }
      d.VV[q] = vv;
    }
  }
}

template <int DIMT>
DEV LSPoint ls_eval(const LSData<DIMT> &d, double q0, double q1, double q2, double a) {
  LSPoint p; p.cost = 0; p.d1 = 0; p.d2 = 0;
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    if (k >= d.nslot) break;
    double x = d.X[k] + a * d.V[k];
    double xc = fmin(fmax(x, d.lo[k]), d.hi[k]);
    p.cost += d.hD[k] * xc * xc + d.F[k] * (fabs(x) - fabs(xc));
    p.d1 += d.DV[k] * xc;
    p.d2 += (x > d.lo[k] && x < d.hi[k]) ? d.DVV[k] : 0.0;
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    if (q >= d.ncslot) break;
    if (!d.on[q]) continue;
    double mu = d.mu[q], U[DIMT];
    double T2 = 0, UV = 0;
#pragma unroll
    for (int j = 0; j < DIMT; j++) {
      U[j] = d.U0[q][j] + a * d.UV[q][j];
      if (j > 0) { T2 += U[j] * U[j]; UV += U[j] * d.UV[q][j]; }
    }
    double iT = fast_rsqrt(T2);            // 1/T without an IEEE divide + sqrt on the critical path
    double N = U[0], T = T2 * iT;
    if (N >= mu * T || (T <= 0 && N >= 0)) {
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
      for (int j = 0; j < DIMT; j++) { double E = d.E[q][j], vj = d.UV[q][j], eu = E * U[j]; p.cost += 0.5 * eu * U[j]; p.d1 += eu * vj; p.d2 += E * vj * vj; }
    } else {
      double Dm = d.Dm[q], NmT = N - mu * T;
      double T1 = UV * iT, T2d = (d.VV[q] - T1 * T1) * iT;
      double g1 = d.UV[q][0] - mu * T1;
      double dn = Dm * NmT;
      p.cost += 0.5 * dn * NmT; p.d1 += dn * g1; p.d2 += Dm * g1 * g1 - dn * mu * T2d;
    }
  }
  wave_sum3(p.cost, p.d1, p.d2);
  p.cost = p.cost + q0 + a * q1 + a * a * q2;
  p.d1 = p.d1 + q1 + 2 * a * q2;
  p.d2 = p.d2 + 2 * q2;
  return p;
}

// returns alpha; q1/q2 = Gauss quadratic coefficients along the direction (for the incremental update)
template <int NVT, int DIMT>
DEV double line_search(Ctx &c, double gauss, double cost0, double *q1_out, double *q2_out) {
  const DevModel &M = *c.M;
  int nv = M.nv;
  double p_sn = 0, p_q1 = 0, p_q2 = 0, p_gs = 0;
  mat_rows_times<NVT>(c, c.search, c.Mv, c.efc_jv);
  PFOR(i, nv) {
    double si = c.search[i];
    p_sn += si * si; p_q1 += si * (c.Ma[i] - c.qfrc_smooth[i]); p_q2 += 0.5 * si * c.Mv[i]; p_gs += c.grad[i] * si;
  }
  SYNC();
  wave_sum4(p_sn, p_q1, p_q2, p_gs);
  double snorm = sqrt(p_sn), q1 = p_q1, q2 = p_q2, gs = p_gs;
  *q1_out = q1; *q2_out = q2;
  PROF(c, 20);
  double scale = 1.0 / (M.meaninertia * (nv > 1 ? nv : 1));
  if (snorm < D_MINVAL) return 0;
  double gtol = M.tolerance * M.ls_tolerance * snorm / scale;
  LSData<DIMT> d;
  ls_load<DIMT>(c, d);
  PROF(c, 21);
  // the point alpha = 0 needs no evaluation: search = -H^-1 grad with H the exact Hessian there, so
  // phi(0) = cost, phi'(0) = grad.search = -phi''(0), and the Newton step from alpha = 0 is exactly 1
  LSPoint p0; p0.cost = cost0; p0.d1 = gs; p0.d2 = -gs;
  if (p0.d1 >= 0) return 0;
  // safeguarded Newton on phi'(alpha) (rtsafe): expand until phi' changes sign, then Newton steps that stay
  // inside the bracket and at least halve the previous step, else bisection; return the best point seen
  double lo = 0, hi = -1, a = 1.0;
  double best_a = 0, best_cost = p0.cost, dxold = a, dx = a;
  for (int it = 0; it < M.ls_iterations; it++) {
    LSPoint p = ls_eval<DIMT>(d, gauss, q1, q2, a);
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
    if (LANE == 0) c.prof[23] += 1;
#endif
    // the bookkeeping is written with selects and a single exit test: every quantity here is wave-uniform, and a uniform
    // branch on a VALU comparison costs a VALU -> SALU round trip each
    int better = p.cost < best_cost;
    best_cost = better ? p.cost : best_cost; best_a = better ? a : best_a;
    int conv = fabs(p.d1) < gtol;
    int neg = p.d1 < 0;
    lo = neg ? a : lo; hi = neg ? hi : a;
    int pos2 = p.d2 > 0;
    double newton = a - p.d1 * fast_rcp(p.d2);
    // not bracketed yet: Newton step if it moves forward, else double the step
    double an_e = pos2 ? newton : 2 * a;
    an_e = (an_e > a) ? an_e : 2 * a;
    // bracketed: Newton step if it stays inside and at least halves the previous step, else bisection
    double nw = pos2 ? newton : lo - 1;
    int ok = (nw > lo) && (nw < hi) && (fabs(2 * p.d1) <= fabs(dxold * p.d2));
    double dx_b = ok ? fabs(nw - a) : 0.5 * (hi - lo);
    double an_b = ok ? nw : lo + dx_b;
    int bracketed = !(hi < 0);
    double an = bracketed ? an_b : an_e;
    dxold = dx; dx = bracketed ? dx_b : an_e - a;
    if (conv || an == a) break;
    a = an;
  }
  PROF(c, 22);
  return best_a;
}

#include "solver_reg.h"

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
#if !defined(MJPC_EMU) && MJPC_SOLVER_REG
  if constexpr (NVT > 0) { solve_constraints_reg<NVT>(c); return; }
#endif
  PROF(c, 7);
  // warm start: the better of qacc_smooth and qacc_warmstart (evaluated last, so its force/state stay valid)
  double gauss, cost;
  double cost_sm;
#if MJPC_HELPER
  double cost_ws = solver_eval<NVT>(c, c.qacc_ws, &gauss);
  if (!flag_wait(c.misc + HX_CSM, c.hseq / 256 + 1)) c.warning |= WARN_SYNC;     // priced by the last helper meanwhile
  cost_sm = c.red[2];
#else
  cost_sm = solver_eval<NVT>(c, c.qacc_smooth, 0);
  double cost_ws = solver_eval<NVT>(c, c.qacc_ws, &gauss);
#endif
  if (cost_ws > cost_sm) {
    PFOR(i, nv) c.qacc[i] = c.qacc_smooth[i];
    SYNC();
    cost = solver_eval<NVT>(c, c.qacc, &gauss);
  } else {
    PFOR(i, nv) c.qacc[i] = c.qacc_ws[i];
    SYNC();
    cost = cost_ws;
  }
  PROF(c, 12);
  newton_gradient<NVT>(c, 0);
  newton_direction<NVT>(c);
  PFOR(i, nv) c.search[i] = -c.Mgrad[i];
  SYNC();
  double scale = 1.0 / (M.meaninertia * (nv > 1 ? nv : 1));
  for (int iter = 0; iter < M.iterations; iter++) {
    PROF(c, 13);
    double q1, q2;
    double alpha = (M.maxdim <= 3) ? line_search<NVT, 3>(c, gauss, cost, &q1, &q2) : line_search<NVT, 6>(c, gauss, cost, &q1, &q2);
    PROF(c, 14);
    if (alpha == 0) break;
    PFOR(i, nv) { c.qacc[i] += alpha * c.search[i]; c.Ma[i] += alpha * c.Mv[i]; }
    PFOR(r, c.nefc) c.efc_jar[r] += alpha * c.efc_jv[r];
    SYNC();
    gauss = gauss + alpha * q1 + alpha * alpha * q2;
    double oldcost = cost;
    cost = gauss + wave_sum(constraint_update_any(c, 1));
    SYNC();
    PROF(c, 12);
    // same stopping rule as the reference's Newton loop; the Hessian build / factorisation of an iteration that is
    // about to stop is skipped (its direction would never be used)
    double improvement = scale * (oldcost - cost);
    int stop = improvement < M.tolerance || (c.warning & WARN_SYNC) != 0;      // a lost hand-shake ends the solve (the candidate fails)
    newton_gradient<NVT>(c, stop);
    c.solver_iter++;
    double pg = 0;
    PFOR(i, nv) pg += c.grad[i] * c.grad[i];
    double gradient = scale * sqrt(wave_sum(pg));
    if (stop || gradient < M.tolerance) break;
    newton_direction<NVT>(c);
    PFOR(i, nv) c.search[i] = -c.Mgrad[i];
    SYNC();
  }
  if (LANE == 0) { c.misc[5] += c.solver_iter; if (c.ncon > c.misc[6]) c.misc[6] = c.ncon; if (c.nefc > c.misc[7]) c.misc[7] = c.nefc; }
  // J^T force of the final state: the last newton_gradient() evaluated grad = Ma - qfrc_smooth - J^T force there
  PFOR(i, nv) c.qfrc_constraint[i] = (c.Ma[i] - c.qfrc_smooth[i]) - c.grad[i];
  SYNC();
}
