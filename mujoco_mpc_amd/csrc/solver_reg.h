// solver_reg.h — the Newton solve for a compile-time dof count (gfx950 only), run by the owner wave alone.
//
// Same minimisation as solver.h (MuJoCo's primal Newton inside mj_step, mjpc/trajectory.cc:158), organised around what one
// wavefront keeps in registers instead of around scaled-row tables in LDS:
//   * Hessian blocks: lane (i, g) — i = lane % nv, g = lane / nv < G = 64 / nv — owns H[i][g*CB .. g*CB + CB) (CB = ceil(nv / G):
//     6 columns for the A1, 14 for the humanoid, 33 for the hand).  `hq` = M + sum over the general rows in their quadratic zone of
//     D_r J_r^T J_r lives there for the whole solve and is UPDATED by the rows whose zone changed in a line search (one rank-1 term
//     each), never rebuilt.  Elliptic contacts are not quadratic in the cone zone: their dim x dim blocks (cone: P P^T - Q Q^T +
//     diag(T^2) in the factored form constraint_update() leaves behind; quadratic zone: diag(D)) are added to a copy of hq every
//     iteration, the copy goes to qH in LDS and the register factorisation (linalg.h) picks its rows up from there.
//   * the exact line search keeps every row / contact of a lane in registers (as before) and its last evaluation IS the
//     constraint update: residuals, forces, zones and cone factors are written from those registers (ls_commit), no second pass.
//   * gradient = Ma - qfrc_smooth - J^T force straight from the forces (lane (i, g) sums every G-th row).
// Pyramidal models: no other wave is involved.  Elliptic models: what an iteration has of work that is parallel over rows and
// contacts (J^T force, the contacts' cone blocks) is one job per iterate for the candidate's other waves (worker_job below).
#pragma once
#ifndef MJPC_EMU
#ifndef LS_PREDICT
#define LS_PREDICT 1
#endif

template <int NVT> struct HLay {
  static constexpr int G0 = 64 / NVT;
  static constexpr int G = G0 < 1 ? 1 : (G0 > NVT ? NVT : G0);     // column groups
  static constexpr int CB = (NVT + G - 1) / G;                      // columns per group (the last group may run into the spare column)
};
DEV int readlane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
#define LSR_STRIDE 7
#define LSC_STRIDE 17
#define LS_ROWREC(c) ((c).efc_JA + 64)
#define LS_CONREC(c) ((c).efc_JA + 64 + (c).M->nefcmax * LSR_STRIDE)

// h += sum over the set bits b of mask:  w(lane b) * J[rbase + b][hi] * J[rbase + b][j0 .. j0 + CB)      (two rows per trip)
template <int NVT>
DEV void hblock_add_rows(const Ctx &c, double *h, unsigned long long mask, int rbase, double wlane, int hi, int j0) {
  constexpr int nvp = NVP_OF(NVT), CB = HLay<NVT>::CB;
  while (mask) {
    int b0 = (int)__builtin_ctzll(mask); mask &= mask - 1;
    int b1 = b0; double sel1 = 0.0;
    if (mask) { b1 = (int)__builtin_ctzll(mask); mask &= mask - 1; sel1 = 1.0; }
    double w0 = readlane_d(wlane, b0), w1 = readlane_d(wlane, b1) * sel1;
    const double *r0p = c.efc_J + (rbase + b0) * nvp, *r1p = c.efc_J + (rbase + b1) * nvp;
    double x0 = r0p[hi], x1 = r1p[hi];
    double y0[CB], y1[CB];
#pragma unroll
    for (int q = 0; q < CB; q++) { y0[q] = r0p[j0 + q]; y1[q] = r1p[j0 + q]; }
    __builtin_amdgcn_sched_barrier(0);
#ifdef MJPC_LEAN_LDS
    if constexpr (HLay<NVT>::G == 1) {
      // one column group in the dense tier: the factorisation takes lane i's registers as row i, i.e. it reads entry (i, j), j > i,
      // where the full-capacity kernel (through qH, lower triangle) reads lane j's entry (j, i) = (w x_j) x_i.  Same bits here, so
      // that the capacity tiers stay bit-identical: above the diagonal the product is formed the way lane j forms it
      double s0 = w0 * x0, s1 = w1 * x1;
#pragma unroll
      for (int q = 0; q < CB; q++) {
        const bool up = j0 + q > hi;
        h[q] = __builtin_fma(up ? w0 * y0[q] : s0, up ? x0 : y0[q], h[q]);
        h[q] = __builtin_fma(up ? w1 * y1[q] : s1, up ? x1 : y1[q], h[q]);
      }
    } else
#endif
    {
      double s0 = w0 * x0, s1 = w1 * x1;
#pragma unroll
      for (int q = 0; q < CB; q++) { h[q] = __builtin_fma(s0, y0[q], h[q]); h[q] = __builtin_fma(s1, y1[q], h[q]); }
    }
  }
}

// hq = M + sum of D_r J_r^T J_r over the general (several Jacobian entries) rows that are in their quadratic zone - also the rows
// of elliptic contacts: in its quadratic zone a contact's block is diag(D); only the cone (middle) zone needs the workers' job
template <int NVT>
DEV void hblock_init(const Ctx &c, double *hq, int hi, int j0) {
  constexpr int nvp = NVP_OF(NVT), CB = HLay<NVT>::CB;
#pragma unroll
  for (int q = 0; q < CB; q++) hq[q] = c.qM[hi * nvp + j0 + q];
  const int nefc = c.nefc, ns = c.nsingle;
  for (int base = 0; base < nefc; base += NLANE) {
    int r = base + LANE, rc = r < nefc ? r : nefc - 1;
    int st = c.efc_state[rc], type = c.efc_type[rc];
    double D = c.efc_D[rc];
    int flag = r < nefc && r >= ns && st == STATE_QUADRATIC;      // (the rows of an elliptic contact in its quadratic zone are plain D rows)
    (void)type;
    unsigned long long mask = __builtin_amdgcn_ballot_w64(flag != 0);
    hblock_add_rows<NVT>(c, hq, mask, base, D, hi, j0);
  }
}

// grad = Ma - qfrc_smooth - J^T force (also copied to Mgrad for the solve); single-entry rows through their per-dof folds (sgl)
template <int NVT>
DEV void newton_grad_reg(Ctx &c, int hi, int hg, bool hact) {
  constexpr int nvp = NVP_OF(NVT), G = HLay<NVT>::G;
  const int ns = c.nsingle, n = c.nefc - ns;
  double part = 0;
  const int trips = (n + G - 1) / G;
  for (int t0 = 0; t0 < trips; t0 += 8) {
    double f[8], j[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int rr = (t0 + u) * G + hg;
      int r = ns + (rr < n ? rr : 0);
      f[u] = c.efc_force[r]; j[u] = c.efc_J[r * nvp + hi];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; u++) { int rr = (t0 + u) * G + hg; double pr = j[u] * f[u]; part += (rr < n) ? pr : 0.0; }
    __builtin_amdgcn_sched_barrier(0);
  }
  double *sc = c.efc_JA;                   // dead during this solve: the scaled-row tables belong to the generic path
  if (hact) sc[hg * NVT + hi] = part;
  SYNC();
  if (LANE < NVT) {
    int i = LANE;
    double jtf = 0;
#pragma unroll
    for (int g = 0; g < G; g++) jtf += sc[g * NVT + i];
    double gr = c.Ma[i] - c.qfrc_smooth[i] - (c.sgl[i] + c.sgl[2 * NVT + i]) - jtf;
    c.grad[i] = gr;
    c.Mgrad[i] = gr;
  }
  SYNC();
}

// ---- the worker waves' job on an iterate (elliptic models): the contacts' cone blocks ---------------------------------------
// H gets  J_c^T (P P^T - Q Q^T + diag(T^2) + w0 e0 e0^T) J_c  from every elliptic contact c that is not in its satisfied zone (the
// record constraint_update() / ls_commit() left at CON_H; quadratic zone: P = Q = 0, T_j^2 = D_j, w0 = D_0 in Q_0's slot, so no
// branch on the zone).  J_c only has entries on the dofs that move the contact's two bodies (n_c = 9 of 18 for a foot of the A1
// on the floor), so the block is built where it is non-zero, and by as many lanes as it has ROWS: ls_records_build lists, once
// per step, the pairs (contact, local row a) of all elliptic contacts (CONE_ITEMS); a lane takes one pair, forms
// t = B_c J_c[:, dof_a] once and walks the columns b <= a of its row, adding  t . J_c[:, dof_b]  to entry (dof_a, dof_b) of the
// worker's partial Hessian with an LDS fp64 atomic add (ds_add_f64: rows of different contacts meet in the entries of their common
// dofs).  Worker w of np takes the passes w, w + np, ... of 64 pairs.  The owner posts ONE packed word (sequence number, number of
// workers, kind), meanwhile updates its register blocks, builds the gradient and stores its own part of H.
//   workers 0 .. MJPC_NH-1 = the helper waves (in every job); worker MJPC_NH = the side wave from job MJPC_SIDE_JOB of a step on.
// kind: 0 release (solve over) | 1 cone blocks
#define HX_JOBW 18
#define HX_SIDEFROM 21
#define HX_NITEMS 45
#define HX_WDONE HX_HDONE
#define JOBW(seq, np, kind) (((seq) << 4) | ((np) << 2) | (kind))
#define MJPC_NW (MJPC_NH + 1)
// behind the line-search records: the contacts' dof lists (one byte per dof, nv per contact), the (contact, row) pairs (two bytes
// each), then one partial Hessian per worker (nv x nvp each, lower triangle).  The owner never adds them up: its own part goes to
// qH and the factorisation sums qH and the partials while it loads its rows (ldl_load_row)
#define CONE_BASE(c) ((c).efc_JA + 64 + (c).M->nefcmax * LSR_STRIDE + (c).M->nconmax * LSC_STRIDE)
#define CONE_DOFS(c) ((unsigned char *)CONE_BASE(c))
#define CONE_ITEMS(c) ((unsigned short *)(CONE_BASE(c) + ((c).M->nconmax * NVT + 7) / 8))
#define CONE_PARTIAL(c, w) (CONE_BASE(c) + ((c).M->nconmax * NVT + 7) / 8 + ((c).M->nconmax * NVT + 3) / 4 + (w) * (NVT * NVP_OF(NVT)))

// the passes p0, p0 + pstep, ... of 64 (contact, row) pairs into the partial `part` (zeroed by the caller)
template <int NVT, int DIMT>
DEV void cone_rows(const Ctx &c, double *part, int p0, int pstep) {
  constexpr int nvp = NVP_OF(NVT);
  const int nitems = uniform_i(c.misc[HX_NITEMS]), stride = c.M->con_stride;
  const unsigned short *items = CONE_ITEMS(c);
  const double *crec = LS_CONREC(c);
  for (int it0 = p0 * NLANE; it0 < nitems; it0 += pstep * NLANE) {
    const int it = it0 + LANE;
    const bool valid = it < nitems;
    const int item = items[valid ? it : 0];
    const int ci = item & 255, a = item >> 8;
    const int info = ((const int *)(crec + ci * LSC_STRIDE + 14))[0];        // on | dim << 8 | first row << 16
    const int r0 = info >> 16, dim = (info >> 8) & 255;
    const unsigned char *dofs = CONE_DOFS(c) + ci * NVT;
    const int i = dofs[a];
    const int st = c.efc_state[r0];
    const double *cf = c.contact + ci * stride + CON_H;
    double P[DIMT], Q[DIMT], T[DIMT], ji[DIMT];
#pragma unroll
    for (int k = 0; k < DIMT; k++) { P[k] = cf[k]; Q[k] = cf[6 + k]; T[k] = cf[12 + k]; }
    const double *Jr = c.efc_J + r0 * nvp;
#pragma unroll
    for (int k = 0; k < DIMT; k++) ji[k] = Jr[(DIMT <= 3 || k < dim ? k : 0) * nvp + i];
    __builtin_amdgcn_sched_barrier(0);
    const bool active = valid && st == STATE_CONE;
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
    if (MJPC_PROFILE_WAVE != 0 && WAVE_ID() == MJPC_PROFILE_WAVE) { int na_ = __builtin_popcountll(__builtin_amdgcn_ballot_w64(active)); if (LANE == 0) { c.prof[23] += na_; c.prof[22] += (it0 == p0 * NLANE); } }      // active rows / jobs of this worker (diagnostics)
#endif
    if constexpr (DIMT > 3) {
#pragma unroll
      for (int k = 3; k < DIMT; k++) { const bool in = k < dim; P[k] = in ? P[k] : 0.0; Q[k] = in ? Q[k] : 0.0; T[k] = in ? T[k] : 0.0; ji[k] = in ? ji[k] : 0.0; }
    }
    double pi = P[0] * ji[0], qi = Q[1] * ji[1];
#pragma unroll
    for (int k = 1; k < DIMT; k++) { pi += P[k] * ji[k]; if (k > 1) qi += Q[k] * ji[k]; }
    double t[DIMT];
    t[0] = P[0] * pi + Q[0] * ji[0];                      // Q_0's slot: 0 in the cone zone, D_0 in the quadratic zone
#pragma unroll
    for (int k = 1; k < DIMT; k++) t[k] = P[k] * pi - Q[k] * qi + (T[k] * T[k]) * ji[k];
    double *prow = part + i * nvp;
    PROFW(c, 16);
    // columns b <= a of the lane's row, two per trip
    for (int b = 0; __builtin_amdgcn_ballot_w64(active && b <= a) != 0; b += 2) {
      const bool on0 = active && b <= a, on1 = active && b + 1 <= a;
      const int j0 = dofs[on0 ? b : 0], j1 = dofs[on1 ? b + 1 : 0];
      double x0[DIMT], x1[DIMT];
#pragma unroll
      for (int k = 0; k < DIMT; k++) { x0[k] = Jr[(DIMT <= 3 || k < dim ? k : 0) * nvp + j0]; x1[k] = Jr[(DIMT <= 3 || k < dim ? k : 0) * nvp + j1]; }
      __builtin_amdgcn_sched_barrier(0);
      double v0 = t[0] * x0[0], v1 = t[0] * x1[0];
#pragma unroll
      for (int k = 1; k < DIMT; k++) { v0 += t[k] * x0[k]; v1 += t[k] * x1[k]; }
      if (on0) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(prow + j0), v0);
      if (on1) __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(prow + j1), v1);
    }
  }
}

#if MJPC_HELPER
DEV int jobw_load(const Ctx &c) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(c.misc + HX_JOBW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
// owner: post the job of the iterate whose zones / cone records are in LDS; returns the number of workers it was cut for
template <int NVT>
DEV int job_post(Ctx &c, int kind, int &side_in) {
  if (c.M->cone != 1) return 0;
  const int seq = ++c.hseq;
  // WHICH jobs the side wave shares is fixed by the job's number, never by who happened to be ready: the partition of the sums -
  // and with it every rounding - must not depend on timing.  Should the side wave still be busy at its first job, the owner waits.
  if (MJPC_SIDE_JOB > 0 && !side_in && (seq & 255) >= MJPC_SIDE_JOB) {
    const int base = seq & ~255;
    int ok = 0;
    for (int n = 0; n < (1 << 21); n++) {
      int from = __builtin_amdgcn_readfirstlane(__hip_atomic_load(c.misc + HX_SIDEFROM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
      if (from > base) { ok = 1; break; }
    }
    if (!ok) c.warning |= WARN_SYNC;
    side_in = 1;
  }
  const int np = MJPC_NH + (side_in ? 1 : 0);
  flag_set(c.misc + HX_JOBW, JOBW(seq, np, kind));
  return np;
}
DEV void job_wait(Ctx &c, int np) {
  const int seq = c.hseq;
  for (int k = 0; k < np; k++) if (!flag_wait(c.misc + HX_WDONE + k, seq)) c.warning |= WARN_SYNC;
}
template <int NVT, int DIMT>
DEV void worker_job(Ctx &c, int W, int np, int seq) {
  constexpr int nvp = NVP_OF(NVT);
  PROFW(c, 12);
  double *part = CONE_PARTIAL(c, W);
  PFOR(e, NVT * nvp) part[e] = 0;
  PROFW(c, 15);
  cone_rows<NVT, DIMT>(c, part, W, np);
  PROFW(c, 13);
  flag_set(c.misc + HX_WDONE + W, seq);
  PROFW(c, 14);
}
// worker W: jobs with a sequence number above `last` until the release (a job cut for fewer workers than W + 1 is not this wave's)
template <int NVT, int DIMT>
DEV void worker_loop_d(Ctx &c, int W, int last) {
  // the contact records the cone blocks walk are helper 0's first job of the solve phase
  if (W != 0 && c.nefc > 0 && !flag_wait(c.misc + HX_LSREC, (last >> 8) + 1)) return;
  for (;;) {
    int word = 0, ok = 0;
    for (int n = 0; n < (1 << 21); n++) { word = jobw_load(c); if ((word >> 4) > last) { ok = 1; break; } }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (!ok) return;                                          // timed out: the owner reports the failure
    const int kind = word & 3, np = (word >> 2) & 3;
    if (kind == 0) return;
    last = word >> 4;
    if (W < np) worker_job<NVT, DIMT>(c, W, np, last);
  }
}
template <int NVT>
DEV void worker_loop(Ctx &c, int W, int last) {
  if (c.M->cone != 1) return;
  if (c.M->maxdim <= 3) worker_loop_d<NVT, 3>(c, W, last); else worker_loop_d<NVT, 6>(c, W, last);
}
// the side wave joins once its own work of the solve phase is done: it announces the first job it could take
template <int NVT>
DEV void side_worker(Ctx &c, int t) {
  if (c.M->cone != 1 || MJPC_SIDE_JOB <= 0) return;
  const int base = t * 256;
  int word = jobw_load(c);
  int last = base;
  if ((word >> 4) > base) {
    if ((word & 3) == 0) return;                              // the solve is already over
    last = word >> 4;                                         // that job was cut without this wave
  }
  if (LANE == 0) __hip_atomic_store(c.misc + HX_SIDEFROM, last + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  worker_loop<NVT>(c, MJPC_NH, last);
}
#endif

// qH = hq + diag of the single-entry rows: the owner's part of the Hessian.  The elliptic contacts' blocks are the workers'
// partials, added by the factorisation (np > 0); a build without worker waves adds them here.
template <int NVT, int DIMT>
DEV void newton_assemble(Ctx &c, const double *hq, int hi, int hg, int j0, bool hact) {
  constexpr int nvp = NVP_OF(NVT), CB = HLay<NVT>::CB;
  double a[CB];
  const double dg = c.sgl[NVT + hi] + c.sgl[3 * NVT + hi];
#pragma unroll
  for (int q = 0; q < CB; q++) a[q] = hq[q] + ((j0 + q == hi) ? dg : 0.0);
#if !MJPC_HELPER
  if (c.M->cone == 1) {
    // single-wave build: the owner is its own only worker
    double *part = CONE_PARTIAL(c, 0);
    PFOR(e, NVT * nvp) part[e] = 0;
    cone_rows<NVT, DIMT>(c, part, 0, 1);
    SYNC();
#pragma unroll
    for (int q = 0; q < CB; q++) a[q] += (hact && j0 + q <= hi) ? part[hi * nvp + j0 + q] : 0.0;
  }
#endif
  if (hact) {
#pragma unroll
    for (int q = 0; q < CB; q++) if (j0 + q < NVT) c.qH[hi * nvp + j0 + q] = a[q];
  }
  SYNC();
}
// Newton direction from qH (+ the workers' partials): Mgrad <- H^-1 grad
template <int NVT>
DEV void newton_direction_np(Ctx &c, int np) {
  if (np > 0) {
    LDLExtra ex;
    ex.row = nullptr;
    ex.n = np;
#pragma unroll
    for (int w = 0; w < 3; w++) ex.x[w] = CONE_PARTIAL(c, w < np ? w : 0);
    newton_direction_sum<NVT>(c, ex);
  } else newton_direction<NVT>(c);
}

// One column group (nv > 32: the hand): lane i's Hessian block IS row i, the layout the register factorisation works in - the
// Hessian never goes through LDS (no qH in such a kernel's layout): Mgrad <- (hq + diag + the workers' partials)^-1 grad
template <int NVT>
DEV void newton_direction_rows(Ctx &c, const double *hq, int hi, bool hact, int np) {
  if constexpr (HLay<NVT>::G != 1) return;             // (only instantiated for one column group)
  double row[NVT];
  const double dg = c.sgl[NVT + hi] + c.sgl[3 * NVT + hi];
#pragma unroll
  for (int q = 0; q < NVT; q++) row[q] = hact ? hq[q] + ((q == hi) ? dg : 0.0) : 0.0;
  LDLExtra ex;
  ex.row = row;
  ex.n = np;
#pragma unroll
  for (int w = 0; w < 3; w++) ex.x[w] = CONE_PARTIAL(c, w < np ? w : 0);
  chol_factor_solve_reg<NVT>(c.qH, c.Mgrad, NVP_OF(NVT), c.M->tree_ok && !c.cross, &ex);
  PROF(c, 17);
}

// line-search data of a lane plus what the commit needs (friction coefficients, first row / dim of the contact, the single-entry
// rows' Jacobian entry, the zone the row was in)
template <int DIMT>
struct LSReg {
  double lo[LS_RPL], hi[LS_RPL], hD[LS_RPL], F[LS_RPL], X[LS_RPL], V[LS_RPL], DV[LS_RPL], DVV[LS_RPL], rJ[LS_RPL];
  int rinfo[LS_RPL];     // bit 0 row of the cost sum (exists, not elliptic) | bit 1 friction-type | bit 2 single-entry | bit 3 was quadratic | dof << 8
  // elliptic contact of the lane, along the search direction (U_j = fr_j jar_j, V_j = fr_j jv_j): everything an evaluation needs is
  // a polynomial in alpha:  N = N0 + a NV;  sum_{j>0} U_j^2 = t0 + 2 a t1 + a^2 t2;  sum_{j>0} U_j V_j = t1 + a t2;  the quadratic
  // zone's cost  sum_j 1/2 E_j U_j^2 = c0 + a c1 + a^2 c2
  double N0[LS_CPL], NV[LS_CPL], t0[LS_CPL], t1[LS_CPL], t2[LS_CPL], c0[LS_CPL], c1[LS_CPL], c2[LS_CPL], mu[LS_CPL], Dm[LS_CPL];
  int on[LS_CPL], cdim[LS_CPL], crow[LS_CPL], cwasq[LS_CPL];
  int nslot, ncslot;
};

// Per-step constants of the line search, one record per row / contact in LDS (behind the gradient scratch in efc_JA).  They do
// not change between the Newton iterations of a step: a helper wave builds them while the owner prices the warm start, and an
// iteration's ls_load_reg() only fetches them next to the two things that do change (the residuals and their slopes).
//   row record      lo, hi, D / 2, friction loss, J entry (single-entry rows), info
//   contact record  E_j = D_j / fr_j^2 (6), fr_j (6; fr_0 = mu), mu, Dm, info (on | dim << 8 | first row << 16)
template <int NVT>
DEV void ls_records_build(Ctx &c) {
  constexpr int nvp = NVP_OF(NVT);
  const int nefc = c.nefc, ncon = c.ncon, ns = c.nsingle, nfr = c.M->nfric;
  double *rrec = LS_ROWREC(c), *crec = LS_CONREC(c);
  PFOR(r, nefc) {
    int type = c.efc_type[r], dof = c.efc_dof[r];
    double D = c.efc_D[r], f = c.efc_floss[r], Rf = c.efc_R[r] * f;
    int single = r < ns;
    double rj = single ? (r < nfr ? 1.0 : c.efc_J[r * nvp + dof]) : 0.0;
    int quad = type != CNSTR_CONTACT_ELLIPTIC;
    int fric = quad && type <= CNSTR_FRICTION_TENDON;
    double *o = rrec + r * LSR_STRIDE;
    o[0] = fric ? -Rf : (quad ? -1e300 : -1.0); o[1] = fric ? Rf : (quad ? 0.0 : 1.0);
    o[2] = quad ? 0.5 * D : 0.0; o[3] = fric ? f : 0.0; o[4] = rj;
    ((int *)(o + 5))[0] = quad | (fric << 1) | ((quad && single) << 2) | ((single ? dof : 0) << 8);
    o[6] = D;                                      // (elliptic rows: their rank-1 weight when the contact enters / leaves its quadratic zone)
  }
  int nc_lane = 0;
  PFOR(ci, ncon) {
    int dim = c.con_i[ci * CONI_STRIDE], i = c.con_i[ci * CONI_STRIDE + 3];
    const double *cc = c.contact + ci * c.M->con_stride;
    int on = dim > 1 && c.efc_type[i] == CNSTR_CONTACT_ELLIPTIC;
    double mu = cc[CON_MU];
    double *o = crec + ci * LSC_STRIDE;
    for (int j = 0; j < 6; j++) {
      int use = on && j < dim;
      double fr = j == 0 ? mu : cc[CON_FRICTION + j - 1];
      double Dj = c.efc_D[use ? i + j : i];
      o[j] = use ? Dj * fast_rcp(fr * fr) : 0.0;
      o[6 + j] = use ? fr : 0.0;
    }
    o[12] = on ? mu : 0.0;
    o[13] = on ? c.efc_D[i] * fast_rcp(mu * mu * (1 + mu * mu)) : 0.0;
    ((int *)(o + 14))[0] = on | (dim << 8) | (i << 16);
    // the dofs that move either body of the contact, ascending: the support of its Jacobian rows (cone_local_add)
    int nc = 0;
    if (on) {
      unsigned long long m = MDM()[MI(geom_bodyid)[c.con_i[ci * CONI_STRIDE + 1]]] | MDM()[MI(geom_bodyid)[c.con_i[ci * CONI_STRIDE + 2]]];
      unsigned char *dl = CONE_DOFS(c) + ci * NVT;
      while (m) { int dd = (int)__builtin_ctzll(m); m &= m - 1; dl[nc++] = (unsigned char)dd; }
    }
    ((int *)(o + 14))[2] = nc;
    nc_lane = nc;
  }
  // ... and the (contact, local row) pairs of all of them in the list the workers' passes walk (cone_rows), contact-major
  // (nconmax <= 64: one contact per lane)
  {
    int total, off = wave_excl_scan(nc_lane, &total);
    unsigned short *il = CONE_ITEMS(c) + off;
    for (int a = 0; a < nc_lane; a++) il[a] = (unsigned short)(LANE | (a << 8));
    if (LANE == 0) c.misc[HX_NITEMS] = total;
  }
  SYNC();
}

template <int NVT, int DIMT>
DEV void ls_load_reg(Ctx &c, LSReg<DIMT> &d) {
  const int nefc = c.nefc, ncon = c.ncon, last = c.nefc - 1;
  const double *rrec = LS_ROWREC(c), *crec = LS_CONREC(c);
  d.nslot = (nefc + NLANE - 1) / NLANE; d.ncslot = (ncon + NLANE - 1) / NLANE;
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    d.lo[k] = -1; d.hi[k] = 1; d.hD[k] = 0; d.F[k] = 0; d.X[k] = 0; d.V[k] = 0; d.DV[k] = 0; d.DVV[k] = 0; d.rJ[k] = 0; d.rinfo[k] = 0;
    if (k < d.nslot) {
      int r = LANE + NLANE * k, rc = r < nefc ? r : last;
      const double *o = rrec + rc * LSR_STRIDE;
      double lo = o[0], hi = o[1], hD = o[2], F = o[3], rj = o[4];
      int info = ((const int *)(o + 5))[0], st = c.efc_state[rc];
      double v = c.efc_jv[rc], x = c.efc_jar[rc];
      __builtin_amdgcn_sched_barrier(0);
      const bool ok = r < nefc;
      // X / V are kept for every existing row (the commit writes jar = X + alpha V for the elliptic rows too); with D = F = 0 such
      // a row adds nothing to the sums whatever its zone
      d.X[k] = ok ? x : 0.0; d.V[k] = ok ? v : 0.0;
      d.lo[k] = ok ? lo : -1.0; d.hi[k] = ok ? hi : 1.0; d.hD[k] = ok ? hD : 0.0; d.F[k] = ok ? F : 0.0; d.rJ[k] = rj;
      d.rinfo[k] = ok ? (info | ((st == STATE_QUADRATIC) << 3)) : 0;
      double dv = 2 * d.hD[k] * d.V[k];
      d.DV[k] = dv; d.DVV[k] = dv * d.V[k];
    }
  }
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    d.on[q] = 0; d.mu[q] = 0; d.Dm[q] = 0; d.cdim[q] = 0; d.crow[q] = 0; d.cwasq[q] = 0;
    d.N0[q] = 0; d.NV[q] = 0; d.t0[q] = 0; d.t1[q] = 0; d.t2[q] = 0; d.c0[q] = 0; d.c1[q] = 0; d.c2[q] = 0;
    if (q < d.ncslot) {
      int ci = LANE + NLANE * q, cic = ci < ncon ? ci : ncon - 1;
      const double *o = crec + cic * LSC_STRIDE;
      int info = ((const int *)(o + 14))[0];
      double E[DIMT], fr[DIMT];
#pragma unroll
      for (int j = 0; j < DIMT; j++) { E[j] = o[j]; fr[j] = o[6 + j]; }
      double mu = o[12], Dm = o[13];
      int on = (ci < ncon) && (info & 1), dim = (info >> 8) & 255, i = info >> 16;
      double jv[DIMT], jr[DIMT];
#pragma unroll
      for (int j = 0; j < DIMT; j++) { int rj = i + j < nefc ? i + j : last; jv[j] = c.efc_jv[rj]; jr[j] = c.efc_jar[rj]; }
      int st0 = c.efc_state[i];
      __builtin_amdgcn_sched_barrier(0);
      d.on[q] = on; d.mu[q] = on ? mu : 0.0; d.Dm[q] = on ? Dm : 0.0; d.cdim[q] = dim; d.crow[q] = i;
      d.cwasq[q] = st0 == STATE_QUADRATIC;
      // (E_j = fr_j = 0 beyond the contact's dim and for a lane without an elliptic contact: those slots stay out of every sum)
      double t0 = 0, t1 = 0, t2 = 0, c0 = 0, c1 = 0, c2 = 0, n0 = 0, nv_ = 0;
#pragma unroll
      for (int j = 0; j < DIMT; j++) {
        double u = jr[j] * fr[j], v = jv[j] * fr[j], eu = E[j] * u, ev = E[j] * v;
        c0 += 0.5 * eu * u; c1 += eu * v; c2 += 0.5 * ev * v;
        if (j == 0) { n0 = u; nv_ = v; } else { t0 += u * u; t1 += u * v; t2 += v * v; }
      }
      d.N0[q] = on ? n0 : 0.0; d.NV[q] = on ? nv_ : 0.0; d.t0[q] = on ? t0 : 0.0; d.t1[q] = on ? t1 : 0.0; d.t2[q] = on ? t2 : 0.0;
      d.c0[q] = on ? c0 : 0.0; d.c1[q] = on ? c1 : 0.0; d.c2[q] = on ? c2 : 0.0;
    }
  }
}

template <int DIMT>
DEV LSPoint ls_eval_reg(const LSReg<DIMT> &d, double q0, double q1, double q2, double a) {
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
    const double mu = d.mu[q];
    const double N = d.N0[q] + a * d.NV[q];
    const double UV = d.t1[q] + a * d.t2[q];                      // sum_{j>0} U_j V_j
    const double T2 = fmax(d.t0[q] + a * (d.t1[q] + UV), 0.0);    // sum_{j>0} U_j^2
    double iT = fast_rsqrt(T2);            // 1/T without an IEEE divide + sqrt on the critical path
    double T = T2 * iT;
    if (N >= mu * T || (T <= 0 && N >= 0)) {
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
      const double c2a = d.c2[q] * a;
      p.cost += d.c0[q] + a * (d.c1[q] + c2a); p.d1 += d.c1[q] + 2 * c2a; p.d2 += 2 * d.c2[q];
    } else {
      double Dm = d.Dm[q], NmT = N - mu * T;
      double T1 = UV * iT, T2d = (d.t2[q] - T1 * T1) * iT;
      double g1 = d.NV[q] - mu * T1;
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

// the constraint update at alpha, from the line-search registers: jar, force, zone of every row, the per-dof folds of the
// single-entry rows, the cone factors of the elliptic contacts.  chg_mask[k] / chg_w[k]: the general rows of slot k whose
// quadratic-zone membership changed and the signed weight (+-D) of their rank-1 term in the Hessian.
template <int NVT, int DIMT>
DEV double ls_commit(Ctx &c, const LSReg<DIMT> &d, double a, unsigned long long *chg_mask, double *chg_w, int *ncone_out) {
  const int nefc = c.nefc, stride = c.M->con_stride;
  double cost = 0;
  int ncone = 0;
  // (contacts first: they read the residuals of the iterate the search started from, which the row pass below overwrites)
#pragma unroll
  for (int q = 0; q < LS_CPL; q++) {
    if (q >= d.ncslot) break;
    if (!d.on[q]) continue;
    int ci = LANE + NLANE * q, dim = d.cdim[q], i = d.crow[q];
    double *cc = c.contact + ci * stride;
    double mu = d.mu[q], U[DIMT], F[DIMT], Ej[DIMT], frj[DIMT];
    const int wasquad = d.cwasq[q];
    {
      // the contact's own rows once more (the evaluations only carried polynomials in alpha): old residuals + alpha * slopes
      const double *o = LS_CONREC(c) + ci * LSC_STRIDE;
      double jr[DIMT], jv[DIMT];
#pragma unroll
      for (int j = 0; j < DIMT; j++) { int rj = i + j < nefc ? i + j : nefc - 1; jr[j] = c.efc_jar[rj]; jv[j] = c.efc_jv[rj]; Ej[j] = o[j]; frj[j] = o[6 + j]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < DIMT; j++) U[j] = (jr[j] + a * jv[j]) * frj[j];
    }
    double T2 = 0;
#pragma unroll
    for (int j = 0; j < DIMT; j++) { F[j] = 0; if (j > 0) T2 += U[j] * U[j]; }
    double iT = fast_rsqrt(T2);
    double N = U[0], T = T2 * iT;
    int st;
    if (N >= mu * T || (T <= 0 && N >= 0)) {
      st = STATE_SATISFIED;
    } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
#pragma unroll
      for (int j = 0; j < DIMT; j++) { double eu = Ej[j] * U[j]; cost += 0.5 * eu * U[j]; F[j] = -(eu * frj[j]); }        // -D_j jar_j with jar_j = U_j / fr_j, D_j = E_j fr_j^2
      st = STATE_QUADRATIC;
      if (!wasquad) {       // the zone-independent block record (see constraint_update): P = Q = 0, T_j = sqrt(D_j), w0 = D_0
#pragma unroll
        for (int j = 0; j < DIMT; j++) if (j < dim) {
          double Dj = Ej[j] * frj[j] * frj[j];
          cc[CON_H + j] = 0;
          cc[CON_H + 6 + j] = j == 0 ? Dj : 0.0;
          cc[CON_H + 12 + j] = Dj * fast_rsqrt(Dj);
        }
      }
    } else {
      double Dm = d.Dm[q], NmT = N - mu * T;
      double f0 = -Dm * NmT * mu;
      cost += 0.5 * Dm * NmT * NmT;
      F[0] = f0;
#pragma unroll
      for (int j = 1; j < DIMT; j++) F[j] = -f0 * iT * U[j] * frj[j];
      st = STATE_CONE;
      double kap = -mu * NmT * Dm * iT;
      double sD = Dm * fast_rsqrt(Dm), sk = kap > 0 ? kap * fast_rsqrt(kap) : 0.0;
      cc[CON_H] = sD * frj[0];
      cc[CON_H + 6] = 0;
      cc[CON_H + 12] = -sD * NmT;
#pragma unroll
      for (int j = 1; j < DIMT; j++) if (j < dim) {
        double u = U[j] * iT;
        cc[CON_H + j] = -sD * frj[j] * mu * u;
        cc[CON_H + 6 + j] = sk * frj[j] * u;
        cc[CON_H + 12 + j] = sk * frj[j];
      }
    }
#pragma unroll
    for (int j = 0; j < DIMT; j++) if (j < dim) { c.efc_force[i + j] = F[j]; c.efc_state[i + j] = st; }
    ncone += st == STATE_CONE;
  }
  *ncone_out = __builtin_popcountll(__builtin_amdgcn_ballot_w64(ncone != 0));      // (LS_CPL == 1 on the device: one contact per lane)
#pragma unroll
  for (int k = 0; k < LS_RPL; k++) {
    chg_mask[k] = 0; chg_w[k] = 0;
    if (k >= d.nslot) break;
    int r = LANE + NLANE * k;
    int info = d.rinfo[k];
    // a row of an elliptic contact: the zone its contact's lane has just written (same wave: LDS keeps program order) and the row's D
    const int rc_ = r < nefc ? r : nefc - 1;
    const int st_new = c.efc_state[rc_];
    const double D_ell = LS_ROWREC(c)[rc_ * LSR_STRIDE + 6];
    int quad = info & 1, fric = (info >> 1) & 1, single = (info >> 2) & 1, wasq = (info >> 3) & 1, dof = info >> 8;
    double x = d.X[k] + a * d.V[k];
    if (r < nefc) c.efc_jar[r] = x;
    double lo = d.lo[k], hi = d.hi[k], D = 2 * d.hD[k], f = d.F[k];
    double xc = fmin(fmax(x, lo), hi);
    int inside = x > lo && x < hi;
    int chg = 0;
    cost += d.hD[k] * xc * xc + f * (fabs(x) - fabs(xc));
    if (quad) {
      double force = fric ? (inside ? -D * x : (x <= lo ? f : -f)) : -D * xc;
      c.efc_force[r] = force;
      c.efc_state[r] = inside ? STATE_QUADRATIC : (fric ? (x <= lo ? STATE_LINEARNEG : STATE_LINEARPOS) : STATE_SATISFIED);
      if (single) {
        int kk = fric ? 0 : 2;
        c.sgl[kk * NVT + dof] = d.rJ[k] * force;
        c.sgl[(kk + 1) * NVT + dof] = inside ? D : 0.0;
      } else {
        chg = inside != wasq;
        chg_w[k] = inside ? D : -D;
      }
    } else if (r < nefc) {
      const int nowq = st_new == STATE_QUADRATIC;
      chg = nowq != wasq;
      chg_w[k] = nowq ? D_ell : -D_ell;
    }
    chg_mask[k] = __builtin_amdgcn_ballot_w64(chg != 0);
  }
  return wave_sum(cost);
}

template <int NVT, int DIMT>
DEV void solve_constraints_reg_d(Ctx &c) {
  const DevModel &M = *c.M;
  constexpr int nv = NVT, nvp = NVP_OF(NVT);
  constexpr int G = HLay<NVT>::G, CB = HLay<NVT>::CB;
  // one column group in the dense tier's lean layout (the hand): factor straight from the row registers, no Hessian in LDS
  // (with all 512 registers and room in LDS the round trip through qH is the faster form: measured on the hand, 6.94 vs 7.62 ms)
#ifdef MJPC_LEAN_LDS
  constexpr bool ROWS = G == 1 && MJPC_HELPER;
#else
  constexpr bool ROWS = false;
#endif
  const int hg_ = LANE / NVT;
  const bool hact = hg_ < G;
  const int hg = hact ? hg_ : 0, hi = hact ? LANE - hg_ * NVT : 0, j0 = hg * CB;
  c.solver_iter = 0;
  PROF(c, 7);
  // warm start: the better of qacc_smooth and qacc_warmstart (evaluated last, so its force/state stay valid)
  double gauss, cost, cost_sm;
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
  int np = 0, side_in = 0;
  // contacts in their cone (middle) zone: only those have an iterate-dependent block (the workers' job); none => no job at all
  int ncone = 0;
  if (c.M->cone == 1) {
    const int ci = LANE < c.ncon ? LANE : 0;
    const int dim = c.con_i[ci * CONI_STRIDE], r0c = c.con_i[ci * CONI_STRIDE + 3];
    const int stc = c.efc_state[r0c];
    ncone = __builtin_popcountll(__builtin_amdgcn_ballot_w64(LANE < c.ncon && dim > 1 && stc == STATE_CONE));
  }
#if MJPC_HELPER
  if (ncone) np = job_post<NVT>(c, 1, side_in);
#else
  ls_records_build<NVT>(c);
#endif
  double hq[CB];
  hblock_init<NVT>(c, hq, hi, j0);
  PROF(c, 15);
  newton_grad_reg<NVT>(c, hi, hg, hact);
  PROF(c, 9);
  if constexpr (!ROWS) { newton_assemble<NVT, DIMT>(c, hq, hi, hg, j0, hact); PROF(c, 19); }
#if MJPC_HELPER
  if (np) { job_wait(c, np); PROF(c, 16); }
#endif
  if constexpr (!ROWS) newton_direction_np<NVT>(c, np); else newton_direction_rows<NVT>(c, hq, hi, hact, np);
  PFOR(i, nv) c.search[i] = -c.Mgrad[i];
  SYNC();
  const double scale = 1.0 / (M.meaninertia * (nv > 1 ? nv : 1));
  for (int iter = 0; iter < M.iterations; iter++) {
    PROF(c, 13);
    // ---- exact line search along `search`
    double p_sn = 0, p_q1 = 0, p_q2 = 0, p_gs = 0;
    mat_rows_times<NVT>(c, c.search, c.Mv, c.efc_jv);
    PFOR(i, nv) {
      double si = c.search[i];
      p_sn += si * si; p_q1 += si * (c.Ma[i] - c.qfrc_smooth[i]); p_q2 += 0.5 * si * c.Mv[i]; p_gs += c.grad[i] * si;
    }
    SYNC();
    wave_sum4(p_sn, p_q1, p_q2, p_gs);
    const double snorm = sqrt(p_sn), q1 = p_q1, q2 = p_q2, gs = p_gs;
    PROF(c, 20);
    if (snorm < D_MINVAL || gs >= 0) break;
    const double gtol = M.tolerance * M.ls_tolerance * snorm / scale;
    LSReg<DIMT> d;
#if MJPC_HELPER
    if (iter == 0 && !flag_wait(c.misc + HX_LSREC, c.hseq / 256 + 1)) c.warning |= WARN_SYNC;      // helper 0 built the records meanwhile
#endif
    ls_load_reg<NVT, DIMT>(c, d);
    PROF(c, 21);
    double lo = 0, hi_a = -1, a = 1.0;
    double best_a = 0, best_cost = cost, dxold = a, dx = a;
    LSPoint p; p.cost = cost; p.d1 = gs; p.d2 = -gs;
    int moved = 0;
    const double pred_tol = 1e-3 * M.tolerance / scale;
    for (int it = 0; it < M.ls_iterations; it++) {
      p = ls_eval_reg<DIMT>(d, gauss, q1, q2, a);
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
      if (LANE == 0) c.prof[23] += 1;
#endif
      int better = p.cost < best_cost;
      best_cost = better ? p.cost : best_cost; best_a = better ? a : best_a;
      int conv = fabs(p.d1) < gtol;
      int neg = p.d1 < 0;
      lo = neg ? a : lo; hi_a = neg ? hi_a : a;
      int pos2 = p.d2 > 0;
      double newton = a - p.d1 * fast_rcp(p.d2);
      double an_e = pos2 ? newton : 2 * a;
      an_e = (an_e > a) ? an_e : 2 * a;
      double nw = pos2 ? newton : lo - 1;
      int ok = (nw > lo) && (nw < hi_a) && (fabs(2 * p.d1) <= fabs(dxold * p.d2));
      double dx_b = ok ? fabs(nw - a) : 0.5 * (hi_a - lo);
      double an_b = ok ? nw : lo + dx_b;
      int bracketed = !(hi_a < 0);
      double an = bracketed ? an_b : an_e;
      dxold = dx; dx = bracketed ? dx_b : an_e - a;
      if (conv || an == a) break;
#if LS_PREDICT
      // a plain Newton step on phi' that promises less than a thousandth of the solver's stopping threshold: take it unseen (the
      // commit below prices the point; the evaluation that would only confirm |phi'| < gtol is skipped)
      const int plain = pos2 && (bracketed ? ok : newton > a);
      if (plain && 0.5 * p.d1 * p.d1 * fast_rcp(p.d2) < pred_tol) { a = an; moved = 1; break; }
#endif
      a = an;
    }
    PROF(c, 22);
    const double alpha = moved ? a : best_a;
    PROF(c, 14);
    if (alpha == 0) break;
    // ---- move there: the commit is the constraint update at alpha (it also returns the constraint cost)
    unsigned long long chg_mask[LS_RPL]; double chg_w[LS_RPL];
    const double ccost = ls_commit<NVT, DIMT>(c, d, alpha, chg_mask, chg_w, &ncone);
    PFOR(i, nv) { c.qacc[i] += alpha * c.search[i]; c.Ma[i] += alpha * c.Mv[i]; }
    SYNC();
    gauss = gauss + alpha * q1 + alpha * alpha * q2;
    const double oldcost = cost;
    cost = gauss + ccost;
    PROF(c, 12);
    const double improvement = scale * (oldcost - cost);
    const int stop = improvement < M.tolerance || (c.warning & WARN_SYNC) != 0;
#if MJPC_HELPER
    np = (stop || !ncone) ? 0 : job_post<NVT>(c, 1, side_in);
#endif
    if (!stop) {
#pragma unroll
      for (int k = 0; k < LS_RPL; k++) { if (k >= d.nslot) break; hblock_add_rows<NVT>(c, hq, chg_mask[k], NLANE * k, chg_w[k], hi, j0); }
    }
    PROF(c, 15);
    newton_grad_reg<NVT>(c, hi, hg, hact);
    PROF(c, 9);
    c.solver_iter++;
    double pg = 0;
    PFOR(i, nv) pg += c.grad[i] * c.grad[i];
    const double gradient = scale * sqrt(wave_sum(pg));
    const int done = stop || gradient < M.tolerance;
    if constexpr (!ROWS) { if (!done) { newton_assemble<NVT, DIMT>(c, hq, hi, hg, j0, hact); PROF(c, 19); } }
#if MJPC_HELPER
    if (np) { job_wait(c, np); PROF(c, 16); }           // (also when the gradient says stop: no job is left behind unfinished)
#endif
    if (done) break;
    if constexpr (!ROWS) newton_direction_np<NVT>(c, np); else newton_direction_rows<NVT>(c, hq, hi, hact, np);
    PFOR(i, nv) c.search[i] = -c.Mgrad[i];
    SYNC();
  }
  if (LANE == 0) { c.misc[5] += c.solver_iter; if (c.ncon > c.misc[6]) c.misc[6] = c.ncon; if (c.nefc > c.misc[7]) c.misc[7] = c.nefc; }
  PFOR(i, nv) c.qfrc_constraint[i] = (c.Ma[i] - c.qfrc_smooth[i]) - c.grad[i];
  SYNC();
}

template <int NVT>
DEV void solve_constraints_reg(Ctx &c) {
  if (c.M->maxdim <= 3) solve_constraints_reg_d<NVT, 3>(c); else solve_constraints_reg_d<NVT, 6>(c);
}
#endif
