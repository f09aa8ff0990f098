// core.h — one candidate rollout, written SPMD over the lanes of ONE wavefront.
//
// Replaces (for every candidate i in parallel) what a ThreadPool worker runs in the reference:
//   mjpc/planners/sampling/planner.cc:355-376   copy nominal policy, AddNoiseToPolicy, Rollout
//   mjpc/trajectory.cc:100-210                  NoisyRollout: policy -> ctrl -> mj_step -> record
//   mjpc/trajectory.cc:312-326                  UpdateReturn
// mj_step / mj_forward (MuJoCo 3.1.4, third-party) are re-designed here as lane-parallel phases over
// LDS-resident state: tree-level kinematics, subtree reductions, pairwise mass-matrix entries,
// scan-compacted collision + constraint rows, a primal Newton solver with wave reductions in the
// exact line search, and Euler with implicit joint damping.
#pragma once
#include "dmath.h"
#include "model.h"
#include "linalg.h"
#include "philox.h"

// ---- where the kernel parameters, the LDS block and the candidate index come from ---------------------
// Phases are __noinline__ (each gets its own register allocation; a fully inlined rollout needs all 512
// registers and serialises every LDS load).  They re-derive what they need from three ambient sources:
// the kernarg segment pointer argument (scalar loads), the workgroup's dynamic LDS symbol, and blockIdx.
#ifdef MJPC_EMU
static thread_local double *g_emu_lds = nullptr;
static thread_local int g_emu_r = 0;
typedef const KParams *KP;
DEV const KParams *kp_generic(KP k) { return k; }
DEV double *lds_base() { return g_emu_lds; }
DEV int cand_index() { return g_emu_r; }
DEV int uniform_i(int v) { return v; }
#else
extern __shared__ __align__(16) double g_lds[];
// The kernarg segment pointer is taken in the kernel and handed to every phase as a constant-address-space
// (4) argument, so K->... stays scalar loads.  (__builtin_amdgcn_kernarg_segment_ptr() inside a
// __noinline__ callee returns null on gfx950 / ROCm 7.2 — measured.)
typedef const __attribute__((address_space(4))) KParams *KP;
DEV const KParams *kp_generic(KP k) {
  // arguments of callable functions arrive in VGPRs: make the pointer provably wave-uniform again
  unsigned long long v = (unsigned long long)k;
  unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  KP u = (KP)(((unsigned long long)hi << 32) | lo);
  return (const KParams *)u;
}
DEV double *lds_base() { return g_lds; }
DEV int cand_index() { return (int)blockIdx.x; }
DEV int uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
#define NPROF 24
// diagnostic builds only: lane 0 accumulates s_memtime deltas per phase slot in LDS
// (multi-wave builds: only the owner wave stamps, so the slots show ITS timeline including the waits for the other waves)
#ifndef MJPC_PROFILE_WAVE
#define MJPC_PROFILE_WAVE 0      // which wavefront of the workgroup records its timeline (0 = owner)
#endif
#define PROF_STAMP(pr, i) do { long long t_ = (long long)__builtin_amdgcn_s_memtime(); (pr)[i] += t_ - (pr)[NPROF]; (pr)[NPROF] = t_; } while (0)
#define PROF(c, i) do { if (MJPC_PROFILE_WAVE == 0 && LANE == 0 && (c).role == 0) PROF_STAMP((c).prof, i); } while (0)
#define PROFW(c, i) do { if (MJPC_PROFILE_WAVE != 0 && LANE == 0 && WAVE_ID() == MJPC_PROFILE_WAVE) PROF_STAMP((c).prof, i); } while (0)
#else
#define PROF(c, i) ((void)0)
#define PROFW(c, i) ((void)0)
#endif

// activation states and their derivatives live behind ctrl in LDS (host.h make_layout)
#define C_ACT(c) ((c).ctrl + (c).M->nu + 1)
#define C_ACTDOT(c) ((c).ctrl + (c).M->nu + 1 + (c).M->na)

struct Ctx {
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
  long long *prof;
#endif
  const DevModel *M;
  const KParams *K;
  double *mcd; int *mci; const double *gdb; const int *gib;   // LDS copy of the model buffers / their HBM bases
  double *qpos, *qvel, *ctrl, *qacc, *qacc_ws, *qacc_smooth, *qfrc_smooth, *qfrc_bias, *qfrc_constraint, *actuator_force;
  double *mocap_pos, *mocap_quat;
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat, *site_xpos;
  double *subtree_com, *cinert, *crb, *cdof, *cvel, *cdof_dot, *cacc, *cfrc, *cfrc_sub, *subtree_linvel, *bodytmp;
  double *qM, *qL, *qH, *Linv, *Hinv;
  double *efc_J, *efc_JA, *efc_D, *efc_R, *efc_aref, *efc_force, *efc_jar, *efc_jv, *efc_floss, *efc_pos, *efc_margin, *efc_diag;
  double *contact;
  double *Ma, *grad, *Mgrad, *search, *Mv, *vtmp, *sgl;
  double *knot_times, *knot_values, *residual, *terms, *red, *xfrc, *scr_a, *scr_b;
  int *efc_type, *efc_id, *efc_state, *efc_dof, *con_i, *active, *misc, *hpair;
  double time;
  int ncon, nefc, nsingle, warning, solver_iter, cross;
  int hseq;            // hand-shake sequence number with the solver's helper wave
  int role;            // 0: owns the rollout scalars (misc[0..8], time); 1: side wave, only reports warnings (misc[11])
};

DEV void ctx_init(Ctx &c, const KParams *K, double *base) {
  const Lay &L = K->L;
  c.K = K;
  // model tables are read from the workgroup's LDS copy of the two model buffers (same layout; ph_init fills it):
  // MD(f) / MI(f) turn the HBM table pointer M.f into its LDS twin
  c.M = &K->M;
  c.mcd = base + L.mc_d; c.mci = (int *)(base + L.mc_i); c.gdb = K->dbase; c.gib = K->ibase;
#define P_(f) c.f = base + L.f
  P_(qpos); P_(qvel); P_(ctrl); P_(qacc); P_(qacc_ws); P_(qacc_smooth); P_(qfrc_smooth); P_(qfrc_bias);
  P_(qfrc_constraint); P_(actuator_force); P_(mocap_pos); P_(mocap_quat);
  P_(xpos); P_(xquat); P_(xmat); P_(xipos); P_(ximat); P_(xanchor); P_(xaxis); P_(geom_xpos); P_(geom_xmat); P_(site_xpos);
  P_(subtree_com); P_(cinert); P_(crb); P_(cdof); P_(cvel); P_(cdof_dot); P_(cacc); P_(cfrc); P_(cfrc_sub);
  P_(subtree_linvel); P_(bodytmp); P_(qM); P_(qL); P_(qH); P_(Linv); P_(Hinv);
  c.efc_J = base + L.efc_J - K->M.nfric * K->M.nvp;      // rows [nfric, nefcmax) are stored: a friction-loss row is the unit vector of its dof
  P_(efc_JA); P_(efc_D); P_(efc_R); P_(efc_aref); P_(efc_force); P_(efc_jar); P_(efc_jv); P_(efc_floss);
  P_(efc_pos); P_(efc_margin); P_(efc_diag); P_(contact);
  P_(Ma); P_(grad); P_(Mgrad); P_(search); P_(Mv); P_(vtmp); P_(sgl);
#ifdef MJPC_LEAN_LDS      // dense tier: the spline knots stay in HBM / L2 (the nominal's times; this candidate's values, written by ph_init)
  c.knot_times = const_cast<double *>(K->knot_times);
  c.knot_values = K->knots + (size_t)cand_index() * K->P * K->M.nu;
#else
  P_(knot_times); P_(knot_values);
#endif
  P_(residual); P_(terms); P_(red); P_(xfrc); P_(scr_a); P_(scr_b);
#undef P_
  int *ib = (int *)(base + L.ints);
  c.efc_type = ib + L.i_efc_type; c.efc_id = ib + L.i_efc_id; c.efc_state = ib + L.i_efc_state; c.efc_dof = ib + L.i_efc_dof;
  c.con_i = ib + L.i_con; c.active = ib + L.i_active; c.misc = ib + L.i_misc; c.hpair = ib + L.i_hpair;
  c.time = 0; c.ncon = 0; c.nefc = 0; c.nsingle = 0; c.warning = 0; c.solver_iter = 0; c.cross = 0; c.role = 0; c.hseq = 0;
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
  c.prof = (long long *)(base + L.prof);
#endif
}

// rollout scalars shared between phases live in LDS: misc[0..4] = ncon, nefc, nsingle, warning, solver_iter; red[0] = time
DEV void ctx_open(Ctx &c, KP Kc, int role = 0) {
  ctx_init(c, kp_generic(Kc), lds_base());
  c.role = role;
  c.ncon = uniform_i(c.misc[0]); c.nefc = uniform_i(c.misc[1]); c.nsingle = uniform_i(c.misc[2]);
  c.warning = uniform_i(c.misc[3]) | uniform_i(c.misc[11]); c.solver_iter = uniform_i(c.misc[4]); c.cross = uniform_i(c.misc[8]);
  c.time = c.red[0];
}
#ifdef MJPC_NO_MODEL_CACHE     // engine_dense.hip: tables read from HBM / L2 (the 26-29 KB LDS copy would keep a second workgroup off the CU)
#define MD(f) (c.M->f)
#define MI(f) (c.M->f)
#define MDM() ((const unsigned long long *)c.M->body_dofmask)
#define MPM() ((const unsigned long long *)c.M->body_patmask)
#else
#define MD(f) (c.mcd + (int)(c.M->f - c.gdb))
#define MI(f) (c.mci + (int)(c.M->f - c.gib))
#define MDM() ((const unsigned long long *)(c.mcd + (int)((const double *)c.M->body_dofmask - c.gdb)))
#define MPM() ((const unsigned long long *)(c.mcd + (int)((const double *)c.M->body_patmask - c.gdb)))
#endif
// the "hot" tables (host.h: everything packed before hot_i / hot_d - body, joint and dof-tree tables and the derived level /
// subtree / chain lists): in LDS also for the flavour that has no room for the whole copy (MJPC_HOT_CACHE, rollout_direct.hip)
#if defined(MJPC_NO_MODEL_CACHE) && !defined(MJPC_HOT_CACHE)
#define MDH(f) (c.M->f)
#define MIH(f) (c.M->f)
#else
#define MDH(f) (c.mcd + (int)(c.M->f - c.gdb))
#define MIH(f) (c.mci + (int)(c.M->f - c.gib))
#endif
DEV void ctx_close(Ctx &c) {
  SYNC();
  if (c.role != 0) {            // the side wave never writes the owner's scalars
    if (LANE == 0 && c.warning) {
#ifdef MJPC_EMU
      c.misc[11] |= c.warning;
#else
      __hip_atomic_fetch_or(c.misc + 11, c.warning, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // several side / helper waves report here
#endif
    }
    SYNC();
    return;
  }
  if (LANE == 0) {
    c.misc[0] = c.ncon; c.misc[1] = c.nefc; c.misc[2] = c.nsingle; c.misc[3] = c.warning; c.misc[4] = c.solver_iter; c.misc[8] = c.cross;
    c.red[0] = c.time;
  }
  SYNC();
}

// ======================================================================================
// spline policy: TimeSpline::Sample (mjpc/spline/spline.cc:103-156,240-277), one component
// ======================================================================================
DEV double spline_slope(const double *times, const double *values, int P, int dim, int node, int k) {
  if (node == 0) return (values[dim + k] - values[k]) / (times[1] - times[0]);
  if (node == P - 1) return (values[node * dim + k] - values[(node - 1) * dim + k]) / (times[node] - times[node - 1]);
  return 0.5 * (values[(node + 1) * dim + k] - values[node * dim + k]) / (times[node + 1] - times[node]) +
         0.5 * (values[node * dim + k] - values[(node - 1) * dim + k]) / (times[node] - times[node - 1]);
}
DEV double spline_sample(const double *times, const double *values, int P, int dim, int interp, double time, int k) {
  if (P == 0) return 0.0;
  int upper = 0;
  while (upper < P && !(time < times[upper])) upper++;
  if (upper == P) return values[(P - 1) * dim + k];
  if (upper == 0) return values[k];
  int lower = upper - 1;
  double lo = times[lower], up = times[upper];
  double t = (time - lo) / (up - lo);
  if (interp == 0) return values[lower * dim + k];
  if (interp == 1) return values[lower * dim + k] * (1 - t) + values[upper * dim + k] * t;
  double c0 = 2.0 * t*t*t - 3.0 * t*t + 1.0;
  double c1 = (t*t*t - 2.0 * t*t + t) * (up - lo);
  double c2 = -2.0 * t*t*t + 3 * t*t;
  double c3 = (t*t*t - t*t) * (up - lo);
  double p0 = values[lower * dim + k], p1 = values[upper * dim + k];
  double m0 = spline_slope(times, values, P, dim, lower, k);
  double m1 = spline_slope(times, values, P, dim, upper, k);
  return c0 * p0 + c1 * m0 + c2 * p1 + c3 * m1;
}

#include "kinematics.h"
#include "collide.h"
#include "constraint.h"
#include "dynamics.h"
#include "residuals.h"
#include "noslip.h"
// ======================================================================================
// phases of one step (mj_step = position, velocity, solve, [residual], integrate) — all __noinline__
// ======================================================================================
DEV int bad_values(const double *x, int n) {
  int b = 0;
  PFOR(i, n) { double v = x[i]; if (!(v == v) || v > 1e10 || v < -1e10) b = 1; }
  return wave_or_i(b);
}

struct Rows { double *states, *actions, *times, *residual, *costs, *trace; int ds, nr, ntr; };
DEV Rows out_rows(const KParams *K) {
  const DevModel &M = K->M;
  Rows R;
  size_t r = (size_t)cand_index(), H = (size_t)K->H;
  R.ds = M.nq + M.nv + M.na; R.nr = M.task.num_residual; R.ntr = 3 * M.task.num_trace;
  R.states = K->states + r * H * R.ds; R.actions = K->actions + r * H * M.nu; R.times = K->times + r * H;
  R.residual = K->residual + r * H * R.nr; R.costs = K->costs + r * H; R.trace = K->trace + r * H * R.ntr;
  return R;
}

// candidate policy (planner.cc:313-339) + initial state (trajectory.cc:120-137)
DEV_NOINLINE void ph_init(KP Kc) {
  Ctx c;
  const KParams *K = kp_generic(Kc);
  ctx_init(c, K, lds_base());
  // the workgroup's LDS copy of the model tables (everything below reads them through c.M)
  {
    double *mcd = lds_base() + K->L.mc_d; int *mci = (int *)(lds_base() + K->L.mc_i);
    PFOR(e, K->cache_d) mcd[e] = K->dbase[e];
    PFOR(e, K->cache_i) mci[e] = K->ibase[e];
    SYNC();
  }
  const DevModel &M = *c.M;
  Rows R = out_rows(K);
  int nq = M.nq, nv = M.nv, nu = M.nu, P = K->P, r = cand_index();
  int gi = K->offset + r;
#ifndef MJPC_LEAN_LDS
  PFOR(p, P) c.knot_times[p] = K->knot_times[p];
#endif
  double std = K->sigma0;
  if (K->sigma1 > 0 && K->noise_sel[r]) std = K->sigma1;
  PFOR(e, P * nu) {
    int k = e % nu;
    double v = K->knot_values[e];
    double lo = MD(actuator_ctrlrange)[2 * k], hi = MD(actuator_ctrlrange)[2 * k + 1];
    if (K->cand_knots) v = K->cand_knots[(size_t)r * P * nu + e];          // robust planner: explicit candidate policy
    else if (gi != K->nominal_index) {
      if (K->noise_std) v = add_mul3_rn(v, 1.0, K->noise_std[e], K->noise_eps[(size_t)r * P * nu + e]);   // Cross-Entropy: absolute std
      else { double scale = 0.5 * (hi - lo); v = add_mul3_rn(v, scale, std, K->noise_eps[(size_t)r * P * nu + e]); }   // bit-exact candidate policy
      v = d_clip(v, lo, hi);
    }
    c.knot_values[e] = v;
    K->knots[(size_t)r * P * nu + e] = v;
  }
  PFOR(i, M.nmocap) {
    d_copy3(c.mocap_pos + 3 * i, K->mocap + 7 * i);
    d_copy4(c.mocap_quat + 4 * i, K->mocap + 7 * i + 3);
  }
  PFOR(i, nq) { c.qpos[i] = K->state[i]; R.states[i] = K->state[i]; }
  PFOR(i, nv) { c.qvel[i] = K->state[nq + i]; R.states[nq + i] = K->state[nq + i]; c.qacc_ws[i] = 0; }
  if (M.na) PFOR(i, M.na) { C_ACT(c)[i] = K->state[nq + nv + i]; R.states[nq + nv + i] = K->state[nq + nv + i]; C_ACTDOT(c)[i] = 0; }
  PFOR(e, nv * M.nvp) c.qM[e] = 0;
  if (K->L.Linv - K->L.qH >= nv * M.nvp) PFOR(e, nv * M.nvp) c.qH[e] = 0;      // (no qH in the layout of a one-column-group register solve)
#ifndef MJPC_LEAN_LDS
  PFOR(e, M.nhpair + nv) c.hpair[e] = MI(hpair_i)[e] | (MI(hpair_j)[e] << 8);
#endif
#ifndef MJPC_LEAN_LDS
  PFOR(k, 6 * M.nbody) c.xfrc[k] = 0;
#endif
  PFOR(k, nu) c.ctrl[k] = 0;      // data->ctrl after Reset (planner.cc:124-130); only visible when H == 1
  if (LANE == 0) {
    R.times[0] = K->time;
    for (int k = 0; k < 3; k++) { c.xpos[k] = 0; c.xipos[k] = 0; c.subtree_linvel[k] = 0; }
    c.xquat[0] = 1; c.xquat[1] = 0; c.xquat[2] = 0; c.xquat[3] = 0;
    for (int k = 0; k < 9; k++) { c.xmat[k] = (k % 4 == 0) ? 1.0 : 0.0; c.ximat[k] = c.xmat[k]; }
    for (int k = 0; k < 6; k++) { c.cvel[k] = 0; c.cfrc[k] = 0; c.cacc[k] = (k >= 3) ? -M.gravity[k - 3] : 0.0; }
    for (int k = 0; k < MISC_INTS; k++) c.misc[k] = 0;
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
    for (int q = 0; q < NPROF; q++) c.prof[q] = 0;
    c.prof[NPROF] = (long long)__builtin_amdgcn_s_memtime();
#endif
  }
  c.time = K->time;
  ctx_close(c);
}

// role 0, head of a step: policy -> ctrl (policy.cc:52-59), mj_checkPos / mj_checkVel, then kinematics and the
// com-based quantities every other phase reads.  A bad state raises misc[10] (both roles leave the loop).
template <int NVT>
DEV_NOINLINE void ph_head(KP Kc, int t, int last) {
  Ctx c; ctx_open(c, Kc);
  const KParams *K = c.K;
  const DevModel &M = *c.M;
  Rows R = out_rows(K);
  int nu = M.nu;
  if (!last) {
    PFOR(k, nu) {
      double a = spline_sample(c.knot_times, c.knot_values, K->P, nu, K->interp, c.time, k);
      a = d_clip(a, MD(actuator_ctrlrange)[2 * k], MD(actuator_ctrlrange)[2 * k + 1]);
      c.ctrl[k] = a; R.actions[t * nu + k] = a;
    }
    SYNC();
    if (bad_values(c.qpos, M.nq)) c.warning |= WARN_BADQPOS;
    if (bad_values(c.qvel, M.nv)) c.warning |= WARN_BADQVEL;
    if (c.warning) { if (LANE == 0) c.misc[10] = 1; ctx_close(c); return; }
    if (K->xfrc_std > 0) {              // NoisyRollout (trajectory.cc:147-155): Ornstein-Uhlenbeck force/torque noise on every body
      double rate = exp(-M.timestep / K->xfrc_rate), scale = K->xfrc_std * sqrt(1 - rate * rate);
      unsigned gi = (unsigned)(K->offset + cand_index());
      int n6 = 6 * M.nbody;
      PFOR(i, n6) c.xfrc[i] = rate * c.xfrc[i] + scale * philox_normal(K->seed, K->stream ^ XFRC_STREAM, gi, (unsigned)(t * n6 + i));
    }
  }
  PROF(c, 0);
  kinematics(c); PROF(c, 1);
#if !MJPC_SIDE_COM
  com_pos(c);
#endif
  PROF(c, 2);                 // (MJPC_SIDE_COM: the com-based quantities are the side wave's first job of the next phase, ph_smooth)
  ctx_close(c);
}
// role 0: contacts and constraint rows (needs positions + qvel only).  With helper waves the contact-free rows and their
// impedance are built by the last helper meanwhile (ph_noncontact); it publishes nsingle / n_nc in misc[25..26], flag misc[24].
DEV_NOINLINE void ph_constraints(KP Kc, int t) {
  Ctx c; ctx_open(c, Kc);
  collision(c); PROF(c, 4);
  int nsingle, n_nc;
#if MJPC_HELPER
#if MJPC_SIDE_COM
  if (!flag_wait(c.misc + HX_COM, t + 1)) c.warning |= WARN_SYNC;        // cdof / subtree_com for the contact Jacobians (side wave)
#endif
  if (!flag_wait(c.misc + 24, t + 1)) c.warning |= WARN_SYNC;
  nsingle = uniform_i(c.misc[25]); n_nc = uniform_i(c.misc[26]);
  c.nsingle = nsingle;
  make_contact_rows(c, n_nc); PROF(c, 5);
  make_impedance(c, n_nc, c.nefc, 1); PROF(c, 7);
#else
  make_noncontact_rows(c, &nsingle, &n_nc);
  c.nsingle = nsingle;
  make_contact_rows(c, n_nc); PROF(c, 5);
  make_impedance(c, 0, c.nefc, 1); PROF(c, 7);
#endif
  ctx_close(c);
}
#if MJPC_HELPER
DEV_NOINLINE void ph_noncontact(KP Kc, int t) {
  Ctx c; ctx_open(c, Kc, 1);
  int nsingle, n_nc;
#if MJPC_SIDE_COM
  // the anchors' point Jacobians of connect constraints need cdof / subtree_com, which the side wave is producing
  if (c.M->neq_connect) { if (!flag_wait(c.misc + HX_COM, t + 1)) c.warning |= WARN_SYNC; }
#endif
  make_noncontact_rows(c, &nsingle, &n_nc);
  if (LANE == 0) { c.misc[25] = nsingle; c.misc[26] = n_nc; }
  flag_set(c.misc + 24, t + 1);
  PROFW(c, 1);
  make_impedance(c, 0, n_nc, 0);
  PROFW(c, 4);
  // then its part of the side wave's subtree sums (velocity_stage)
  if (!flag_wait(c.misc + HX_SWEEP, t + 1)) c.warning |= WARN_SYNC;
  subtree_sums(c, 1);
  flag_set(c.misc + HX_SUBSUM, t + 1);
  PROFW(c, 5);
  ctx_close(c);
}
#endif
// role 1, concurrently with ph_constraints: joint-space inertia + its factor, smooth dynamics (qfrc_smooth, qacc_smooth)
template <int NVT>
DEV_NOINLINE void ph_smooth(KP Kc, int t) {
  Ctx c; ctx_open(c, Kc, 1);
#if MJPC_HELPER
#if MJPC_SIDE_COM
  // the com-based quantities (subtree_com, cinert, cdof) are produced here, off the owner's critical path: the owner goes from
  // the kinematics straight into collision detection and needs them only for the contact Jacobians
  kinematics_rest(c);
  com_pos(c);
  flag_set(c.misc + HX_COM, t + 1);
#endif
  velocity_stage<NVT>(c, Kc, t + 1);            // helper 0 builds and factors M meanwhile (ph_inertia)
#else
  crb_and_factor<NVT>(c); PROF(c, 3);
  velocity_stage<NVT>(c, Kc, 0); PROF(c, 6);
#endif
  ctx_close(c);
}
#if MJPC_HELPER
// helper 0, concurrently with ph_constraints / ph_smooth: joint-space inertia and its factor
template <int NVT>
DEV_NOINLINE void ph_inertia(KP Kc, int t) {
  Ctx c; ctx_open(c, Kc, 1);
#if MJPC_SIDE_COM
  if (!flag_wait(c.misc + HX_COM, t + 1)) c.warning |= WARN_SYNC;
#endif
  crb_and_factor<NVT>(c);
  flag_set(c.misc + 22, t + 1);
  ctx_close(c);
}
#endif
// (inlining it into the kernel trades the ~100 callee-saved register saves per call for register pressure in the hot loops:
// measured 1 % slower with one candidate per CU; MJPC_INLINE_SOLVE is the A/B switch)
#ifdef MJPC_INLINE_SOLVE
#define DEV_SOLVE_PHASE DEV
#else
#define DEV_SOLVE_PHASE DEV_NOINLINE
#endif
template <int NVT>
DEV_SOLVE_PHASE void ph_solve(KP Kc, int last, int t) {
  Ctx c; ctx_open(c, Kc);
  c.hseq = t * 256;
  solve_constraints<NVT>(c); PROF(c, 8);
#if MJPC_HELPER
  // release the waves that wait for jobs: the workers of elliptic models (one packed word, solver_reg.h) / the Hessian builders of
  // the generic path
  if (MJPC_SOLVER_REG && NVT > 0) { if (c.M->cone == 1) flag_set(c.misc + HX_JOBW, JOBW(++c.hseq, 0, 0)); }
  else {
    if (LANE == 0) c.misc[HX_KIND] = 0;
    flag_set(c.misc + HX_JOB, ++c.hseq);
  }
#endif
  // (a step that already overflowed a buffer fails with that code alone: what the solver made of the truncated rows does not matter)
  if (!last && !(c.warning & (WARN_CONTACTFULL | WARN_CNSTRFULL)) && bad_values(c.qacc, c.M->nv)) c.warning |= WARN_BADQACC;
  ctx_close(c);
}
// role 0, models with noslip_iterations > 0: the noslip pass behind the Newton solve, a phase of its own (the solve phase keeps its
// register allocation; the job waves have been released by then, the pass is the owner's alone)
template <int NVT>
DEV_NOINLINE void ph_noslip(KP Kc, int last) {
  Ctx c; ctx_open(c, Kc);
  if (!(c.warning & (WARN_CONTACTFULL | WARN_CNSTRFULL | WARN_SYNC | WARN_BADQACC))) {
    noslip_pass<NVT>(c);
    if (!last && bad_values(c.qacc, c.M->nv)) c.warning |= WARN_BADQACC;
  }
  ctx_close(c);
}
#if MJPC_HELPER
template <int NVT>
DEV_NOINLINE void ph_solve_helper(KP Kc, int t) {
  Ctx c; ctx_open(c, Kc, 1);
  // helper k = wave k + 1 (each index is its own instantiation: the column / entry ranges are compile-time)
  int k = WAVE_ID() - 1;
  static_for<0, MJPC_NH>([&](auto Kc_) { constexpr int KK = decltype(Kc_)::value; if (k == KK) solver_helper_loop<NVT, KK>(c, t * 256); });
}
#endif

// residual (sensor callback at mjSTAGE_ACC), trace, cost; returns (cost, warning)
struct CostOut { double cost; int warning; };
DEV_NOINLINE CostOut ph_residual_cost(KP Kc, int t, int last) {        // role 1, concurrently with ph_solve
  Ctx c; ctx_open(c, Kc, 1);
  const KParams *K = c.K;
  const DevModel &M = *c.M;
  const DevTask &T = M.task;
  Rows R = out_rows(K);
  if (last) PFOR(k, M.nu) R.actions[t * M.nu + k] = (K->H > 1) ? c.ctrl[k] : 0.0;     // trajectory.cc:190-195
  if (t == 0 && cand_index() == 0 && K->frame) {          // host Task::Transition reads these after a simulation step
    int nb = M.nbody, ns = M.nsite;
    double *f = K->frame;
    PFOR(i, 3 * nb) { f[i] = c.xpos[i]; f[12 * nb + 3 * ns + i] = c.subtree_com[i]; f[15 * nb + 3 * ns + i] = c.subtree_linvel[i]; }
    PFOR(i, 9 * nb) f[3 * nb + i] = c.xmat[i];
    PFOR(i, 3 * ns) f[12 * nb + i] = c.site_xpos[i];
  }
  task_residual(c, c.residual); PROF(c, 9); PROFW(c, 9);
  PFOR(i, R.nr) R.residual[t * R.nr + i] = c.residual[i];
  PFOR(i, T.num_trace) {
    int id = MI(task.trace_objid)[i], ty = MI(task.trace_objtype)[i];
    const double *src = ty == 6 ? c.site_xpos + 3 * id : (ty == 5 ? c.geom_xpos + 3 * id : (ty == 1 ? c.xipos + 3 * id : c.xpos + 3 * id));
    d_copy3(R.trace + t * R.ntr + 3 * i, src);
  }
  CostOut o;
  o.cost = cost_value(c, c.residual);      // UpdateReturn (trajectory.cc:312-326) folded into the loop
  if (LANE == 0) R.costs[t] = o.cost;
  o.warning = c.warning;
  PROF(c, 10); PROFW(c, 10);
  ctx_close(c);
  return o;
}

// role 1, after the residual and still under the solver's shadow: factor M + h*diag(damping) for the implicit-damping
// Euler step into qL / Linv (M's own factor is no longer needed once qacc_smooth exists)
template <int NVT>
DEV void prefactor_body(Ctx &c) {
  const DevModel &M = *c.M;
  {
    int nv = M.nv, nvp = M.nvp;
    double h = M.timestep;
    PFOR(e, nv * nvp) { int i = e / nvp, j = e - i * nvp; c.qL[e] = c.qM[e] + ((i == j) ? h * MD(dof_damping)[i] : 0.0); }
    if (M.nidrv) {      // implicitfast: tendon damping and the velocity terms of the actuator biases (skipped while the force is clamped)
      SYNC();
      if (LANE == 0)
        for (int k = 0; k < M.nidrv; k++) {
          int i = MI(idrv_e)[3 * k], j = MI(idrv_e)[3 * k + 1], a = MI(idrv_e)[3 * k + 2];
          if (a >= 0 && MI(actuator_forcelimited)[a]) {
            double f = c.actuator_force[a];
            if (f <= MD(actuator_forcerange)[2 * a] || f >= MD(actuator_forcerange)[2 * a + 1]) continue;
          }
          double v = h * MD(idrv_c)[k];
          c.qL[i * nvp + j] += v;
          if (i != j) c.qL[j * nvp + i] += v;
        }
      SYNC();
    }
    chol_factor<NVT>(c.qL, c.Linv, c.scr_a, nv, nvp, c.M->tree_ok);
  }
}
// (a model with a noslip pass keeps M's own factor until the pass is over: its prefactor is the first thing of ph_integrate)
template <int NVT>
DEV_NOINLINE void ph_prefactor(KP Kc) {
  Ctx c; ctx_open(c, Kc, 1);
  const DevModel &M = *c.M;
  if (M.any_damping && !M.int_dense && M.noslip_iterations <= 0) prefactor_body<NVT>(c);
  PROFW(c, 11);
  ctx_close(c);
}
#if MJPC_HELPER
// role 1, once its own work of the solve phase is done: one more worker for the owner's per-iterate jobs (elliptic models)
template <int NVT>
DEV_NOINLINE void ph_side_worker(KP Kc, int t) {
  Ctx c; ctx_open(c, Kc, 1);
  if constexpr (NVT > 0 && MJPC_SOLVER_REG) { if (c.nefc > 0) side_worker<NVT>(c, t); }
}
#endif


// ---- the implicit integrators' dense path (DevModel::int_dense): implicitfast with fluid forces, and mjINT_IMPLICIT -------------
// A = M - h dF/dv is built dense in qL and LU-solved (mj_implicit: mjd_smooth_vel + mju_factorLUSparse, no pivoting):
//   + h diag(damping) + the tendon-damping / actuator-bias entries (idrv)                      [symmetric]
//   + h J^T R diag(visc_k + 2 quad_k |lvel_k|) R^T J per body for the inertia-box fluid forces  [symmetric, mjd_inertiaBoxFluid]
//   + h d qfrc_bias / d qvel for mjINT_IMPLICIT (mjd_rne_vel)                                   [not symmetric]
// qfrc_bias is an exact quadratic polynomial of qvel, so its derivative is taken as the central difference with step ONE of the
// bias forces themselves (no truncation error): lane (j, +-) runs its own serial RNE with qvel +- e_j, its per-body cvel / cacc /
// cfrc in a scratch carved from the constraint rows (dead after the solve), 18 doubles per body and lane, in batches of lanes.
DEV void implicit_rne_columns(Ctx &c, double *A, double h) {
  const DevModel &M = *c.M;
  const int nv = M.nv, nvp = M.nvp, nb = M.nbody;
  double *S = lds_base() + c.K->L.efc_J;
  int B = M.int_scratch / (18 * nb);                            // lanes (= columns) per batch
  B = B > NLANE ? NLANE : B;
#define S_(b, k) S[((b) * 18 + (k)) * B + LANE]
  for (int j0 = 0; j0 < nv; j0 += B) {
    const int j = j0 + LANE;
    if (LANE < B && j < nv) {
      for (int pass = 0; pass < 2; pass++) {
        const double sgn = pass ? -1.0 : 1.0;
        for (int b = 1; b < nb; b++) {
          const int p = MIH(body_parentid)[b];
          double cv[6], ca[6];
          for (int k = 0; k < 6; k++) { cv[k] = p ? S_(p, k) : 0.0; ca[k] = p ? S_(p, 6 + k) : (k >= 3 ? -M.gravity[k - 3] : 0.0); }
          int bda = MIH(body_dofadr)[b];
          for (int jn = MIH(body_jntadr)[b]; jn < MIH(body_jntadr)[b] + MIH(body_jntnum)[b]; jn++) {
            const int type = MIH(jnt_type)[jn];
            if (type == 0) {                                    // free: the translational dofs have no cdof_dot
              for (int k = 0; k < 3; k++) { double vq = c.qvel[bda + k] + (bda + k == j ? sgn : 0.0); for (int q = 0; q < 6; q++) cv[q] += c.cdof[6 * (bda + k) + q] * vq; }
              bda += 3;
            }
            if (type == 0 || type == 1) {                       // rotational triple: all three cdof_dot from the velocity before them
              double cd[3][6];
              for (int k = 0; k < 3; k++) d_crossmotion(cd[k], cv, c.cdof + 6 * (bda + k));
              for (int k = 0; k < 3; k++) {
                double vq = c.qvel[bda + k] + (bda + k == j ? sgn : 0.0);
                for (int q = 0; q < 6; q++) { ca[q] += cd[k][q] * vq; cv[q] += c.cdof[6 * (bda + k) + q] * vq; }
              }
              bda += 3;
            } else {
              double cd[6], vq = c.qvel[bda] + (bda == j ? sgn : 0.0);
              d_crossmotion(cd, cv, c.cdof + 6 * bda);
              for (int q = 0; q < 6; q++) { ca[q] += cd[q] * vq; cv[q] += c.cdof[6 * bda + q] * vq; }
              bda++;
            }
          }
          double t1[6], t2[6], t3[6];
          d_mulinertvec(t1, c.cinert + 10 * b, ca);
          d_mulinertvec(t2, c.cinert + 10 * b, cv);
          d_crossforce(t3, cv, t2);
          for (int k = 0; k < 6; k++) { S_(b, k) = cv[k]; S_(b, 6 + k) = ca[k]; S_(b, 12 + k) = t1[k] + t3[k]; }
        }
        for (int b = nb - 1; b > 0; b--) {
          const int p = MIH(body_parentid)[b];
          if (p > 0) for (int k = 0; k < 6; k++) S_(p, 12 + k) += S_(b, 12 + k);
        }
        for (int i = 0; i < nv; i++) {                          // column j of the matrix belongs to this lane
          const int bi = MIH(dof_bodyid)[i];
          double bias = 0;
          for (int k = 0; k < 6; k++) bias += c.cdof[6 * i + k] * S_(bi, 12 + k);
          A[i * nvp + j] += sgn * (h * 0.5) * bias;
        }
      }
    }
    SYNC();
  }
#undef S_
}
template <int NVT>
DEV void implicit_dense_solve(Ctx &c) {
  const DevModel &M = *c.M;
  const int nv = M.nv, nvp = M.nvp, nb = M.nbody;
  const double h = M.timestep;
  double *A = c.qL;
  PFOR(e, nv * nvp) { int i = e / nvp, j = e - i * nvp; A[e] = (j < nv) ? c.qM[(i >= j) ? e : j * nvp + i] + ((i == j) ? h * MD(dof_damping)[i] : 0.0) : 0.0; }
  SYNC();
  if (M.nidrv && LANE == 0)
    for (int k = 0; k < M.nidrv; k++) {
      int i = MI(idrv_e)[3 * k], j = MI(idrv_e)[3 * k + 1], a = MI(idrv_e)[3 * k + 2];
      if (a >= 0 && MI(actuator_forcelimited)[a]) {
        double f = c.actuator_force[a];
        if (f <= MD(actuator_forcerange)[2 * a] || f >= MD(actuator_forcerange)[2 * a + 1]) continue;
      }
      double v = h * MD(idrv_c)[k];
      A[i * nvp + j] += v;
      if (i != j) A[j * nvp + i] += v;
    }
  SYNC();
  if (M.fluid) {
    // per body: the diagonal of -d lfrc / d lvel in the inertial frame (scratch: 6 per body, from efc_J on)
    double *DL = lds_base() + c.K->L.efc_J;
    PFOR(b, nb) {
      double dl[6] = {0, 0, 0, 0, 0, 0};
      double mass = MDH(body_mass)[b];
      if (b > 0 && mass >= D_MINVAL) {
        const double *I = MDH(body_inertia) + 3 * b, *R = c.ximat + 9 * b;
        double box[3], off[3], vw[3], lvel[6];
        box[0] = sqrt(d_div(fmax(D_MINVAL, I[1] + I[2] - I[0]), mass) * 6.0);
        box[1] = sqrt(d_div(fmax(D_MINVAL, I[0] + I[2] - I[1]), mass) * 6.0);
        box[2] = sqrt(d_div(fmax(D_MINVAL, I[0] + I[1] - I[2]), mass) * 6.0);
        d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MIH(body_rootid)[b]);
        d_cross(vw, c.cvel + 6 * b, off);
        for (int k = 0; k < 3; k++) vw[k] += c.cvel[6 * b + 3 + k] - M.wind[k];
        d_mulmattvec3(lvel, R, c.cvel + 6 * b); d_mulmattvec3(lvel + 3, R, vw);
        if (M.viscosity > 0) {
          double diam = (box[0] + box[1] + box[2]) / 3.0;
          for (int k = 0; k < 3; k++) { dl[k] += D_PI * diam * diam * diam * M.viscosity; dl[3 + k] += 3.0 * D_PI * diam * M.viscosity; }
        }
        if (M.density > 0) {
          double b0 = box[0], b1 = box[1], b2 = box[2];
          dl[3] += 2 * 0.5 * M.density * b1 * b2 * fabs(lvel[3]);
          dl[4] += 2 * 0.5 * M.density * b0 * b2 * fabs(lvel[4]);
          dl[5] += 2 * 0.5 * M.density * b0 * b1 * fabs(lvel[5]);
          dl[0] += 2 * M.density * b0 * (b1 * b1 * b1 * b1 + b2 * b2 * b2 * b2) * fabs(lvel[0]) / 64.0;
          dl[1] += 2 * M.density * b1 * (b0 * b0 * b0 * b0 + b2 * b2 * b2 * b2) * fabs(lvel[1]) / 64.0;
          dl[2] += 2 * M.density * b2 * (b0 * b0 * b0 * b0 + b1 * b1 * b1 * b1) * fabs(lvel[2]) / 64.0;
        }
      }
      for (int k = 0; k < 6; k++) DL[6 * b + k] = dl[k];
    }
    SYNC();
    PFOR(e, nv * nv) {
      const int i = e / nv, j = e - i * nv;
      double acc = 0;
      for (int b = 1; b < nb; b++) {
        const unsigned long long dm = MDM()[b];
        if (!((dm >> i) & 1ull) || !((dm >> j) & 1ull)) continue;
        const double *R = c.ximat + 9 * b, *dl = DL + 6 * b;
        double off[3], t[3], wi[6], wj[6], li[6], lj[6];
        d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MIH(body_rootid)[b]);
        const double *ci = c.cdof + 6 * i, *cj = c.cdof + 6 * j;
        d_cross(t, ci, off); for (int k = 0; k < 3; k++) { wi[k] = ci[k]; wi[3 + k] = ci[3 + k] + t[k]; }
        d_cross(t, cj, off); for (int k = 0; k < 3; k++) { wj[k] = cj[k]; wj[3 + k] = cj[3 + k] + t[k]; }
        d_mulmattvec3(li, R, wi); d_mulmattvec3(li + 3, R, wi + 3);
        d_mulmattvec3(lj, R, wj); d_mulmattvec3(lj + 3, R, wj + 3);
        for (int k = 0; k < 6; k++) acc += li[k] * dl[k] * lj[k];
      }
      A[i * nvp + j] += h * acc;
    }
    SYNC();
  }
  if (M.int_dense == 2) implicit_rne_columns(c, A, h);
  // LU without pivoting, in place (unit lower factor below the diagonal), then the two substitutions on Mgrad
  for (int k = 0; k < nv; k++) {
    SYNC();
    const double piv = A[k * nvp + k];
    PFOR(ii, nv - k - 1) { int i = k + 1 + ii; A[i * nvp + k] = d_div(A[i * nvp + k], piv); }
    SYNC();
    PFOR(e, (nv - k - 1) * (nv - k - 1)) {
      int i = k + 1 + e / (nv - k - 1), j = k + 1 + e % (nv - k - 1);
      A[i * nvp + j] -= A[i * nvp + k] * A[k * nvp + j];
    }
  }
  SYNC();
  double *x = c.Mgrad;
  if (LANE == 0) {
    for (int i = 0; i < nv; i++) { double s = x[i]; for (int j = 0; j < i; j++) s -= A[i * nvp + j] * x[j]; x[i] = s; }
    for (int i = nv - 1; i >= 0; i--) { double s = x[i]; for (int j = i + 1; j < nv; j++) s -= A[i * nvp + j] * x[j]; x[i] = d_div(s, A[i * nvp + i]); }
  }
  SYNC();
}

// mj_Euler with implicit joint damping, then record state[t+1]
template <int NVT>
DEV_NOINLINE void ph_integrate(KP Kc, int t) {
  Ctx c; ctx_open(c, Kc);
  const KParams *K = c.K;
  const DevModel &M = *c.M;
  Rows R = out_rows(K);
  int nq = M.nq, nv = M.nv, nvp = M.nvp;
  double h = M.timestep;
  if (M.noslip_iterations <= 0) { PFOR(i, nv) c.qacc_ws[i] = c.qacc[i]; }      // (with a noslip pass the warm start is the Newton solution it saved)
  else if (M.any_damping && !M.int_dense) { prefactor_body<NVT>(c); SYNC(); }
  if (M.int_dense) {
    PFOR(i, nv) c.Mgrad[i] = c.qfrc_smooth[i] + c.qfrc_constraint[i];
    SYNC();
    implicit_dense_solve<NVT>(c);
    PFOR(i, nv) c.qvel[i] += h * c.Mgrad[i];
  } else if (M.any_damping) {
    PFOR(i, nv) c.Mgrad[i] = c.qfrc_smooth[i] + c.qfrc_constraint[i];
    chol_solve<NVT>(c.qL, c.Linv, c.Mgrad, nv, nvp, c.M->tree_ok);       // factor of M + h*B from ph_prefactor
    PFOR(i, nv) c.qvel[i] += h * c.Mgrad[i];
  } else {
    PFOR(i, nv) c.qvel[i] += h * c.qacc[i];
  }
  SYNC();
  PFOR(j, M.njnt) {
    int qa = MIH(jnt_qposadr)[j], da = MIH(jnt_dofadr)[j], type = MIH(jnt_type)[j];
    if (type == 0) {
      for (int k = 0; k < 3; k++) c.qpos[qa + k] += h * c.qvel[da + k];
      d_quatintegrate(c.qpos + qa + 3, c.qvel + da + 3, h);
    } else if (type == 1) {
      d_quatintegrate(c.qpos + qa, c.qvel + da, h);
    } else c.qpos[qa] += h * c.qvel[da];
  }
  if (M.na) {       // mj_advance / mj_nextActivation: act += h act_dot (filterexact: exact decay over h), clamped to actrange
    PFOR(i, M.nu) {
      int dt = MI(actuator_dyntype)[i];
      if (dt) {
        int a = MI(actuator_actadr)[i];
        double act = C_ACT(c)[a], ad = C_ACTDOT(c)[a];
        if (dt == DYN_FILTEREXACT) { double tau = fmax(D_MINVAL, MD(actuator_dynprm)[i]); act += ad * tau * (1 - exp(-h / tau)); }
        else act += h * ad;
        if (MI(actuator_actlimited)[i]) act = d_clip(act, MD(actuator_actrange)[2 * i], MD(actuator_actrange)[2 * i + 1]);
        C_ACT(c)[a] = act;
        R.states[(t + 1) * R.ds + nq + nv + a] = act;
      }
    }
  }
  c.time += h;
  SYNC();
  PFOR(i, nq) R.states[(t + 1) * R.ds + i] = c.qpos[i];
  PFOR(i, nv) R.states[(t + 1) * R.ds + nq + i] = c.qvel[i];
  if (LANE == 0) R.times[t + 1] = c.time;
  PROF(c, 11);
  ctx_close(c);
}

// checkpoint layout (doubles): [0] valid, [1] step, [2] time, [3] cost sum before that step, [4..6] diag counters, then qpos, qvel,
// qacc_warmstart, act
#define CKPT_HEAD 7
DEV_NOINLINE void ph_finish(KP Kc, double total, int failure, int t_fail, double total_before) {        // role 1 (it holds the cost sum)
  Ctx c; ctx_open(c, Kc, 1);
  const KParams *K = c.K;
  int r = cand_index(), H = K->H;
  if (K->tier > 0 && K->ckpt) {
    // dense tier: a candidate stopped by a full contact / row buffer hands its state at the failing step to the retry launch
    // (positions and velocities are those the failing step started from: integration had not happened; not with force noise,
    // whose Ornstein-Uhlenbeck state was already advanced)
    double *ck = K->ckpt + (size_t)r * K->ckpt_stride;
    const DevModel &M = *c.M;
    int resumable = failure && (c.warning & (WARN_CONTACTFULL | WARN_CNSTRFULL)) && !(c.warning & ~(WARN_CONTACTFULL | WARN_CNSTRFULL)) && !(K->xfrc_std > 0);
    if (LANE == 0) {
      ck[0] = resumable ? 1.0 : 0.0; ck[1] = (double)t_fail; ck[2] = c.time; ck[3] = total_before;
      ck[4] = (double)c.misc[5]; ck[5] = (double)c.misc[6]; ck[6] = (double)c.misc[7];
    }
    if (resumable) {
      PFOR(i, M.nq) ck[CKPT_HEAD + i] = c.qpos[i];
      PFOR(i, M.nv) { ck[CKPT_HEAD + M.nq + i] = c.qvel[i]; ck[CKPT_HEAD + M.nq + M.nv + i] = c.qacc_ws[i]; }
      if (M.na) PFOR(i, M.na) ck[CKPT_HEAD + M.nq + 2 * M.nv + i] = C_ACT(c)[i];
    }
  }
  if (LANE == 0) {
    K->returns[r] = failure ? 1.0e6 : total / (H > 1 ? H : 1);
    K->failure[r] = failure ? (c.warning ? c.warning : 1) : 0;
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
    if (K->prof) for (int q = 0; q < NPROF; q++) K->prof[(size_t)r * NPROF + q] = c.prof[q];
#endif
    if (K->diag) { K->diag[4 * r] = c.misc[5]; K->diag[4 * r + 1] = c.misc[6]; K->diag[4 * r + 2] = c.misc[7]; K->diag[4 * r + 3] = c.warning; }
  }
}

// owner wave, retry launch: state of the checkpointed step into LDS (everything else was set up by ph_init)
DEV_NOINLINE void ph_resume(KP Kc) {
  Ctx c; ctx_open(c, Kc);
  const KParams *K = c.K;
  const DevModel &M = *c.M;
  const double *ck = K->ckpt + (size_t)cand_index() * K->ckpt_stride;
  PFOR(i, M.nq) c.qpos[i] = ck[CKPT_HEAD + i];
  PFOR(i, M.nv) { c.qvel[i] = ck[CKPT_HEAD + M.nq + i]; c.qacc_ws[i] = ck[CKPT_HEAD + M.nq + M.nv + i]; }
  if (M.na) PFOR(i, M.na) C_ACT(c)[i] = ck[CKPT_HEAD + M.nq + 2 * M.nv + i];
  if (LANE == 0) { c.misc[5] = (int)ck[4]; c.misc[6] = (int)ck[5]; c.misc[7] = (int)ck[6]; }
  c.time = ck[2];
  ctx_close(c);
}

// ======================================================================================
// the whole rollout of the workgroup's candidate  (trajectory.cc:100-210 + 312-326)
// ======================================================================================
template <int NVT>
DEV void rollout(KP Kc) {
  int H = Kc->H;
  const bool r0 = ROLE0, r1 = ROLE1;
  if (r0) ph_init(Kc);
  XBAR();
  const int *misc = (const int *)(lds_base() + Kc->L.ints) + Kc->L.i_misc;
  double total = 0, total_before = 0;
  int failure = 0, t0 = 0, t_fail = 0;
  if (Kc->retry && Kc->ckpt) {
    // retry launch of the capacity tiers: pick the candidate up at the step where the dense tier ran out of room
    const double *ck = Kc->ckpt + (size_t)cand_index() * Kc->ckpt_stride;
    if (ck[0] != 0.0) {
      t0 = (int)ck[1]; total = ck[3];
      if (r0) ph_resume(Kc);
      XBAR();
    }
  }
#if defined(MJPC_PROFILE) && !defined(MJPC_EMU)
  long long *rprof = (long long *)(lds_base() + Kc->L.prof);
#define RPROF(i) do { if (LANE == 0 && WAVE_ID() == MJPC_PROFILE_WAVE) PROF_STAMP(rprof, i); } while (0)
#else
#define RPROF(i) ((void)0)
#endif
  for (int t = t0; t < H; t++) {
    int last = (t == H - 1);
    t_fail = t; total_before = total;
    if (r0) ph_head<NVT>(Kc, t, last);
    XBAR(); RPROF(2);
    if (uniform_i(misc[10])) { failure = 1; break; }
    if (r0) ph_constraints(Kc, t);
    if (r1) ph_smooth<NVT>(Kc, t);
#if MJPC_HELPER
    if (ROLEH && WAVE_ID() == 1) ph_inertia<NVT>(Kc, t);
    if (ROLEH && WAVE_ID() == MJPC_WAVES - 2) ph_noncontact(Kc, t);
#endif
    XBAR(); RPROF(3);
    if (r0) { ph_solve<NVT>(Kc, last, t); if (Kc->M.noslip_iterations > 0) ph_noslip<NVT>(Kc, last); }
#if MJPC_HELPER
    if (ROLEH) ph_solve_helper<NVT>(Kc, t);
#endif
    if (r1) {
      CostOut o = ph_residual_cost(Kc, t, last); total += o.cost;
      if (!last) ph_prefactor<NVT>(Kc);
#if MJPC_HELPER
      ph_side_worker<NVT>(Kc, t);
#endif
    }
    XBAR(); RPROF(6);
    if (uniform_i(misc[3]) | uniform_i(misc[11])) { failure = 1; break; }
    if (r0 && !last) ph_integrate<NVT>(Kc, t);
  }
  XBAR();
  if (r1) ph_finish(Kc, total, failure, t_fail, total_before);
}
