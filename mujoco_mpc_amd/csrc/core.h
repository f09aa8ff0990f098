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

// ======================================================================================
// position stage
// ======================================================================================
// Forward kinematics in three stages, so that only the parent -> child composition sits in the per-level chain:
//   A (lane per joint)  local joint rotations: sin/cos of the hinge angles, ball quaternions normalised  -> jq (scratch)
//   B (levels)          xquat / xpos / xmat of the bodies, joint axes and anchors
//   C (lane per body / geom / site)  inertial, geom and site frames
// Same arithmetic per quantity as the one-stage form.
DEV void kin_joint_local(Ctx &c, int j, double *jq) {
  const DevModel &M = *c.M;
  int type = MI(jnt_type)[j], qa = MI(jnt_qposadr)[j];
  if (type == 3) {
    double ax[3];
    d_copy3(ax, MD(jnt_axis) + 3 * j);
    d_axisangle2quat(jq + 4 * j, ax, c.qpos[qa] - MD(qpos0)[qa]);
  } else if (type == 1) {
    d_normalize4(c.qpos + qa);
    d_copy4(jq + 4 * j, c.qpos + qa);
  } else if (type == 0) {
    d_normalize4(c.qpos + qa + 3);
  }
}
// pose of body i from its parent's pose held in registers (have_parent: the parent is not the world); `store`: i is the body
// this lane is responsible for, its pose and its joints' anchors / axes go to LDS
DEV void kin_compose(Ctx &c, int i, const double *jq, const double *ppos, const double *pquat, const double *pmat, int have_parent,
                     double *xpos, double *xquat, double *xm, int store) {
  const DevModel &M = *c.M;
  int jntnum = MI(body_jntnum)[i], jntadr = MI(body_jntadr)[i];
  int mid = MI(body_mocapid)[i];
  if (mid >= 0) {
    d_copy3(xpos, c.mocap_pos + 3 * mid);
    d_copy4(xquat, c.mocap_quat + 4 * mid);
    d_normalize4(xquat);
  } else if (jntnum == 1 && MI(jnt_type)[jntadr] == 0) {
    int qa = MI(jnt_qposadr)[jntadr];
    d_copy3(xpos, c.qpos + qa);
    d_copy4(xquat, c.qpos + qa + 3);              // normalised in place by kin_joint_local
    if (store) {
      d_copy3(c.xanchor + 3 * jntadr, xpos);
      d_copy3(c.xaxis + 3 * jntadr, MD(jnt_axis) + 3 * jntadr);
    }
  } else {
    if (have_parent) {
      d_mulmatvec3(xpos, pmat, MD(body_pos) + 3 * i);
      d_add3(xpos, xpos, ppos);
      d_mulquat(xquat, pquat, MD(body_quat) + 4 * i);
    } else {
      d_copy3(xpos, MD(body_pos) + 3 * i);
      d_copy4(xquat, MD(body_quat) + 4 * i);
    }
    if (jntnum > 0) {
      double m[9];
      d_quat2mat(m, xquat);                       // rotation of the frame the next joint is expressed in
      for (int j = jntadr; j < jntadr + jntnum; j++) {
        int qa = MI(jnt_qposadr)[j], type = MI(jnt_type)[j];
        double vec[3], ax[3], jp[3], xax[3], xan[3];
        d_copy3(ax, MD(jnt_axis) + 3 * j); d_copy3(jp, MD(jnt_pos) + 3 * j);
        d_mulmatvec3(xax, m, ax);
        d_mulmatvec3(vec, m, jp);
        d_add3(xan, vec, xpos);
        if (store) { d_copy3(c.xaxis + 3 * j, xax); d_copy3(c.xanchor + 3 * j, xan); }
        if (type == 2) {
          d_addtoscl3(xpos, xax, c.qpos[qa] - MD(qpos0)[qa]);
        } else {
          double qloc[4], t[4];
          d_copy4(qloc, jq + 4 * j);
          d_mulquat(t, xquat, qloc);
          d_copy4(xquat, t);
          d_quat2mat(m, xquat);
          d_mulmatvec3(vec, m, jp);
          d_sub3(xpos, xan, vec);
        }
      }
    }
  }
  d_normalize4(xquat);
  d_quat2mat(xm, xquat);
  if (store) {
    d_copy3(c.xpos + 3 * i, xpos);
    d_copy4(c.xquat + 4 * i, xquat);
    for (int k = 0; k < 9; k++) c.xmat[9 * i + k] = xm[k];
  }
}

// one candidate per CU: the side wave computes the com-based quantities while the owner is already in collision detection
// (C2 -1.9 %); with two workgroups per CU the waves share their SIMDs and the extra hand-shake only costs (+0.6 .. 1.7 %)
#if MJPC_HELPER && !defined(MJPC_LEAN_LDS)
#define MJPC_SIDE_COM 1
#else
#define MJPC_SIDE_COM 0
#endif
// the poses nothing in collision detection reads: inertial frames and sites (with MJPC_SIDE_COM they are the side wave's work)
DEV void kin_frames_sites(Ctx &c) {
  const DevModel &M = *c.M;
  PFOR(i, M.nbody) {
    if (i == 0) continue;
    double v[3], q[4], ip[3], iq[4], xm[9];
    d_copy3(ip, MD(body_ipos) + 3 * i); d_copy4(iq, MD(body_iquat) + 4 * i);
    d_mulmatvec3(v, c.xmat + 9 * i, ip);
    d_add3(c.xipos + 3 * i, v, c.xpos + 3 * i);
    d_mulquat(q, c.xquat + 4 * i, iq);
    d_quat2mat(xm, q);
    for (int k = 0; k < 9; k++) c.ximat[9 * i + k] = xm[k];
  }
  PFOR(s, M.nsite) {
    int b = MI(site_bodyid)[s];
    double v[3], sp[3];
    d_copy3(sp, MD(site_pos) + 3 * s);
    d_mulmatvec3(v, c.xmat + 9 * b, sp);
    d_add3(c.site_xpos + 3 * s, v, c.xpos + 3 * b);
  }
}
DEV void kinematics_rest(Ctx &c) { kin_frames_sites(c); SYNC(); }

DEV void kinematics(Ctx &c) {
  const DevModel &M = *c.M;
  double *jq = c.cdof_dot;                        // scratch: rebuilt by the velocity stage after the next barrier
  PFOR(j, M.njnt) kin_joint_local(c, j, jq);
  SYNC();
  // one lane per body walks its own ancestor chain with the running pose in registers: the ancestors' poses are recomputed per
  // lane (same arithmetic, same results) instead of being handed down through LDS with a barrier per tree level
  PFOR(b, M.nbody) {
    if (b == 0) continue;
    double ppos[3], pquat[4], pmat[9], xpos[3], xquat[4], xm[9];
    int have = 0;
    for (int q = MI(chain_adr)[b]; q < MI(chain_adr)[b + 1]; q++) {
      int a = MI(chain_list)[q];
      kin_compose(c, a, jq, ppos, pquat, pmat, have, xpos, xquat, xm, a == b);
      d_copy3(ppos, xpos); d_copy4(pquat, xquat);
      for (int k = 0; k < 9; k++) pmat[k] = xm[k];
      have = 1;
    }
  }
  SYNC();
#if !MJPC_SIDE_COM
  kin_frames_sites(c);
#endif
  PFOR(g, M.ngeom) {
    int b = MI(geom_bodyid)[g];
    double v[3], q[4], gp[3], gq[4], xm[9];
    d_copy3(gp, MD(geom_pos) + 3 * g); d_copy4(gq, MD(geom_quat) + 4 * g);
    d_mulmatvec3(v, c.xmat + 9 * b, gp);
    d_add3(c.geom_xpos + 3 * g, v, c.xpos + 3 * b);
    d_mulquat(q, c.xquat + 4 * b, gq);
    d_quat2mat(xm, q);
    for (int k = 0; k < 9; k++) c.geom_xmat[9 * g + k] = xm[k];
  }
  SYNC();
}
DEV void com_pos(Ctx &c) {
  const DevModel &M = *c.M;
  PFOR(b, M.nbody) {
    double s[3] = {0, 0, 0};
    for (int k = MI(subtree_adr)[b]; k < MI(subtree_adr)[b + 1]; k++) {
      int cb = MI(subtree_list)[k];
      d_addtoscl3(s, c.xipos + 3 * cb, MD(body_mass)[cb]);
    }
    double sm = MD(body_subtreemass)[b];
    if (sm < D_MINVAL) d_copy3(c.subtree_com + 3 * b, c.xipos + 3 * b);
    else d_scl3(c.subtree_com + 3 * b, s, 1.0 / sm);
  }
  SYNC();
  PFOR(b, M.nbody) {
    if (b == 0) { for (int k = 0; k < 10; k++) c.cinert[k] = 0; continue; }
    double off[3], ine[3], r[10];
    d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MI(body_rootid)[b]);
    d_copy3(ine, MD(body_inertia) + 3 * b);
    d_inertcom(r, ine, c.ximat + 9 * b, off, MD(body_mass)[b]);
    for (int k = 0; k < 10; k++) c.cinert[10 * b + k] = r[k];
  }
  PFOR(j, M.njnt) {
    int b = MI(jnt_bodyid)[j], da = MI(jnt_dofadr)[j], type = MI(jnt_type)[j];
    double off[3];
    d_sub3(off, c.subtree_com + 3 * MI(body_rootid)[b], c.xanchor + 3 * j);
    int skip = 0;
    if (type == 0) {
      for (int k = 0; k < 18; k++) c.cdof[6 * da + k] = 0;
      for (int k = 0; k < 3; k++) c.cdof[6 * (da + k) + 3 + k] = 1;
      skip = 3;
    }
    if (type == 0 || type == 1) {
      const double *xm = c.xmat + 9 * b;
      for (int k = 0; k < 3; k++) {
        double ax[3] = {xm[k], xm[k + 3], xm[k + 6]}, cr[3];
        double *cd = c.cdof + 6 * (da + k + skip);
        d_cross(cr, ax, off);
        d_copy3(cd, ax); d_copy3(cd + 3, cr);
      }
    } else if (type == 2) {
      c.cdof[6 * da] = 0; c.cdof[6 * da + 1] = 0; c.cdof[6 * da + 2] = 0;
      d_copy3(c.cdof + 6 * da + 3, c.xaxis + 3 * j);
    } else {
      double cr[3];
      d_cross(cr, c.xaxis + 3 * j, off);
      d_copy3(c.cdof + 6 * da, c.xaxis + 3 * j);
      d_copy3(c.cdof + 6 * da + 3, cr);
    }
  }
  SYNC();
}

template <int NVT>
DEV void crb_and_factor(Ctx &c) {
  const DevModel &M = *c.M;
  int nv = M.nv, nvp = M.nvp;
  PFOR(e, M.nbody * 10) {
    int b = e / 10, k = e - 10 * b;
    double s = 0;
    if (b > 0) for (int q = MI(subtree_adr)[b]; q < MI(subtree_adr)[b + 1]; q++) s += c.cinert[10 * MI(subtree_list)[q] + k];
    c.crb[e] = s;
  }
  SYNC();
  PROFW(c, 1);
  PFOR(p, M.nmpair) {
    int i = MI(mpair_i)[p], j = MI(mpair_j)[p];
    double buf[6];
    d_mulinertvec(buf, c.crb + 10 * MI(dof_bodyid)[i], c.cdof + 6 * i);
    const double *cj = c.cdof + 6 * j;
    double v = cj[0]*buf[0] + cj[1]*buf[1] + cj[2]*buf[2] + cj[3]*buf[3] + cj[4]*buf[4] + cj[5]*buf[5];
    if (i == j) v += MD(dof_armature)[i];
    c.qM[i * nvp + j] = v; c.qM[j * nvp + i] = v;
  }
  SYNC();
  PFOR(e, nv * nvp) c.qL[e] = c.qM[e];
  PROFW(c, 4);
  chol_factor<NVT>(c.qL, c.Linv, c.vtmp, nv, nvp, c.M->tree_ok);
  PROFW(c, 5);
}

// ======================================================================================
// collision: bounding-sphere filter -> ordered compaction -> analytic narrow phase
// ======================================================================================
struct NPCon { double dist, pos[3], frame[6]; };
// The per-lane contact list (at most 4) must stay in registers: a run-time index would send the whole array to scratch
// memory, so slots are written / read through compile-time indices and value selects (no loops: the indices must be
// constants before the first SROA run).
DEV void np_sel(NPCon &d, const NPCon &v, bool p) {
  d.dist = p ? v.dist : d.dist;
  d.pos[0] = p ? v.pos[0] : d.pos[0]; d.pos[1] = p ? v.pos[1] : d.pos[1]; d.pos[2] = p ? v.pos[2] : d.pos[2];
  d.frame[0] = p ? v.frame[0] : d.frame[0]; d.frame[1] = p ? v.frame[1] : d.frame[1]; d.frame[2] = p ? v.frame[2] : d.frame[2];
  d.frame[3] = p ? v.frame[3] : d.frame[3]; d.frame[4] = p ? v.frame[4] : d.frame[4]; d.frame[5] = p ? v.frame[5] : d.frame[5];
}
DEV void np_put(NPCon *con, int idx, const NPCon &v) {
  np_sel(con[0], v, idx == 0); np_sel(con[1], v, idx == 1); np_sel(con[2], v, idx == 2); np_sel(con[3], v, idx == 3);
}
DEV NPCon np_get(const NPCon *con, int idx) {
  NPCon v = con[0];
  np_sel(v, con[1], idx == 1); np_sel(v, con[2], idx == 2); np_sel(v, con[3], idx == 3);
  return v;
}

DEV int np_sphere_sphere(NPCon *con, double margin, const double *p1, double r1, const double *p2, double r2) {
  double dif[3];
  d_sub3(dif, p2, p1);
  double cdist = d_norm3(dif), dist = cdist - r1 - r2;
  if (dist > margin) return 0;
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  if (cdist < D_MINVAL) con->frame[0] = 1; else d_scl3(con->frame, dif, 1.0 / cdist);
  con->dist = dist;
  d_addscl3(con->pos, p1, con->frame, r1 + 0.5 * dist);
  return 1;
}
DEV int np_plane_sphere(NPCon *con, double margin, const double *pp, const double *n, const double *cen, double r) {
  double dif[3];
  d_sub3(dif, cen, pp);
  double dist = d_dot3(dif, n) - r;
  if (dist > margin) return 0;
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  d_copy3(con->frame, n);
  con->dist = dist;
  d_addscl3(con->pos, cen, n, -(r + 0.5 * dist));
  return 1;
}
DEV int np_plane_capsule(NPCon *con, double margin, const double *pp, const double *pm, const double *cp, const double *cm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, axis[3] = {cm[2], cm[5], cm[8]}, seg[3], e[3];
  int cnt = 0;
  d_scl3(seg, axis, size[1]);
  d_add3(e, cp, seg);
  NPCon t;
  if (np_plane_sphere(&t, margin, pp, n, e, size[0])) { d_copy3(t.frame + 3, axis); con[0] = t; cnt++; }
  d_sub3(e, cp, seg);
  if (np_plane_sphere(&t, margin, pp, n, e, size[0])) { d_copy3(t.frame + 3, axis); np_put(con, cnt, t); cnt++; }
  return cnt;
}
DEV int np_plane_box(NPCon *con, double margin, const double *pp, const double *pm, const double *bp, const double *bm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, dif[3];
  d_sub3(dif, bp, pp);
  double dist = d_dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double vec[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]};
    double corner[3];
    d_mulmatvec3(corner, bm, vec);
    double ldist = d_dot3(n, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    if (cnt >= 4) break;
    NPCon t, *q = &t;
    q->dist = dist + ldist;
    for (int k = 0; k < 6; k++) q->frame[k] = 0;
    d_copy3(q->frame, n);
    d_add3(corner, corner, bp);
    d_addscl3(q->pos, corner, n, -0.5 * q->dist);
    np_put(con, cnt, t);
    cnt++;
    if (cnt >= 4) break;
  }
  return cnt;
}
DEV int np_plane_cylinder(NPCon *con, double margin, const double *pp, const double *pm, const double *cp, const double *cm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, axis[3] = {cm[2], cm[5], cm[8]};
  double prjaxis = d_dot3(n, axis);
  if (prjaxis > 0) { d_scl3(axis, axis, -1); prjaxis = -prjaxis; }
  double vec[3];
  d_sub3(vec, cp, pp);
  double dist0 = d_dot3(vec, n);
  d_scl3(vec, axis, prjaxis); d_sub3(vec, vec, n);
  double len2 = d_dot3(vec, vec);
  if (len2 >= D_MINVAL) d_scl3(vec, vec, size[0] / sqrt(len2));
  else { vec[0] = cm[0] * size[0]; vec[1] = cm[3] * size[0]; vec[2] = cm[6] * size[0]; }
  double prjvec = d_dot3(vec, n);
  d_scl3(axis, axis, size[1]); prjaxis *= size[1];
  int cnt = 0;
  if (dist0 + prjaxis + prjvec <= margin) {
    NPCon *q = con; cnt = 1;
    q->dist = dist0 + prjaxis + prjvec;
    d_add3(q->pos, cp, vec); d_add3(q->pos, q->pos, axis); d_addtoscl3(q->pos, n, -0.5 * q->dist);
    for (int k = 0; k < 6; k++) q->frame[k] = 0;
    d_copy3(q->frame, n);
  } else return 0;
  if (dist0 - prjaxis + prjvec <= margin) {
    NPCon *q = con + 1; cnt = 2;
    q->dist = dist0 - prjaxis + prjvec;
    d_add3(q->pos, cp, vec); d_sub3(q->pos, q->pos, axis); d_addtoscl3(q->pos, n, -0.5 * q->dist);
    for (int k = 0; k < 6; k++) q->frame[k] = 0;
    d_copy3(q->frame, n);
  }
  double prjvec1 = -0.5 * prjvec;
  if (dist0 + prjaxis + prjvec1 <= margin) {
    double vec1[3];
    d_cross(vec1, vec, axis);
    d_normalize3(vec1);
    d_scl3(vec1, vec1, size[0] * sqrt(3.0) / 2);
    for (int s = -1; s <= 1; s += 2) {
      NPCon t, *q = &t;
      q->dist = dist0 + prjaxis + prjvec1;
      d_add3(q->pos, cp, axis); d_addtoscl3(q->pos, vec, -0.5); d_addtoscl3(q->pos, vec1, (double)s);
      d_addtoscl3(q->pos, n, -0.5 * q->dist);
      for (int k = 0; k < 6; k++) q->frame[k] = 0;
      d_copy3(q->frame, n);
      np_put(con, cnt, t);
      cnt++;
    }
  }
  return cnt;
}
DEV int np_sphere_capsule(NPCon *con, double margin, const double *sp, double sr, const double *cp, const double *cm, const double *csize) {
  double axis[3] = {cm[2], cm[5], cm[8]}, vec[3], pt[3];
  d_sub3(vec, sp, cp);
  double x = d_clip(d_dot3(axis, vec), -csize[1], csize[1]);
  d_addscl3(pt, cp, axis, x);
  return np_sphere_sphere(con, margin, sp, sr, pt, csize[0]);
}
DEV int np_capsule_capsule(NPCon *con, double margin, const double *p1, const double *m1, const double *s1,
                           const double *p2, const double *m2, const double *s2) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3];
  d_sub3(dif, p1, p2);
  double len1 = s1[1], len2 = s2[1];
  double ma = d_dot3(a1, a1), mb = -d_dot3(a1, a2), mc = d_dot3(a2, a2);
  double u = -d_dot3(a1, dif), v = d_dot3(a2, dif);
  double det = ma * mc - mb * mb;
  if (fabs(det) >= D_MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > len1) { x1 = len1; x2 = (v - mb * len1) / mc; }
    else if (x1 < -len1) { x1 = -len1; x2 = (v + mb * len1) / mc; }
    if (x2 > len2) { x2 = len2; x1 = d_clip((u - mb * len2) / ma, -len1, len1); }
    else if (x2 < -len2) { x2 = -len2; x1 = d_clip((u + mb * len2) / ma, -len1, len1); }
    double v1[3], v2[3];
    d_addscl3(v1, p1, a1, x1); d_addscl3(v2, p2, a2, x2);
    return np_sphere_sphere(con, margin, v1, s1[0], v2, s2[0]);
  }
  int cnt = 0;
  for (int s = -1; s <= 1 && cnt < 2; s += 2) {
    double e[3], w[3], pt[3];
    d_addscl3(e, p1, a1, s * len1);
    d_sub3(w, e, p2);
    double x = d_clip(d_dot3(a2, w), -len2, len2);
    d_addscl3(pt, p2, a2, x);
    NPCon t;
    if (np_sphere_sphere(&t, margin, e, s1[0], pt, s2[0])) { np_put(con, cnt, t); cnt++; }
  }
  return cnt;
}
DEV int np_sphere_box(NPCon *con, double margin, const double *sp, double sr, const double *bp, const double *bm, const double *bs) {
  double dif[3], cc[3], clamped[3];
  d_sub3(dif, sp, bp);
  d_mulmattvec3(cc, bm, dif);
  int inside = 1;
  for (int i = 0; i < 3; i++) { clamped[i] = d_clip(cc[i], -bs[i], bs[i]); if (clamped[i] != cc[i]) inside = 0; }
  double nloc[3], dist;
  if (!inside) {
    double dd[3]; d_sub3(dd, cc, clamped);
    double len = d_norm3(dd);
    dist = len - sr;
    if (dist > margin) return 0;
    d_scl3(nloc, dd, 1.0 / len);
  } else {
    int k = 0; double best = 1e300;
    for (int i = 0; i < 3; i++) { double pen = bs[i] - fabs(cc[i]); if (pen < best) { best = pen; k = i; } }
    nloc[0] = nloc[1] = nloc[2] = 0;
    double sgn = cc[k] >= 0 ? 1.0 : -1.0;
    if (k == 0) nloc[0] = sgn; else if (k == 1) nloc[1] = sgn; else nloc[2] = sgn;
    d_copy3(clamped, cc);
    if (k == 0) clamped[0] = sgn * bs[0]; else if (k == 1) clamped[1] = sgn * bs[1]; else clamped[2] = sgn * bs[2];
    dist = -best - sr;
  }
  double nw[3], surf[3];
  d_mulmatvec3(nw, bm, nloc);
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  d_scl3(con->frame, nw, -1);
  con->dist = dist;
  d_mulmatvec3(surf, bm, clamped); d_add3(surf, surf, bp);
  d_addscl3(con->pos, surf, nw, 0.5 * dist);
  return 1;
}

// capsule (geom1) vs box (geom2): closest point of the segment to the box by a fixed-count bisection of the monotone derivative
// (the CPU checker restates the same construction with the same operations), then sphere-box there and at the far end cap
DEV double capsule_box_g(const double *p0, const double *a, double h, const double *b, double s) {
  double g = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double q = p0[i] + (s * h) * a[i];
    double e = fabs(q) - b[i];
    g += (e > 0) ? (q > 0 ? e : -e) * a[i] : 0.0;
  }
  return g;
}
DEV int np_capsule_box(NPCon *con, double margin, const double *cp, const double *cm, const double *cs,
                       const double *bp, const double *bm, const double *bs) {
  double axis[3] = {cm[2], cm[5], cm[8]}, dif[3], p0[3], a[3];
  d_sub3(dif, cp, bp);
  d_mulmattvec3(p0, bm, dif);
  d_mulmattvec3(a, bm, axis);
  double h = cs[1], sstar;
  if (capsule_box_g(p0, a, h, bs, -1.0) >= 0) sstar = -1.0;
  else if (capsule_box_g(p0, a, h, bs, 1.0) <= 0) sstar = 1.0;
  else {
    double lo = -1.0, hi = 1.0;
    for (int it = 0; it < 48; it++) {
      double mid = 0.5 * (lo + hi);
      if (capsule_box_g(p0, a, h, bs, mid) < 0) lo = mid; else hi = mid;
    }
    sstar = 0.5 * (lo + hi);
  }
  int cnt = 0;
  double pt[3];
  NPCon t;
  d_addscl3(pt, cp, axis, sstar * h);
  if (np_sphere_box(&t, margin, pt, cs[0], bp, bm, bs)) { con[0] = t; cnt++; }
  double s2 = sstar <= 0 ? 1.0 : -1.0;
  d_addscl3(pt, cp, axis, s2 * h);
  if (np_sphere_box(&t, margin, pt, cs[0], bp, bm, bs)) { np_put(con, cnt, t); cnt++; }
  return cnt;
}

// box (geom1 = A) vs box (geom2 = B): separating-axis test, then reference-face clipping (<= 4 contacts) or one edge-edge
// contact (DESIGN.md section 6 describes the construction).  Axis-indexed accesses go through selects so that
// nothing needs a run-time-indexed private array.
#define BB_TOL 1e-9
DEV double sel3(const double *v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }
DEV void col3(double *r, const double *m, int k) { r[0] = k == 0 ? m[0] : (k == 1 ? m[1] : m[2]); r[1] = k == 0 ? m[3] : (k == 1 ? m[4] : m[5]); r[2] = k == 0 ? m[6] : (k == 1 ? m[7] : m[8]); }
struct BBSel { double x, y, d; int ok; };
// candidate q of the face case: 0-3 incident vertices, 4-7 reference corners, 8-23 edge crossings
struct BBFace { double c0[3], e1[3], e2[3], hu, hv, det; };
DEV BBSel bb_candidate(const BBFace &f, int q) {
  BBSel o; o.ok = 0; o.x = 0; o.y = 0; o.d = 0;
  if (q < 4) {
    double s1 = (q == 0 || q == 3) ? -1.0 : 1.0, s2 = (q < 2) ? -1.0 : 1.0;
    double x = f.c0[0] + s1 * f.e1[0] + s2 * f.e2[0], y = f.c0[1] + s1 * f.e1[1] + s2 * f.e2[1], d = f.c0[2] + s1 * f.e1[2] + s2 * f.e2[2];
    if (fabs(x) <= f.hu + BB_TOL && fabs(y) <= f.hv + BB_TOL) { o.ok = 1; o.x = x; o.y = y; o.d = d; }
  } else if (q < 8) {
    int k = q - 4;
    if (fabs(f.det) > 1e-14) {
      double cx = (k == 0 || k == 3) ? -f.hu : f.hu, cy = (k < 2) ? -f.hv : f.hv;
      double dx = cx - f.c0[0], dy = cy - f.c0[1];
      double al = (dx * f.e2[1] - dy * f.e2[0]) / f.det, be = (f.e1[0] * dy - f.e1[1] * dx) / f.det;
      if (fabs(al) <= 1.0 + BB_TOL && fabs(be) <= 1.0 + BB_TOL) { o.ok = 1; o.x = cx; o.y = cy; o.d = f.c0[2] + al * f.e1[2] + be * f.e2[2]; }
    }
  } else {
    int k = (q - 8) >> 2, e = (q - 8) & 3, k2 = (k + 1) & 3;
    double a1 = (k == 0 || k == 3) ? -1.0 : 1.0, a2 = (k < 2) ? -1.0 : 1.0, b1 = (k2 == 0 || k2 == 3) ? -1.0 : 1.0, b2 = (k2 < 2) ? -1.0 : 1.0;
    double px = f.c0[0] + a1 * f.e1[0] + a2 * f.e2[0], py = f.c0[1] + a1 * f.e1[1] + a2 * f.e2[1], pd = f.c0[2] + a1 * f.e1[2] + a2 * f.e2[2];
    double qx = f.c0[0] + b1 * f.e1[0] + b2 * f.e2[0], qy = f.c0[1] + b1 * f.e1[1] + b2 * f.e2[1], qd = f.c0[2] + b1 * f.e1[2] + b2 * f.e2[2];
    double dx = qx - px, dy = qy - py, dd = qd - pd;
    int xline = e < 2;
    double lim = (e & 1) ? 1.0 : -1.0;
    double num = xline ? lim * f.hu - px : lim * f.hv - py, den = xline ? dx : dy;
    if (!(fabs(den) < 1e-14)) {
      double s = num / den;
      if (!(s <= 0.0 || s >= 1.0)) {
        double ox = xline ? lim * f.hu : px + s * dx, oy = xline ? py + s * dy : lim * f.hv;
        if (!((xline ? fabs(oy) - f.hv : fabs(ox) - f.hu) > BB_TOL)) { o.ok = 1; o.x = ox; o.y = oy; o.d = pd + s * dd; }
      }
    }
  }
  return o;
}
DEV int np_box_box(NPCon *con, double margin, const double *pa, const double *ma, const double *sa,
                   const double *pb, const double *mb, const double *sb) {
  double R[9], AR[9], t[3], tb[3], dif[3];
  d_sub3(dif, pb, pa);
  d_mulmattvec3(t, ma, dif);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double r = ma[i] * mb[j] + ma[3 + i] * mb[3 + j] + ma[6 + i] * mb[6 + j];
      R[3 * i + j] = r; AR[3 * i + j] = fabs(r);
    }
#pragma unroll
  for (int j = 0; j < 3; j++) tb[j] = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
  double best = -1e300; int code = -1, sep = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double s = fabs(t[i]) - (sa[i] + sb[0] * AR[3 * i] + sb[1] * AR[3 * i + 1] + sb[2] * AR[3 * i + 2]);
    sep |= s > margin;
    if (s > best) { best = s; code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double s = fabs(tb[j]) - (sb[j] + sa[0] * AR[j] + sa[1] * AR[3 + j] + sa[2] * AR[6 + j]);
    sep |= s > margin;
    if (s > best) { best = s; code = 3 + j; }
  }
  if (sep) return 0;                 // separated along a face normal: most non-touching pairs leave here
  double ebest = -1e300; int ecode = -1;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double l2 = 1.0 - R[3 * i + j] * R[3 * i + j];
      if (l2 < 1e-6) continue;
      double proj = t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j];
      double ra = sa[i1] * AR[3 * i2 + j] + sa[i2] * AR[3 * i1 + j];
      double rb = sb[j1] * AR[3 * i + j2] + sb[j2] * AR[3 * i + j1];
      double s = (fabs(proj) - (ra + rb)) / sqrt(l2);
      sep |= s > margin;
      if (s > ebest) { ebest = s; ecode = 3 * i + j; }
    }
  if (sep) return 0;
  if (ecode >= 0 && ebest > best + 0.05 * fabs(best) + BB_TOL) {
    int i = ecode / 3, j = ecode - 3 * i;
    double ai[3], bj[3], n[3];
    col3(ai, ma, i); col3(bj, mb, j);
    d_cross(n, ai, bj);
    d_normalize3(n);
    if (d_dot3(n, dif) < 0) d_scl3(n, n, -1);
    double ea[3], eb[3];
    d_copy3(ea, pa); d_copy3(eb, pb);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double ak[3] = {ma[k], ma[3 + k], ma[6 + k]}, bk[3] = {mb[k], mb[3 + k], mb[6 + k]};
      if (k != i) d_addtoscl3(ea, ak, d_dot3(n, ak) > 0 ? sa[k] : -sa[k]);
      if (k != j) d_addtoscl3(eb, bk, d_dot3(n, bk) > 0 ? -sb[k] : sb[k]);
    }
    double w[3]; d_sub3(w, eb, ea);
    double cc = (i == 0 ? sel3(R, j) : (i == 1 ? sel3(R + 3, j) : sel3(R + 6, j)));
    double d1 = d_dot3(w, ai), d2 = d_dot3(w, bj), den = 1.0 - cc * cc;
    double sai = sel3(sa, i), sbj = sel3(sb, j);
    double u = d_clip((d1 - cc * d2) / den, -sai, sai);
    double v = d_clip((cc * d1 - d2) / den, -sbj, sbj);
    double qa[3], qb[3];
    d_addscl3(qa, ea, ai, u); d_addscl3(qb, eb, bj, v);
    d_sub3(w, qb, qa);
    double dist = d_dot3(w, n);
    if (dist > margin) return 0;
    for (int k = 0; k < 6; k++) con->frame[k] = 0;
    d_copy3(con->frame, n);
    con->dist = dist;
    con->pos[0] = 0.5 * (qa[0] + qb[0]); con->pos[1] = 0.5 * (qa[1] + qb[1]); con->pos[2] = 0.5 * (qa[2] + qb[2]);
    return 1;
  }
  // ---- face case
  int refA = code < 3, ax = refA ? code : code - 3;
  double pr[3], mr[9], sr[3], pq[3], mq[9], sq[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { pr[k] = refA ? pa[k] : pb[k]; sr[k] = refA ? sa[k] : sb[k]; pq[k] = refA ? pb[k] : pa[k]; sq[k] = refA ? sb[k] : sa[k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) { mr[k] = refA ? ma[k] : mb[k]; mq[k] = refA ? mb[k] : ma[k]; }
  double sgn = (refA ? sel3(t, ax) : -sel3(tb, ax)) >= 0 ? 1.0 : -1.0;
  int u1 = (ax + 1) % 3, u2 = (ax + 2) % 3;
  double n[3], ru[3], rv[3];
  col3(n, mr, ax); d_scl3(n, n, sgn);
  col3(ru, mr, u1); col3(rv, mr, u2);
  double hn = sel3(sr, ax);
  BBFace f;
  f.hu = sel3(sr, u1); f.hv = sel3(sr, u2);
  double nl[3];
  d_mulmattvec3(nl, mq, n);
  int k = 0; double amax = fabs(nl[0]);
  if (fabs(nl[1]) > amax) { amax = fabs(nl[1]); k = 1; }
  if (fabs(nl[2]) > amax) { amax = fabs(nl[2]); k = 2; }
  int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
  double qk[3], q1[3], q2[3], cen[3], rel[3];
  col3(qk, mq, k); col3(q1, mq, k1); col3(q2, mq, k2);
  double sqk = sel3(sq, k), sq1 = sel3(sq, k1), sq2 = sel3(sq, k2);
  d_addscl3(cen, pq, qk, sel3(nl, k) > 0 ? -sqk : sqk);
  d_sub3(rel, cen, pr);
  f.c0[0] = d_dot3(rel, ru); f.c0[1] = d_dot3(rel, rv); f.c0[2] = d_dot3(rel, n) - hn;
  f.e1[0] = sq1 * d_dot3(q1, ru); f.e1[1] = sq1 * d_dot3(q1, rv); f.e1[2] = sq1 * d_dot3(q1, n);
  f.e2[0] = sq2 * d_dot3(q2, ru); f.e2[1] = sq2 * d_dot3(q2, rv); f.e2[2] = sq2 * d_dot3(q2, n);
  f.det = f.e1[0] * f.e2[1] - f.e1[1] * f.e2[0];
  // selection passes regenerate the candidates instead of storing 24 of them: deepest, farthest from it, extreme on either side
  BBSel s0, s1, s2, s3; s0.ok = s1.ok = s2.ok = s3.ok = 0;
  double bd = 1e300;
  for (int q = 0; q < 24; q++) { BBSel cd = bb_candidate(f, q); if (cd.ok && cd.d <= margin && cd.d < bd) { bd = cd.d; s0 = cd; } }
  if (!s0.ok) return 0;
  double far = 1e-16;
  for (int q = 0; q < 24; q++) {
    BBSel cd = bb_candidate(f, q);
    if (cd.ok && cd.d <= margin) { double r2 = (cd.x - s0.x) * (cd.x - s0.x) + (cd.y - s0.y) * (cd.y - s0.y); if (r2 > far) { far = r2; s1 = cd; } }
  }
  if (s1.ok) {
    double lx = s1.x - s0.x, ly = s1.y - s0.y, amx = 1e-12, amn = -1e-12;
    for (int q = 0; q < 24; q++) {
      BBSel cd = bb_candidate(f, q);
      if (cd.ok && cd.d <= margin) {
        double ar = lx * (cd.y - s0.y) - ly * (cd.x - s0.x);
        if (ar > amx) { amx = ar; s2 = cd; }
        if (ar < amn) { amn = ar; s3 = cd; }
      }
    }
  }
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    BBSel cd = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
    if (!cd.ok) continue;
    NPCon o;
    double hgt = hn + cd.d - 0.5 * cd.d;
    o.pos[0] = pr[0] + cd.x * ru[0] + cd.y * rv[0] + hgt * n[0];
    o.pos[1] = pr[1] + cd.x * ru[1] + cd.y * rv[1] + hgt * n[1];
    o.pos[2] = pr[2] + cd.x * ru[2] + cd.y * rv[2] + hgt * n[2];
    o.dist = cd.d;
    for (int e = 0; e < 6; e++) o.frame[e] = 0;
    if (refA) d_copy3(o.frame, n); else d_scl3(o.frame, n, -1);
    np_put(con, cnt, o);
    cnt++;
  }
  return cnt;
}

// sphere (geom1) vs cylinder (geom2): closed form in the cylinder's frame (distance = hypot of the radial and axial excess)
DEV int np_sphere_cylinder(NPCon *con, double margin, const double *sp, double sr, const double *cp, const double *cm, const double *cs) {
  double dif[3], p[3];
  d_sub3(dif, sp, cp);
  d_mulmattvec3(p, cm, dif);
  double R = cs[0], h = cs[1];
  double rho = sqrt(p[0] * p[0] + p[1] * p[1]);
  double ux = rho > D_MINVAL ? p[0] / rho : 1.0, uy = rho > D_MINVAL ? p[1] / rho : 0.0;
  double er = rho - R, ez = fabs(p[2]) - h, sz = p[2] >= 0 ? 1.0 : -1.0;
  double q[3], nl[3], dist;
  if (er <= 0 && ez <= 0) {
    if (-er < -ez) { q[0] = ux * R; q[1] = uy * R; q[2] = p[2]; nl[0] = ux; nl[1] = uy; nl[2] = 0; dist = er - sr; }
    else { q[0] = p[0]; q[1] = p[1]; q[2] = sz * h; nl[0] = 0; nl[1] = 0; nl[2] = sz; dist = ez - sr; }
  } else {
    double cr = er > 0 ? R : rho;
    q[0] = ux * cr; q[1] = uy * cr; q[2] = ez > 0 ? sz * h : p[2];
    double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    double len = d_norm3(d);
    dist = len - sr;
    if (dist > margin) return 0;
    d_scl3(nl, d, 1.0 / len);
  }
  if (dist > margin) return 0;
  double nw[3], surf[3];
  d_mulmatvec3(nw, cm, nl);
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  d_scl3(con->frame, nw, -1);
  con->dist = dist;
  d_mulmatvec3(surf, cm, q); d_add3(surf, surf, cp);
  d_addscl3(con->pos, surf, nw, 0.5 * dist);
  return 1;
}
DEV double capsule_cylinder_g(const double *p0, const double *a, double hc, double R, double h, double s) {
  double q[3] = {p0[0] + (s * hc) * a[0], p0[1] + (s * hc) * a[1], p0[2] + (s * hc) * a[2]};
  double rho = sqrt(q[0] * q[0] + q[1] * q[1]);
  double g = 0, er = rho - R, ez = fabs(q[2]) - h;
  if (er > 0) g += er * (q[0] * a[0] + q[1] * a[1]) / rho;
  if (ez > 0) g += (q[2] > 0 ? ez : -ez) * a[2];
  return g;
}
DEV int np_capsule_cylinder(NPCon *con, double margin, const double *kp, const double *km, const double *ks,
                            const double *cp, const double *cm, const double *cs) {
  double axis[3] = {km[2], km[5], km[8]}, dif[3], p0[3], a[3];
  d_sub3(dif, kp, cp);
  d_mulmattvec3(p0, cm, dif);
  d_mulmattvec3(a, cm, axis);
  double hc = ks[1], sstar;
  if (capsule_cylinder_g(p0, a, hc, cs[0], cs[1], -1.0) >= 0) sstar = -1.0;
  else if (capsule_cylinder_g(p0, a, hc, cs[0], cs[1], 1.0) <= 0) sstar = 1.0;
  else {
    double lo = -1.0, hi = 1.0;
    for (int it = 0; it < 48; it++) {
      double mid = 0.5 * (lo + hi);
      if (capsule_cylinder_g(p0, a, hc, cs[0], cs[1], mid) < 0) lo = mid; else hi = mid;
    }
    sstar = 0.5 * (lo + hi);
  }
  int cnt = 0;
  double pt[3];
  NPCon t;
  d_addscl3(pt, kp, axis, sstar * hc);
  if (np_sphere_cylinder(&t, margin, pt, ks[0], cp, cm, cs)) { con[0] = t; cnt++; }
  double s2 = sstar <= 0 ? 1.0 : -1.0;
  d_addscl3(pt, kp, axis, s2 * hc);
  if (np_sphere_cylinder(&t, margin, pt, ks[0], cp, cm, cs)) { np_put(con, cnt, t); cnt++; }
  return cnt;
}

// ---- convex pairs without an analytic collider (cylinder-cylinder, cylinder-box): Minkowski portal refinement ("XenoCollide",
// the algorithm class MuJoCo reaches through libccd for these pairs; tolerance 1e-6, 50 iterations).  Both geoms are inflated
// by margin / 2, one contact per pair; the penetration is measured where the ray from the centres' difference through the
// origin leaves the Minkowski difference, the contact sits half way between the two witness points.
#define MPR_TOLERANCE 1e-6
#define MPR_ITERATIONS 50
struct MShape { int type; const double *pos, *mat, *size; double margin; const double *vert; int nvert; };
struct MSup { double v[3], v1[3], v2[3]; };
DEV void mpr_support1(const MShape &s, const double *dir, double *out) {
  double l[3], v[3];
  d_mulmattvec3(l, s.mat, dir);
  if (s.type == 6) {
    for (int k = 0; k < 3; k++) v[k] = l[k] >= 0 ? s.size[k] : -s.size[k];
  } else if (s.type == 5) {
    double n = sqrt(l[0] * l[0] + l[1] * l[1]);
    if (n > D_MINVAL) { v[0] = s.size[0] * l[0] / n; v[1] = s.size[0] * l[1] / n; } else { v[0] = 0; v[1] = 0; }
    v[2] = l[2] >= 0 ? s.size[1] : -s.size[1];
  } else if (s.type == 7) {       // convex mesh: the vertex farthest along l (first of equals); vertices are read from HBM / L2
    double best = -1e300; int bi = 0;
    for (int i = 0; i < s.nvert; i++) {
      double t = s.vert[3 * i] * l[0] + s.vert[3 * i + 1] * l[1] + s.vert[3 * i + 2] * l[2];
      if (t > best) { best = t; bi = i; }
    }
    v[0] = s.vert[3 * bi]; v[1] = s.vert[3 * bi + 1]; v[2] = s.vert[3 * bi + 2];
  } else if (s.type == 4) {
    double a = s.size[0] * s.size[0] * l[0], b = s.size[1] * s.size[1] * l[1], c = s.size[2] * s.size[2] * l[2];
    double n = sqrt(a * l[0] + b * l[1] + c * l[2]);
    if (n > D_MINVAL) { v[0] = a / n; v[1] = b / n; v[2] = c / n; } else { v[0] = 0; v[1] = 0; v[2] = 0; }
  } else if (s.type == 3) {
    v[0] = s.size[0] * l[0]; v[1] = s.size[0] * l[1]; v[2] = s.size[0] * l[2] + (l[2] >= 0 ? s.size[1] : -s.size[1]);
  } else {
    v[0] = s.size[0] * l[0]; v[1] = s.size[0] * l[1]; v[2] = s.size[0] * l[2];
  }
  d_mulmatvec3(out, s.mat, v);
  d_add3(out, out, s.pos);
  d_addtoscl3(out, dir, s.margin);
}
DEV void mpr_support(const MShape &a, const MShape &b, const double *dir, MSup &s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  mpr_support1(a, dir, s.v1);
  mpr_support1(b, nd, s.v2);
  d_sub3(s.v, s.v1, s.v2);
}
DEV void mpr_portal_dir(const MSup &p1, const MSup &p2, const MSup &p3, double *dir) {
  double a[3], b[3];
  d_sub3(a, p2.v, p1.v);
  d_sub3(b, p3.v, p1.v);
  d_cross(dir, a, b);
  d_normalize3(dir);
}
DEV int mpr_reach_tolerance(const MSup &p1, const MSup &p2, const MSup &p3, const MSup &v4, const double *dir) {
  double dv4 = d_dot3(v4.v, dir);
  double d1 = dv4 - d_dot3(p1.v, dir), d2 = dv4 - d_dot3(p2.v, dir), d3 = dv4 - d_dot3(p3.v, dir);
  double dm = fmin(d1, fmin(d2, d3));
  return dm <= MPR_TOLERANCE;
}
DEV void mpr_expand_portal(const MSup &p0, MSup &p1, MSup &p2, MSup &p3, const MSup &v4) {
  double v4v0[3];
  d_cross(v4v0, v4.v, p0.v);
  if (d_dot3(p1.v, v4v0) > 0) {
    if (d_dot3(p2.v, v4v0) > 0) p1 = v4; else p3 = v4;
  } else {
    if (d_dot3(p3.v, v4v0) > 0) p2 = v4; else p1 = v4;
  }
}
// closest point of the triangle (a, b, c) to the origin and its barycentric weights
DEV void mpr_closest_on_triangle(const double *a, const double *b, const double *c, double *w, double *bw) {
  double ab[3], ac[3];
  d_sub3(ab, b, a); d_sub3(ac, c, a);
  double d1 = -d_dot3(ab, a), d2 = -d_dot3(ac, a);
  if (d1 <= 0 && d2 <= 0) { d_copy3(w, a); bw[0] = 1; bw[1] = 0; bw[2] = 0; return; }
  double d3 = -d_dot3(ab, b), d4 = -d_dot3(ac, b);
  if (d3 >= 0 && d4 <= d3) { d_copy3(w, b); bw[0] = 0; bw[1] = 1; bw[2] = 0; return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); d_addscl3(w, a, ab, v); bw[0] = 1 - v; bw[1] = v; bw[2] = 0; return; }
  double d5 = -d_dot3(ab, c), d6 = -d_dot3(ac, c);
  if (d6 >= 0 && d5 <= d6) { d_copy3(w, c); bw[0] = 0; bw[1] = 0; bw[2] = 1; return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double v = d2 / (d2 - d6); d_addscl3(w, a, ac, v); bw[0] = 1 - v; bw[1] = 0; bw[2] = v; return; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double bc[3]; d_sub3(bc, c, b);
    double v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    d_addscl3(w, b, bc, v); bw[0] = 0; bw[1] = 1 - v; bw[2] = v; return;
  }
  double den = 1.0 / (va + vb + vc);
  d_addscl3(w, a, ab, vb * den);
  d_addtoscl3(w, ac, vc * den);
  bw[1] = vb * den; bw[2] = vc * den; bw[0] = 1 - bw[1] - bw[2];
}
// 1 contact (frame[0..2] = normal from geom 1 to geom 2) or 0
DEV int np_convex(NPCon *con, double margin, const MShape &A, const MShape &B) {
  const double *p1 = A.pos, *p2 = B.pos;
  MSup q0, q1, q2, q3, v4;
  double dir[3], va[3], vb[3], depth, nrm[3], pos[3];
  d_sub3(q0.v, p1, p2); d_copy3(q0.v1, p1); d_copy3(q0.v2, p2);
  if (d_dot3(q0.v, q0.v) < D_MINVAL * D_MINVAL) q0.v[0] += 1e-9;
  d_scl3(dir, q0.v, -1); d_normalize3(dir);
  mpr_support(A, B, dir, q1);
  if (d_dot3(q1.v, dir) <= 0) return 0;
  d_cross(dir, q0.v, q1.v);
  int found = 0;
  if (d_dot3(dir, dir) < D_MINVAL * D_MINVAL) found = 2;
  else {
    d_normalize3(dir);
    mpr_support(A, B, dir, q2);
    if (d_dot3(q2.v, dir) <= 0) return 0;
    d_sub3(va, q1.v, q0.v); d_sub3(vb, q2.v, q0.v);
    d_cross(dir, va, vb); d_normalize3(dir);
    if (d_dot3(dir, q0.v) > 0) { MSup t = q1; q1 = q2; q2 = t; d_scl3(dir, dir, -1); }
    int ok = 0;
    for (int it = 0; it < MPR_ITERATIONS; it++) {
      mpr_support(A, B, dir, v4);
      if (d_dot3(v4.v, dir) <= 0) return 0;
      int cont = 0;
      d_cross(va, q1.v, v4.v);
      if (d_dot3(va, q0.v) < 0) { q2 = v4; cont = 1; }
      if (!cont) {
        d_cross(va, v4.v, q2.v);
        if (d_dot3(va, q0.v) < 0) { q1 = v4; cont = 1; }
      }
      if (!cont) { q3 = v4; ok = 1; break; }
      d_sub3(va, q1.v, q0.v); d_sub3(vb, q2.v, q0.v);
      d_cross(dir, va, vb); d_normalize3(dir);
    }
    if (!ok) return 0;
  }
  if (found == 2) {
    depth = d_norm3(q1.v);
    d_copy3(nrm, q1.v); d_normalize3(nrm);
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (q1.v1[k] + q1.v2[k]);
  } else {
    int hit = 0;
    for (int it = 0; it < MPR_ITERATIONS; it++) {
      mpr_portal_dir(q1, q2, q3, dir);
      if (d_dot3(dir, q1.v) >= 0) { hit = 1; break; }
      mpr_support(A, B, dir, v4);
      if (d_dot3(v4.v, dir) < 0 || mpr_reach_tolerance(q1, q2, q3, v4, dir)) return 0;
      mpr_expand_portal(q0, q1, q2, q3, v4);
    }
    if (!hit) return 0;
    for (int it = 0; ; it++) {
      mpr_portal_dir(q1, q2, q3, dir);
      mpr_support(A, B, dir, v4);
      if (mpr_reach_tolerance(q1, q2, q3, v4, dir) || it >= MPR_ITERATIONS) {
        double w[3], bw[3];
        mpr_closest_on_triangle(q1.v, q2.v, q3.v, w, bw);
        depth = d_norm3(w);
        if (depth < D_MINVAL) d_copy3(nrm, dir); else d_scl3(nrm, w, 1.0 / depth);
        for (int k = 0; k < 3; k++)
          pos[k] = 0.5 * (bw[0] * (q1.v1[k] + q1.v2[k]) + bw[1] * (q2.v1[k] + q2.v2[k]) + bw[2] * (q3.v1[k] + q3.v2[k]));
        break;
      }
      mpr_expand_portal(q0, q1, q2, q3, v4);
    }
  }
  double dist = margin - depth;
  if (dist > margin) return 0;
  NPCon t;
  t.dist = dist;
  d_copy3(t.pos, pos);
  d_copy3(t.frame, nrm); t.frame[3] = 0; t.frame[4] = 0; t.frame[5] = 0;
  np_put(con, 0, t);
  return 1;
}

// plane against an ellipsoid / a convex mesh: its support point against the plane normal
DEV int np_plane_convex(NPCon *con, double margin, const double *pp, const double *pm, MShape E) {
  double n[3] = {pm[2], pm[5], pm[8]}, nd[3] = {-pm[2], -pm[5], -pm[8]}, sp[3], dif[3];
  E.margin = 0;
  mpr_support1(E, nd, sp);
  d_sub3(dif, sp, pp);
  double dist = d_dot3(dif, n);
  if (dist > margin) return 0;
  NPCon t;
  t.dist = dist;
  d_addscl3(t.pos, sp, n, -0.5 * dist);
  d_copy3(t.frame, n); t.frame[3] = 0; t.frame[4] = 0; t.frame[5] = 0;
  np_put(con, 0, t);
  return 1;
}

// The rarely-met pair types (capsule-box, box-box, the cylinder and ellipsoid pairs).  n = -1: no collider and possibly touching.
// Only the out-of-line flavour of the narrow-phase batch (narrow_batch<true>) contains this code: the batch loop of
// collision() itself stays free of it and of any call inside the loop body's live ranges.
struct NPOut { NPCon c[4]; int n; };
DEV NPOut narrow_heavy(Ctx &c, int g1, int g2, double margin) {
  const DevModel &M = *c.M;
  NPOut o;
  int t1 = MI(geom_type)[g1], t2 = MI(geom_type)[g2];
  double p1[3], p2[3], m1[9], m2[9], s1[3], s2[3];
  d_copy3(p1, c.geom_xpos + 3 * g1); d_copy3(p2, c.geom_xpos + 3 * g2);
  for (int k = 0; k < 9; k++) { m1[k] = c.geom_xmat[9 * g1 + k]; m2[k] = c.geom_xmat[9 * g2 + k]; }
  d_copy3(s1, MD(geom_size) + 3 * g1); d_copy3(s2, MD(geom_size) + 3 * g2);
  o.n = -1;
  if (t1 == 3 && t2 == 6) o.n = np_capsule_box(o.c, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == 6 && t2 == 6) o.n = np_box_box(o.c, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == 2 && t2 == 5) o.n = np_sphere_cylinder(o.c, margin, p1, s1[0], p2, m2, s2);
  else if (t1 == 3 && t2 == 5) o.n = np_capsule_cylinder(o.c, margin, p1, m1, s1, p2, m2, s2);
  else if ((t1 == 4 || t2 == 4 || t1 == 7 || t2 == 7) && t1 != 1 && t2 != 1) {
    // ellipsoids and convex meshes: support point against a plane, the portal-refinement collider against everything else
    MShape A = {t1, p1, m1, s1, 0.5 * margin, nullptr, 0}, B = {t2, p2, m2, s2, 0.5 * margin, nullptr, 0};
    if (t1 == 7) { int k = M.geom_dataid[g1]; A.vert = M.mesh_vert + 3 * M.mesh_vertadr[k]; A.nvert = M.mesh_vertnum[k]; }
    if (t2 == 7) { int k = M.geom_dataid[g2]; B.vert = M.mesh_vert + 3 * M.mesh_vertadr[k]; B.nvert = M.mesh_vertnum[k]; }
    o.n = t1 == 0 ? np_plane_convex(o.c, margin, p1, m1, B) : np_convex(o.c, margin, A, B);
  }
  else if (t1 == 5 && (t2 == 5 || t2 == 6)) {
    // cylinder-cylinder / cylinder-box: the cylinder's bounding capsule decides "certainly apart" (exact, cheap); otherwise the
    // portal-refinement collider
    NPCon tmp[4];
    int n = t2 == 5 ? np_capsule_capsule(tmp, margin, p1, m1, s1, p2, m2, s2) : np_capsule_box(tmp, margin, p1, m1, s1, p2, m2, s2);
    MShape A = {t1, p1, m1, s1, 0.5 * margin, nullptr, 0}, B = {t2, p2, m2, s2, 0.5 * margin, nullptr, 0};
    o.n = n == 0 ? 0 : np_convex(o.c, margin, A, B);
  }
  return o;
}

// squared distance from point q to the segment p +- h a (|a| = 1)
DEV double seg_point_dist2(const double *p, const double *a, double h, const double *q) {
  double w[3];
  d_sub3(w, q, p);
  double x = d_clip(d_dot3(a, w), -h, h);
  d_addtoscl3(w, a, -x);
  return d_dot3(w, w);
}

// returns the number of contacts; -2: a pair type handled by narrow_heavy().
// A cylinder that is not against a plane is first replaced by its bounding capsule (same radius and half length) and runs
// through the SAME sphere-capsule / capsule-capsule code as the real capsules of the wave (no extra divergent code path):
// "certainly apart" is exact, and only a cylinder whose bounding capsule touches goes out of line.  A capsule / cylinder
// against a box first tests its segment against the box's bounding sphere.
DEV int narrow_phase(Ctx &c, int g1, int g2, double margin, NPCon *con) {
  const DevModel &M = *c.M;
  const int t1 = MI(geom_type)[g1], t2 = MI(geom_type)[g2];
  const int cyl = (t1 == 5 || (t2 == 5 && t1 != 0));
  const int e1 = t1 == 5 ? 3 : t1, e2 = (t2 == 5 && t1 != 0) ? 3 : t2;
  if (e1 == 6 || e1 == 4 || e2 == 4 || e1 == 1 || e2 == 1 || e1 == 7 || e2 == 7) return -2;      // box-box (and what create() refuses)
  double p1[3], p2[3], m1[9], m2[9], s1[3], s2[3];
  d_copy3(p1, c.geom_xpos + 3 * g1); d_copy3(p2, c.geom_xpos + 3 * g2);
  for (int k = 0; k < 9; k++) { m1[k] = c.geom_xmat[9 * g1 + k]; m2[k] = c.geom_xmat[9 * g2 + k]; }
  d_copy3(s1, MD(geom_size) + 3 * g1); d_copy3(s2, MD(geom_size) + 3 * g2);
  int n;
  if (e1 == 0) {
    double nrm[3] = {m1[2], m1[5], m1[8]};
    if (e2 == 2) n = np_plane_sphere(con, margin, p1, nrm, p2, s2[0]);
    else if (e2 == 3) n = np_plane_capsule(con, margin, p1, m1, p2, m2, s2);
    else if (e2 == 6) n = np_plane_box(con, margin, p1, m1, p2, m2, s2);
    else n = np_plane_cylinder(con, margin, p1, m1, p2, m2, s2);
    return n;
  }
  if (e2 == 6 && e1 == 3) {         // capsule / cylinder against a box: cheap conservative separations, else out of line
    double a1[3] = {m1[2], m1[5], m1[8]};
    double r = s1[0] + MD(geom_rbound)[g2] + margin;
    if (seg_point_dist2(p1, a1, s1[1], p2) > r * r) return 0;         // segment against the box's bounding sphere
    double dif[3], q[3], al[3];
    d_sub3(dif, p1, p2);
    d_mulmattvec3(q, m2, dif);
    d_mulmattvec3(al, m2, a1);
    double rr = s1[0] + margin;                                        // the box's three face normals as separating axes
    if (fabs(q[0]) - s1[1] * fabs(al[0]) > s2[0] + rr || fabs(q[1]) - s1[1] * fabs(al[1]) > s2[1] + rr ||
        fabs(q[2]) - s1[1] * fabs(al[2]) > s2[2] + rr) return 0;
    // ... and the three axes  segment direction x box axis  (the segment projects to a point on them)
    {
      double l0 = sqrt(al[1] * al[1] + al[2] * al[2]), l1 = sqrt(al[0] * al[0] + al[2] * al[2]), l2 = sqrt(al[0] * al[0] + al[1] * al[1]);
      if (fabs(q[2] * al[1] - q[1] * al[2]) > s2[1] * fabs(al[2]) + s2[2] * fabs(al[1]) + rr * l0) return 0;      // a x e0 = (0, a2, -a1)
      if (fabs(q[0] * al[2] - q[2] * al[0]) > s2[0] * fabs(al[2]) + s2[2] * fabs(al[0]) + rr * l1) return 0;      // a x e1 = (-a2, 0, a0)
      if (fabs(q[1] * al[0] - q[0] * al[1]) > s2[0] * fabs(al[1]) + s2[1] * fabs(al[0]) + rr * l2) return 0;      // a x e2 = (a1, -a0, 0)
    }
    return -2;
  }
  if (cyl) {
    // cheap conservative separations before the bounding-capsule test proper (no divide, no square root): each segment against
    // the other geom's bounding sphere
    double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
    double h1 = e1 == 2 ? 0.0 : s1[1], h2 = e2 == 2 ? 0.0 : s2[1];
    double ra = s1[0] + h1 + s2[0] + margin, rb = s1[0] + s2[0] + h2 + margin;
    if (seg_point_dist2(p2, a2, h2, p1) > ra * ra) return 0;
    if (seg_point_dist2(p1, a1, h1, p2) > rb * rb) return 0;
  }
  if (e1 == 2) {
    if (e2 == 2) n = np_sphere_sphere(con, margin, p1, s1[0], p2, s2[0]);
    else if (e2 == 3) n = np_sphere_capsule(con, margin, p1, s1[0], p2, m2, s2);
    else n = np_sphere_box(con, margin, p1, s1[0], p2, m2, s2);
  } else {
    n = np_capsule_capsule(con, margin, p1, m1, s1, p2, m2, s2);
  }
  return (cyl && n > 0) ? -2 : n;
}

DEV void contact_param(Ctx &c, int g1, int g2, double *cc, int *dim) {
  const DevModel &M = *c.M; (void)M;
  int p1 = MI(geom_priority)[g1], p2 = MI(geom_priority)[g2];
  double fri[3];
  if (p1 != p2) {
    int g = p1 > p2 ? g1 : g2;
    *dim = MI(geom_condim)[g];
    for (int i = 0; i < 2; i++) cc[CON_SOLREF + i] = MD(geom_solref)[2 * g + i];
    for (int i = 0; i < 5; i++) cc[CON_SOLIMP + i] = MD(geom_solimp)[5 * g + i];
    d_copy3(fri, MD(geom_friction) + 3 * g);
  } else {
    int d1 = MI(geom_condim)[g1], d2 = MI(geom_condim)[g2];
    *dim = d1 > d2 ? d1 : d2;
    double s1 = MD(geom_solmix)[g1], s2 = MD(geom_solmix)[g2], mix;
    if (s1 >= D_MINVAL && s2 >= D_MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < D_MINVAL && s2 < D_MINVAL) mix = 0.5;
    else if (s1 < D_MINVAL) mix = 0.0;
    else mix = 1.0;
    double r10 = MD(geom_solref)[2 * g1], r20 = MD(geom_solref)[2 * g2];
    for (int i = 0; i < 2; i++) {
      double a = MD(geom_solref)[2 * g1 + i], b = MD(geom_solref)[2 * g2 + i];
      cc[CON_SOLREF + i] = (r10 > 0 && r20 > 0) ? mix * a + (1 - mix) * b : fmin(a, b);
    }
    for (int i = 0; i < 5; i++) cc[CON_SOLIMP + i] = mix * MD(geom_solimp)[5 * g1 + i] + (1 - mix) * MD(geom_solimp)[5 * g2 + i];
    for (int i = 0; i < 3; i++) fri[i] = fmax(MD(geom_friction)[3 * g1 + i], MD(geom_friction)[3 * g2 + i]);
  }
  cc[CON_FRICTION] = fri[0]; cc[CON_FRICTION + 1] = fri[0]; cc[CON_FRICTION + 2] = fri[1];
  cc[CON_FRICTION + 3] = fri[2]; cc[CON_FRICTION + 4] = fri[2];
}

// one batch of (at most) NLANE active pairs: narrow phase per lane, ordered compaction, contact records.
// returns 0: done; 1 (HEAVY == false only): some pair needs narrow_heavy(), nothing was written; 2: contact buffer full
template <bool HEAVY>
DEV int narrow_batch(Ctx &c, int base, int nactive) {
  const DevModel &M = *c.M;
  int a = base + LANE, n = 0, g1 = 0, g2 = 0;
  double margin = 0, gap = 0;
  NPCon con[4] = {};
  if (a < nactive) {
    int p = c.active[a];
    g1 = MI(pair_g1)[p]; g2 = MI(pair_g2)[p];
    margin = fmax(MD(geom_margin)[g1], MD(geom_margin)[g2]);
    gap = fmax(MD(geom_gap)[g1], MD(geom_gap)[g2]);
    n = narrow_phase(c, g1, g2, margin, con);
    if constexpr (HEAVY) {
      if (n == -2) {
        NPOut h = narrow_heavy(c, g1, g2, margin);
        n = h.n; con[0] = h.c[0]; con[1] = h.c[1]; con[2] = h.c[2]; con[3] = h.c[3];
      }
      if (n < 0) { c.warning |= WARN_UNSUPPORTED; n = 0; }      // lane-local here; made wave-uniform below
    }
  }
  if constexpr (HEAVY) c.warning = wave_or_i(c.warning);
  else if (wave_any(n == -2)) return 1;
  int tot, off = wave_excl_scan(n, &tot);
  if (c.ncon + tot > M.nconmax) { c.warning |= WARN_CONTACTFULL; return 2; }
  for (int k = 0; k < n; k++) {
    int ci = c.ncon + off + k;
    double *cc = c.contact + ci * c.M->con_stride;
    int dim;
    contact_param(c, g1, g2, cc, &dim);
    const NPCon cur = np_get(con, k);
    double fr[9];
    for (int q = 0; q < 6; q++) fr[q] = cur.frame[q];
    d_makeframe(fr);
    cc[CON_DIST] = cur.dist;
    d_copy3(cc + CON_POS, cur.pos);
    for (int q = 0; q < 9; q++) cc[CON_FRAME + q] = fr[q];
    cc[CON_INCLUDEMARGIN] = margin - gap;
    cc[CON_MU] = 0;
    int *ci_ = c.con_i + ci * CONI_STRIDE;
    ci_[0] = dim; ci_[1] = g1; ci_[2] = g2; ci_[3] = 0;
  }
  c.ncon += tot;
  return 0;
}
// the batches from `base` on with every collider available (out of line: own registers, called from outside collision()'s loop)
struct BatchOut { int ncon, warning; };
DEV_NOINLINE BatchOut narrow_rest_heavy(const KParams *Kg, int base, int nactive, int ncon, int warning) {
  Ctx c;
  ctx_init(c, Kg, lds_base());
  c.ncon = ncon; c.warning = warning;
  for (; base < nactive; base += NLANE) if (narrow_batch<true>(c, base, nactive) == 2) break;
  BatchOut o;
  o.ncon = c.ncon; o.warning = c.warning;
  return o;
}

DEV void collision(Ctx &c) {
  const DevModel &M = *c.M;
  c.ncon = 0;
  if (M.disableflags & (1 << 4)) return;
  // (1) broad phase: ordered compaction of the pairs whose bounding volumes overlap
  int nactive = 0;
  for (int base = 0; base < M.npair; base += NLANE) {
    int p = base + LANE, pass = 0;
    if (p < M.npair) {
      int g1 = MI(pair_g1)[p], g2 = MI(pair_g2)[p];
      double margin = fmax(MD(geom_margin)[g1], MD(geom_margin)[g2]);
      double r1 = MD(geom_rbound)[g1], r2 = MD(geom_rbound)[g2];
      double dif[3];
      d_sub3(dif, c.geom_xpos + 3 * g2, c.geom_xpos + 3 * g1);
      pass = 1;
      if (MI(geom_type)[g1] == 0) {
        const double *mat = c.geom_xmat + 9 * g1;
        double n[3] = {mat[2], mat[5], mat[8]};
        if (d_dot3(dif, n) > margin + r2) pass = 0;
      } else if (r1 > 0 && r2 > 0) {
        double bound = r1 + r2 + margin;
        if (d_dot3(dif, dif) > bound * bound) pass = 0;
      }
    }
    int tot, off = wave_excl_scan(pass, &tot);
    if (pass && nactive + off < MAX_ACTIVE_PAIRS) c.active[nactive + off] = p;
    nactive += tot;
  }
  if (nactive > MAX_ACTIVE_PAIRS) { c.warning |= WARN_CONTACTFULL; nactive = MAX_ACTIVE_PAIRS; }
  SYNC();
  // (2) narrow phase, one lane per active pair, contacts appended in pair order.  The loop only knows the cheap colliders; at
  // the first batch in which some pair needs an expensive one it stops (nothing of that batch is kept) and the out-of-line
  // flavour finishes the list from there.  The call sits behind the loop, so the loop's registers are not shaped by it.
  int heavy_from = -1;
  for (int base = 0; base < nactive; base += NLANE) {
    int st = narrow_batch<false>(c, base, nactive);
    if (st == 1) heavy_from = base;
    if (st != 0) break;
  }
  if (heavy_from >= 0) {
    BatchOut o = narrow_rest_heavy(c.K, heavy_from, nactive, c.ncon, c.warning);
    c.ncon = o.ncon; c.warning = o.warning;
  }
  SYNC();
}

// ======================================================================================
// constraint rows
// ======================================================================================
DEV double impedance(const double *solimp_in, double pos, double margin) {
  double si0 = d_clip(solimp_in[0], 0.0001, 0.9999), si1 = d_clip(solimp_in[1], 0.0001, 0.9999);
  double si2 = fmax(0.0, solimp_in[2]), si3 = d_clip(solimp_in[3], 0.0001, 0.9999), si4 = fmax(1.0, solimp_in[4]);
  if (si0 == si1 || si2 <= D_MINVAL) return 0.5 * (si0 + si1);
  double x = d_div(pos - margin, si2);
  if (x < 0) x = -x;
  if (x >= 1) return si1;
  if (x == 0) return si0;
  double y;
  if (si4 == 1) y = x;
  else {
    // one evaluation for both halves of the sigmoid: u = x below the midpoint, 1 - x above it
    int low = x <= si3;
    double m = low ? si3 : 1 - si3, u = low ? x : 1 - x;
    double a = d_div(1.0, d_pow_small(m, si4 - 1));
    double w = a * d_pow_small(u, si4);
    y = low ? w : 1 - w;
  }
  return si0 + y * (si1 - si0);
}

// rows that need no contact (friction loss, joint limits, fixed-tendon limits): rows [0, n_nc), incl. their Jacobian.
// A helper wave builds them (and their impedance) while the owner wave is still in the collision phase.
// rotation axis (unit; (1,0,0) for a null rotation) and angle of a unit quaternion, as mju_quat2Vel(quat, 1) followed by
// mju_normalize3 give them
DEV double ball_angle(double *axis, const double *quat) {
  axis[0] = quat[1]; axis[1] = quat[2]; axis[2] = quat[3];
  double s = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  if (s < D_MINVAL) { axis[0] = 1; axis[1] = 0; axis[2] = 0; } else { axis[0] /= s; axis[1] /= s; axis[2] /= s; }
  double speed = 2 * atan2(s, quat[0]);
  if (speed > D_PI) speed -= 2 * D_PI;
  double v[3] = {axis[0] * speed, axis[1] * speed, axis[2] * speed};
  double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n < D_MINVAL) { axis[0] = 1; axis[1] = 0; axis[2] = 0; } else { axis[0] = v[0] / n; axis[1] = v[1] / n; axis[2] = v[2] / n; }
  return n;
}

DEV void make_noncontact_rows(Ctx &c, int *nsingle_out, int *n_nc_out) {
  const DevModel &M = *c.M;
  int nv = M.nv, nvp = M.nvp;
  int nefc = M.nfric;
  PFOR(e, 4 * nv) c.sgl[e] = 0;
  // friction-loss rows are static: rows [0, nfric)
  PFOR(r, M.nfric) {
    int d = MI(fric_dof)[r];
    c.efc_type[r] = CNSTR_FRICTION_DOF; c.efc_id[r] = d; c.efc_dof[r] = d;
    c.efc_floss[r] = MD(dof_frictionloss)[d]; c.efc_pos[r] = 0; c.efc_margin[r] = 0;
    c.efc_diag[r] = MD(dof_invweight0)[d];
  }
  // joint limits: ordered compaction, lower side before upper side
  for (int base = 0; base < M.nlimit; base += NLANE) {
    int q = base + LANE, cnt = 0, j = 0;
    double dist[2] = {0, 0}; int side[2] = {0, 0};
    if (q < M.nlimit) {
      j = MI(limit_jnt)[q];
      double value = c.qpos[MI(jnt_qposadr)[j]], margin = MD(jnt_margin)[j];
      for (int s = -1; s <= 1; s += 2) {
        double dd = s * (MD(jnt_range)[2 * j + (s + 1) / 2] - value);
        if (dd < margin) { dist[cnt] = dd; side[cnt] = s; cnt++; }
      }
    }
    int tot, off = wave_excl_scan(cnt, &tot);
    if (nefc + tot > M.nefcmax) { c.warning |= WARN_CNSTRFULL; break; }
    for (int k = 0; k < cnt; k++) {
      int r = nefc + off + k;
      c.efc_type[r] = CNSTR_LIMIT_JOINT; c.efc_id[r] = j; c.efc_dof[r] = MI(jnt_dofadr)[j];
      c.efc_floss[r] = (double)(-side[k]);      // J entry, consumed below
      c.efc_pos[r] = dist[k]; c.efc_margin[r] = MD(jnt_margin)[j];
      c.efc_diag[r] = MD(dof_invweight0)[MI(jnt_dofadr)[j]];
    }
    nefc += tot;
  }
  int nlim_end = nefc;
  // ball-joint limits (mj_instantiateLimit): rotation angle of the joint quaternion against max(range), J = -axis at its three
  // dofs: general rows, kept behind the single-entry rows (MuJoCo interleaves them in joint order; same constraint set)
  for (int base = 0; base < M.nlimit_ball; base += NLANE) {
    int q = base + LANE, cnt = 0, j = 0;
    double dist = 0;
    if (q < M.nlimit_ball) {
      j = MI(limit_ball)[q];
      double axis[3];
      double angle = ball_angle(axis, c.qpos + MI(jnt_qposadr)[j]);
      dist = fmax(MD(jnt_range)[2 * j], MD(jnt_range)[2 * j + 1]) - angle;
      if (dist < MD(jnt_margin)[j]) cnt = 1;
    }
    int tot, off = wave_excl_scan(cnt, &tot);
    if (nefc + tot > M.nefcmax) { c.warning |= WARN_CNSTRFULL; break; }
    if (cnt) {
      int r = nefc + off;
      c.efc_type[r] = CNSTR_LIMIT_JOINT; c.efc_id[r] = j; c.efc_dof[r] = MI(jnt_dofadr)[j];
      c.efc_floss[r] = 0;
      c.efc_pos[r] = dist; c.efc_margin[r] = MD(jnt_margin)[j];
      c.efc_diag[r] = MD(dof_invweight0)[MI(jnt_dofadr)[j]];
    }
    nefc += tot;
  }
  int nball_end = nefc;
  // fixed-tendon limits (general rows: several Jacobian entries), lower side before upper side
  int ntl0 = nefc;
  for (int base = 0; base < M.ntendon; base += NLANE) {
    int t = base + LANE, cnt = 0;
    double dist[2] = {0, 0}; int side[2] = {0, 0};
    if (t < M.ntendon && MI(tendon_limited)[t]) {
      double value = 0, margin = MD(tendon_margin)[t];
      for (int w = MI(tendon_adr)[t]; w < MI(tendon_adr)[t] + MI(tendon_num)[t]; w++) value += MD(wrap_prm)[w] * c.qpos[MI(wrap_qposadr)[w]];
      for (int s = -1; s <= 1; s += 2) {
        double dd = s * (MD(tendon_range)[2 * t + (s + 1) / 2] - value);
        if (dd < margin) { dist[cnt] = dd; side[cnt] = s; cnt++; }
      }
    }
    int tot, off = wave_excl_scan(cnt, &tot);
    if (nefc + tot > M.nefcmax) { c.warning |= WARN_CNSTRFULL; break; }
    for (int k = 0; k < cnt; k++) {
      int r = nefc + off + k;
      c.efc_type[r] = CNSTR_LIMIT_TENDON; c.efc_id[r] = t;
      c.efc_floss[r] = (double)(-side[k]);      // sign of the Jacobian, consumed below
      c.efc_pos[r] = dist[k]; c.efc_margin[r] = MD(tendon_margin)[t];
      c.efc_diag[r] = MD(tendon_invweight0)[t];
    }
    nefc += tot;
  }
  int ntl_end = nefc;
  SYNC();
  PFOR(e, (ntl_end - M.nfric) * nvp) c.efc_J[M.nfric * nvp + e] = 0;
  SYNC();
  PFOR(rr, nlim_end - M.nfric) {
    int r = M.nfric + rr;
    c.efc_J[r * nvp + MI(jnt_dofadr)[c.efc_id[r]]] = c.efc_floss[r]; c.efc_floss[r] = 0;
  }
  PFOR(rr, nball_end - nlim_end) {
    int r = nlim_end + rr, j = c.efc_id[r], da = MI(jnt_dofadr)[j];
    double axis[3];
    ball_angle(axis, c.qpos + MI(jnt_qposadr)[j]);
    for (int k = 0; k < 3; k++) c.efc_J[r * nvp + da + k] = -axis[k];
  }
  PFOR(rr, ntl_end - ntl0) {
    int r = ntl0 + rr, t = c.efc_id[r];
    double sg = c.efc_floss[r];
    for (int w = MI(tendon_adr)[t]; w < MI(tendon_adr)[t] + MI(tendon_num)[t]; w++) c.efc_J[r * nvp + MI(wrap_dofadr)[w]] = sg * MD(wrap_prm)[w];
    c.efc_floss[r] = 0;
  }
  SYNC();
  *nsingle_out = nlim_end; *n_nc_out = ntl_end;
}

// contact rows [n_nc, nefc): dim rows per contact (2(dim-1) pyramid edges), their Jacobian, the cross-branch flag
DEV void make_contact_rows(Ctx &c, int n_nc) {
  const DevModel &M = *c.M;
  int nv = M.nv, nvp = M.nvp;
  int nefc = n_nc;
  // contacts: dim rows each
  for (int base = 0; base < c.ncon; base += NLANE) {
    int ci = base + LANE, dim = 0;
    if (ci < c.ncon) {
      dim = c.con_i[ci * CONI_STRIDE];
      if (dim > 1 && M.cone != 1) dim = 2 * (dim - 1);     // pyramidal cone: 2(dim-1) edge rows
    }
    int tot, off = wave_excl_scan(dim, &tot);
    if (nefc + tot > M.nefcmax) { c.warning |= WARN_CNSTRFULL; c.ncon = base; break; }
    if (ci < c.ncon) {
      int r0 = nefc + off;
      c.con_i[ci * CONI_STRIDE + 3] = r0;
      int g1 = c.con_i[ci * CONI_STRIDE + 1], g2 = c.con_i[ci * CONI_STRIDE + 2];
      int b1 = MI(geom_bodyid)[g1], b2 = MI(geom_bodyid)[g2];
      double tran = MD(body_invweight0)[2 * b1] + MD(body_invweight0)[2 * b2];
      double rot = MD(body_invweight0)[2 * b1 + 1] + MD(body_invweight0)[2 * b2 + 1];
      const double *cc = c.contact + ci * c.M->con_stride;
      int cdim = c.con_i[ci * CONI_STRIDE];
      int pyr = (cdim > 1 && M.cone != 1);
      for (int k = 0; k < dim; k++) {
        c.efc_type[r0 + k] = cdim == 1 ? CNSTR_CONTACT_FRICTIONLESS : (pyr ? CNSTR_CONTACT_PYRAMIDAL : CNSTR_CONTACT_ELLIPTIC);
        c.efc_id[r0 + k] = EFC_CON_ID(ci, cdim, r0);      // contact id, its dim and first row in one word: no dependent con_i hop later
        c.efc_floss[r0 + k] = 0; c.efc_pos[r0 + k] = cc[CON_DIST]; c.efc_margin[r0 + k] = cc[CON_INCLUDEMARGIN];
        if (pyr) { double mu = cc[CON_FRICTION + k / 2]; c.efc_diag[r0 + k] = tran + mu * mu * (k < 4 ? tran : rot); }
        else c.efc_diag[r0 + k] = k < 3 ? tran : rot;
      }
    }
    nefc += tot;
  }
  c.nefc = nefc;
  // cross-branch contacts (both bodies movable, neither dof chain contains the other) break M's sparsity pattern in H
  int crossflag = 0;
  PFOR(ci, c.ncon) {
    unsigned long long m1 = MPM()[MI(geom_bodyid)[c.con_i[ci * CONI_STRIDE + 1]]];
    unsigned long long m2 = MPM()[MI(geom_bodyid)[c.con_i[ci * CONI_STRIDE + 2]]];
    unsigned long long u = m1 | m2;
    if (u != m1 && u != m2) crossflag = 1;
  }
  c.cross = wave_or_i(crossflag) | M.limit_cross;
  SYNC();
  // Jacobian
  PFOR(e, (nefc - n_nc) * nvp) c.efc_J[n_nc * nvp + e] = 0;
  SYNC();
  // element e = ci * nv + d; the quotient / remainder advance incrementally (one runtime division per lane instead of one per element)
  int je_ci = LANE / nv, je_d = LANE - je_ci * nv;
  const int je_sq = NLANE / nv, je_sr = NLANE - je_sq * nv;
  for (int e = LANE; e < c.ncon * nv; e += NLANE, je_ci += je_sq, je_d += je_sr) {
    if (je_d >= nv) { je_d -= nv; je_ci++; }
    const int ci = je_ci, d = je_d;
    const int *cin = c.con_i + ci * CONI_STRIDE;
    int dim = cin[0], r0 = cin[3];
    int pyr = (dim > 1 && M.cone != 1);
    int b1 = MI(geom_bodyid)[cin[1]], b2 = MI(geom_bodyid)[cin[2]];
    unsigned long long bit = 1ull << d;
    int in1 = (MDM()[b1] & bit) != 0, in2 = (MDM()[b2] & bit) != 0;
    if (!in1 && !in2) continue;
    const double *cc = c.contact + ci * c.M->con_stride;
    const double *cd = c.cdof + 6 * d;
    double jp[3] = {0, 0, 0}, jr[3] = {0, 0, 0};
    if (in2) {
      double off[3], t[3];
      d_sub3(off, cc + CON_POS, c.subtree_com + 3 * MI(body_rootid)[b2]);
      d_cross(t, cd, off);
      jp[0] += cd[3] + t[0]; jp[1] += cd[4] + t[1]; jp[2] += cd[5] + t[2];
      jr[0] += cd[0]; jr[1] += cd[1]; jr[2] += cd[2];
    }
    if (in1) {
      double off[3], t[3];
      d_sub3(off, cc + CON_POS, c.subtree_com + 3 * MI(body_rootid)[b1]);
      d_cross(t, cd, off);
      jp[0] -= cd[3] + t[0]; jp[1] -= cd[4] + t[1]; jp[2] -= cd[5] + t[2];
      jr[0] -= cd[0]; jr[1] -= cd[1]; jr[2] -= cd[2];
    }
    if (pyr) {
      double jn = cc[CON_FRAME] * jp[0] + cc[CON_FRAME + 1] * jp[1] + cc[CON_FRAME + 2] * jp[2];
      for (int k = 1; k < dim; k++) {
        const double *ax = cc + CON_FRAME + 3 * (k % 3);
        const double *jj = k < 3 ? jp : jr;
        double jk = ax[0] * jj[0] + ax[1] * jj[1] + ax[2] * jj[2], mu = cc[CON_FRICTION + k - 1];
        c.efc_J[(r0 + 2 * (k - 1)) * nvp + d] = jn + mu * jk;
        c.efc_J[(r0 + 2 * (k - 1) + 1) * nvp + d] = jn - mu * jk;
      }
    } else {
      for (int k = 0; k < dim; k++) {
        const double *ax = cc + CON_FRAME + 3 * (k % 3);
        const double *jj = k < 3 ? jp : jr;
        c.efc_J[(r0 + k) * nvp + d] = ax[0] * jj[0] + ax[1] * jj[1] + ax[2] * jj[2];
      }
    }
  }
  SYNC();
}

// efc_vel, impedance, R, D, aref
// rows [r0, r1); with_contacts: also the contact pass (cone mu, per-row R of the friction rows)
DEV void make_impedance(Ctx &c, int r0, int r1, int with_contacts) {
  const DevModel &M = *c.M;
  int nv = M.nv, nvp = M.nvp;
  PFOR(rr, r1 - r0) {
    int r = r0 + rr;
    int type = c.efc_type[r], id = c.efc_id[r];
    double vel = 0;
    if (type == CNSTR_FRICTION_DOF) vel = c.qvel[id];      // J = unit vector of the dof (no stored row)
    else
    for (int i0 = 0; i0 < nv; i0 += 9) {     // blocks of 9 loads in flight (nv = 18, 27 divide evenly), same summation order
      double jj[9], qq[9];
#pragma unroll
      for (int k = 0; k < 9; k++) { int i = i0 + k, ic = i < nv ? i : nv - 1; jj[k] = c.efc_J[r * nvp + ic]; qq[k] = c.qvel[ic]; }
#pragma unroll
      for (int k = 0; k < 9; k++) vel += (i0 + k < nv) ? jj[k] * qq[k] : 0.0;
    }
    double solref[2], solimp[5];
    int first = 1;
    if (type == CNSTR_FRICTION_DOF) {
      for (int k = 0; k < 2; k++) solref[k] = MD(dof_solref)[2 * id + k];
      for (int k = 0; k < 5; k++) solimp[k] = MD(dof_solimp)[5 * id + k];
    } else if (type == CNSTR_LIMIT_JOINT) {
      for (int k = 0; k < 2; k++) solref[k] = MD(jnt_solref)[2 * id + k];
      for (int k = 0; k < 5; k++) solimp[k] = MD(jnt_solimp)[5 * id + k];
    } else if (type == CNSTR_LIMIT_TENDON) {
      for (int k = 0; k < 2; k++) solref[k] = MD(tendon_solref_lim)[2 * id + k];
      for (int k = 0; k < 5; k++) solimp[k] = MD(tendon_solimp_lim)[5 * id + k];
    } else {
      const double *cc = c.contact + EFC_CON_CI(id) * c.M->con_stride;
      for (int k = 0; k < 2; k++) solref[k] = cc[CON_SOLREF + k];
      for (int k = 0; k < 5; k++) solimp[k] = cc[CON_SOLIMP + k];
      first = (r == EFC_CON_R0(id)) || type == CNSTR_CONTACT_PYRAMIDAL;
    }
    double imp = impedance(solimp, c.efc_pos[r], c.efc_margin[r]);
    double dmax = d_clip(solimp[1], 0.0001, 0.9999);
    double K, B;
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * M.timestep), dr = solref[1];
      K = d_div(1.0, fmax(D_MINVAL, dmax * dmax * tc * tc * dr * dr));
      B = d_div(2.0, fmax(D_MINVAL, dmax * tc));
    } else {
      K = d_div(-solref[0], fmax(D_MINVAL, dmax * dmax));
      B = d_div(-solref[1], fmax(D_MINVAL, dmax));
    }
    if (type == CNSTR_FRICTION_DOF || !first) K = 0;
    c.efc_R[r] = fmax(D_MINVAL, d_div(1 - imp, imp) * c.efc_diag[r]);
    c.efc_aref[r] = -B * vel - K * imp * (c.efc_pos[r] - c.efc_margin[r]);
  }
  SYNC();
  if (with_contacts) PFOR(ci, c.ncon) {
    int dim = c.con_i[ci * CONI_STRIDE];
    if (dim > 1) {
      double *cc = c.contact + ci * c.M->con_stride;
      double *R = c.efc_R + c.con_i[ci * CONI_STRIDE + 3];
      double R1 = d_div(R[0], fmax(D_MINVAL, M.impratio));
      cc[CON_MU] = cc[CON_FRICTION] * d_sqrt(d_div(R1, R[0]));
      if (M.cone != 1) {        // pyramidal: every edge row gets Rpy = 2 mu^2 R0
        double Rpy = 2 * cc[CON_MU] * cc[CON_MU] * R[0];
        for (int k = 0; k < 2 * (dim - 1); k++) R[k] = Rpy;
      } else {
        R[1] = R1;
        for (int k = 2; k < dim; k++)
          R[k] = d_div(R[1] * cc[CON_FRICTION] * cc[CON_FRICTION], cc[CON_FRICTION + k - 1] * cc[CON_FRICTION + k - 1]);
      }
    }
  }
  SYNC();
  PFOR(rr, r1 - r0) c.efc_D[r0 + rr] = d_div(1.0, c.efc_R[r0 + rr]);
  SYNC();
}

// ======================================================================================
// velocity stage: com velocities, subtree momentum, RNE bias, passive, actuation
// ======================================================================================
DEV void vel_body(Ctx &c, int i) {
  const DevModel &M = *c.M;
  double cvel[6];
  for (int k = 0; k < 6; k++) cvel[k] = c.cvel[6 * MI(body_parentid)[i] + k];
  int bda = MI(body_dofadr)[i];
  for (int j = MI(body_jntadr)[i]; j < MI(body_jntadr)[i] + MI(body_jntnum)[i]; j++) {
    int type = MI(jnt_type)[j];
    if (type == 0) {
      for (int k = 0; k < 18; k++) c.cdof_dot[6 * bda + k] = 0;
      for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) cvel[q] += c.cdof[6 * (bda + k) + q] * c.qvel[bda + k];
      bda += 3;
    }
    if (type == 0 || type == 1) {
      for (int k = 0; k < 3; k++) {
        double r[6];
        d_crossmotion(r, cvel, c.cdof + 6 * (bda + k));
        for (int q = 0; q < 6; q++) c.cdof_dot[6 * (bda + k) + q] = r[q];
      }
      for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) cvel[q] += c.cdof[6 * (bda + k) + q] * c.qvel[bda + k];
      bda += 3;
    } else {
      double r[6];
      d_crossmotion(r, cvel, c.cdof + 6 * bda);
      for (int q = 0; q < 6; q++) c.cdof_dot[6 * bda + q] = r[q];
      for (int q = 0; q < 6; q++) cvel[q] += c.cdof[6 * bda + q] * c.qvel[bda];
      bda++;
    }
  }
  for (int k = 0; k < 6; k++) c.cvel[6 * i + k] = cvel[k];
  // RNE forward part: cacc, cfrc_body
  double a[6];
  for (int k = 0; k < 6; k++) a[k] = c.cacc[6 * MI(body_parentid)[i] + k];
  bda = MI(body_dofadr)[i];
  for (int k = 0; k < MI(body_dofnum)[i]; k++)
    for (int q = 0; q < 6; q++) a[q] += c.cdof_dot[6 * (bda + k) + q] * c.qvel[bda + k];
  for (int k = 0; k < 6; k++) c.cacc[6 * i + k] = a[k];
  double t1[6], t2[6], t3[6];
  d_mulinertvec(t1, c.cinert + 10 * i, a);
  d_mulinertvec(t2, c.cinert + 10 * i, cvel);
  d_crossforce(t3, cvel, t2);
  for (int k = 0; k < 6; k++) c.cfrc[6 * i + k] = t1[k] + t3[k];
  // body momentum for subtree_linvel
  double off[3], v[3];
  d_sub3(off, c.xipos + 3 * i, c.subtree_com + 3 * MI(body_rootid)[i]);
  d_cross(v, cvel, off);
  d_add3(v, v, cvel + 3);
  d_scl3(c.bodytmp + 3 * i, v, MD(body_mass)[i]);
}

// com velocity and RNE acceleration of body i from its parent's (in registers), same operation order as vel_body; `store`: i is
// the lane's own body: cvel, cdof_dot of its dofs, cacc, cfrc_body and its momentum go to LDS (deep trees, see velocity_stage)
DEV void vel_compose(Ctx &c, int i, double *cvel, double *a, int store) {
  const DevModel &M = *c.M;
  int bda = MI(body_dofadr)[i];
  for (int j = MI(body_jntadr)[i]; j < MI(body_jntadr)[i] + MI(body_jntnum)[i]; j++) {
    int type = MI(jnt_type)[j];
    if (type == 0) {
      if (store) for (int k = 0; k < 18; k++) c.cdof_dot[6 * bda + k] = 0;
      for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) a[q] += 0.0 * c.qvel[bda + k];
      for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) cvel[q] += c.cdof[6 * (bda + k) + q] * c.qvel[bda + k];
      bda += 3;
    }
    if (type == 0 || type == 1) {
      for (int k = 0; k < 3; k++) {
        double r[6];
        d_crossmotion(r, cvel, c.cdof + 6 * (bda + k));
        if (store) for (int q = 0; q < 6; q++) c.cdof_dot[6 * (bda + k) + q] = r[q];
        for (int q = 0; q < 6; q++) a[q] += r[q] * c.qvel[bda + k];
      }
      for (int k = 0; k < 3; k++) for (int q = 0; q < 6; q++) cvel[q] += c.cdof[6 * (bda + k) + q] * c.qvel[bda + k];
      bda += 3;
    } else {
      double r[6];
      d_crossmotion(r, cvel, c.cdof + 6 * bda);
      if (store) for (int q = 0; q < 6; q++) c.cdof_dot[6 * bda + q] = r[q];
      for (int q = 0; q < 6; q++) a[q] += r[q] * c.qvel[bda];
      for (int q = 0; q < 6; q++) cvel[q] += c.cdof[6 * bda + q] * c.qvel[bda];
      bda++;
    }
  }
  if (!store) return;
  for (int k = 0; k < 6; k++) { c.cvel[6 * i + k] = cvel[k]; c.cacc[6 * i + k] = a[k]; }
  double t1[6], t2[6], t3[6];
  d_mulinertvec(t1, c.cinert + 10 * i, a);
  d_mulinertvec(t2, c.cinert + 10 * i, cvel);
  d_crossforce(t3, cvel, t2);
  for (int k = 0; k < 6; k++) c.cfrc[6 * i + k] = t1[k] + t3[k];
  double off[3], v[3];
  d_sub3(off, c.xipos + 3 * i, c.subtree_com + 3 * MI(body_rootid)[i]);
  d_cross(v, cvel, off);
  d_add3(v, v, cvel + 3);
  d_scl3(c.bodytmp + 3 * i, v, MD(body_mass)[i]);
}

// subtree sums of the body forces / momenta left by the sweep.  part 0: cfrc_sub components 0..2; part 1: components 3..5 and
// subtree_linvel
#define HX_SWEEP 27      // sweep done (side wave -> last helper), value t + 1
#define HX_SUBSUM 19     // the helper's part of the subtree sums done, value t + 1
#define HX_COM 44        // com-based quantities of this step done (side wave -> owner, helper 0), value t + 1
DEV void subtree_sums(Ctx &c, int part) {
  const DevModel &M = *c.M;
  int per = part ? 6 : 3;
  PFOR(e, M.nbody * per) {
    int b = e / per, k = e - per * b;
    if (k < 3) {
      int kc = k + 3 * part;
      double s = 0;
      if (b > 0) for (int q = MI(subtree_adr)[b]; q < MI(subtree_adr)[b + 1]; q++) s += c.cfrc[6 * MI(subtree_list)[q] + kc];
      c.cfrc_sub[6 * b + kc] = s;
    } else {
      int kk = k - 3;
      double s = 0;
      for (int q = MI(subtree_adr)[b]; q < MI(subtree_adr)[b + 1]; q++) s += c.bodytmp[3 * MI(subtree_list)[q] + kk];
      c.subtree_linvel[3 * b + kk] = s / fmax(D_MINVAL, MD(body_subtreemass)[b]);
    }
  }
}

// mfact_seq != 0: M's factor is produced by a helper wave; wait for its sequence number (misc[22]) before the solve
template <int NVT>
DEV void velocity_stage(Ctx &c, int mfact_seq) {
  const DevModel &M = *c.M;
  int nv = M.nv;
#ifdef MJPC_LEAN_LDS
  // the RNE intermediates share their LDS with the solver's scaled rows here: the world body's entries are rewritten every step
  if (LANE < 6) { c.cfrc[LANE] = 0; c.cacc[LANE] = (LANE >= 3) ? -M.gravity[LANE - 3] : 0.0; }
  SYNC();
#endif
  if constexpr (NVT == 27) {
    // the humanoid's 8 tree levels: one lane per body walks its ancestor chain with the running velocity / acceleration in
    // registers, like kinematics (-1.3 % of its step; the A1's 4 levels are cheaper as a level sweep, and keeping both forms in
    // one instantiation costs it +0.7 %, hence the compile-time choice)
    PFOR(b, M.nbody) {
      if (b == 0) continue;
      double cvel[6], acc[6];
      for (int k = 0; k < 6; k++) { cvel[k] = c.cvel[k]; acc[k] = c.cacc[k]; }      // the world body: 0 and -gravity
      for (int q = MI(chain_adr)[b]; q < MI(chain_adr)[b + 1]; q++) {
        int a = MI(chain_list)[q];
        vel_compose(c, a, cvel, acc, a == b);
      }
    }
    SYNC();
  } else {
    for (int l = 0; l < M.nlevel; l++) {
      int a = MI(level_adr)[l], n = MI(level_adr)[l + 1] - a;
      PFOR(k, n) vel_body(c, MI(level_body)[a + k]);
      SYNC();
    }
  }
  PROFW(c, 1);
#if MJPC_HELPER
  // the subtree sums are shared with the last helper wave (idle by now, ph_noncontact): it takes the torque half of cfrc_sub and
  // the subtree momenta, this wave the force half; every element is summed by one lane in list order, as before
  flag_set(c.misc + HX_SWEEP, mfact_seq);
  subtree_sums(c, 0);
#else
  subtree_sums(c, 0); subtree_sums(c, 1);
#endif
  PROFW(c, 4);
  // actuator forces
  PFOR(i, M.nu) {
    double ctrl = c.ctrl[i];
    if (MI(actuator_ctrllimited)[i]) ctrl = d_clip(ctrl, MD(actuator_ctrlrange)[2 * i], MD(actuator_ctrlrange)[2 * i + 1]);
    double force = MD(actuator_gainprm)[3 * i] * ctrl;
    if (MI(actuator_biastype)[i] == 1) {
      // transmission length / velocity: gear * qpos (joint) or sum of gear * coef * qpos over the tendon's joints
      double length = 0, velocity = 0;
      for (int e = MI(act_adr)[i]; e < MI(act_adr)[i + 1]; e++) {
        double cf = MD(act_coef)[e];
        length += cf * c.qpos[MI(act_qpos)[e]]; velocity += cf * c.qvel[MI(act_dof)[e]];
      }
      force += MD(actuator_biasprm)[3 * i] + MD(actuator_biasprm)[3 * i + 1] * length + MD(actuator_biasprm)[3 * i + 2] * velocity;
    }
    if (MI(actuator_forcelimited)[i]) force = d_clip(force, MD(actuator_forcerange)[2 * i], MD(actuator_forcerange)[2 * i + 1]);
    c.actuator_force[i] = force;
  }
#if MJPC_HELPER
  if (!flag_wait(c.misc + HX_SUBSUM, mfact_seq)) c.warning |= WARN_SYNC;
#endif
  SYNC();
  PFOR(d, nv) {
    const double *cd = c.cdof + 6 * d, *cf = c.cfrc_sub + 6 * MI(dof_bodyid)[d];
    double bias = cd[0]*cf[0] + cd[1]*cf[1] + cd[2]*cf[2] + cd[3]*cf[3] + cd[4]*cf[4] + cd[5]*cf[5];
    c.qfrc_bias[d] = bias;
    double act = 0;
    for (int e = 0; e < M.nact; e++) if (MI(act_dof)[e] == d) act += MD(act_coef)[e] * c.actuator_force[MI(act_of)[e]];     // moment^T force
    c.qfrc_smooth[d] = act - bias - MD(dof_damping)[d] * c.qvel[d];   // joint springs are added below
  }
  SYNC();
  PFOR(j, M.njnt) {
    double k = MD(jnt_stiffness)[j];
    int type = MI(jnt_type)[j];
    if (k != 0 && (type == 2 || type == 3)) {
      int qa = MI(jnt_qposadr)[j];
      c.qfrc_smooth[MI(jnt_dofadr)[j]] -= k * (c.qpos[qa] - MD(qpos_spring)[qa]);
    }
  }
  SYNC();
  if (M.ntendon_passive > 0) {
    // tendon springs (dead band) and dampers, mj_passive: one lane per dof gathers J^T force over the (few) passive tendons
    PFOR(d, nv) {
      double acc = c.qfrc_smooth[d];
      for (int e = 0; e < M.ntendon_passive; e++) {
        int t = MI(tpass_id)[e];
        double coef = 0, length = 0, velocity = 0;
        for (int w = MI(tendon_adr)[t]; w < MI(tendon_adr)[t] + MI(tendon_num)[t]; w++) {
          double cf = MD(wrap_prm)[w];
          length += cf * c.qpos[MI(wrap_qposadr)[w]]; velocity += cf * c.qvel[MI(wrap_dofadr)[w]];
          if (MI(wrap_dofadr)[w] == d) coef += cf;
        }
        if (coef == 0) continue;
        const double *pr = MD(tpass_prm) + 4 * e;
        double frc = 0;
        if (length > pr[3]) frc = pr[0] * (pr[3] - length); else if (length < pr[2]) frc = pr[0] * (pr[2] - length);
        frc -= pr[1] * velocity;
        acc += coef * frc;
      }
      c.qfrc_smooth[d] = acc;
    }
    SYNC();
  }
  if (c.K->xfrc_std > 0) {
    // mj_xfrcAccumulate: J^T [force; torque], force applied at the body's inertial frame origin; bodies in ascending order
    PFOR(d, nv) {
      const double *cd = c.cdof + 6 * d;
      int bd = MI(dof_bodyid)[d];
      double acc = c.qfrc_smooth[d];
      for (int q = MI(subtree_adr)[bd]; q < MI(subtree_adr)[bd + 1]; q++) {
        int b = MI(subtree_list)[q];
        const double *f = c.xfrc + 6 * b;
        double off[3], tt[3];
        d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MI(body_rootid)[b]);
        d_cross(tt, cd, off);
        acc += (cd[3] + tt[0]) * f[0] + (cd[4] + tt[1]) * f[1] + (cd[5] + tt[2]) * f[2] + cd[0] * f[3] + cd[1] * f[4] + cd[2] * f[5];
      }
      c.qfrc_smooth[d] = acc;
    }
    SYNC();
  }
  PFOR(d, nv) c.qacc_smooth[d] = c.qfrc_smooth[d];
  PROFW(c, 5);
  if (mfact_seq && !flag_wait(c.misc + 22, mfact_seq)) c.warning |= WARN_SYNC;
  PROFW(c, 7);
  chol_solve<NVT>(c.qL, c.Linv, c.qacc_smooth, nv, M.nvp, M.tree_ok);
  PROFW(c, 8);
}

#include "solver.h"

// ======================================================================================
// task residuals (device restatement of the reference's ResidualFn::Residual)
// ======================================================================================
DEV double ray_geom(const double *pos, const double *mat, const double *size, const double *pnt, const double *vec, int type) {
  double dif[3], lp[3], lv[3];
  d_sub3(dif, pnt, pos);
  d_mulmattvec3(lp, mat, dif);
  d_mulmattvec3(lv, mat, vec);
  if (type == 0) {
    if (lv[2] > -D_MINVAL) return -1;
    double x = -lp[2] / lv[2];
    if (x < 0) return -1;
    double p0 = lp[0] + x * lv[0], p1 = lp[1] + x * lv[1];
    if ((size[0] <= 0 || fabs(p0) <= size[0]) && (size[1] <= 0 || fabs(p1) <= size[1])) return x;
    return -1;
  }
  if (type == 2) {
    double a = d_dot3(lv, lv), b = d_dot3(lv, lp), cq = d_dot3(lp, lp) - size[0] * size[0];
    double det = b * b - a * cq;
    if (det < D_MINVAL || a < D_MINVAL) return -1;
    det = sqrt(det);
    double x0 = (-b - det) / a, x1 = (-b + det) / a;
    if (x0 >= 0) return x0;
    if (x1 >= 0) return x1;
    return -1;
  }
  if (type == 6) {
    double best = -1;
    for (int i = 0; i < 3; i++) {
      double lvi = i == 0 ? lv[0] : (i == 1 ? lv[1] : lv[2]);
      double lpi = i == 0 ? lp[0] : (i == 1 ? lp[1] : lp[2]);
      double szi = i == 0 ? size[0] : (i == 1 ? size[1] : size[2]);
      if (fabs(lvi) <= D_MINVAL) continue;
      int j = (i + 1) % 3, k = (i + 2) % 3;
      double lvj = j == 0 ? lv[0] : (j == 1 ? lv[1] : lv[2]), lpj = j == 0 ? lp[0] : (j == 1 ? lp[1] : lp[2]);
      double lvk = k == 0 ? lv[0] : (k == 1 ? lv[1] : lv[2]), lpk = k == 0 ? lp[0] : (k == 1 ? lp[1] : lp[2]);
      double szj = j == 0 ? size[0] : (j == 1 ? size[1] : size[2]), szk = k == 0 ? size[0] : (k == 1 ? size[1] : size[2]);
      for (int side = -1; side <= 1; side += 2) {
        double x = (side * szi - lpi) / lvi;
        if (x < 0) continue;
        double pj = lpj + x * lvj, pk = lpk + x * lvk;
        if (fabs(pj) <= szj && fabs(pk) <= szk) if (best < 0 || x < best) best = x;
      }
    }
    return best;
  }
  return -1;
}

// mjpc/utilities.cc:538-556 Ground(): mj_ray straight down from 0.5 m above, geom group 0
DEV double ray_ground(Ctx &c, const double *pos) {
  const DevModel &M = *c.M;
  double down[3] = {0, 0, -1}, query[3] = {pos[0], pos[1], pos[2] + 0.5};
  double dist = -1;
  for (int r = 0; r < M.nray; r++) {
    int g = MI(ray_geom)[r];
    double gp[3], gm[9], gs[3];
    d_copy3(gp, c.geom_xpos + 3 * g); d_copy3(gs, MD(geom_size) + 3 * g);
    for (int k = 0; k < 9; k++) gm[k] = c.geom_xmat[9 * g + k];
    double x = ray_geom(gp, gm, gs, query, down, MI(geom_type)[g]);
    if (x >= 0 && (dist < 0 || x < dist)) dist = x;
  }
  if (dist < 0) c.warning |= WARN_RAY;        // the reference aborts here (utilities.cc:549-552); a candidate fails instead
  return pos[2] + 0.5 - dist;
}

DEV int reinterpret_int(double v) { union { double d; int i[2]; } u; u.d = v; return u.i[0]; }

enum { QI_TORSO = 0, QI_HEAD = 1, QI_GOAL = 2, QI_FOOT = 3, QI_GAIT = 7, QI_GAIT_SWITCH = 8, QI_FLIP_DIR = 9,
       QI_BIPED_TYPE = 10, QI_CADENCE = 11, QI_AMPLITUDE = 12, QI_DUTY = 13, QI_HEADING = 14, QI_HOME = 15,
       QI_CROUCH = 16, QI_MODE = 17 };
enum { QD_MODE_START = 0, QD_POSITION = 1, QD_HEADING = 4, QD_SPEED = 6, QD_ANGVEL = 7, QD_GROUND = 8,
       QD_ORIENT = 9, QD_GAIT = 13, QD_PHASE_START = 14, QD_PHASE_START_TIME = 15, QD_PHASE_VEL = 16,
       QD_GRAVITY = 17, QD_JUMP_VEL = 18, QD_FLIGHT_TIME = 19, QD_JUMP_ACC = 20, QD_CROUCH_TIME = 21,
       QD_LEAP_TIME = 22, QD_JUMP_TIME = 23, QD_CROUCH_VEL = 24, QD_LAND_TIME = 25, QD_LAND_ACC = 26,
       QD_FLIGHT_ROT_VEL = 27, QD_JUMP_ROT_VEL = 28, QD_JUMP_ROT_ACC = 29, QD_LAND_ROT_ACC = 30 };

DEV double q_gait_phase(int gait, int foot) {   // quadruped.h:77-85
  const double tab[20] = {0, 0, 0, 0, 0, 0.75, 0.5, 0.25, 0, 0.5, 0.5, 0, 0, 0.33, 0.33, 0.66, 0, 0.4, 0.05, 0.35};
  return tab[4 * gait + foot];
}
DEV double q_step_height(double time, double footphase, double duty_ratio) {   // quadruped.cc:650-659
  double angle = fmod(time + D_PI - footphase, 2 * D_PI) - D_PI;
  double value = 0;
  if (duty_ratio < 1) { angle *= 0.5 / (1 - duty_ratio); value = cos(d_clip(angle, -D_PI / 2, D_PI / 2)); }
  return fabs(value) < 1e-6 ? 0.0 : value;
}
DEV double q_flip_height(const double *D, double time) {   // quadruped.cc:674-690
  double jt = D[QD_JUMP_TIME], ft = D[QD_FLIGHT_TIME], lt = D[QD_LAND_TIME];
  if (time >= jt + ft + lt) return 0.25 + D[QD_GROUND];
  double h = 0;
  if (time < jt) h = 0.25 + time * D[QD_CROUCH_VEL] + 0.5 * time * time * D[QD_JUMP_ACC];
  else if (time >= jt && time < jt + ft) { time -= jt; h = 0.5 + D[QD_JUMP_VEL] * time - 0.5 * 9.81 * time * time; }
  else if (time >= jt + ft) { time -= jt + ft; h = 0.5 - D[QD_JUMP_VEL] * time + 0.5 * D[QD_LAND_ACC] * time * time; }
  return h + D[QD_GROUND];
}
DEV void q_flip_quat(const double *D, const double *P, const int *I, double *quat, double time) {   // quadruped.cc:695-714
  double angle = 0, jt = D[QD_JUMP_TIME], ft = D[QD_FLIGHT_TIME], lt = D[QD_LAND_TIME], ct = D[QD_CROUCH_TIME];
  if (time >= jt + ft + lt) angle = 2 * D_PI;
  else if (time >= ct && time < jt) { time -= ct; angle = 0.5 * D[QD_JUMP_ROT_ACC] * time * time + D[QD_JUMP_ROT_VEL] * time; }
  else if (time >= jt && time < jt + ft) { time -= jt; angle = D_PI / 2 + D[QD_FLIGHT_ROT_VEL] * time; }
  else if (time >= jt + ft) { time -= jt + ft; angle = 1.75 * D_PI + D[QD_FLIGHT_ROT_VEL] * time - 0.5 * D[QD_LAND_ROT_ACC] * time * time; }
  int flip_dir = reinterpret_int(P[I[QI_FLIP_DIR]]);
  double axis[3] = {0, flip_dir ? 1.0 : -1.0, 0}, q[4], o[4];
  d_axisangle2quat(q, axis, angle);
  d_copy4(o, D + QD_ORIENT);
  d_mulquat(quat, o, q);
}

// mjpc/tasks/quadruped/quadruped.cc:33-221
DEV void residual_quadruped(Ctx &c, double *residual) {
  const DevModel &M = *c.M;
  const int *I = MI(task.int_data);
  const double *D = MD(task.dbl_data), *P = MD(task.parameters);
  int mode = I[QI_MODE], torso = I[QI_TORSO], nu = M.nu;
  int is_biped = mode == 1;
  double height_goal = is_biped ? 0.6 : 0.25;
  double avg[3];
  {
    const double *fFL = c.geom_xpos + 3 * I[QI_FOOT + 0], *fHL = c.geom_xpos + 3 * I[QI_FOOT + 1];
    const double *fFR = c.geom_xpos + 3 * I[QI_FOOT + 2], *fHR = c.geom_xpos + 3 * I[QI_FOOT + 3];
    if (mode == 1) {
      int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]);
      if (handstand) d_add3(avg, fFL, fFR); else d_add3(avg, fHL, fHR);
      d_scl3(avg, avg, 0.5);
    } else {
      d_add3(avg, fHL, fHR); d_add3(avg, avg, fFL); d_add3(avg, avg, fFR); d_scl3(avg, avg, 0.25);
    }
  }
  const double *torso_pos = c.xipos + 3 * torso;
  const double *goal_pos = c.mocap_pos + 3 * I[QI_GOAL];
  // ---- Gait (4 residuals at offset 7): one lane per foot, each casts its own ray
  PFOR(f, 4) {
    double r = 0;
    int skip = 0;
    if (is_biped) {
      int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]) != 0;
      int front_hand = !handstand && (f == 0 || f == 2);
      int back_hand = handstand && (f == 1 || f == 3);
      skip = front_hand || back_hand;
    }
    if (!skip) {
      int gait = is_biped ? 2 : reinterpret_int(D[QD_GAIT]);
      double phase = D[QD_PHASE_START] + (c.time - D[QD_PHASE_START_TIME]) * D[QD_PHASE_VEL];
      double step = P[I[QI_AMPLITUDE]] * q_step_height(phase, 2 * D_PI * q_gait_phase(gait, f), P[I[QI_DUTY]]);
      double fp[3], query[3];
      d_copy3(fp, c.geom_xpos + 3 * I[QI_FOOT + f]);
      d_copy3(query, fp);
      if (mode == 3) {
        double v[3];
        d_sub3(v, goal_pos, fp); v[2] = 0; d_normalize3(v);
        d_addtoscl3(query, v, 0.15);
      }
      double ground_height = ray_ground(c, query);
      double height_difference = fp[2] - (ground_height + 0.02 + step);
      if (mode == 3) height_difference = fmin(0.0, height_difference);
      r = step ? height_difference : 0;
    }
    residual[7 + f] = r;
  }
  // ---- Effort (12 at 13) and Posture (12 at 25)
  PFOR(i, nu) {
    residual[13 + i] = c.actuator_force[i] * 2e-2;
    const double *home = M.key_qpos + I[QI_HOME] * M.nq;
    double p = c.qpos[7 + i] - home[7 + i];
    if (mode == 4) {
      double flip_time = c.time - D[QD_MODE_START];
      if (flip_time < D[QD_CROUCH_TIME]) p = c.qpos[7 + i] - M.key_qpos[I[QI_CROUCH] * M.nq + 7 + i];
      else if (flip_time >= D[QD_CROUCH_TIME] && flip_time < D[QD_JUMP_TIME] + D[QD_FLIGHT_TIME]) p = 0;
    }
    int j = i % 3;
    p *= (j == 0) ? 2.0 : 1.0;
    if (mode == 1) {
      int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]) != 0;
      if (handstand) { if (i == 4 || i == 5 || i == 10 || i == 11) p *= 0.03; }
      else { if (i == 1 || i == 2 || i == 7 || i == 8) p *= 0.03; }
    }
    residual[13 + nu + i] = p;
  }
  // ---- everything else: lane 0
  if (LANE == 0) {
    const double *xm = c.xmat + 9 * torso;
    int k = 0;
    if (mode != 4) {
      if (mode == 1) { int hs = reinterpret_int(P[I[QI_BIPED_TYPE]]) ? -1 : 1; residual[k++] = xm[6] - hs; }
      else residual[k++] = xm[8] - 1;
      residual[k++] = 0; residual[k++] = 0;
    } else {
      double quat[4], r3[3];
      q_flip_quat(D, P, I, quat, c.time - D[QD_MODE_START]);
      d_subquat(r3, c.xquat + 4 * torso, quat);
      residual[0] = r3[0]; residual[1] = r3[1]; residual[2] = r3[2]; k = 3;
    }
    if (mode == 3) residual[k++] = 0;
    else if (mode == 4) residual[k++] = torso_pos[2] - q_flip_height(D, c.time - D[QD_MODE_START]);
    else residual[k++] = (torso_pos[2] - avg[2]) - height_goal;
    const double *head = c.site_xpos + 3 * I[QI_HEAD];
    double target[3] = {goal_pos[0], goal_pos[1], goal_pos[2]};
    if (mode == 2) {   // Walk(), quadruped.cc:619-636
      double tm = c.time - D[QD_MODE_START];
      if (fabs(D[QD_ANGVEL]) < 0.01) {
        double fwd[2] = {D[QD_HEADING], D[QD_HEADING + 1]};
        d_normalize2(fwd);
        target[0] = D[QD_POSITION] + D[QD_HEADING] + tm * D[QD_SPEED] * fwd[0];
        target[1] = D[QD_POSITION + 1] + D[QD_HEADING + 1] + tm * D[QD_SPEED] * fwd[1];
      } else {
        double angle = tm * D[QD_ANGVEL], cs = cos(angle), sn = sin(angle);
        target[0] = cs * D[QD_HEADING] - sn * D[QD_HEADING + 1] + D[QD_POSITION];
        target[1] = sn * D[QD_HEADING] + cs * D[QD_HEADING + 1] + D[QD_POSITION + 1];
      }
    }
    residual[k++] = head[0] - target[0];
    residual[k++] = head[1] - target[1];
    residual[k++] = mode == 3 ? 2 * (head[2] - target[2]) : 0;
    // Balance (2 at 11)
    const double *compos = c.subtree_com + 3 * torso, *comvel = c.subtree_linvel + 3 * torso;
    double fall_time = sqrt(2 * height_goal / 9.81);
    residual[11] = compos[0] + comvel[0] * fall_time - avg[0];
    residual[12] = compos[1] + comvel[1] * fall_time - avg[1];
    // Yaw (2) and "Angmom" (3) after effort + posture
    int o = 13 + 2 * nu;
    double th[2] = {xm[0], xm[3]};
    if (mode == 1) { int hs = reinterpret_int(P[I[QI_BIPED_TYPE]]) ? 1 : -1; th[0] = hs * xm[2]; th[1] = hs * xm[5]; }
    d_normalize2(th);
    double heading_goal = P[I[QI_HEADING]];
    residual[o] = th[0] - cos(heading_goal);
    residual[o + 1] = th[1] - sin(heading_goal);
    residual[o + 2] = comvel[0]; residual[o + 3] = comvel[1]; residual[o + 4] = comvel[2];
  }
}

// mjpc/tasks/humanoid/tracking/tracking.cc:94-216 (int_data: motion, first key, length, 16 site ids, 16 mocap ids)
DEV void residual_humanoid_track(Ctx &c, double *residual) {
  const DevModel &M = *c.M;
  const int *I = MI(task.int_data);
  const double kFps = 30.0;
  int start = I[1], length = I[2], nv = M.nv, nu = M.nu;
  double current_index = (c.time - MD(task.dbl_data)[0]) * kFps + start;
  int last_key_index = start + length - 1;
  double ci = current_index < 0 ? 0 : (current_index > last_key_index ? (double)last_key_index : current_index);
  int k0 = (int)floor(ci), k1 = k0 + 1 < last_key_index ? k0 + 1 : last_key_index;
  double w1 = ci - k0, w0 = 1.0 - w1;
  PFOR(i, nv - 6) residual[i] = c.qvel[6 + i];
  PFOR(i, nu) residual[nv - 6 + i] = c.ctrl[i];
  int o = nv - 6 + nu;
  // interpolated markers (vtmp-free scratch: bodytmp holds 16x3 markers) and averages
  PFOR(b, 16) {
    int mid = I[19 + b];
    const double *p0 = M.key_mpos + M.nmocap * 3 * k0 + 3 * mid, *p1 = M.key_mpos + M.nmocap * 3 * k1 + 3 * mid;
    double mp[3];
    d_scl3(mp, p0, w0); d_addtoscl3(mp, p1, w1);
    d_copy3(c.bodytmp + 3 * b, mp);
    // velocity residual: finite-difference marker velocity minus framelinvel of the tracking site
    int sid = I[3 + b], body = MI(site_bodyid)[sid];
    double v[3], off[3], lin[3];
    d_sub3(v, p1, p0); d_scl3(v, v, kFps);
    d_sub3(off, c.site_xpos + 3 * sid, c.subtree_com + 3 * MI(body_rootid)[body]);
    d_cross(lin, c.cvel + 6 * body, off);
    d_add3(lin, lin, c.cvel + 6 * body + 3);
    d_sub3(residual + o + 3 + 48 + 3 * b, v, lin);
  }
  SYNC();
  double avg_m[3] = {0, 0, 0}, avg_s[3] = {0, 0, 0};
  for (int b = 0; b < 16; b++) { d_add3(avg_m, avg_m, c.bodytmp + 3 * b); d_add3(avg_s, avg_s, c.site_xpos + 3 * I[3 + b]); }
  d_scl3(avg_m, avg_m, 1.0 / 16); d_scl3(avg_s, avg_s, 1.0 / 16);
  if (LANE == 0) d_sub3(residual + o, avg_m, avg_s);
  PFOR(b, 16) {
    double bm[3], bs[3];
    d_sub3(bm, c.bodytmp + 3 * b, avg_m);
    d_sub3(bs, c.site_xpos + 3 * I[3 + b], avg_s);
    d_sub3(residual + o + 3 + 3 * b, bm, bs);
  }
}

// velocity of a body's inertial-frame origin in the world frame (framelinvel objtype="body")
DEV void body_linvel(Ctx &c, int body, double *lin) {
  const DevModel &M = *c.M;
  double off[3];
  d_sub3(off, c.xipos + 3 * body, c.subtree_com + 3 * MI(body_rootid)[body]);
  d_cross(lin, c.cvel + 6 * body, off);
  d_add3(lin, lin, c.cvel + 6 * body + 3);
}
// mjpc/tasks/humanoid/stand/stand.cc:41-94.  int_data = [site sp0, sp1, sp2, sp3, body head, body torso]
DEV void residual_humanoid_stand(Ctx &c, double *residual) {
  const DevModel &M = *c.M;
  const int *I = MI(task.int_data);
  int nv = M.nv, nu = M.nu;
  if (LANE == 0) {
    const double *f1 = c.site_xpos + 3 * I[0], *f2 = c.site_xpos + 3 * I[1], *f3 = c.site_xpos + 3 * I[2], *f4 = c.site_xpos + 3 * I[3];
    const double *head = c.xipos + 3 * I[4];
    residual[0] = (head[2] - 0.25 * (f1[2] + f2[2] + f3[2] + f4[2])) - MD(task.parameters)[0];
    const double *com = c.subtree_com + 3 * I[5], *comvel = c.subtree_linvel + 3 * I[5];
    double cpx = com[0] + comvel[0] * 0.2, cpy = com[1] + comvel[1] * 0.2;
    double fx = (((f1[0] + f2[0]) + f3[0]) + f4[0]) * 0.25 - cpx, fy = (((f1[1] + f2[1]) + f3[1]) + f4[1]) * 0.25 - cpy;
    residual[1] = sqrt(fx * fx + fy * fy);
    residual[2] = comvel[0]; residual[3] = comvel[1];
  }
  PFOR(i, nv - 6) residual[4 + i] = c.qvel[6 + i];
  PFOR(i, nu) residual[4 + nv - 6 + i] = c.ctrl[i];
}
// mjpc/tasks/humanoid/walk/walk.cc:44-166.  int_data = [body torso, pelvis, foot_right, foot_left, waist_lower]
DEV void residual_humanoid_walk(Ctx &c, double *residual) {
  const DevModel &M = *c.M;
  const int *I = MI(task.int_data);
  const double *P = MD(task.parameters);
  int nq = M.nq, nu = M.nu;
  int torso = I[0], pelvis = I[1], fr = I[2], fl = I[3], wl = I[4];
  if (LANE == 0) {
    double torso_height = c.xipos[3 * torso + 2];
    residual[0] = torso_height - P[0];
    const double *foot_right = c.xipos + 3 * fr, *foot_left = c.xipos + 3 * fl;
    residual[1] = 0.5 * (foot_left[2] + foot_right[2]) - c.xipos[3 * pelvis + 2] - 0.2;
    const double *subcom = c.subtree_com + 3 * torso, *subcomvel = c.subtree_linvel + 3 * torso;
    double cp[3], axis[3], center[3], vec[3], pcp[3];
    for (int k = 0; k < 3; k++) cp[k] = subcom[k] + subcomvel[k] * 0.3;
    cp[2] = 1.0e-3;
    d_sub3(axis, foot_right, foot_left);
    axis[2] = 1.0e-3;
    double length = 0.5 * d_normalize3(axis) - 0.05;
    d_add3(center, foot_right, foot_left);
    d_scl3(center, center, 0.5);
    d_sub3(vec, cp, center);
    double t = d_dot3(vec, axis);
    t = fmax(-length, fmin(length, t));
    d_scl3(vec, axis, t);
    d_add3(pcp, vec, center);
    double standing = torso_height / sqrt(torso_height * torso_height + 0.45 * 0.45) - 0.4;
    residual[2] = (cp[0] - pcp[0]) * standing; residual[3] = (cp[1] - pcp[1]) * standing;
    const double *xt = c.xmat + 9 * torso, *xp = c.xmat + 9 * pelvis, *xr = c.xmat + 9 * fr, *xl = c.xmat + 9 * fl;
    residual[4] = xt[8] - 1.0;
    residual[5] = 0.3 * (xp[8] - 1.0);
    for (int k = 0; k < 3; k++) {
      double zr = k == 2 ? 1.0 : 0.0;
      residual[6 + k] = (xr[3 * k + 2] - zr) * (0.1 * standing);
      residual[9 + k] = (xl[3 * k + 2] - zr) * (0.1 * standing);
    }
    int o = 12 + nq - 7;
    double fwx = ((xt[0] + xp[0]) + xr[0]) + xl[0], fwy = ((xt[3] + xp[3]) + xr[3]) + xl[3];
    double n = sqrt(fwx * fwx + fwy * fwy);
    if (n < D_MINVAL) { fwx = 1; fwy = 0; } else { double sc = 1.0 / n; fwx *= sc; fwy *= sc; }     // mju_normalize
    double tv[3], rv[3], lv[3];
    body_linvel(c, torso, tv); body_linvel(c, fr, rv); body_linvel(c, fl, lv);
    const double *wlv = c.subtree_linvel + 3 * wl;
    double cvx = (wlv[0] + tv[0]) * 0.5, cvy = (wlv[1] + tv[1]) * 0.5;
    residual[o] = standing * (cvx * fwx + cvy * fwy - P[1]);
    residual[o + 1] = ((cvx + rv[0] * -0.5) + lv[0] * -0.5) * standing;
    residual[o + 2] = ((cvy + rv[1] * -0.5) + lv[1] * -0.5) * standing;
  }
  PFOR(i, nq - 7) residual[12 + i] = c.qpos[7 + i];
  PFOR(i, nu) residual[12 + nq - 7 + 3 + i] = c.ctrl[i];
}

// mjpc/tasks/shadow_reorient/hand.cc:37-84.  int_data = [palm site, cube body, goal body, keyframe]; framepos / framequat /
// framelinvel sensors with objtype="body" read the body's inertial frame
DEV void residual_shadow(Ctx &c, double *residual) {
  const DevModel &M = *c.M;
  const int *I = MI(task.int_data);
  int palm = I[0], cube = I[1], goal = I[2], key = I[3], nu = M.nu;
  if (LANE == 0) {
    d_sub3(residual, c.xipos + 3 * cube, c.site_xpos + 3 * palm);
    double gq[4], cq[4], iq[4], r3[3], lin[3];
    d_copy4(iq, MD(body_iquat) + 4 * goal); d_mulquat(gq, c.xquat + 4 * goal, iq);
    d_copy4(iq, MD(body_iquat) + 4 * cube); d_mulquat(cq, c.xquat + 4 * cube, iq);
    d_normalize4(gq);
    d_subquat(r3, gq, cq);
    residual[3] = r3[0]; residual[4] = r3[1]; residual[5] = r3[2];
    body_linvel(c, cube, lin);
    residual[6] = lin[0]; residual[7] = lin[1]; residual[8] = lin[2];
  }
  PFOR(i, nu) residual[9 + i] = c.actuator_force[i];
  // the 26-wide slices start at 7 / 6 and straddle the cube's free joint (hand.cc:75-80)
  PFOR(i, 26) {
    residual[9 + nu + i] = c.qpos[7 + i] - M.key_qpos[key * M.nq + 7 + i];
    residual[9 + nu + 26 + i] = c.qvel[6 + i];
  }
}

DEV void task_residual(Ctx &c, double *residual) {
  const DevModel &M = *c.M;
  int id = M.task.task_id;
  if (id == 0) {          // particle_residual.h:33-43
    PFOR(i, M.nq) residual[i] = c.qpos[i] - (i < 2 ? c.mocap_pos[i] : 0.0);
    PFOR(i, M.nv) residual[2 + i] = c.qvel[i];
  } else if (id == 1) {   // cartpole.cc:36-49
    if (LANE == 0) {
      residual[0] = cos(c.qpos[1]) - 1;
      residual[1] = c.qpos[0] - MD(task.parameters)[0];
      residual[2] = c.qvel[1];
      residual[3] = c.ctrl[0];
    }
  } else if (id == 3) {   // copy state (rollout_test.cc:40-60)
    PFOR(i, M.nq) residual[i] = c.qpos[i];
    PFOR(i, M.nv) residual[M.nq + i] = c.qvel[i];
  } else if (id == 2) {
    residual_quadruped(c, residual);
  } else if (id == 4) {
    residual_humanoid_track(c, residual);
  } else if (id == 5) {
    residual_humanoid_stand(c, residual);
  } else if (id == 6) {
    residual_humanoid_walk(c, residual);
  } else if (id == 7) {
    residual_shadow(c, residual);
  } else if (id == 8) {   // walker.cc:39-57: control, torso height - goal, torso z axis z - 1, subtree x velocity - goal
    int nu = M.nu, b = MI(task.int_data)[0];
    PFOR(i, nu) residual[i] = c.ctrl[i];
    if (LANE == 0) {
      residual[nu] = c.xpos[3 * b + 2] - MD(task.parameters)[0];
      residual[nu + 1] = c.xmat[9 * b + 8] - 1.0;
      residual[nu + 2] = c.subtree_linvel[3 * b] - MD(task.parameters)[1];
    }
  } else if (id == 9) {   // acrobot.cc:34-49: goal - tip (z, x), joint velocities, control
    if (LANE == 0) {
      int g = MI(task.int_data)[0], t = MI(task.int_data)[1];
      residual[0] = c.site_xpos[3 * g + 2] - c.site_xpos[3 * t + 2];
      residual[1] = c.site_xpos[3 * g] - c.site_xpos[3 * t];
      residual[2] = c.qvel[0];
      residual[3] = c.qvel[1];
      residual[4] = c.ctrl[0];
    }
  }
  c.warning = wave_or_i(c.warning);      // a ray miss is raised by the lane that cast it
  SYNC();
}

// ======================================================================================
// cost: Norm (mjpc/norm.cc:50-210, value only) and CostValue (mjpc/task.cc:71-110)
// ======================================================================================
DEV double norm_value(const double *x, const double *params, int n, int type) {
  double y = 0, p = params[0], q = params[1];
  switch (type) {
    case -1: y = x[0]; break;
    case 0: for (int i = 0; i < n; i++) y += x[i] * x[i]; y *= 0.5; break;
    case 1: { double cq = 0; for (int i = 0; i < n; i++) cq += x[i] * x[i];
              double a = pow(cq, q / 2) + pow(p, q); y = pow(a, 1 / q) - p; break; }
    case 2: { double s = 0; for (int i = 0; i < n; i++) s += x[i] * x[i]; y = sqrt(s + p * p) - p; break; }
    case 3: for (int i = 0; i < n; i++) y += p * p * (cosh(x[i] / p) - 1.0); break;
    case 5: for (int i = 0; i < n; i++) y += pow(fabs(x[i]), p); break;
    case 6: for (int i = 0; i < n; i++) { double s = sqrt(x[i] * x[i] + p * p); y += s - p; } break;
    case 7: for (int i = 0; i < n; i++) { double a = fabs(x[i]); double d = pow(a, q); double e = d + pow(p, q); y += pow(e, 1 / q) - p; } break;
    case 8: for (int i = 0; i < n; i++) { if (p > 0) { double s = exp(x[i] / p); y += p * log(1 + s); } else y += x[i] > 0 ? x[i] : 0; } break;
    default: break;
  }
  return y;
}
DEV double cost_value(Ctx &c, const double *residual) {
  const DevTask &T = c.M->task;
  PFOR(k, T.num_term) {
    int fs = 0, ps = 0;
    for (int j = 0; j < k; j++) { fs += MI(task.dim_norm_residual)[j]; ps += MI(task.num_norm_parameter)[j]; }
    double prm[2] = {0, 0};
    for (int j = 0; j < MI(task.num_norm_parameter)[k] && j < 2; j++) prm[j] = MD(task.norm_parameter)[ps + j];
    c.terms[k] = MD(task.weight)[k] * norm_value(residual + fs, prm, MI(task.dim_norm_residual)[k], MI(task.norm)[k]);
  }
  SYNC();
  double cost = 0;
  for (int k = 0; k < T.num_term; k++) cost += c.terms[k];     // ascending k, like task.cc:99-102
  SYNC();
  if (fabs(T.risk) < 1e-6) return cost;
  return (exp(T.risk * cost) - 1.0) / T.risk;
}

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
  R.ds = M.nq + M.nv; R.nr = M.task.num_residual; R.ntr = 3 * M.task.num_trace;
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
  PFOR(e, nv * M.nvp) { c.qM[e] = 0; c.qH[e] = 0; }
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
  velocity_stage<NVT>(c, t + 1);                // helper 0 builds and factors M meanwhile (ph_inertia)
#else
  crb_and_factor<NVT>(c); PROF(c, 3);
  velocity_stage<NVT>(c, 0); PROF(c, 6);
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
  if (LANE == 0) c.misc[HX_KIND] = 0;           // release the helper wave
  flag_set(c.misc + HX_JOB, ++c.hseq);
#endif
  if (!last && bad_values(c.qacc, c.M->nv)) c.warning |= WARN_BADQACC;
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
DEV_NOINLINE void ph_prefactor(KP Kc) {
  Ctx c; ctx_open(c, Kc, 1);
  const DevModel &M = *c.M;
  if (M.any_damping) {
    int nv = M.nv, nvp = M.nvp;
    double h = M.timestep;
    PFOR(e, nv * nvp) { int i = e / nvp, j = e - i * nvp; c.qL[e] = c.qM[e] + ((i == j) ? h * MD(dof_damping)[i] : 0.0); }
    chol_factor<NVT>(c.qL, c.Linv, c.scr_a, nv, nvp, c.M->tree_ok);
  }
  PROFW(c, 11);
  ctx_close(c);
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
  PFOR(i, nv) c.qacc_ws[i] = c.qacc[i];
  if (M.any_damping) {
    PFOR(i, nv) c.Mgrad[i] = c.qfrc_smooth[i] + c.qfrc_constraint[i];
    chol_solve<NVT>(c.qL, c.Linv, c.Mgrad, nv, nvp, c.M->tree_ok);       // factor of M + h*B from ph_prefactor
    PFOR(i, nv) c.qvel[i] += h * c.Mgrad[i];
  } else {
    PFOR(i, nv) c.qvel[i] += h * c.qacc[i];
  }
  SYNC();
  PFOR(j, M.njnt) {
    int qa = MI(jnt_qposadr)[j], da = MI(jnt_dofadr)[j], type = MI(jnt_type)[j];
    if (type == 0) {
      for (int k = 0; k < 3; k++) c.qpos[qa + k] += h * c.qvel[da + k];
      d_quatintegrate(c.qpos + qa + 3, c.qvel + da + 3, h);
    } else if (type == 1) {
      d_quatintegrate(c.qpos + qa, c.qvel + da, h);
    } else c.qpos[qa] += h * c.qvel[da];
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
// qacc_warmstart
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
    if (r0) ph_solve<NVT>(Kc, last, t);
#if MJPC_HELPER
    if (ROLEH) ph_solve_helper<NVT>(Kc, t);
#endif
    if (r1) { CostOut o = ph_residual_cost(Kc, t, last); total += o.cost; if (!last) ph_prefactor<NVT>(Kc); }
    XBAR(); RPROF(6);
    if (uniform_i(misc[3]) | uniform_i(misc[11])) { failure = 1; break; }
    if (r0 && !last) ph_integrate<NVT>(Kc, t);
  }
  XBAR();
  if (r1) ph_finish(Kc, total, failure, t_fail, total_before);
}
