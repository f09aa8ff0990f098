// model.h — device-side model / task description, LDS layout and kernel parameter block.
//
// HBM layout: the model is two flat device buffers (one int32, one fp64) plus this struct of
// pointers into them; it is read-only and shared by every candidate.  Every workgroup copies the hot prefix of both
// buffers (everything but the keyframe tables) into its LDS once per rollout (Lay::mc_d / mc_i) and the phases read
// the tables from there: a dependent table walk costs LDS latency instead of an L2/HBM round trip per hop.  Per-candidate mutable state ("mjData") is NOT in HBM: it lives in LDS for the whole
// horizon (Lay gives the carve-up); only the Trajectory record (states/actions/times/residual/
// costs/trace per step, mjpc/trajectory.h:74-86) is streamed out, row-major per candidate.
#pragma once
#include <stdint.h>

#define CON_STRIDE_ELLIPTIC 45  // doubles per contact in LDS incl. the 18 cone-Hessian factors at CON_H (odd: conflict-free field access)
#define NVP_OF(n) ((((n) + 1)) | 1)   // row stride of nv-wide tables: odd (conflict-free column access) with >= 1 spare column
#define CON_STRIDE_PLAIN 27     // without the cone Hessian (pyramidal / frictionless models)
#define CON_DIST 0
#define CON_POS 1
#define CON_FRAME 4
#define CON_INCLUDEMARGIN 13
#define CON_FRICTION 14
#define CON_SOLREF 19
#define CON_SOLIMP 21
#define CON_MU 26
#define CON_H 27
#ifndef MJPC_SIDE_JOB
#define MJPC_SIDE_JOB 0      // elliptic models: the side wave shares the owner's per-iterate jobs from this job of a step on (1 = the first; 0 = never: measured slower on the A1)
#endif
#define CONI_STRIDE 4          // ints per contact: dim, geom1, geom2, efc_address
#define MAX_ACTIVE_PAIRS 192
// efc_id of a contact row: contact index | contact dim << 8 | first row of the contact << 16
#define EFC_CON_ID(ci, dim, r0) ((ci) | ((dim) << 8) | ((r0) << 16))
#define EFC_CON_CI(id) ((id) & 255)
#define EFC_CON_DIM(id) (((id) >> 8) & 255)
#define EFC_CON_R0(id) ((id) >> 16)

// Dof-tree topologies known at compile time, keyed by nv.  When the model's dof_parentid equals the table (DevModel::tree_ok,
// checked on the host) the register L^T D L factorisations eliminate the independent branches of one tree level together and
// skip the structural zeros of M's pattern; any other tree of the same nv uses the dense elimination order.
// parent(k) is the ELIMINATION-tree parent: the model's dof_parentid, except for "hub" links (real_parent differs): dofs of another
// kinematic tree that the Hessian couples with this subtree all the time.  In the hand task every finger chain can touch the
// cube, so the cube's free joint is made the (virtual) ancestor of the wrist: the pattern  tree + hub  then covers M and every
// hand-cube contact, and only a contact between two fingers needs the dense order.
template <int N> struct DofTree { static constexpr bool known = false; static constexpr int parent(int) { return -1; } static constexpr int real_parent(int) { return -1; } };
template <> struct DofTree<18> {      // floating base + 4 chains of 3 (quadruped)
  static constexpr bool known = true;
  static constexpr int parent(int k) { constexpr int p[18] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 5, 9, 10, 5, 12, 13, 5, 15, 16}; return p[k]; }
  static constexpr int real_parent(int k) { return parent(k); }
};
template <> struct DofTree<27> {      // floating base + 3-dof waist carrying two 6-dof legs, two 3-dof arms on the base (humanoid)
  static constexpr bool known = true;
  static constexpr int parent(int k) {
    constexpr int p[27] = {-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 8, 15, 16, 17, 18, 19, 5, 21, 22, 5, 24, 25};
    return p[k];
  }
  static constexpr int real_parent(int k) { return parent(k); }
};
template <> struct DofTree<33> {      // goal ball (0-2) | cube free joint (3-8) = hub | wrist (9, 10) carrying 4 + 4 + 4 + 5 + 5 finger dofs (shadow hand)
  static constexpr bool known = true;
  static constexpr int parent(int k) {
    constexpr int p[33] = {-1, 0, 1, -1, 3, 4, 5, 6, 7, /* hub link */ 8, 9, 10, 11, 12, 13, 10, 15, 16, 17, 10, 19, 20, 21, 10, 23, 24, 25, 26, 10, 28, 29, 30, 31};
    return p[k];
  }
  static constexpr int real_parent(int k) { return k == 9 ? -1 : parent(k); }
};
template <int N> constexpr bool dof_tree_matches(const int *dof_parentid) {
  if (!DofTree<N>::known) return false;
  for (int k = 0; k < N; k++) if (dof_parentid[k] != DofTree<N>::real_parent(k)) return false;
  return true;
}
// elimination-tree parent of dof k for a model with nv dofs whose tree matched (tree_ok); the model's own parent otherwise
static inline int pattern_parent(int nv, int tree_ok, const int *dof_parentid, int k) {
  if (!tree_ok) return dof_parentid[k];
  if (nv == 18) return DofTree<18>::parent(k);
  if (nv == 27) return DofTree<27>::parent(k);
  if (nv == 33) return DofTree<33>::parent(k);
  return dof_parentid[k];
}

enum { DYN_NONE = 0, DYN_INTEGRATOR = 1, DYN_FILTER = 2, DYN_FILTEREXACT = 3 };      // MJPC_DYN_* (mjtDyn)
enum { CNSTR_EQUALITY = 0, CNSTR_FRICTION_DOF = 1, CNSTR_FRICTION_TENDON = 2, CNSTR_LIMIT_JOINT = 3, CNSTR_LIMIT_TENDON = 4, CNSTR_CONTACT_FRICTIONLESS = 5, CNSTR_CONTACT_PYRAMIDAL = 6, CNSTR_CONTACT_ELLIPTIC = 7 };
enum { STATE_SATISFIED = 0, STATE_QUADRATIC = 1, STATE_LINEARNEG = 2, STATE_LINEARPOS = 3, STATE_CONE = 4 };
enum { WARN_BADQPOS = 1, WARN_BADQVEL = 2, WARN_BADQACC = 4, WARN_CONTACTFULL = 8, WARN_CNSTRFULL = 16, WARN_RAY = 32, WARN_SYNC = 64, WARN_UNSUPPORTED = 128 };

struct DevTask {
  int task_id, num_residual, num_term, num_trace, num_parameter, num_int, num_dbl;
  double risk;
  const int *dim_norm_residual, *norm, *num_norm_parameter, *trace_objtype, *trace_objid, *int_data;
  const double *weight, *norm_parameter, *parameters, *dbl_data;
};

struct DevModel {
  int nq, nv, nu, nbody, njnt, ngeom, nsite, nmocap, nkey, nvp, ntendon;
  int nlevel, npair, nfric, nlimit, nray, nmpair, nhpair, nzpair, nconmax, nefcmax, any_damping;
  int int_dense;       // 0: Euler / implicitfast inside the factorisation pattern (ph_prefactor); 1: implicitfast with fluid forces, 2: full implicit (mjINT_IMPLICIT) - dense M - h dF/dv built and LU-solved in ph_integrate
  int int_scratch;     // doubles of LDS the dense path may use from efc_J on (the constraint rows are dead after the solve)
  int nlimit_ball;      // limited ball joints (limit_ball[])
  int limit_cross;      // some limited tendon couples dofs outside the Hessian's pattern: every build is a dense one
  int ntendon_passive;  // tendons with a spring or a damper (tpass_*)
  int smooth_extras;    // ntendon_passive + nsiteact + ngravcomp + fluid + nactfrc: one test in velocity_stage for all the rare extras
  int ntfric;           // tendons with friction loss (tfric_*): one static friction row each, behind the single-entry rows
  int cone, iterations, ls_iterations, disableflags, con_stride, maxdim, tree_ok, nact;
  double timestep, gravity[3], impratio, tolerance, ls_tolerance, meaninertia;
  const int *body_parentid, *body_rootid, *body_mocapid, *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr;
  const double *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_subtreemass, *body_inertia, *body_invweight0;
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid;
  const double *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin, *jnt_solref, *jnt_solimp, *qpos0, *qpos_spring;
  const int *dof_bodyid, *dof_parentid;
  const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0, *dof_solref, *dof_solimp;
  const int *geom_type, *geom_condim, *geom_bodyid, *geom_priority;
  const double *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solmix, *geom_solref, *geom_solimp, *geom_margin, *geom_gap, *geom_rbound;
  const int *site_bodyid;
  const double *site_pos, *site_quat;
  // actuator transmissions flattened on the host: actuator i owns entries [act_adr[i], act_adr[i+1]) = (dof, qpos address,
  // coefficient = gear [* tendon coefficient]); act_of[e] = the actuator of entry e
  const int *act_adr, *act_dof, *act_qpos, *act_of, *actuator_ctrllimited, *actuator_forcelimited, *actuator_biastype;
  int fluid; double density, viscosity, wind[3];      // inertia-box fluid model (mj_passive), fluid = density > 0 || viscosity > 0
  int nactfrc;                              // joints with a clamp on the total actuator force: actfrc_dof, actfrc_range [lo, hi]
  const int *actfrc_dof; const double *actfrc_range;
  int ngravcomp;                            // bodies with gravity compensation: gc_body, gc_force [3 each, world frame]
  const int *gc_body; const double *gc_force;
  int nsiteact;                             // site transmissions: sact_i [actuator, site, body], sact_g [force 3, torque 3 in the body frame]
  const int *sact_i; const double *sact_g;
  int nrsact;                               // site transmissions with a reference site: rsact_i [actuator, site, refsite, body, refbody, common-dof mask lo, hi],
  const int *rsact_i, *rsact_of;            // rsact_g [the translational gear in the reference BODY's frame], rsact_of[actuator] = entry or -1
  const double *rsact_g;
  int noslip_iterations; double noslip_tolerance;      // mj_solNoSlip after the Newton solve (noslip.h)
  int na;                                   // activation states (one per stateful actuator); tables below only when na > 0
  const int *actuator_dyntype, *actuator_actadr, *actuator_actlimited;
  const double *actuator_dynprm, *actuator_actrange;
  const int *dact_adr, *dact_e;             // entries grouped by dof: dof d owns dact_e[dact_adr[d] .. dact_adr[d+1])
  const double *act_coef;
  const double *actuator_gainprm, *actuator_biasprm, *actuator_gear, *actuator_ctrlrange, *actuator_forcerange;
  const int *tendon_adr, *tendon_num, *tendon_limited, *wrap_dofadr, *wrap_qposadr;
  const double *wrap_prm, *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_invweight0;
  const int *tpass_id;                      // tendons with passive forces
  const double *tpass_prm;                  // [4 each] stiffness, damping, spring dead band lo / hi
  int neq, neqrow, neq_connect;             // active equality constraints (eq_tab: type, obj1, obj2, first row; eq_prm: data 11, solref 2, solimp 5)
  const int *eq_tab; const double *eq_prm;
  int nidrv;                                // implicitfast: entries of -dF/dv beyond joint damping (idrv_e: i, j, actuator or -1; idrv_c: coefficient)
  const int *idrv_e; const double *idrv_c;
  const int *tfric_id;                      // tendons with friction loss
  const double *tfric_prm;                  // [8 each] frictionloss, solref[2], solimp[5]
  const double *key_qpos, *key_mpos;
  const int *geom_dataid, *mesh_vertadr, *mesh_vertnum;     // convex meshes: read from HBM / L2 (never in the LDS copy)
  const double *mesh_vert;
  const int *hfield_nrow, *hfield_ncol, *hfield_adr;        // height fields: read from HBM / L2
  const double *hfield_size, *hfield_data;
  // derived on the host at create()
  const int *level_adr, *level_body;        // bodies grouped by tree depth (depth >= 1)
  const int *subtree_adr, *subtree_list;    // bodies of each subtree, self first, ascending ids
  const int *chain_adr, *chain_list;        // ancestors of each body from the root's child down to the body itself
  const int *mpair_i, *mpair_j;             // (dof i, ancestor-or-self dof j): the non-zeros of M
  const int *hpair_i, *hpair_j;             // the Hessian's pattern: (dof i, elimination-tree ancestor-or-self j), then the nv gradient entries (i, nv)
  const int *zpair_i, *zpair_j;             // the rest of the lower triangle (structural zeros of that pattern)
  const unsigned long long *body_dofmask;   // bit d set <=> dof d moves body
  const unsigned long long *body_patmask;   // the same along the elimination tree (hub dofs included): cross-branch test of a contact
  const int *pair_gg; const double *pair_bp; // statically filtered geom pairs (type1 <= type2): geom1 | geom2 << 16; [margin, rbound1 (-1: plane), rbound2]
  const int *fric_dof, *limit_jnt, *limit_ball, *ray_geom;
  DevTask task;
};

// LDS carve-up, offsets in doubles (ints live behind `ints`, offsets in ints)
#define MISC_INTS 48     // per-candidate scalars and hand-shake flags in LDS (core.h / solver.h: misc[])
struct Lay {
  int qpos, qvel, ctrl, qacc, qacc_ws, qacc_smooth, qfrc_smooth, qfrc_bias, qfrc_constraint, actuator_force;
  int mocap_pos, mocap_quat;
  int xpos, xquat, xmat, xipos, ximat, xanchor, xaxis, geom_xpos, geom_xmat, site_xpos;
  int subtree_com, cinert, crb, cdof, cvel, cdof_dot, cacc, cfrc, cfrc_sub, subtree_linvel, bodytmp;
  int qM, qL, qH, Linv, Hinv;
  int efc_J, efc_JA, efc_D, efc_R, efc_aref, efc_force, efc_jar, efc_jv, efc_floss, efc_pos, efc_margin, efc_diag;
  int contact;
  int Ma, grad, Mgrad, search, Mv, vtmp, sgl;
  int knot_times, knot_values, residual, terms, red, prof, scr_a, scr_b;
  int xfrc;            // external body forces (NoisyRollout of the robust planner), 6 per body
  int noslip;          // noslip pass: M^-1 J^T rows [nefcmax x nvp], contact blocks [nconmax x 36], b and diagonal [2 x nefcmax] (0 without noslip)
  int mc_d, mc_i;      // LDS copy of the model tables: fp64 part, int part (both offsets in doubles)
  int ints;            // start of the int region (in doubles)
  int i_efc_type, i_efc_id, i_efc_state, i_efc_dof, i_con, i_active, i_misc, i_hpair;
  int total_doubles;   // LDS bytes = 8 * total_doubles
};

struct KParams {
  DevModel M;
  Lay L;
  const int *ibase; const double *dbase;   // the two model buffers in HBM; [0, cache_i) / [0, cache_d) are LDS-cached
  int cache_i, cache_d;
  // plan inputs (device pointers)
  const double *state, *mocap, *knot_times, *knot_values, *noise_eps, *noise_std, *cand_knots;
  const double *userdata; int nuserdata;     // mjData.userdata of the plan's state, constant over the rollouts (no built-in residual reads it yet)
  double xfrc_std, xfrc_rate;
  const int *noise_sel;
  double time, sigma0, sigma1;
  unsigned long long seed, stream;
  int P, interp, H, N, offset, nlocal, use_device_noise, nominal_index;
  // capacity tiers (engine.hip): tier > 0 marks the dense-tier launch; a candidate that overflows its buffers leaves a checkpoint
  // (state at the failing step) in ckpt[] and the retry launch (retry = 1) resumes it from there at full capacity
  double *ckpt; int ckpt_stride, tier;
  int retry;           // 1: re-run only the candidates whose failure[] holds a buffer-overflow bit (capacity tiers, engine.hip)
  int fault;           // test-suite fault injection (0 = none; 1 = drop one helper hand-shake, see solver.h)
  // outputs (device), row-major per local candidate
  double *states, *actions, *times, *residual, *costs, *trace, *knots, *returns;
  int *failure, *diag;
  double *frame;       // kinematic frame of local candidate 0 at step 0: xpos | xmat | site_xpos | subtree_com | subtree_linvel
  long long *prof;     // optional per-candidate phase cycle counters (MJPC_PROFILE builds)
};
