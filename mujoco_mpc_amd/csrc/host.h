// host.h — host-side model compilation shared by the HIP engine (engine.hip) and the 1-lane
// emulation build used by the CPU test tier (tests/emu).  Flattens MjpcHipModel/MjpcHipTask into
// one int32 and one fp64 buffer, derives the static tables the kernel needs, and computes the
// LDS carve-up.
#pragma once
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/mjpc_hip.h"
#include "model.h"

// LDS carve-up of one workgroup (offsets in doubles).  Arrays that are only alive while the constraint rows are built share
// their storage with solve-phase vectors: efc_diag <-> efc_force, efc_margin <-> efc_jv (the next step rebuilds them).
struct PackedModel {
  std::vector<int> ib;
  std::vector<double> db;
  DevModel M;          // pointer fields hold OFFSETS (as intptr) until relocate()
  Lay L;
  size_t task_i0, task_d0, task_i_cap, task_d_cap;   // task region inside ib/db
  size_t cache_i, cache_d;                           // prefix of ib/db copied into LDS by every workgroup
  size_t hot_i = 0, hot_d = 0;                       // the shorter prefix the hot-tables-only flavour copies
  std::string error;
};

namespace mjpc_host {

// diagnostics knobs (include/mjpc_hip_debug.h): set explicitly through mjpc_hip_debug_set, never read from the environment, and
// looked at only while an engine is created
inline std::mutex &debug_mutex() { static std::mutex m; return m; }
inline std::map<std::string, std::string> &debug_table() { static std::map<std::string, std::string> t; return t; }
inline void debug_set(const char *name, const char *value) {
  std::lock_guard<std::mutex> lock(debug_mutex());
  if (value) debug_table()[name] = value; else debug_table().erase(name);
}
inline bool debug_knob(const char *name, std::string *value = nullptr) {
  std::lock_guard<std::mutex> lock(debug_mutex());
  auto it = debug_table().find(name);
  if (it == debug_table().end()) return false;
  if (value) *value = it->second;
  return true;
}

template <class T> static inline const T *as_off(size_t off) { return reinterpret_cast<const T *>(off * sizeof(T) + 1); }

static inline size_t put_i(PackedModel &p, const int *src, size_t n) {
  size_t o = p.ib.size();
  for (size_t i = 0; i < n; i++) p.ib.push_back(src ? src[i] : 0);
  if (n == 0) p.ib.push_back(0);
  return o;
}
static inline size_t put_d(PackedModel &p, const double *src, size_t n) {
  size_t o = p.db.size();
  for (size_t i = 0; i < n; i++) p.db.push_back(src ? src[i] : 0.0);
  if (n == 0) p.db.push_back(0.0);
  return o;
}

// pack the task tables at the current end of ib/db and fill M.task with offsets
static inline void pack_task(PackedModel &p, const MjpcHipTask *t) {
  DevTask &T = p.M.task;
  T.task_id = t->task_id; T.num_residual = t->num_residual; T.num_term = t->num_term; T.num_trace = t->num_trace;
  T.num_parameter = t->num_parameter; T.num_int = t->num_int; T.num_dbl = t->num_dbl; T.risk = t->risk;
  int np = 0;
  for (int k = 0; k < t->num_term; k++) np += t->num_norm_parameter[k];
  T.dim_norm_residual = as_off<int>(put_i(p, t->dim_norm_residual, t->num_term));
  T.norm = as_off<int>(put_i(p, t->norm, t->num_term));
  T.num_norm_parameter = as_off<int>(put_i(p, t->num_norm_parameter, t->num_term));
  T.trace_objtype = as_off<int>(put_i(p, t->trace_objtype, t->num_trace));
  T.trace_objid = as_off<int>(put_i(p, t->trace_objid, t->num_trace));
  T.int_data = as_off<int>(put_i(p, t->int_data, t->num_int));
  T.weight = as_off<double>(put_d(p, t->weight, t->num_term));
  T.norm_parameter = as_off<double>(put_d(p, t->norm_parameter, np));
  T.parameters = as_off<double>(put_d(p, t->parameters, t->num_parameter));
  T.dbl_data = as_off<double>(put_d(p, t->dbl_data, t->num_dbl));
}

// lean: the dense tier's layout (rollout_dense2.hip, MJPC_LEAN_LDS): spline knots and the Hessian entry table are read from HBM / L2
static inline void make_layout(const PackedModel &p, Lay &L, const MjpcHipModel *m, const MjpcHipTask *t, int P_max, size_t cache_d, size_t cache_i, bool lean = false, bool reg_solver = false) {
  const DevModel &M = p.M;
  int nb = m->nbody, nj = m->njnt, nv = m->nv, ng = m->ngeom, ns = m->nsite, nu = m->nu;
  int o = 0;
  int nvp = M.nvp, ne = M.nefcmax, nc = M.nconmax, nr = t->num_residual;
#define A_(f, n) L.f = o; o += (int)(n)
  A_(qpos, m->nq); A_(qvel, nv); A_(ctrl, nu + 1 + 2 * m->na);      // activations and their derivatives sit behind ctrl (C_ACT / C_ACTDOT)
  A_(qacc, nv); A_(qacc_ws, nv); A_(qacc_smooth, nv); A_(qfrc_smooth, nv);
  A_(qfrc_bias, nv); A_(qfrc_constraint, nv); A_(actuator_force, nu + 1); A_(mocap_pos, 3 * m->nmocap + 3); A_(mocap_quat, 4 * m->nmocap + 4);
  A_(xpos, 3 * nb); A_(xquat, 4 * nb); A_(xmat, 9 * nb); A_(xipos, 3 * nb); A_(ximat, 9 * nb); A_(xanchor, 3 * nj + 3); A_(xaxis, 3 * nj + 3);
  A_(geom_xpos, 3 * ng + 3); A_(geom_xmat, 9 * ng + 9); A_(site_xpos, 3 * ns + 3);
  // lean layout: the composite inertias and the RNE intermediates only live between two solves and share the block of the
  // solver's scaled rows (efc_JA), which only lives inside a solve (placed below)
#define B_(f, n) if (!lean) { A_(f, n); }
  A_(subtree_com, 3 * nb); B_(cinert, 10 * nb); B_(crb, 10 * nb); A_(cdof, 6 * nv + 18); A_(cvel, 6 * nb); B_(cdof_dot, 6 * nv + 18);
  B_(cacc, 6 * nb); B_(cfrc, 6 * nb); B_(cfrc_sub, 6 * nb); A_(subtree_linvel, 3 * nb); A_(bodytmp, 3 * nb);
#undef B_
  A_(qM, nv * nvp + 1); A_(qL, nv * nvp + 1);
  A_(qH, (lean && reg_solver && nv > 32) ? 1 : nv * nvp + 1);      // dense tier, one column group (64 / nv == 1): the register solve factors straight from its row registers, no Hessian in LDS
  A_(Linv, nv + 1); A_(Hinv, nv + 1);
  // efc_JA holds the scaled rows of the Newton Hessian: the active contact rows (padded to 8) + one negative row per cone
  // contact (padded to 4)
  int ja_rows = ((ne - M.nfric + 7) & ~7) + (m->cone == MJPC_CONE_ELLIPTIC ? ((nc + 3) & ~3) : 0) + 4 + ((M.ntfric + 3) & ~3);
  
  A_(efc_J, (ne - M.nfric) * nvp + 1);
  // the register Newton path (solver_reg.h) uses the same block for its gradient scratch (64), the line search's row / contact
  // records (7 per row, 17 per contact) and, on elliptic models, the contacts' dof lists (a byte per dof), their (contact, row) pairs (two bytes each) and one partial Hessian
  // per worker wave (nv x nvp each)
  int ja_need = 64 + ne * 7 + nc * 17 + (m->cone == MJPC_CONE_ELLIPTIC ? (nc * nv + 7) / 8 + (nc * nv + 3) / 4 + (MJPC_SIDE_JOB > 0 ? 3 : 2) * nv * nvp : 0);
  // reg_solver: the kernel is a compile-time-nv instantiation, whose Newton solve keeps the Hessian in registers and never
  // builds the scaled-row table: the block only has to hold the records / partials above (hand: 12 KB instead of 35 KB)
  int ja_size = reg_solver ? ja_need + 1 : ja_rows * nvp + 1;
  if (ja_size < ja_need) ja_size = ja_need;
  A_(efc_JA, ja_size);
  if (lean) {
    int q = L.efc_JA;
    L.cinert = q; q += 10 * nb; L.crb = q; q += 10 * nb; L.cdof_dot = q; q += 6 * nv + 18;
    L.cacc = q; q += 6 * nb; L.cfrc = q; q += 6 * nb; L.cfrc_sub = q; q += 6 * nb;
    if (q > o) o = q;
  }
  A_(efc_D, ne); A_(efc_R, ne); A_(efc_aref, ne); A_(efc_force, ne); A_(efc_jar, ne); A_(efc_jv, ne);
  A_(efc_floss, ne); A_(efc_pos, ne);
  L.efc_margin = L.efc_jv; L.efc_diag = L.efc_force;
  A_(contact, nc * M.con_stride + 1);
  A_(Ma, nv + 1); A_(grad, nv + 1); A_(Mgrad, nv + 1); A_(search, nv + 1); A_(Mv, nv + 1); A_(vtmp, nv + 1); A_(sgl, 4 * nv + 1);
  if (lean) { L.knot_times = 0; L.knot_values = 0; } else { A_(knot_times, P_max); A_(knot_values, P_max * nu + 1); }
  A_(residual, nr + 1); A_(terms, t->num_term + 1); A_(red, 8); A_(prof, 26);
  A_(scr_a, nv + 1); A_(scr_b, nv + 1);                       // solve-phase scratch of the side wave / of the helper's cost at qacc_smooth
  if (lean) L.xfrc = 0; else { A_(xfrc, 6 * nb); }             // no force noise on the dense tier (engine.hip)
  if (m->noslip_iterations > 0) { A_(noslip, ne * nvp + nc * 36 + 2 * ne + 1); } else L.noslip = 0;
  A_(mc_d, cache_d + 1); A_(mc_i, (cache_i + 2) / 2);
  L.ints = o;
#undef A_
  int io = 0;
  L.i_efc_type = io; io += ne; L.i_efc_id = io; io += ne; L.i_efc_state = io; io += ne; L.i_efc_dof = io; io += ne;
  L.i_con = io; io += nc * CONI_STRIDE; L.i_active = io; io += (ne + nc + M.ntfric > MAX_ACTIVE_PAIRS ? ne + nc + M.ntfric : MAX_ACTIVE_PAIRS); L.i_misc = io; io += MISC_INTS;
  L.i_hpair = io; if (!lean) io += M.nhpair + nv;      // LDS copy of the Hessian/gradient entry table (i | j << 8)
  L.total_doubles = o + (io + 1) / 2;
}

// use_cache: lay out an LDS copy of the model tables (rollout_cached.hip) or none (rollout_direct.hip)
// use_cache: whole LDS copy of the tables; hot_only (with use_cache == false): only the hot prefix
static inline bool build(PackedModel &p, const MjpcHipModel *m, const MjpcHipTask *t, int P_max, bool use_cache = true, bool lean = false, bool hot_only = false, bool reg_solver = false) {
  p.ib.clear(); p.db.clear(); p.error.clear();
  DevModel &M = p.M;
  memset(&M, 0, sizeof(M));
  int nb = m->nbody, nj = m->njnt, nv = m->nv, ng = m->ngeom, ns = m->nsite, nu = m->nu;
  // models the engine cannot roll out faithfully are refused (never silently approximated)
  if (nv > 64) { p.error = "nv > 64 not supported (dof bitmask)"; return false; }
  { // activation states: one per stateful actuator (integrator / filter / filterexact), addresses 0 .. na-1
    int cnt = 0;
    for (int i = 0; i < nu && m->actuator_dyntype; i++) if (m->actuator_dyntype[i] != MJPC_DYN_NONE) {
      int dt = m->actuator_dyntype[i], a = m->actuator_actadr ? m->actuator_actadr[i] : -1;
      if (dt < 0 || dt > MJPC_DYN_FILTEREXACT) { p.error = "actuator " + std::to_string(i) + ": dyntype " + std::to_string(dt) + " not supported (integrator / filter / filterexact only)"; return false; }
      if (a < 0 || a >= m->na || !m->actuator_dynprm || !m->actuator_actlimited || !m->actuator_actrange) { p.error = "actuator " + std::to_string(i) + ": activation address outside [0, na) or missing actuator_dynprm / actlimited / actrange"; return false; }
      cnt++;
    }
    if (cnt != m->na) { p.error = "na = " + std::to_string(m->na) + " but " + std::to_string(cnt) + " stateful actuators (one activation each)"; return false; } }
  if (m->solver != MJPC_SOL_NEWTON) { p.error = "only the Newton solver (mjSOL_NEWTON) is implemented"; return false; }
  if (m->integrator != MJPC_INT_EULER && m->integrator != MJPC_INT_IMPLICITFAST && m->integrator != MJPC_INT_IMPLICIT) { p.error = "only the Euler (with implicit joint damping), implicitfast and implicit integrators are implemented (not RK4)"; return false; }
  if (m->noslip_iterations < 0) { p.error = "noslip_iterations < 0"; return false; }
  for (int e = 0; e < m->neq; e++) {
    if (!m->eq_type || !m->eq_obj1id || !m->eq_obj2id || !m->eq_active0 || !m->eq_data || !m->eq_solref || !m->eq_solimp) { p.error = "neq > 0 but the eq_* tables are missing"; return false; }
    if (!m->eq_active0[e]) continue;
    int ty = m->eq_type[e], a = m->eq_obj1id[e], b = m->eq_obj2id[e];
    if (ty != MJPC_EQ_CONNECT && ty != MJPC_EQ_WELD && ty != MJPC_EQ_JOINT && ty != MJPC_EQ_TENDON) { p.error = "equality " + std::to_string(e) + ": only connect, weld, joint and tendon equalities are implemented (flex refused)"; return false; }
    if (ty == MJPC_EQ_TENDON && (a < 0 || a >= m->ntendon || b >= m->ntendon)) { p.error = "equality " + std::to_string(e) + ": tendon id out of range"; return false; }
    if ((ty == MJPC_EQ_CONNECT || ty == MJPC_EQ_WELD) && (a < 0 || a >= nb || b < 0 || b >= nb)) { p.error = "equality " + std::to_string(e) + ": body id out of range"; return false; }
    if (ty == MJPC_EQ_JOINT) {
      if (a < 0 || a >= nj || b >= nj) { p.error = "equality " + std::to_string(e) + ": joint id out of range"; return false; }
      if ((m->jnt_type[a] != MJPC_JNT_HINGE && m->jnt_type[a] != MJPC_JNT_SLIDE) || (b >= 0 && m->jnt_type[b] != MJPC_JNT_HINGE && m->jnt_type[b] != MJPC_JNT_SLIDE)) {
        p.error = "equality " + std::to_string(e) + ": joint equalities couple hinge / slide joints"; return false; }
    }
  }
  if (m->unsupported != 0) { p.error = "the model uses features outside the engine's model view (MJPC_UNSUP_* mask " + std::to_string(m->unsupported) + ": ellipsoid fluid model, non-fixed actuator gains, muscle / user actuator dynamics, spatial tendons, ...)"; return false; }
  { const int known = MJPC_DSBL_CONSTRAINT | MJPC_DSBL_EQUALITY | MJPC_DSBL_FRICTIONLOSS | MJPC_DSBL_LIMIT | MJPC_DSBL_CONTACT | MJPC_DSBL_SENSOR | MJPC_DSBL_MIDPHASE;
    if (m->disableflags & ~known) { p.error = "disableflags " + std::to_string(m->disableflags) + ": only constraint / frictionloss / limit / contact can be disabled"; return false; }
    if (m->enableflags & (MJPC_ENBL_OVERRIDE | MJPC_ENBL_MULTICCD)) { p.error = "enableflags: contact override and multiccd are not supported"; return false; } }
  if (m->nuserdata < 0) { p.error = "nuserdata < 0"; return false; }
  if (m->nefcmax > 192) { p.error = "nefcmax > 192 not supported (line-search rows per lane)"; return false; }
  if (m->nconmax > 64) { p.error = "nconmax > 64 not supported (one contact per lane in the solver)"; return false; }
  if (m->iterations > 250) { p.error = "solver iterations > 250 not supported (hand-shake sequence numbers)"; return false; }
  for (int i = 0; i < nu; i++)
    if (m->actuator_trntype[i] == MJPC_TRN_SITE) {
      if (!m->actuator_gear6 || m->actuator_trnid[i] < 0 || m->actuator_trnid[i] >= ns) { p.error = "actuator " + std::to_string(i) + ": site transmission without actuator_gear6 or with a site id out of range"; return false; }
      const int rs = m->actuator_refsite ? m->actuator_refsite[i] : -1;
      if (rs >= ns) { p.error = "actuator " + std::to_string(i) + ": reference site out of range"; return false; }
      if (rs >= 0) {
        const double *g = m->actuator_gear6 + 6 * i;
        if (g[3] != 0 || g[4] != 0 || g[5] != 0) { p.error = "actuator " + std::to_string(i) + ": a rotational gear together with a reference site is not implemented"; return false; }
        if (m->actuator_biastype[i] == MJPC_BIAS_AFFINE && m->actuator_biasprm[3 * i + 2] != 0 && m->integrator != MJPC_INT_EULER) { p.error = "actuator " + std::to_string(i) + ": a velocity bias on a site transmission under an implicit integrator is not implemented"; return false; }
      } else if (m->actuator_biastype[i] != MJPC_BIAS_NONE || (m->actuator_dyntype && m->actuator_dyntype[i] != MJPC_DYN_NONE)) { p.error = "actuator " + std::to_string(i) + ": site transmissions without a reference site are implemented for plain motors (no bias, no activation)"; return false; }
    } else if (m->actuator_trntype[i] != MJPC_TRN_JOINT && m->actuator_trntype[i] != MJPC_TRN_TENDON) {
      p.error = "actuator " + std::to_string(i) + ": only joint, fixed-tendon and site transmissions are supported"; return false; }
  M.na = m->na;
  M.fluid = (m->density > 0 || m->viscosity > 0) ? 1 : 0; M.density = m->density; M.viscosity = m->viscosity;
  for (int k = 0; k < 3; k++) M.wind[k] = m->wind[k];
  // the implicit integrators' velocity derivatives beyond the factorisation pattern's symmetric terms (fluid forces; the bias forces
  // of mjINT_IMPLICIT) go through a dense, LU-solved M - h dF/dv in ph_integrate
  M.int_dense = m->integrator == MJPC_INT_IMPLICIT ? 2 : ((M.fluid && m->integrator == MJPC_INT_IMPLICITFAST) ? 1 : 0);
  M.nq = m->nq; M.nv = nv; M.nu = nu; M.nbody = nb; M.njnt = nj; M.ngeom = ng; M.nsite = ns; M.nmocap = m->nmocap;
  M.nkey = m->nkey; M.nvp = NVP_OF(nv); M.ntendon = m->ntendon;
  M.cone = m->cone; M.iterations = m->iterations; M.ls_iterations = m->ls_iterations;
  // the device only knows "no contacts"; disabled constraint kinds are dropped from the host-made row lists below
  M.disableflags = (m->disableflags & (MJPC_DSBL_CONTACT | MJPC_DSBL_CONSTRAINT)) ? MJPC_DSBL_CONTACT : 0;
  const bool no_fric = m->disableflags & (MJPC_DSBL_CONSTRAINT | MJPC_DSBL_FRICTIONLOSS), no_limit = m->disableflags & (MJPC_DSBL_CONSTRAINT | MJPC_DSBL_LIMIT);
  M.timestep = m->timestep; for (int k = 0; k < 3; k++) M.gravity[k] = m->gravity[k];
  M.impratio = m->impratio; M.tolerance = m->tolerance; M.ls_tolerance = m->ls_tolerance; M.meaninertia = m->meaninertia;
  M.con_stride = (m->cone == MJPC_CONE_ELLIPTIC) ? CON_STRIDE_ELLIPTIC : CON_STRIDE_PLAIN;
  M.maxdim = 1;
  for (int g = 0; g < ng; g++) if (m->geom_condim[g] > M.maxdim) M.maxdim = m->geom_condim[g];
  M.tree_ok = (nv == 18 && dof_tree_matches<18>(m->dof_parentid)) || (nv == 27 && dof_tree_matches<27>(m->dof_parentid)) ||
              (nv == 33 && dof_tree_matches<33>(m->dof_parentid));
  if (debug_knob("dense_factor")) M.tree_ok = 0;           // diagnostics knob: dense elimination order for every factorisation
  M.nconmax = m->nconmax > 0 ? m->nconmax : 32;
  M.nefcmax = m->nefcmax > 0 ? m->nefcmax : 128;
#define PI_(f, n) M.f = as_off<int>(put_i(p, m->f, (size_t)(n)))
#define PD_(f, n) M.f = as_off<double>(put_d(p, m->f, (size_t)(n)))
  PI_(body_parentid, nb); PI_(body_rootid, nb); PI_(body_mocapid, nb); PI_(body_jntnum, nb); PI_(body_jntadr, nb);
  PI_(body_dofnum, nb); PI_(body_dofadr, nb);
  PD_(body_pos, 3 * nb); PD_(body_quat, 4 * nb); PD_(body_ipos, 3 * nb); PD_(body_iquat, 4 * nb); PD_(body_mass, nb);
  PD_(body_subtreemass, nb); PD_(body_inertia, 3 * nb); PD_(body_invweight0, 2 * nb);
  PI_(jnt_type, nj); PI_(jnt_qposadr, nj); PI_(jnt_dofadr, nj); PI_(jnt_bodyid, nj);
  PD_(jnt_pos, 3 * nj); PD_(jnt_axis, 3 * nj); PD_(jnt_stiffness, nj); PD_(jnt_range, 2 * nj); PD_(jnt_margin, nj);
  PD_(jnt_solref, 2 * nj); PD_(jnt_solimp, 5 * nj); PD_(qpos0, m->nq); PD_(qpos_spring, m->nq);
  PI_(dof_bodyid, nv); PI_(dof_parentid, nv);
  PD_(dof_armature, nv);
  // bodies by depth
  { std::vector<int> depth(nb, 0); int maxd = 0;
    for (int b = 1; b < nb; b++) { depth[b] = depth[m->body_parentid[b]] + 1; if (depth[b] > maxd) maxd = depth[b]; }
    std::vector<int> adr(maxd + 1, 0), list;
    for (int l = 1; l <= maxd; l++) { adr[l - 1] = (int)list.size(); for (int b = 1; b < nb; b++) if (depth[b] == l) list.push_back(b); }
    adr[maxd] = (int)list.size();
    M.nlevel = maxd;
    M.level_adr = as_off<int>(put_i(p, adr.data(), adr.size())); M.level_body = as_off<int>(put_i(p, list.data(), list.size())); }
  // subtree lists
  { std::vector<int> adr(nb + 1, 0), list;
    for (int b = 0; b < nb; b++) {
      adr[b] = (int)list.size();
      for (int c = b; c < nb; c++) { int a = c; while (a > b) a = m->body_parentid[a]; if (a == b) list.push_back(c); }
    }
    adr[nb] = (int)list.size();
    M.subtree_adr = as_off<int>(put_i(p, adr.data(), adr.size())); M.subtree_list = as_off<int>(put_i(p, list.data(), list.size())); }
  // ancestor chains (kinematics walks them in registers instead of sweeping the tree level by level)
  { std::vector<int> adr(nb + 1, 0), list;
    for (int b = 0; b < nb; b++) {
      adr[b] = (int)list.size();
      std::vector<int> up;
      for (int a = b; a > 0; a = m->body_parentid[a]) up.push_back(a);
      for (size_t k = up.size(); k-- > 0;) list.push_back(up[k]);
    }
    adr[nb] = (int)list.size();
    list.push_back(0);
    M.chain_adr = as_off<int>(put_i(p, adr.data(), adr.size())); M.chain_list = as_off<int>(put_i(p, list.data(), list.size())); }
  // everything so far = the tables of the dependent-load chains of kinematics / com / inertia / velocity sweep: the "hot" prefix
  // that the flavour without room for the whole copy still keeps in LDS (MIH / MDH in core.h)
  p.hot_i = p.ib.size(); p.hot_d = p.db.size();
  PD_(dof_damping, nv); PD_(dof_frictionloss, nv); PD_(dof_invweight0, nv);
  PD_(dof_solref, 2 * nv); PD_(dof_solimp, 5 * nv);
  PI_(geom_type, ng); PI_(geom_condim, ng); PI_(geom_bodyid, ng); PI_(geom_priority, ng);
  PD_(geom_size, 3 * ng); PD_(geom_pos, 3 * ng); PD_(geom_quat, 4 * ng); PD_(geom_friction, 3 * ng); PD_(geom_solmix, ng);
  PD_(geom_solref, 2 * ng); PD_(geom_solimp, 5 * ng); PD_(geom_margin, ng); PD_(geom_gap, ng); PD_(geom_rbound, ng);
  PI_(site_bodyid, ns); PD_(site_pos, 3 * ns); PD_(site_quat, 4 * ns);
  PI_(actuator_ctrllimited, nu); PI_(actuator_forcelimited, nu); PI_(actuator_biastype, nu);
  PD_(actuator_gainprm, 3 * nu); PD_(actuator_biasprm, 3 * nu); PD_(actuator_gear, nu);
  PD_(actuator_ctrlrange, 2 * nu); PD_(actuator_forcerange, 2 * nu);
  if (m->na > 0) { PI_(actuator_dyntype, nu); PI_(actuator_actadr, nu); PI_(actuator_actlimited, nu); PD_(actuator_dynprm, nu); PD_(actuator_actrange, 2 * nu); }
  PI_(tendon_adr, m->ntendon); PI_(tendon_num, m->ntendon);
  { std::vector<int> tl(m->ntendon, 0);
    if (!no_limit) for (int t = 0; t < m->ntendon; t++) tl[t] = m->tendon_limited[t];
    M.tendon_limited = as_off<int>(put_i(p, tl.data(), tl.size())); }
  PD_(wrap_prm, m->nwrap); PD_(tendon_range, 2 * m->ntendon); PD_(tendon_margin, m->ntendon);
  PD_(tendon_solref_lim, 2 * m->ntendon); PD_(tendon_solimp_lim, 5 * m->ntendon); PD_(tendon_invweight0, m->ntendon);
#undef PI_
#undef PD_
  // actuator transmissions -> (dof, qpos address, coefficient) entries: joint = one entry with the gear, fixed tendon = one per
  // wrapped joint with gear * coefficient (mj_transmission for mjTRN_JOINT / mjTRN_TENDON)
  { std::vector<int> adr(nu + 1, 0), da, qa, of; std::vector<double> cf;
    for (int i = 0; i < nu; i++) {
      adr[i] = (int)da.size();
      int id = m->actuator_trnid[i];
      if (m->actuator_trntype[i] == MJPC_TRN_SITE) continue;      // configuration-dependent moment: sact_* below
      if (m->actuator_trntype[i] == MJPC_TRN_TENDON) {
        if (id < 0 || id >= m->ntendon) { p.error = "actuator " + std::to_string(i) + ": tendon id out of range"; return false; }
        for (int w = m->tendon_adr[id]; w < m->tendon_adr[id] + m->tendon_num[id]; w++) {
          int j = m->wrap_objid[w];
          da.push_back(m->jnt_dofadr[j]); qa.push_back(m->jnt_qposadr[j]); of.push_back(i); cf.push_back(m->actuator_gear[i] * m->wrap_prm[w]);
        }
      } else {
        if (id < 0 || id >= nj) { p.error = "actuator " + std::to_string(i) + ": joint id out of range"; return false; }
        int t = m->jnt_type[id];
        if (t != MJPC_JNT_HINGE && t != MJPC_JNT_SLIDE) { p.error = "actuator " + std::to_string(i) + ": joint transmissions need a hinge or slide joint"; return false; }
        da.push_back(m->jnt_dofadr[id]); qa.push_back(m->jnt_qposadr[id]); of.push_back(i); cf.push_back(m->actuator_gear[i]);
      }
    }
    adr[nu] = (int)da.size(); M.nact = (int)da.size();
    M.act_adr = as_off<int>(put_i(p, adr.data(), adr.size())); M.act_dof = as_off<int>(put_i(p, da.data(), da.size()));
    M.act_qpos = as_off<int>(put_i(p, qa.data(), qa.size())); M.act_of = as_off<int>(put_i(p, of.data(), of.size()));
    // the same entries grouped by dof (ascending entry index inside a dof: the summation order of moment^T force is kept)
    std::vector<int> dadr(nv + 1, 0), dent;
    for (int d = 0; d < nv; d++) { dadr[d] = (int)dent.size(); for (int e = 0; e < (int)da.size(); e++) if (da[e] == d) dent.push_back(e); }
    dadr[nv] = (int)dent.size();
    M.dact_adr = as_off<int>(put_i(p, dadr.data(), dadr.size())); M.dact_e = as_off<int>(put_i(p, dent.data(), dent.size()));
    M.act_coef = as_off<double>(put_d(p, cf.data(), cf.size())); }
  // joint-level clamp of the total actuator force: [dof] and [lo, hi] of the limited scalar joints
  { std::vector<int> fd_; std::vector<double> fr_;
    for (int j = 0; j < nj && m->jnt_actfrclimited && m->jnt_actfrcrange; j++)
      if (m->jnt_actfrclimited[j] && (m->jnt_type[j] == MJPC_JNT_HINGE || m->jnt_type[j] == MJPC_JNT_SLIDE)) {
        fd_.push_back(m->jnt_dofadr[j]); fr_.push_back(m->jnt_actfrcrange[2 * j]); fr_.push_back(m->jnt_actfrcrange[2 * j + 1]);
      }
    M.nactfrc = (int)fd_.size();
    if (M.nactfrc) for (int i = 0; i < nu; i++) if (m->actuator_trntype[i] == MJPC_TRN_SITE) { p.error = "jnt_actfrclimited together with site transmissions is not implemented"; return false; }
    M.actfrc_dof = as_off<int>(put_i(p, fd_.data(), fd_.size())); M.actfrc_range = as_off<double>(put_d(p, fr_.data(), fr_.size())); }
  // gravity compensation: bodies with gravcomp != 0 and the force -gravity * mass * gravcomp (world frame, constant)
  { std::vector<int> gb; std::vector<double> gf;
    for (int b = 1; b < nb && m->body_gravcomp; b++) if (m->body_gravcomp[b] != 0) {
      gb.push_back(b);
      for (int k = 0; k < 3; k++) gf.push_back(-m->gravity[k] * m->body_mass[b] * m->body_gravcomp[b]);
    }
    M.ngravcomp = (int)gb.size();
    M.gc_body = as_off<int>(put_i(p, gb.data(), gb.size())); M.gc_force = as_off<double>(put_d(p, gf.data(), gf.size())); }
  // site transmissions with a reference site (mj_transmission): the translational gear in the reference body's frame and the
  // dofs both sites hang from (their moment entries are cleared)
  { std::vector<int> ri, rof(nu, -1); std::vector<double> rg;
    for (int i = 0; i < nu; i++) if (m->actuator_trntype[i] == MJPC_TRN_SITE && m->actuator_refsite && m->actuator_refsite[i] >= 0) {
      int s = m->actuator_trnid[i], r = m->actuator_refsite[i], bs = m->site_bodyid[s], br = m->site_bodyid[r];
      rof[i] = (int)ri.size() / 7;
      unsigned long long mask = 0;
      int b0 = m->body_weldid[bs], b1 = m->body_weldid[br];
      int d0 = m->body_dofnum[b0] ? m->body_dofadr[b0] + m->body_dofnum[b0] - 1 : -1, d1 = m->body_dofnum[b1] ? m->body_dofadr[b1] + m->body_dofnum[b1] - 1 : -1;
      if (d0 >= 0 && d1 >= 0) {
        while (d0 != d1) { if (d0 < d1) d1 = m->dof_parentid[d1]; else d0 = m->dof_parentid[d0]; if (d0 == -1 || d1 == -1) break; }
        if (d0 == d1) for (int d = d0; d >= 0; d = m->dof_parentid[d]) mask |= 1ull << d;
      }
      ri.push_back(i); ri.push_back(s); ri.push_back(r); ri.push_back(bs); ri.push_back(br); ri.push_back((int)(mask & 0xffffffffu)); ri.push_back((int)(mask >> 32));
      const double *q = m->site_quat + 4 * r, *g = m->actuator_gear6 + 6 * i;
      double R[9] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]),
                     2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]),
                     2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2])};
      for (int k = 0; k < 3; k++) rg.push_back(R[3 * k] * g[0] + R[3 * k + 1] * g[1] + R[3 * k + 2] * g[2]);
    }
    M.nrsact = (int)ri.size() / 7;
    M.rsact_i = as_off<int>(put_i(p, ri.data(), ri.size())); M.rsact_of = as_off<int>(put_i(p, rof.data(), rof.size())); M.rsact_g = as_off<double>(put_d(p, rg.data(), rg.size())); }
  M.noslip_iterations = m->noslip_iterations; M.noslip_tolerance = m->noslip_tolerance;
  // site transmissions (mjTRN_SITE, no refsite): [actuator, site, body] and the gear wrench rotated into the body frame
  { std::vector<int> si; std::vector<double> sg;
    for (int i = 0; i < nu; i++) if (m->actuator_trntype[i] == MJPC_TRN_SITE && !(m->actuator_refsite && m->actuator_refsite[i] >= 0)) {
      int s = m->actuator_trnid[i];
      si.push_back(i); si.push_back(s); si.push_back(m->site_bodyid[s]);
      const double *q = m->site_quat + 4 * s;
      double R[9] = {1 - 2 * (q[2] * q[2] + q[3] * q[3]), 2 * (q[1] * q[2] - q[0] * q[3]), 2 * (q[1] * q[3] + q[0] * q[2]),
                     2 * (q[1] * q[2] + q[0] * q[3]), 1 - 2 * (q[1] * q[1] + q[3] * q[3]), 2 * (q[2] * q[3] - q[0] * q[1]),
                     2 * (q[1] * q[3] - q[0] * q[2]), 2 * (q[2] * q[3] + q[0] * q[1]), 1 - 2 * (q[1] * q[1] + q[2] * q[2])};
      for (int part = 0; part < 2; part++) for (int r = 0; r < 3; r++) {
        const double *g = m->actuator_gear6 + 6 * i + 3 * part;
        sg.push_back(R[3 * r] * g[0] + R[3 * r + 1] * g[1] + R[3 * r + 2] * g[2]);
      }
    }
    M.nsiteact = (int)si.size() / 3;
    M.sact_i = as_off<int>(put_i(p, si.data(), si.size())); M.sact_g = as_off<double>(put_d(p, sg.data(), sg.size())); }
  { std::vector<int> wd(m->nwrap), wq(m->nwrap);
    for (int w = 0; w < m->nwrap; w++) { int j = m->wrap_objid[w]; wd[w] = m->jnt_dofadr[j]; wq[w] = m->jnt_qposadr[j]; }
    M.wrap_dofadr = as_off<int>(put_i(p, wd.data(), wd.size())); M.wrap_qposadr = as_off<int>(put_i(p, wq.data(), wq.size())); }
  // non-zeros of the joint-space inertia
  { std::vector<int> pi, pj;
    for (int i = 0; i < nv; i++) for (int j = i; j >= 0; j = m->dof_parentid[j]) { pi.push_back(i); pj.push_back(j); }
    M.nmpair = (int)pi.size();
    for (int i = 0; i < nv; i++) { pi.push_back(i); pj.push_back(nv); }      // + the gradient entries (column nv of the scaled rows)
    M.mpair_i = as_off<int>(put_i(p, pi.data(), pi.size())); M.mpair_j = as_off<int>(put_i(p, pj.data(), pj.size()));
    // the Hessian's pattern follows the elimination tree (== the dof tree unless the compile-time tree has hub links)
    std::vector<int> hi, hj;
    for (int i = 0; i < nv; i++) for (int j = i; j >= 0; j = pattern_parent(nv, M.tree_ok, m->dof_parentid, j)) { hi.push_back(i); hj.push_back(j); }
    M.nhpair = (int)hi.size();
    for (int i = 0; i < nv; i++) { hi.push_back(i); hj.push_back(nv); }
    M.hpair_i = as_off<int>(put_i(p, hi.data(), hi.size())); M.hpair_j = as_off<int>(put_i(p, hj.data(), hj.size()));
    // the structurally-zero part of the lower triangle (dof pairs on different branches of the elimination tree)
    std::vector<int> zi, zj;
    for (int i = 0; i < nv; i++) for (int j = 0; j < i; j++) {
      bool anc = false;
      for (int a = i; a >= 0; a = pattern_parent(nv, M.tree_ok, m->dof_parentid, a)) if (a == j) anc = true;
      if (!anc) { zi.push_back(i); zj.push_back(j); }
    }
    M.nzpair = (int)zi.size();
    M.zpair_i = as_off<int>(put_i(p, zi.data(), zi.size())); M.zpair_j = as_off<int>(put_i(p, zj.data(), zj.size())); }
  // dof chain bitmask per body
  { std::vector<double> masks(nb);
    for (int b = 0; b < nb; b++) {
      unsigned long long mask = 0;
      for (int a = b; a > 0; a = m->body_parentid[a])
        for (int k = 0; k < m->body_dofnum[a]; k++) mask |= 1ull << (m->body_dofadr[a] + k);
      memcpy(&masks[b], &mask, 8);
    }
    M.body_dofmask = reinterpret_cast<const unsigned long long *>(as_off<double>(put_d(p, masks.data(), nb)));
    // the same along the elimination tree: from the body's deepest dof up through pattern_parent
    std::vector<double> pmasks(nb);
    for (int b = 0; b < nb; b++) {
      int a = b;
      while (a > 0 && m->body_dofnum[a] == 0) a = m->body_parentid[a];
      unsigned long long mask = 0;
      if (a > 0) for (int d = m->body_dofadr[a] + m->body_dofnum[a] - 1; d >= 0; d = pattern_parent(nv, M.tree_ok, m->dof_parentid, d)) mask |= 1ull << d;
      memcpy(&pmasks[b], &mask, 8);
    }
    M.body_patmask = reinterpret_cast<const unsigned long long *>(as_off<double>(put_d(p, pmasks.data(), nb))); }
  // static collision filtering (same weld body, parent-child, <exclude>, contype/conaffinity)
  { std::vector<int> g1s, g2s;
    for (int a = 0; a < ng; a++) for (int b = a + 1; b < ng; b++) {
      int g1 = a, g2 = b;
      if (m->geom_type[g1] > m->geom_type[g2]) { g1 = b; g2 = a; }
      int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
      int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
      if (w1 == w2) continue;
      int pw1 = m->body_weldid[m->body_parentid[w1]], pw2 = m->body_weldid[m->body_parentid[w2]];
      if (w1 != 0 && w2 != 0 && (w1 == pw2 || w2 == pw1)) continue;
      bool excl = false;
      for (int e = 0; e < m->nexclude; e++)
        if (m->exclude_signature[e] == (b1 << 16) + b2 || m->exclude_signature[e] == (b2 << 16) + b1) excl = true;
      if (excl) continue;
      if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) continue;
      if (m->geom_type[g1] == MJPC_GEOM_PLANE && m->geom_type[g2] == MJPC_GEOM_PLANE) continue;
      for (int g : {g1, g2}) {
        int ty = m->geom_type[g];
        if (ty == MJPC_GEOM_HFIELD) {
          int k = m->geom_dataid ? m->geom_dataid[g] : -1;
          bool ok = k >= 0 && k < m->nhfield && m->hfield_data && m->hfield_nrow[k] >= 2 && m->hfield_ncol[k] >= 2 &&
                    m->hfield_adr[k] + m->hfield_nrow[k] * m->hfield_ncol[k] <= m->nhfielddata;
          int other = g == g1 ? g2 : g1;
          if (!ok) { p.error = "geom " + std::to_string(g) + " is a height field that can collide but has no usable data (geom_dataid / hfield_*)"; return false; }
          if (m->geom_type[other] < MJPC_GEOM_SPHERE) { p.error = "height field " + std::to_string(g) + " against a plane / height field (geom " + std::to_string(other) + ") has no collider"; return false; }
        }
        if (ty == MJPC_GEOM_MESH) {
          int k = m->geom_dataid ? m->geom_dataid[g] : -1;
          if (k < 0 || k >= m->nmesh || !m->mesh_vert || m->mesh_vertnum[k] < 4 || m->mesh_vertadr[k] + m->mesh_vertnum[k] > m->nmeshvert) {
            p.error = "geom " + std::to_string(g) + " is a mesh that can collide but has no usable vertex data (geom_dataid / mesh_vert*)";
            return false;
          }
        }
      }
      g1s.push_back(g1); g2s.push_back(g2);
    }
    M.npair = (int)g1s.size();
    if (ng > 65535) { p.error = "more than 65535 geoms"; return false; }
    // one record per pair for the broad phase: both geoms in one word, the pair's margin and the two bounding radii (-1 marks a
    // plane as the first geom): a single round of loads per batch, whatever memory the tables live in
    std::vector<int> gg(g1s.size()); std::vector<double> bp(3 * g1s.size());
    for (size_t k = 0; k < g1s.size(); k++) {
      int a = g1s[k], b = g2s[k];
      gg[k] = a | (b << 16);
      bp[3 * k] = std::max(m->geom_margin[a], m->geom_margin[b]);
      bp[3 * k + 1] = m->geom_type[a] == MJPC_GEOM_PLANE ? -1.0 : m->geom_rbound[a];
      bp[3 * k + 2] = m->geom_rbound[b];
    }
    M.pair_gg = as_off<int>(put_i(p, gg.data(), gg.size())); M.pair_bp = as_off<double>(put_d(p, bp.data(), bp.size())); }
  { std::vector<int> fr, lim, limb, ray;
    for (int i = 0; i < nv; i++) if (!no_fric && m->dof_frictionloss[i] > 0) fr.push_back(i);
    for (int j = 0; j < nj; j++) if (!no_limit && m->jnt_limited[j] && (m->jnt_type[j] == MJPC_JNT_SLIDE || m->jnt_type[j] == MJPC_JNT_HINGE)) lim.push_back(j);
    for (int j = 0; j < nj; j++) if (!no_limit && m->jnt_limited[j] && m->jnt_type[j] == MJPC_JNT_BALL) limb.push_back(j);
    for (int g = 0; g < ng; g++) if (m->geom_group[g] == 0) {
      // Ground() (utilities.cc:531-553) casts a ray at the group-0 geoms: the ray code knows planes, spheres and boxes
      int ty = m->geom_type[g];
      if (t->task_id == MJPC_TASK_QUADRUPED && ty != MJPC_GEOM_PLANE && ty != MJPC_GEOM_SPHERE && ty != MJPC_GEOM_BOX) {
        p.error = "geom " + std::to_string(g) + " (type " + std::to_string(ty) + ") is in group 0 but the ground ray cast only intersects planes, spheres and boxes";
        return false;
      }
      ray.push_back(g);
    }
    M.nfric = (int)fr.size(); M.nlimit = (int)lim.size(); M.nlimit_ball = (int)limb.size(); M.nray = (int)ray.size();
    M.fric_dof = as_off<int>(put_i(p, fr.data(), fr.size())); M.limit_jnt = as_off<int>(put_i(p, lim.data(), lim.size()));
    M.limit_ball = as_off<int>(put_i(p, limb.data(), limb.size()));
    M.ray_geom = as_off<int>(put_i(p, ray.data(), ray.size())); }
  // a limited tendon whose joints do not lie on one branch of the elimination tree puts entries outside the Hessian's pattern
  // (the same holds for the friction row of a tendon with friction loss)
  M.limit_cross = 0;
  for (int t = 0; t < m->ntendon; t++) if ((m->tendon_limited[t] && !no_limit) || (!no_fric && m->tendon_frictionloss && m->tendon_frictionloss[t] > 0))
    for (int w1 = m->tendon_adr[t]; w1 < m->tendon_adr[t] + m->tendon_num[t]; w1++)
      for (int w2 = m->tendon_adr[t]; w2 < w1; w2++) {
        int a = m->jnt_dofadr[m->wrap_objid[w1]], b = m->jnt_dofadr[m->wrap_objid[w2]];
        if (a < b) std::swap(a, b);
        bool anc = false;
        for (int k = a; k >= 0; k = pattern_parent(nv, M.tree_ok, m->dof_parentid, k)) if (k == b) anc = true;
        if (!anc) M.limit_cross = 1;
      }
  // equality constraints: static rows right behind the single-entry rows; table per equality [type, obj1, obj2, first row],
  // parameters [data 11, solref 2, solimp 5]; rows whose dofs do not lie on one branch force dense Hessian builds (limit_cross)
  { std::vector<int> tab; std::vector<double> prm; int rows = 0, ncon = 0;
    const bool no_eq = m->disableflags & (MJPC_DSBL_CONSTRAINT | MJPC_DSBL_EQUALITY);
    for (int e = 0; e < m->neq && !no_eq; e++) if (m->eq_active0[e]) {
      int ty = m->eq_type[e], a = m->eq_obj1id[e], b = m->eq_obj2id[e];
      tab.push_back(ty); tab.push_back(a); tab.push_back(b); tab.push_back(rows);
      for (int k = 0; k < 11; k++) prm.push_back(m->eq_data[11 * e + k]);
      for (int k = 0; k < 2; k++) prm.push_back(m->eq_solref[2 * e + k]);
      for (int k = 0; k < 5; k++) prm.push_back(m->eq_solimp[5 * e + k]);
      rows += ty == MJPC_EQ_CONNECT ? 3 : ty == MJPC_EQ_WELD ? 6 : 1; ncon += ty == MJPC_EQ_CONNECT || ty == MJPC_EQ_WELD;      // ncon: equalities whose Jacobian needs subtree_com
      // dofs of the row(s): every pair must be ancestor-related in the elimination tree, else the pattern does not hold them
      std::vector<int> dofs;
      if (ty == MJPC_EQ_CONNECT || ty == MJPC_EQ_WELD) { for (int bb : {a, b}) for (int x = bb; x > 0; x = m->body_parentid[x]) for (int k = 0; k < m->body_dofnum[x]; k++) dofs.push_back(m->body_dofadr[x] + k); }
      else if (ty == MJPC_EQ_TENDON) { for (int tt : {a, b}) if (tt >= 0) for (int w = m->tendon_adr[tt]; w < m->tendon_adr[tt] + m->tendon_num[tt]; w++) dofs.push_back(m->jnt_dofadr[m->wrap_objid[w]]); }
      else { dofs.push_back(m->jnt_dofadr[a]); if (b >= 0) dofs.push_back(m->jnt_dofadr[b]); }
      for (size_t x = 0; x < dofs.size(); x++) for (size_t y = 0; y < x; y++) {
        int i = dofs[x], j = dofs[y];
        if (i == j) continue;
        if (i < j) std::swap(i, j);
        bool anc = false;
        for (int k = i; k >= 0; k = pattern_parent(nv, M.tree_ok, m->dof_parentid, k)) if (k == j) anc = true;
        if (!anc) M.limit_cross = 1;
      }
    }
    M.neq = (int)tab.size() / 4; M.neqrow = rows; M.neq_connect = ncon;
    M.eq_tab = as_off<int>(put_i(p, tab.data(), tab.size())); M.eq_prm = as_off<double>(put_d(p, prm.data(), prm.size())); }
  // tendons with friction loss: one friction row each (tendon id; frictionloss, solref[2], solimp[5])
  { std::vector<int> ids; std::vector<double> prm;
    static const double def_ref[2] = {0.02, 1.0}, def_imp[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
    for (int t = 0; t < m->ntendon; t++) if (!no_fric && m->tendon_frictionloss && m->tendon_frictionloss[t] > 0) {
      ids.push_back(t); prm.push_back(m->tendon_frictionloss[t]);
      for (int k = 0; k < 2; k++) prm.push_back(m->tendon_solref_fri ? m->tendon_solref_fri[2 * t + k] : def_ref[k]);
      for (int k = 0; k < 5; k++) prm.push_back(m->tendon_solimp_fri ? m->tendon_solimp_fri[5 * t + k] : def_imp[k]);
    }
    M.ntfric = (int)ids.size();
    M.tfric_id = as_off<int>(put_i(p, ids.data(), ids.size())); M.tfric_prm = as_off<double>(put_d(p, prm.data(), prm.size())); }
  // tendons with passive forces
  { std::vector<int> ids; std::vector<double> prm;
    for (int t = 0; t < m->ntendon; t++) {
      double k = m->tendon_stiffness ? m->tendon_stiffness[t] : 0, b = m->tendon_damping ? m->tendon_damping[t] : 0;
      if (k == 0 && b == 0) continue;
      ids.push_back(t); prm.push_back(k); prm.push_back(b);
      prm.push_back(m->tendon_lengthspring ? m->tendon_lengthspring[2 * t] : 0); prm.push_back(m->tendon_lengthspring ? m->tendon_lengthspring[2 * t + 1] : 0);
    }
    M.ntendon_passive = (int)ids.size();
    M.tpass_id = as_off<int>(put_i(p, ids.data(), ids.size())); M.tpass_prm = as_off<double>(put_d(p, prm.data(), prm.size())); }
  M.smooth_extras = M.ntendon_passive + M.nsiteact + M.nrsact + M.ngravcomp + M.fluid + M.nactfrc;
  M.any_damping = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) M.any_damping = 1;
  // implicitfast: velocity derivatives of the smooth forces beyond joint damping, as entries (i >= j, coefficient, actuator or -1) of
  // the matrix added to M before the integration solve: M - h dF/dv  (tendon damping: b c_i c_j; affine actuator bias: -prm2 m_i m_j)
  { std::vector<int> ei; std::vector<double> ec;
    auto add_outer = [&](const std::vector<std::pair<int, double>> &row, double scale, int act) -> bool {
      for (size_t a = 0; a < row.size(); a++) for (size_t b = 0; b <= a; b++) {
        int i = row[a].first, j = row[b].first;
        if (i < j) std::swap(i, j);
        bool anc = false;
        for (int k = i; k >= 0; k = pattern_parent(nv, M.tree_ok, m->dof_parentid, k)) if (k == j) anc = true;
        if (!anc && !M.int_dense) return false;       // (the dense path of ph_integrate takes entries anywhere)
        double cf = scale * row[a].second * row[b].second * ((a != b && row[a].first == row[b].first) ? 2.0 : 1.0);
        ei.push_back(i); ei.push_back(j); ei.push_back(act); ec.push_back(cf);
      }
      return true;
    };
    if (m->integrator == MJPC_INT_IMPLICITFAST || m->integrator == MJPC_INT_IMPLICIT) {
      for (int t = 0; t < m->ntendon; t++) {
        double b = m->tendon_damping ? m->tendon_damping[t] : 0;
        if (b == 0) continue;
        std::vector<std::pair<int, double>> row;
        for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) row.push_back({m->jnt_dofadr[m->wrap_objid[w]], m->wrap_prm[w]});
        if (!add_outer(row, b, -1)) { p.error = "implicitfast: tendon " + std::to_string(t) + " damps dofs on different branches (outside the factorisation pattern)"; return false; }
      }
      for (int i = 0; i < nu; i++) {
        double kv = m->actuator_biastype[i] == MJPC_BIAS_AFFINE ? m->actuator_biasprm[3 * i + 2] : 0;
        if (kv == 0) continue;
        std::vector<std::pair<int, double>> row;
        double gear = m->actuator_gear[i];
        if (m->actuator_trntype[i] == MJPC_TRN_TENDON) {
          int t = m->actuator_trnid[i];
          for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) row.push_back({m->jnt_dofadr[m->wrap_objid[w]], gear * m->wrap_prm[w]});
        } else row.push_back({m->jnt_dofadr[m->actuator_trnid[i]], gear});
        if (!add_outer(row, -kv, i)) { p.error = "implicitfast: actuator " + std::to_string(i) + " couples dofs on different branches (outside the factorisation pattern)"; return false; }
      }
    }
    M.nidrv = (int)ec.size();
    if (M.nidrv) M.any_damping = 1;
    M.idrv_e = as_off<int>(put_i(p, ei.data(), ei.size())); M.idrv_c = as_off<double>(put_d(p, ec.data(), ec.size())); }
  if (M.nefcmax < M.nfric + M.ntfric + M.neqrow + 2) M.nefcmax = M.nfric + M.ntfric + M.neqrow + 2;
  // task region (re-packable by set_task)
  p.task_i0 = p.ib.size(); p.task_d0 = p.db.size();
  pack_task(p, t);
  p.task_i_cap = p.ib.size() - p.task_i0; p.task_d_cap = p.db.size() - p.task_d0;
  // everything so far is LDS-cached; the keyframe tables (large, touched by a few residual terms only) stay in HBM
  p.cache_i = p.ib.size(); p.cache_d = p.db.size();
  M.key_qpos = as_off<double>(put_d(p, m->key_qpos, (size_t)m->nkey * m->nq));
  M.key_mpos = as_off<double>(put_d(p, m->key_mpos, (size_t)m->nkey * 3 * m->nmocap));
  // convex meshes: vertex pools, also outside the LDS copy
  { std::vector<int> did(ng, -1);
    if (m->geom_dataid) for (int g = 0; g < ng; g++) did[g] = m->geom_dataid[g];
    M.geom_dataid = as_off<int>(put_i(p, did.data(), did.size()));
    M.mesh_vertadr = as_off<int>(put_i(p, m->mesh_vertadr, (size_t)(m->nmesh > 0 ? m->nmesh : 0)));
    M.mesh_vertnum = as_off<int>(put_i(p, m->mesh_vertnum, (size_t)(m->nmesh > 0 ? m->nmesh : 0)));
    M.mesh_vert = as_off<double>(put_d(p, m->mesh_vert, (size_t)(m->nmeshvert > 0 ? 3 * m->nmeshvert : 0))); }
  { size_t nh = (size_t)(m->nhfield > 0 ? m->nhfield : 0);
    M.hfield_nrow = as_off<int>(put_i(p, m->hfield_nrow, nh)); M.hfield_ncol = as_off<int>(put_i(p, m->hfield_ncol, nh));
    M.hfield_adr = as_off<int>(put_i(p, m->hfield_adr, nh)); M.hfield_size = as_off<double>(put_d(p, m->hfield_size, 4 * nh));
    M.hfield_data = as_off<double>(put_d(p, m->hfield_data, (size_t)(m->nhfielddata > 0 ? m->nhfielddata : 0))); }
  // ---- LDS layout
  if (!use_cache) { p.cache_i = hot_only ? p.hot_i : 0; p.cache_d = hot_only ? p.hot_d : 0; }      // the kernel reads (the rest of) the tables from HBM / L2
  make_layout(p, p.L, m, t, P_max, p.cache_d, p.cache_i, lean, reg_solver);
  M.int_scratch = p.L.efc_D - p.L.efc_J;            // efc_J and efc_JA are adjacent and dead after the solve
  if (M.int_dense == 2 && M.int_scratch < 18 * nb) { p.error = "implicit integrator: no room for the bias-derivative scratch in LDS (raise nefcmax)"; return false; }
  if (M.int_dense && lean) { p.error = "the implicit integrators' dense path has no dense-tier layout"; return false; }
  return true;
}


// turn offsets into real pointers for buffers living at (ibase, dbase)
static inline DevModel relocate(const PackedModel &p, const int *ibase, const double *dbase) {
  DevModel M = p.M;
  auto fi = [&](const int *&q) { size_t off = (reinterpret_cast<size_t>(q) - 1) / sizeof(int); q = ibase + off; };
  auto fd = [&](const double *&q) { size_t off = (reinterpret_cast<size_t>(q) - 1) / sizeof(double); q = dbase + off; };
  fi(M.body_parentid); fi(M.body_rootid); fi(M.body_mocapid); fi(M.body_jntnum); fi(M.body_jntadr); fi(M.body_dofnum); fi(M.body_dofadr);
  fd(M.body_pos); fd(M.body_quat); fd(M.body_ipos); fd(M.body_iquat); fd(M.body_mass); fd(M.body_subtreemass); fd(M.body_inertia); fd(M.body_invweight0);
  fi(M.jnt_type); fi(M.jnt_qposadr); fi(M.jnt_dofadr); fi(M.jnt_bodyid);
  fd(M.jnt_pos); fd(M.jnt_axis); fd(M.jnt_stiffness); fd(M.jnt_range); fd(M.jnt_margin); fd(M.jnt_solref); fd(M.jnt_solimp); fd(M.qpos0); fd(M.qpos_spring);
  fi(M.dof_bodyid); fi(M.dof_parentid);
  fd(M.dof_armature); fd(M.dof_damping); fd(M.dof_frictionloss); fd(M.dof_invweight0); fd(M.dof_solref); fd(M.dof_solimp);
  fi(M.geom_type); fi(M.geom_condim); fi(M.geom_bodyid); fi(M.geom_priority);
  fd(M.geom_size); fd(M.geom_pos); fd(M.geom_quat); fd(M.geom_friction); fd(M.geom_solmix); fd(M.geom_solref); fd(M.geom_solimp);
  fd(M.geom_margin); fd(M.geom_gap); fd(M.geom_rbound);
  fi(M.site_bodyid); fd(M.site_pos); fd(M.site_quat);
  fi(M.act_adr); fi(M.act_dof); fi(M.act_qpos); fi(M.act_of); fi(M.dact_adr); fi(M.dact_e); fd(M.act_coef); fi(M.actuator_ctrllimited); fi(M.actuator_forcelimited); fi(M.actuator_biastype);
  if (M.na > 0) { fi(M.actuator_dyntype); fi(M.actuator_actadr); fi(M.actuator_actlimited); fd(M.actuator_dynprm); fd(M.actuator_actrange); }
  fd(M.actuator_gainprm); fd(M.actuator_biasprm); fd(M.actuator_gear); fd(M.actuator_ctrlrange); fd(M.actuator_forcerange);
  fd(M.key_qpos); fd(M.key_mpos); fi(M.geom_dataid); fi(M.mesh_vertadr); fi(M.mesh_vertnum); fd(M.mesh_vert);
  fi(M.hfield_nrow); fi(M.hfield_ncol); fi(M.hfield_adr); fd(M.hfield_size); fd(M.hfield_data);
  fi(M.tendon_adr); fi(M.tendon_num); fi(M.tendon_limited); fi(M.wrap_dofadr); fi(M.wrap_qposadr);
  fd(M.wrap_prm); fd(M.tendon_range); fd(M.tendon_margin); fd(M.tendon_solref_lim); fd(M.tendon_solimp_lim); fd(M.tendon_invweight0);
  fi(M.level_adr); fi(M.level_body); fi(M.subtree_adr); fi(M.subtree_list); fi(M.chain_adr); fi(M.chain_list); fi(M.mpair_i); fi(M.mpair_j); fi(M.hpair_i); fi(M.hpair_j); fi(M.zpair_i); fi(M.zpair_j);
  { const double *q = reinterpret_cast<const double *>(M.body_dofmask); fd(q); M.body_dofmask = reinterpret_cast<const unsigned long long *>(q); }
  { const double *q = reinterpret_cast<const double *>(M.body_patmask); fd(q); M.body_patmask = reinterpret_cast<const unsigned long long *>(q); }
  fi(M.pair_gg); fd(M.pair_bp); fi(M.fric_dof); fi(M.limit_jnt); fi(M.limit_ball); fi(M.ray_geom); fi(M.tpass_id); fd(M.tpass_prm); fi(M.tfric_id); fd(M.tfric_prm); fi(M.idrv_e); fd(M.idrv_c); fi(M.eq_tab); fd(M.eq_prm); fi(M.sact_i); fd(M.sact_g); fi(M.rsact_i); fi(M.rsact_of); fd(M.rsact_g); fi(M.gc_body); fd(M.gc_force); fi(M.actfrc_dof); fd(M.actfrc_range);
  DevTask &T = M.task;
  fi(T.dim_norm_residual); fi(T.norm); fi(T.num_norm_parameter); fi(T.trace_objtype); fi(T.trace_objid); fi(T.int_data);
  fd(T.weight); fd(T.norm_parameter); fd(T.parameters); fd(T.dbl_data);
  return M;
}

// re-pack the task into the reserved region (sizes must not exceed the initial ones)
static inline bool repack_task(PackedModel &p, const MjpcHipTask *t) {
  std::vector<int> ib_tail(p.ib.begin() + p.task_i0 + p.task_i_cap, p.ib.end());       // keyframe tables behind the task region
  std::vector<double> db_tail(p.db.begin() + p.task_d0 + p.task_d_cap, p.db.end());
  p.ib.resize(p.task_i0); p.db.resize(p.task_d0);
  pack_task(p, t);
  bool ok = !(p.ib.size() - p.task_i0 > p.task_i_cap || p.db.size() - p.task_d0 > p.task_d_cap);
  if (!ok) p.error = "task grew beyond the size given at create()";
  p.ib.resize(p.task_i0 + p.task_i_cap, 0); p.db.resize(p.task_d0 + p.task_d_cap, 0.0);
  p.ib.insert(p.ib.end(), ib_tail.begin(), ib_tail.end()); p.db.insert(p.db.end(), db_tail.begin(), db_tail.end());
  return ok;
}

}  // namespace mjpc_host
