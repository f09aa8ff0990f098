// dynamics.h — velocity stage (com velocities, RNE bias, passive forces, actuation) and the constraint solver (part of core.h)
// Included by core.h only, in this order: the files share one translation unit and its macros.
#pragma once
// ======================================================================================
// velocity stage: com velocities, subtree momentum, RNE bias, passive, actuation
// ======================================================================================
DEV void vel_body(Ctx &c, int i) {
  const DevModel &M = *c.M;
  double cvel[6];
  for (int k = 0; k < 6; k++) cvel[k] = c.cvel[6 * MIH(body_parentid)[i] + k];
  int bda = MIH(body_dofadr)[i];
  for (int j = MIH(body_jntadr)[i]; j < MIH(body_jntadr)[i] + MIH(body_jntnum)[i]; j++) {
    int type = MIH(jnt_type)[j];
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
  for (int k = 0; k < 6; k++) a[k] = c.cacc[6 * MIH(body_parentid)[i] + k];
  bda = MIH(body_dofadr)[i];
  for (int k = 0; k < MIH(body_dofnum)[i]; k++)
    for (int q = 0; q < 6; q++) a[q] += c.cdof_dot[6 * (bda + k) + q] * c.qvel[bda + k];
  for (int k = 0; k < 6; k++) c.cacc[6 * i + k] = a[k];
  double t1[6], t2[6], t3[6];
  d_mulinertvec(t1, c.cinert + 10 * i, a);
  d_mulinertvec(t2, c.cinert + 10 * i, cvel);
  d_crossforce(t3, cvel, t2);
  for (int k = 0; k < 6; k++) c.cfrc[6 * i + k] = t1[k] + t3[k];
  // body momentum for subtree_linvel
  double off[3], v[3];
  d_sub3(off, c.xipos + 3 * i, c.subtree_com + 3 * MIH(body_rootid)[i]);
  d_cross(v, cvel, off);
  d_add3(v, v, cvel + 3);
  d_scl3(c.bodytmp + 3 * i, v, MDH(body_mass)[i]);
}

// com velocity and RNE acceleration of body i from its parent's (in registers), same operation order as vel_body; `store`: i is
// the lane's own body: cvel, cdof_dot of its dofs, cacc, cfrc_body and its momentum go to LDS (deep trees, see velocity_stage)
DEV void vel_compose(Ctx &c, int i, double *cvel, double *a, int store) {
  const DevModel &M = *c.M;
  int bda = MIH(body_dofadr)[i];
  for (int j = MIH(body_jntadr)[i]; j < MIH(body_jntadr)[i] + MIH(body_jntnum)[i]; j++) {
    int type = MIH(jnt_type)[j];
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
  d_sub3(off, c.xipos + 3 * i, c.subtree_com + 3 * MIH(body_rootid)[i]);
  d_cross(v, cvel, off);
  d_add3(v, v, cvel + 3);
  d_scl3(c.bodytmp + 3 * i, v, MDH(body_mass)[i]);
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
      if (b > 0) for (int q = MIH(subtree_adr)[b]; q < MIH(subtree_adr)[b + 1]; q++) s += c.cfrc[6 * MIH(subtree_list)[q] + kc];
      c.cfrc_sub[6 * b + kc] = s;
    } else {
      int kk = k - 3;
      double s = 0;
      for (int q = MIH(subtree_adr)[b]; q < MIH(subtree_adr)[b + 1]; q++) s += c.bodytmp[3 * MIH(subtree_list)[q] + kk];
      c.subtree_linvel[3 * b + kk] = s / fmax(D_MINVAL, MDH(body_subtreemass)[b]);
    }
  }
}

#ifndef MJPC_CHAIN33
#define MJPC_CHAIN33 1      // the per-lane chain walk of the velocity sweep also for the 33-dof hand (-0.5 %; 0 = level sweep)
#endif
// site transmission with a reference site (mj_transmission): entry d of the moment (Jp_site - Jp_ref)^T w, w = the gear in the world
// frame; zero on the dofs both sites hang from
DEV double rs_moment(Ctx &c, const int *ri, const double *w, int d) {
  const DevModel &M = *c.M;
  const int s = ri[1], r = ri[2], bs = ri[3], br = ri[4];
  const unsigned long long common = (unsigned long long)(unsigned)ri[5] | ((unsigned long long)(unsigned)ri[6] << 32);
  const int in_s = (int)((MDM()[bs] >> d) & 1ull), in_r = (int)((MDM()[br] >> d) & 1ull);
  if ((!in_s && !in_r) || ((common >> d) & 1ull)) return 0.0;
  const double *cd = c.cdof + 6 * d;
  double jp[3] = {0, 0, 0}, off[3], t[3];
  if (in_s) { d_sub3(off, c.site_xpos + 3 * s, c.subtree_com + 3 * MIH(body_rootid)[bs]); d_cross(t, cd, off); for (int q = 0; q < 3; q++) jp[q] += cd[3 + q] + t[q]; }
  if (in_r) { d_sub3(off, c.site_xpos + 3 * r, c.subtree_com + 3 * MIH(body_rootid)[br]); d_cross(t, cd, off); for (int q = 0; q < 3; q++) jp[q] -= cd[3 + q] + t[q]; }
  return jp[0] * w[0] + jp[1] * w[1] + jp[2] * w[2];
}
// force of actuator i for the input u (control, or activation of a stateful actuator): gain * u + affine bias, force range
DEV double actuator_force_of(Ctx &c, int i, double u) {
  double force = MD(actuator_gainprm)[3 * i] * u;
  if (MI(actuator_biastype)[i] == 1) {
    // transmission length / velocity: gear * qpos (joint) or sum of gear * coef * qpos over the tendon's joints
    double length = 0, velocity = 0;
    for (int e = MI(act_adr)[i]; e < MI(act_adr)[i + 1]; e++) {
      double cf = MD(act_coef)[e];
      length += cf * c.qpos[MI(act_qpos)[e]]; velocity += cf * c.qvel[MI(act_dof)[e]];
    }
    if (c.M->nrsact > 0 && MI(rsact_of)[i] >= 0) {      // site relative to a reference site: length = w . (x_site - x_ref), velocity = moment . qvel
      const int k = MI(rsact_of)[i];
      const int *ri = MI(rsact_i) + 7 * k;
      double w[3], vec[3];
      d_mulmatvec3(w, c.xmat + 9 * ri[4], MD(rsact_g) + 3 * k);
      d_sub3(vec, c.site_xpos + 3 * ri[1], c.site_xpos + 3 * ri[2]);
      length = w[0] * vec[0] + w[1] * vec[1] + w[2] * vec[2];
      for (int d = 0; d < c.M->nv; d++) velocity += rs_moment(c, ri, w, d) * c.qvel[d];
    }
    force += MD(actuator_biasprm)[3 * i] + MD(actuator_biasprm)[3 * i + 1] * length + MD(actuator_biasprm)[3 * i + 2] * velocity;
  }
  if (MI(actuator_forcelimited)[i]) force = d_clip(force, MD(actuator_forcerange)[2 * i], MD(actuator_forcerange)[2 * i + 1]);
  return force;
}

// The rare extras of the smooth phase, out of line so that the common path stays compact: joint-level actuator force clamp, fluid
// forces, gravity compensation, site transmissions, tendon springs / dampers.  All of them add to qfrc_smooth in LDS.
DEV_NOINLINE void ph_smooth_extras(KP Kc) {
  Ctx c; ctx_open(c, Kc, 1);
  const DevModel &M = *c.M;
  const int nv = M.nv;
  if (M.nactfrc > 0) {
    // jnt_actfrclimited: the total actuator force on the joint's dof is clamped (end of mj_fwdActuation); applied as a correction
    // to what the gather above added
    PFOR(k, M.nactfrc) {
      int dd = MI(actfrc_dof)[k];
      double act = 0;
      for (int q = MI(dact_adr)[dd]; q < MI(dact_adr)[dd + 1]; q++) { int e = MI(dact_e)[q]; act += MD(act_coef)[e] * c.actuator_force[MI(act_of)[e]]; }
      c.qfrc_smooth[dd] += d_clip(act, MD(actfrc_range)[2 * k], MD(actfrc_range)[2 * k + 1]) - act;
    }
    SYNC();
  }
  if (M.fluid) {
    // fluid forces, inertia-box model (mj_inertiaBoxFluidModel): per body the world force / torque at its com into cfrc (dead by
    // now: the bias forces have been read off cfrc_sub), then J^T of them per dof
    PFOR(b, M.nbody) {
      double *w = c.cfrc + 6 * b;
      for (int k = 0; k < 6; k++) w[k] = 0;
      double mass = MDH(body_mass)[b];
      if (b == 0 || mass < D_MINVAL) continue;
      const double *I = MDH(body_inertia) + 3 * b, *R = c.ximat + 9 * b;
      double box[3], off[3], vw[3], lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0};
      box[0] = sqrt(d_div(fmax(D_MINVAL, I[1] + I[2] - I[0]), mass) * 6.0);
      box[1] = sqrt(d_div(fmax(D_MINVAL, I[0] + I[2] - I[1]), mass) * 6.0);
      box[2] = sqrt(d_div(fmax(D_MINVAL, I[0] + I[1] - I[2]), mass) * 6.0);
      d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MIH(body_rootid)[b]);
      d_cross(vw, c.cvel + 6 * b, off);
      for (int k = 0; k < 3; k++) vw[k] += c.cvel[6 * b + 3 + k] - M.wind[k];
      d_mulmattvec3(lvel, R, c.cvel + 6 * b); d_mulmattvec3(lvel + 3, R, vw);
      if (M.viscosity > 0) {
        double diam = (box[0] + box[1] + box[2]) / 3.0;
        for (int k = 0; k < 3; k++) { lfrc[k] = -D_PI * diam * diam * diam * M.viscosity * lvel[k]; lfrc[3 + k] = -3.0 * D_PI * diam * M.viscosity * lvel[3 + k]; }
      }
      if (M.density > 0) {
        double b0 = box[0], b1 = box[1], b2 = box[2];
        lfrc[3] -= 0.5 * M.density * b1 * b2 * fabs(lvel[3]) * lvel[3];
        lfrc[4] -= 0.5 * M.density * b0 * b2 * fabs(lvel[4]) * lvel[4];
        lfrc[5] -= 0.5 * M.density * b0 * b1 * fabs(lvel[5]) * lvel[5];
        lfrc[0] -= M.density * b0 * (b1 * b1 * b1 * b1 + b2 * b2 * b2 * b2) * fabs(lvel[0]) * lvel[0] / 64.0;
        lfrc[1] -= M.density * b1 * (b0 * b0 * b0 * b0 + b2 * b2 * b2 * b2) * fabs(lvel[1]) * lvel[1] / 64.0;
        lfrc[2] -= M.density * b2 * (b0 * b0 * b0 * b0 + b1 * b1 * b1 * b1) * fabs(lvel[2]) * lvel[2] / 64.0;
      }
      d_mulmatvec3(w, R, lfrc); d_mulmatvec3(w + 3, R, lfrc + 3);        // torque, force in the world frame
    }
    SYNC();
    PFOR(d, nv) {
      double acc = c.qfrc_smooth[d];
      const double *cd = c.cdof + 6 * d;
      for (int b = 1; b < M.nbody; b++) {
        if (!((MDM()[b] >> d) & 1ull)) continue;
        const double *w = c.cfrc + 6 * b;
        double off[3], t[3];
        d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MIH(body_rootid)[b]);
        d_cross(t, cd, off);
        acc += (cd[3] + t[0]) * w[3] + (cd[4] + t[1]) * w[4] + (cd[5] + t[2]) * w[5] + cd[0] * w[0] + cd[1] * w[1] + cd[2] * w[2];
      }
      c.qfrc_smooth[d] = acc;
    }
    SYNC();
  }
  if (M.ngravcomp > 0) {
    // gravity compensation (mj_passive): a constant world force at the body's com, through the point Jacobian
    PFOR(d, nv) {
      double acc = c.qfrc_smooth[d];
      const double *cd = c.cdof + 6 * d;
      for (int k = 0; k < M.ngravcomp; k++) {
        int b = MI(gc_body)[k];
        if (!((MDM()[b] >> d) & 1ull)) continue;
        double off[3], t[3];
        d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MIH(body_rootid)[b]);
        d_cross(t, cd, off);
        const double *f = MD(gc_force) + 3 * k;
        acc += (cd[3] + t[0]) * f[0] + (cd[4] + t[1]) * f[1] + (cd[5] + t[2]) * f[2];
      }
      c.qfrc_smooth[d] = acc;
    }
    SYNC();
  }
  if (M.nsiteact > 0) {
    // site transmissions: qfrc += J_site^T (R gear_force; R gear_torque) force, the site Jacobian from cdof about the root's com
    PFOR(d, nv) {
      double acc = c.qfrc_smooth[d];
      const double *cd = c.cdof + 6 * d;
      for (int k = 0; k < M.nsiteact; k++) {
        int a = MI(sact_i)[3 * k], s = MI(sact_i)[3 * k + 1], b = MI(sact_i)[3 * k + 2];
        if (!((MDM()[b] >> d) & 1ull)) continue;
        double f[3], tq[3], off[3], t[3];
        d_mulmatvec3(f, c.xmat + 9 * b, MD(sact_g) + 6 * k); d_mulmatvec3(tq, c.xmat + 9 * b, MD(sact_g) + 6 * k + 3);
        d_sub3(off, c.site_xpos + 3 * s, c.subtree_com + 3 * MIH(body_rootid)[b]);
        d_cross(t, cd, off);
        acc += c.actuator_force[a] * ((cd[3] + t[0]) * f[0] + (cd[4] + t[1]) * f[1] + (cd[5] + t[2]) * f[2] + cd[0] * tq[0] + cd[1] * tq[1] + cd[2] * tq[2]);
      }
      c.qfrc_smooth[d] = acc;
    }
    SYNC();
  }
  if (M.nrsact > 0) {
    // site transmissions with a reference site: qfrc += moment^T force
    PFOR(d, nv) {
      double acc = c.qfrc_smooth[d];
      for (int k = 0; k < M.nrsact; k++) {
        const int *ri = MI(rsact_i) + 7 * k;
        double w[3];
        d_mulmatvec3(w, c.xmat + 9 * ri[4], MD(rsact_g) + 3 * k);
        acc += rs_moment(c, ri, w, d) * c.actuator_force[ri[0]];
      }
      c.qfrc_smooth[d] = acc;
    }
    SYNC();
  }
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
}

// mfact_seq != 0: M's factor is produced by a helper wave; wait for its sequence number (misc[22]) before the solve
template <int NVT>
DEV void velocity_stage(Ctx &c, KP Kc, int mfact_seq) {
  const DevModel &M = *c.M;
  int nv = M.nv;
#ifdef MJPC_LEAN_LDS
  // the RNE intermediates share their LDS with the solver's scaled rows here: the world body's entries are rewritten every step
  if (LANE < 6) { c.cfrc[LANE] = 0; c.cacc[LANE] = (LANE >= 3) ? -M.gravity[LANE - 3] : 0.0; }
  SYNC();
#endif
  if constexpr (NVT == 27 || (NVT == 33 && MJPC_CHAIN33)) {
    // the humanoid's 8 tree levels: one lane per body walks its ancestor chain with the running velocity / acceleration in
    // registers, like kinematics (-1.3 % of its step; the A1's 4 levels are cheaper as a level sweep, and keeping both forms in
    // one instantiation costs it +0.7 %, hence the compile-time choice)
    PFOR(b, M.nbody) {
      if (b == 0) continue;
      double cvel[6], acc[6];
      for (int k = 0; k < 6; k++) { cvel[k] = c.cvel[k]; acc[k] = c.cacc[k]; }      // the world body: 0 and -gravity
      for (int q = MIH(chain_adr)[b]; q < MIH(chain_adr)[b + 1]; q++) {
        int a = MIH(chain_list)[q];
        vel_compose(c, a, cvel, acc, a == b);
      }
    }
    SYNC();
  } else {
    for (int l = 0; l < M.nlevel; l++) {
      int a = MIH(level_adr)[l], n = MIH(level_adr)[l + 1] - a;
      PFOR(k, n) vel_body(c, MIH(level_body)[a + k]);
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
    c.actuator_force[i] = actuator_force_of(c, i, ctrl);
  }
  if (M.na) {        // stateful actuators: act_dot from the clamped control, the force from the current activation instead
    PFOR(i, M.nu) {
      int dt = MI(actuator_dyntype)[i];
      if (dt) {
        double ctrl = c.ctrl[i];
        if (MI(actuator_ctrllimited)[i]) ctrl = d_clip(ctrl, MD(actuator_ctrlrange)[2 * i], MD(actuator_ctrlrange)[2 * i + 1]);
        int a = MI(actuator_actadr)[i];
        double act = C_ACT(c)[a];
        C_ACTDOT(c)[a] = dt == DYN_INTEGRATOR ? ctrl : d_div(ctrl - act, fmax(D_MINVAL, MD(actuator_dynprm)[i]));
        c.actuator_force[i] = actuator_force_of(c, i, act);
      }
    }
  }
#if MJPC_HELPER
  if (!flag_wait(c.misc + HX_SUBSUM, mfact_seq)) c.warning |= WARN_SYNC;
#endif
  SYNC();
  PFOR(d, nv) {
    const double *cd = c.cdof + 6 * d, *cf = c.cfrc_sub + 6 * MIH(dof_bodyid)[d];
    double bias = cd[0]*cf[0] + cd[1]*cf[1] + cd[2]*cf[2] + cd[3]*cf[3] + cd[4]*cf[4] + cd[5]*cf[5];
    c.qfrc_bias[d] = bias;
    double act = 0;
    for (int q = MI(dact_adr)[d]; q < MI(dact_adr)[d + 1]; q++) { int e = MI(dact_e)[q]; act += MD(act_coef)[e] * c.actuator_force[MI(act_of)[e]]; }     // moment^T force
    c.qfrc_smooth[d] = act - bias - MD(dof_damping)[d] * c.qvel[d];   // joint springs are added below
  }
  SYNC();
  PFOR(j, M.njnt) {
    double k = MDH(jnt_stiffness)[j];
    int type = MIH(jnt_type)[j];
    if (k != 0 && (type == 2 || type == 3)) {
      int qa = MIH(jnt_qposadr)[j];
      c.qfrc_smooth[MIH(jnt_dofadr)[j]] -= k * (c.qpos[qa] - MDH(qpos_spring)[qa]);
    }
  }
  SYNC();
  if (M.smooth_extras) ph_smooth_extras(Kc);      // one host-made flag for the rare extras (out of line)
  if (c.K->xfrc_std > 0) {
    // mj_xfrcAccumulate: J^T [force; torque], force applied at the body's inertial frame origin; bodies in ascending order
    PFOR(d, nv) {
      const double *cd = c.cdof + 6 * d;
      int bd = MIH(dof_bodyid)[d];
      double acc = c.qfrc_smooth[d];
      for (int q = MIH(subtree_adr)[bd]; q < MIH(subtree_adr)[bd + 1]; q++) {
        int b = MIH(subtree_list)[q];
        const double *f = c.xfrc + 6 * b;
        double off[3], tt[3];
        d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MIH(body_rootid)[b]);
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

