/*
 * oracle/physics.c — TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Sequential fp64 restatement of what the reference's rollout executes inside
 * `mj_step(model, data)` (mjpc/trajectory.cc:158) and `mj_forward` (mjpc/trajectory.cc:198).
 * The arithmetic lives in MuJoCo 3.1.4 (third-party, pinned by /root/reference/CMakeLists.txt:58-61,
 * absent from /root/reference and from this image): this file restates MuJoCo's published
 * computation pipeline stage by stage:
 *   position: kinematics, com-based inertias, CRBA, factorisation, collision, constraint rows
 *   velocity: com velocities, passive forces, reference acceleration, RNE bias
 *   actuation, smooth acceleration, primal Newton constraint solve (soft constraints,
 *   friction-loss, limits, frictionless / elliptic contacts), Euler with implicit damping.
 * PARITY UNPINNED at this boundary; analytic checks live in tests/.
 */
#include <stdlib.h>
#include <stdio.h>
#include "oracle.h"
#include "omath.h"

int oracle_collide_pair(const OModel *om, OData *d, int g1, int g2, double margin, OContact *out, int *unsupported);

/* ------------------------------------------------------------------------------------ */
static void *dalloc(OData *d, size_t bytes) {
  void *p = calloc(1, bytes ? bytes : 8);
  d->blocks[d->nblocks++] = p;
  return p;
}
#define DD(field, n) d->field = (double *)dalloc(d, sizeof(double) * (size_t)(n))
#define DI(field, n) d->field = (int *)dalloc(d, sizeof(int) * (size_t)(n))

OData *oracle_make_data(const OModel *om) {
  const MjpcHipModel *m = &om->m;
  OData *d = (OData *)calloc(1, sizeof(OData));
  int nv = m->nv, nb = m->nbody, ne = om->nefcmax;
  DD(qpos, m->nq); DD(qvel, nv); DD(ctrl, m->nu); DD(mocap_pos, 3 * m->nmocap + 3);
  DD(mocap_quat, 4 * m->nmocap + 4); DD(userdata, m->nuserdata + 1);
  DD(qacc, nv); DD(qacc_newton, nv); DD(qacc_warmstart, nv); DD(qacc_smooth, nv); DD(qfrc_smooth, nv);
  DD(qfrc_bias, nv); DD(qfrc_passive, nv); DD(qfrc_actuator, nv); DD(qfrc_constraint, nv);
  DD(actuator_force, m->nu + 1); DD(act, m->na + 1); DD(act_dot, m->na + 1);
  DD(xpos, 3 * nb); DD(xquat, 4 * nb); DD(xmat, 9 * nb); DD(xipos, 3 * nb); DD(ximat, 9 * nb);
  DD(xanchor, 3 * m->njnt + 3); DD(xaxis, 3 * m->njnt + 3);
  DD(geom_xpos, 3 * m->ngeom + 3); DD(geom_xmat, 9 * m->ngeom + 9);
  DD(site_xpos, 3 * m->nsite + 3); DD(site_xmat, 9 * m->nsite + 9);
  DD(subtree_com, 3 * nb); DD(cinert, 10 * nb); DD(cdof, 6 * nv + 6); DD(cvel, 6 * nb);
  DD(cdof_dot, 6 * nv + 6); DD(crb, 10 * nb); DD(subtree_linvel, 3 * nb); DD(cacc, 6 * nb); DD(cfrc, 6 * nb); DD(xfrc_applied, 6 * nb);
  DD(qM, nv * nv + 1); DD(qL, nv * nv + 1); DD(qH, nv * nv + 1); DD(qLD2, nv * nv + 1);
  d->contact = (OContact *)dalloc(d, sizeof(OContact) * (size_t)(om->nconmax + 8));
  DI(efc_type, ne); DI(efc_id, ne); DI(efc_state, ne);
  DD(efc_J, ne * nv + 1); DD(efc_pos, ne); DD(efc_margin, ne); DD(efc_frictionloss, ne);
  DD(efc_diagApprox, ne); DD(efc_D, ne); DD(efc_R, ne); DD(efc_vel, ne); DD(efc_aref, ne);
  DD(efc_force, ne); DD(efc_jar, ne); DD(efc_jv, ne); DD(efc_KBIP, 4 * ne);
  DD(efc_solref, 2 * ne); DD(efc_solimp, 5 * ne);
  if (m->noslip_iterations > 0) { DD(efc_AR, ne * ne + 1); DD(efc_b, ne); DD(efc_MiJT, ne * nv + 1); }
  DD(Ma, nv); DD(grad, nv); DD(Mgrad, nv); DD(search, nv); DD(Mv, nv); DD(work, 16 * nv + 64);
  DD(sensordata, om->t.num_residual + 1);
  /* defaults: qpos0, mocap at body pose */
  o_copy(d->qpos, m->qpos0, m->nq);
  for (int i = 0; i < nb; i++) if (m->body_mocapid[i] >= 0) {
    o_copy3(d->mocap_pos + 3 * m->body_mocapid[i], m->body_pos + 3 * i);
    o_copy(d->mocap_quat + 4 * m->body_mocapid[i], m->body_quat + 4 * i, 4);
  }
  return d;
}
void oracle_free_data(OData *d) {
  if (!d) return;
  for (int i = 0; i < d->nblocks; i++) free(d->blocks[i]);
  free(d);
}

/* ---- dense Cholesky helpers (A = L L^T, lower triangle of L stored row-major) -------- */
static int chol_factor(double *L, const double *A, int n) {
  int rank = n;
  for (int i = 0; i < n * n; i++) L[i] = A[i];
  for (int j = 0; j < n; j++) {
    double t = L[j * n + j];
    for (int k = 0; k < j; k++) t -= L[j * n + k] * L[j * n + k];
    if (t < O_MINVAL) { t = O_MINVAL; rank--; }
    double dj = sqrt(t);
    L[j * n + j] = dj;
    double inv = 1.0 / dj;
    for (int i = j + 1; i < n; i++) {
      double s = L[i * n + j];
      for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = s * inv;
    }
  }
  return rank;
}
static void chol_solve(double *x, const double *L, const double *b, int n) {
  for (int i = 0; i < n; i++) {
    double s = b[i];
    for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
    x[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = x[i];
    for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
    x[i] = s / L[i * n + i];
  }
}

/* ---- position stage ------------------------------------------------------------------ */
static void kinematics(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  o_zero(d->xpos, 3); o_zero(d->xipos, 3);
  d->xquat[0] = 1; d->xquat[1] = d->xquat[2] = d->xquat[3] = 0;
  o_zero(d->xmat, 9); d->xmat[0] = d->xmat[4] = d->xmat[8] = 1;
  o_copy(d->ximat, d->xmat, 9);
  for (int i = 1; i < m->nbody; i++) {
    double xpos[3], xquat[4];
    int pid = m->body_parentid[i];
    int jntnum = m->body_jntnum[i], jntadr = m->body_jntadr[i];
    if (m->body_mocapid[i] >= 0) {
      int id = m->body_mocapid[i];
      o_copy3(xpos, d->mocap_pos + 3 * id);
      o_copy(xquat, d->mocap_quat + 4 * id, 4);
      o_normalize4(xquat);
    } else if (jntnum == 1 && m->jnt_type[jntadr] == MJPC_JNT_FREE) {
      int qa = m->jnt_qposadr[jntadr];
      o_normalize4(d->qpos + qa + 3);
      o_copy3(xpos, d->qpos + qa);
      o_copy(xquat, d->qpos + qa + 3, 4);
      o_copy3(d->xanchor + 3 * jntadr, xpos);
      o_copy3(d->xaxis + 3 * jntadr, m->jnt_axis + 3 * jntadr);
    } else {
      if (pid) {
        o_mulmatvec3(xpos, d->xmat + 9 * pid, m->body_pos + 3 * i);
        o_add3(xpos, xpos, d->xpos + 3 * pid);
        o_mulquat(xquat, d->xquat + 4 * pid, m->body_quat + 4 * i);
      } else {
        o_copy3(xpos, m->body_pos + 3 * i);
        o_copy(xquat, m->body_quat + 4 * i, 4);
      }
      for (int j = jntadr; j < jntadr + jntnum; j++) {
        int qa = m->jnt_qposadr[j];
        double vec[3];
        o_rotvecquat(d->xaxis + 3 * j, m->jnt_axis + 3 * j, xquat);
        o_rotvecquat(vec, m->jnt_pos + 3 * j, xquat);
        o_add3(d->xanchor + 3 * j, vec, xpos);
        if (m->jnt_type[j] == MJPC_JNT_SLIDE) {
          o_addtoscl3(xpos, d->xaxis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
        } else if (m->jnt_type[j] == MJPC_JNT_BALL || m->jnt_type[j] == MJPC_JNT_HINGE) {
          double qloc[4], t[4];
          if (m->jnt_type[j] == MJPC_JNT_BALL) {
            o_normalize4(d->qpos + qa);
            o_copy(qloc, d->qpos + qa, 4);
          } else {
            o_axisangle2quat(qloc, m->jnt_axis + 3 * j, d->qpos[qa] - m->qpos0[qa]);
          }
          o_mulquat(t, xquat, qloc);
          o_copy(xquat, t, 4);
          o_rotvecquat(vec, m->jnt_pos + 3 * j, xquat);
          o_sub3(xpos, d->xanchor + 3 * j, vec);
        }
      }
    }
    o_normalize4(xquat);
    o_copy3(d->xpos + 3 * i, xpos);
    o_copy(d->xquat + 4 * i, xquat, 4);
    o_quat2mat(d->xmat + 9 * i, xquat);
    /* inertial frame */
    double v[3], q[4];
    o_mulmatvec3(v, d->xmat + 9 * i, m->body_ipos + 3 * i);
    o_add3(d->xipos + 3 * i, v, xpos);
    o_mulquat(q, xquat, m->body_iquat + 4 * i);
    o_quat2mat(d->ximat + 9 * i, q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    double v[3], q[4];
    o_mulmatvec3(v, d->xmat + 9 * b, m->geom_pos + 3 * g);
    o_add3(d->geom_xpos + 3 * g, v, d->xpos + 3 * b);
    o_mulquat(q, d->xquat + 4 * b, m->geom_quat + 4 * g);
    o_quat2mat(d->geom_xmat + 9 * g, q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    double v[3], q[4];
    o_mulmatvec3(v, d->xmat + 9 * b, m->site_pos + 3 * s);
    o_add3(d->site_xpos + 3 * s, v, d->xpos + 3 * b);
    o_mulquat(q, d->xquat + 4 * b, m->site_quat + 4 * s);
    o_quat2mat(d->site_xmat + 9 * s, q);
  }
}

static void com_pos(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  for (int i = 0; i < m->nbody; i++) o_scl3(d->subtree_com + 3 * i, d->xipos + 3 * i, m->body_mass[i]);
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    o_add3(d->subtree_com + 3 * p, d->subtree_com + 3 * p, d->subtree_com + 3 * i);
  }
  for (int i = 0; i < m->nbody; i++) {
    if (m->body_subtreemass[i] < O_MINVAL) o_copy3(d->subtree_com + 3 * i, d->xipos + 3 * i);
    else o_scl3(d->subtree_com + 3 * i, d->subtree_com + 3 * i, 1.0 / m->body_subtreemass[i]);
  }
  o_zero(d->cinert, 10);
  for (int i = 1; i < m->nbody; i++) {
    double off[3];
    o_sub3(off, d->xipos + 3 * i, d->subtree_com + 3 * m->body_rootid[i]);
    o_inertcom(d->cinert + 10 * i, m->body_inertia + 3 * i, d->ximat + 9 * i, off, m->body_mass[i]);
  }
  for (int j = 0; j < m->njnt; j++) {
    int b = m->jnt_bodyid[j], da = m->jnt_dofadr[j];
    double off[3];
    o_sub3(off, d->subtree_com + 3 * m->body_rootid[b], d->xanchor + 3 * j);
    int skip = 0;
    switch (m->jnt_type[j]) {
      case MJPC_JNT_FREE:
        o_zero(d->cdof + 6 * da, 18);
        for (int k = 0; k < 3; k++) d->cdof[6 * (da + k) + 3 + k] = 1;
        skip = 3;
        /* fallthrough */
      case MJPC_JNT_BALL:
        for (int k = 0; k < 3; k++) {
          const double *xm = d->xmat + 9 * b;
          double ax[3] = {xm[k], xm[k + 3], xm[k + 6]};
          double *c = d->cdof + 6 * (da + k + skip);
          o_copy3(c, ax); o_cross(c + 3, ax, off);
        }
        break;
      case MJPC_JNT_SLIDE:
        o_zero(d->cdof + 6 * da, 3);
        o_copy3(d->cdof + 6 * da + 3, d->xaxis + 3 * j);
        break;
      default: /* hinge */
        o_copy3(d->cdof + 6 * da, d->xaxis + 3 * j);
        o_cross(d->cdof + 6 * da + 3, d->xaxis + 3 * j, off);
    }
  }
}

static void crb_and_factor(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  o_copy(d->crb, d->cinert, 10 * m->nbody);
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int k = 0; k < 10; k++) d->crb[10 * p + k] += d->crb[10 * i + k];
  }
  o_zero(d->qM, nv * nv);
  for (int i = 0; i < nv; i++) {
    double buf[6];
    o_mulinertvec(buf, d->crb + 10 * m->dof_bodyid[i], d->cdof + 6 * i);
    d->qM[i * nv + i] = m->dof_armature[i];
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      double v = o_dot(d->cdof + 6 * j, buf, 6);
      d->qM[i * nv + j] += v;
      if (j != i) d->qM[j * nv + i] = d->qM[i * nv + j];
    }
  }
  chol_factor(d->qL, d->qM, nv);
}

/* jacobian of point `p` attached to body b: jacp[3*nv], jacr[3*nv] (either may be NULL) */
static void jac_point(const OModel *om, const OData *d, double *jacp, double *jacr, const double *p, int b) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  if (jacp) o_zero(jacp, 3 * nv);
  if (jacr) o_zero(jacr, 3 * nv);
  if (b <= 0) return;
  double off[3];
  o_sub3(off, p, d->subtree_com + 3 * m->body_rootid[b]);
  /* find last dof of the chain: walk up until a body with dofs */
  while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parentid[b];
  if (b <= 0) return;
  int i = m->body_dofadr[b] + m->body_dofnum[b] - 1;
  for (; i >= 0; i = m->dof_parentid[i]) {
    const double *c = d->cdof + 6 * i;
    if (jacr) { jacr[i] = c[0]; jacr[nv + i] = c[1]; jacr[2 * nv + i] = c[2]; }
    if (jacp) {
      double t[3]; o_cross(t, c, off);
      jacp[i] = c[3] + t[0]; jacp[nv + i] = c[4] + t[1]; jacp[2 * nv + i] = c[5] + t[2];
    }
  }
}

/* mj_contactParam-style mixing */
static void contact_param(const MjpcHipModel *m, int g1, int g2, OContact *c, double *margin, double *gap) {
  int p1 = m->geom_priority[g1], p2 = m->geom_priority[g2];
  *margin = fmax(m->geom_margin[g1], m->geom_margin[g2]);
  *gap = fmax(m->geom_gap[g1], m->geom_gap[g2]);
  double fri[3];
  if (p1 != p2) {
    int g = p1 > p2 ? g1 : g2;
    c->dim = m->geom_condim[g];
    o_copy(c->solref, m->geom_solref + 2 * g, 2);
    o_copy(c->solimp, m->geom_solimp + 5 * g, 5);
    o_copy3(fri, m->geom_friction + 3 * g);
  } else {
    c->dim = m->geom_condim[g1] > m->geom_condim[g2] ? m->geom_condim[g1] : m->geom_condim[g2];
    double s1 = m->geom_solmix[g1], s2 = m->geom_solmix[g2], mix;
    if (s1 >= O_MINVAL && s2 >= O_MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < O_MINVAL && s2 < O_MINVAL) mix = 0.5;
    else if (s1 < O_MINVAL) mix = 0.0;
    else mix = 1.0;
    const double *r1 = m->geom_solref + 2 * g1, *r2 = m->geom_solref + 2 * g2;
    if (r1[0] > 0 && r2[0] > 0) for (int i = 0; i < 2; i++) c->solref[i] = mix * r1[i] + (1 - mix) * r2[i];
    else for (int i = 0; i < 2; i++) c->solref[i] = fmin(r1[i], r2[i]);
    for (int i = 0; i < 5; i++) c->solimp[i] = mix * m->geom_solimp[5 * g1 + i] + (1 - mix) * m->geom_solimp[5 * g2 + i];
    for (int i = 0; i < 3; i++) fri[i] = fmax(m->geom_friction[3 * g1 + i], m->geom_friction[3 * g2 + i]);
  }
  c->friction[0] = fri[0]; c->friction[1] = fri[0]; c->friction[2] = fri[1];
  c->friction[3] = fri[2]; c->friction[4] = fri[2];
}

static void collision(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  d->ncon = 0;
  if (m->disableflags & (MJPC_DSBL_CONTACT | MJPC_DSBL_CONSTRAINT)) return;
  for (int p = 0; p < om->npair; p++) {
    int g1 = om->pair_g1[p], g2 = om->pair_g2[p];
    OContact proto;
    double margin, gap;
    memset(&proto, 0, sizeof(proto));
    contact_param(m, g1, g2, &proto, &margin, &gap);
    /* bounding-sphere / plane-sphere filter */
    double r1 = m->geom_rbound[g1], r2 = m->geom_rbound[g2];
    const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
    if (m->geom_type[g1] == MJPC_GEOM_PLANE) {
      const double *mat = d->geom_xmat + 9 * g1;
      double n[3] = {mat[2], mat[5], mat[8]}, dif[3];
      o_sub3(dif, p2, p1);
      if (o_dot3(dif, n) > margin + r2) continue;
    } else if (r1 > 0 && r2 > 0) {
      double dif[3]; o_sub3(dif, p2, p1);
      double bound = r1 + r2 + margin;
      if (o_dot3(dif, dif) > bound * bound) continue;
    }
    OContact con[8];
    int before = d->unsupported;
    int n = oracle_collide_pair(om, d, g1, g2, margin, con, &d->unsupported);
    if (d->unsupported != before) d->warning |= MJPC_WARN_UNSUPPORTED;   /* a pair without a collider came within reach: fail loudly */
    for (int k = 0; k < n; k++) {
      if (d->ncon >= om->nconmax) { d->warning |= MJPC_WARN_CONTACTFULL; return; }   /* contact buffer full */
      OContact *c = d->contact + d->ncon++;
      *c = proto;
      c->dist = con[k].dist;
      o_copy3(c->pos, con[k].pos);
      o_copy(c->frame, con[k].frame, 9);
      o_makeframe(c->frame);
      c->includemargin = margin - gap;
      c->geom1 = g1; c->geom2 = g2;
    }
  }
}

/* impedance d(r) and derived K,B  (MuJoCo "solref/solimp" soft-constraint model) */
static double impedance(const double *solimp_in, double pos, double margin) {
  double si[5]; o_copy(si, solimp_in, 5);
  const double MINIMP = 0.0001, MAXIMP = 0.9999;
  si[0] = o_clip(si[0], MINIMP, MAXIMP); si[1] = o_clip(si[1], MINIMP, MAXIMP);
  si[2] = fmax(0, si[2]); si[3] = o_clip(si[3], MINIMP, MAXIMP); si[4] = fmax(1, si[4]);
  if (si[0] == si[1] || si[2] <= O_MINVAL) return 0.5 * (si[0] + si[1]);
  double x = (pos - margin) / si[2];
  if (x < 0) x = -x;
  if (x >= 1) return si[1];
  if (x == 0) return si[0];
  double y;
  if (si[4] == 1) y = x;
  else if (x <= si[3]) { double a = 1 / pow(si[3], si[4] - 1); y = a * pow(x, si[4]); }
  else { double b = 1 / pow(1 - si[3], si[4] - 1); y = 1 - b * pow(1 - x, si[4]); }
  return si[0] + y * (si[1] - si[0]);
}

/* mju_quat2Vel(quat, 1) + mju_normalize3: rotation axis (unit; (1,0,0) for a null rotation) and angle in (-pi, pi] of a unit
 * quaternion */
double oracle_ball_angle(double axis[3], const double *quat) {
  axis[0] = quat[1]; axis[1] = quat[2]; axis[2] = quat[3];
  double s = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  if (s < O_MINVAL) { axis[0] = 1; axis[1] = 0; axis[2] = 0; } else { axis[0] /= s; axis[1] /= s; axis[2] /= s; }
  double speed = 2 * atan2(s, quat[0]);
  if (speed > O_PI) speed -= 2 * O_PI;
  /* angleAxis = axis * speed, then its norm / direction: a negative speed flips the axis */
  double v[3] = {axis[0] * speed, axis[1] * speed, axis[2] * speed};
  double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (n < O_MINVAL) { axis[0] = 1; axis[1] = 0; axis[2] = 0; } else { axis[0] = v[0] / n; axis[1] = v[1] / n; axis[2] = v[2] / n; }
  return n;
}

static int add_row(const OModel *om, OData *d, int type, int id) {
  if (d->nefc >= om->nefcmax) { d->warning |= MJPC_WARN_CNSTRFULL; return -1; }   /* constraint buffer full */
  int r = d->nefc++;
  o_zero(d->efc_J + r * om->m.nv, om->m.nv);
  d->efc_type[r] = type; d->efc_id[r] = id;
  d->efc_pos[r] = 0; d->efc_margin[r] = 0; d->efc_frictionloss[r] = 0;
  return r;
}

static void make_constraint(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  d->nefc = 0; d->nf = 0; d->nl = 0;
  const int no_fric = m->disableflags & (MJPC_DSBL_CONSTRAINT | MJPC_DSBL_FRICTIONLOSS), no_limit = m->disableflags & (MJPC_DSBL_CONSTRAINT | MJPC_DSBL_LIMIT);
  const int no_eq = m->disableflags & (MJPC_DSBL_CONSTRAINT | MJPC_DSBL_EQUALITY);
  /* equality constraints (mj_instantiateEquality): connect = the anchor of body 1 minus the anchor of body 2 (3 rows), joint =
   * (q1 - q1_0) - poly(q2 - q2_0) (1 row); always active, two-sided */
  for (int e = 0; e < m->neq && !no_eq; e++) if (m->eq_active0[e]) {
    const double *data = m->eq_data + 11 * e;
    if (m->eq_type[e] == MJPC_EQ_CONNECT) {
      int b1 = m->eq_obj1id[e], b2 = m->eq_obj2id[e];
      double p1[3], p2[3], *j1 = d->work, *j2 = d->work + 3 * nv;
      o_mulmatvec3(p1, d->xmat + 9 * b1, data); o_add3(p1, p1, d->xpos + 3 * b1);
      o_mulmatvec3(p2, d->xmat + 9 * b2, data + 3); o_add3(p2, p2, d->xpos + 3 * b2);
      jac_point(om, d, j1, NULL, p1, b1); jac_point(om, d, j2, NULL, p2, b2);
      for (int k = 0; k < 3; k++) {
        int r = add_row(om, d, O_CNSTR_EQUALITY, e); if (r < 0) return;
        for (int i = 0; i < nv; i++) d->efc_J[r * nv + i] = j1[k * nv + i] - j2[k * nv + i];
        d->efc_pos[r] = p1[k] - p2[k];
        o_copy(d->efc_solref + 2 * r, m->eq_solref + 2 * e, 2); o_copy(d->efc_solimp + 5 * r, m->eq_solimp + 5 * e, 5);
        d->efc_diagApprox[r] = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
      }
    } else if (m->eq_type[e] == MJPC_EQ_WELD) {
      /* weld (body semantic): rows 0-2 = (x1 + R1 data[3:6]) - (x2 + R2 data[0:3]); rows 3-5 = torquescale * vec(conj(q2) q1 relpose),
       * whose Jacobian is torquescale * 0.5 * vec(conj(q2) (0, w1 - w2) q1 relpose) per dof column; data[6:10] = relpose,
       * data[10] = torquescale */
      int b1 = m->eq_obj1id[e], b2 = m->eq_obj2id[e];
      double p1[3], p2[3], *j1 = d->work, *j2 = d->work + 6 * nv, cpos[6], quat[4], quat1[4], quat2[4], ts = data[10];
      o_mulmatvec3(p1, d->xmat + 9 * b1, data + 3); o_add3(p1, p1, d->xpos + 3 * b1);
      o_mulmatvec3(p2, d->xmat + 9 * b2, data); o_add3(p2, p2, d->xpos + 3 * b2);
      jac_point(om, d, j1, j1 + 3 * nv, p1, b1); jac_point(om, d, j2, j2 + 3 * nv, p2, b2);
      o_mulquat(quat, d->xquat + 4 * b1, data + 6);
      quat1[0] = d->xquat[4 * b2]; for (int k = 1; k < 4; k++) quat1[k] = -d->xquat[4 * b2 + k];
      o_mulquat(quat2, quat1, quat);
      for (int k = 0; k < 3; k++) { cpos[k] = p1[k] - p2[k]; cpos[3 + k] = ts * quat2[1 + k]; }
      for (int i = 0; i < nv; i++) {
        double ax[4] = {0, j1[3 * nv + i] - j2[3 * nv + i], j1[4 * nv + i] - j2[4 * nv + i], j1[5 * nv + i] - j2[5 * nv + i]}, t[4], q3[4];
        o_mulquat(t, quat1, ax); o_mulquat(q3, t, quat);
        for (int k = 0; k < 3; k++) { j1[k * nv + i] -= j2[k * nv + i]; j1[(3 + k) * nv + i] = ts * (0.5 * q3[1 + k]); }
      }
      for (int k = 0; k < 6; k++) {
        int r = add_row(om, d, O_CNSTR_EQUALITY, e); if (r < 0) return;
        o_copy(d->efc_J + r * nv, j1 + k * nv, nv);
        d->efc_pos[r] = cpos[k];
        o_copy(d->efc_solref + 2 * r, m->eq_solref + 2 * e, 2); o_copy(d->efc_solimp + 5 * r, m->eq_solimp + 5 * e, 5);
        d->efc_diagApprox[r] = m->body_invweight0[2 * b1 + (k > 2)] + m->body_invweight0[2 * b2 + (k > 2)];
      }
    } else if (m->eq_type[e] == MJPC_EQ_JOINT) {
      int j1 = m->eq_obj1id[e], j2 = m->eq_obj2id[e];
      int r = add_row(om, d, O_CNSTR_EQUALITY, e); if (r < 0) return;
      double pos = d->qpos[m->jnt_qposadr[j1]] - m->qpos0[m->jnt_qposadr[j1]];
      d->efc_J[r * nv + m->jnt_dofadr[j1]] = 1;
      d->efc_diagApprox[r] = m->dof_invweight0[m->jnt_dofadr[j1]];
      if (j2 >= 0) {
        double dif = d->qpos[m->jnt_qposadr[j2]] - m->qpos0[m->jnt_qposadr[j2]];
        pos -= data[0] + dif * (data[1] + dif * (data[2] + dif * (data[3] + dif * data[4])));
        d->efc_J[r * nv + m->jnt_dofadr[j2]] = -(data[1] + dif * (2 * data[2] + dif * (3 * data[3] + dif * 4 * data[4])));
        d->efc_diagApprox[r] += m->dof_invweight0[m->jnt_dofadr[j2]];
      } else pos -= data[0];
      d->efc_pos[r] = pos;
      o_copy(d->efc_solref + 2 * r, m->eq_solref + 2 * e, 2); o_copy(d->efc_solimp + 5 * r, m->eq_solimp + 5 * e, 5);
    } else if (m->eq_type[e] == MJPC_EQ_TENDON) {     /* (L1 - L1_0) - poly(L2 - L2_0), lengths of fixed tendons, L_0 at qpos0 */
      int t1 = m->eq_obj1id[e], t2 = m->eq_obj2id[e];
      int r = add_row(om, d, O_CNSTR_EQUALITY, e); if (r < 0) return;
      double pos = 0, dif = 0, deriv = 0;
      for (int w = m->tendon_adr[t1]; w < m->tendon_adr[t1] + m->tendon_num[t1]; w++) {
        int qa = m->jnt_qposadr[m->wrap_objid[w]];
        pos += m->wrap_prm[w] * (d->qpos[qa] - m->qpos0[qa]);
      }
      d->efc_diagApprox[r] = m->tendon_invweight0[t1];
      if (t2 >= 0) {
        for (int w = m->tendon_adr[t2]; w < m->tendon_adr[t2] + m->tendon_num[t2]; w++) {
          int qa = m->jnt_qposadr[m->wrap_objid[w]];
          dif += m->wrap_prm[w] * (d->qpos[qa] - m->qpos0[qa]);
        }
        pos -= data[0] + dif * (data[1] + dif * (data[2] + dif * (data[3] + dif * data[4])));
        deriv = data[1] + dif * (2 * data[2] + dif * (3 * data[3] + dif * 4 * data[4]));
        d->efc_diagApprox[r] += m->tendon_invweight0[t2];
      } else pos -= data[0];
      for (int w = m->tendon_adr[t1]; w < m->tendon_adr[t1] + m->tendon_num[t1]; w++) d->efc_J[r * nv + m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w];
      if (t2 >= 0) for (int w = m->tendon_adr[t2]; w < m->tendon_adr[t2] + m->tendon_num[t2]; w++) d->efc_J[r * nv + m->jnt_dofadr[m->wrap_objid[w]]] -= deriv * m->wrap_prm[w];
      d->efc_pos[r] = pos;
      o_copy(d->efc_solref + 2 * r, m->eq_solref + 2 * e, 2); o_copy(d->efc_solimp + 5 * r, m->eq_solimp + 5 * e, 5);
    } else { d->unsupported++; }
  }
  /* friction loss */
  for (int i = 0; i < nv; i++) if (!no_fric && m->dof_frictionloss[i] > 0) {
    int r = add_row(om, d, O_CNSTR_FRICTION_DOF, i); if (r < 0) return;
    d->efc_J[r * nv + i] = 1;
    d->efc_frictionloss[r] = m->dof_frictionloss[i];
    o_copy(d->efc_solref + 2 * r, m->dof_solref + 2 * i, 2);
    o_copy(d->efc_solimp + 5 * r, m->dof_solimp + 5 * i, 5);
    d->efc_diagApprox[r] = m->dof_invweight0[i];
    d->nf++;
  }
  /* tendon friction loss (mjCNSTR_FRICTION_TENDON): one friction row along the tendon, J = the tendon's coefficients */
  for (int t = 0; t < m->ntendon; t++) if (!no_fric && m->tendon_frictionloss && m->tendon_frictionloss[t] > 0) {
    static const double def_ref[2] = {0.02, 1.0}, def_imp[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
    int r = add_row(om, d, O_CNSTR_FRICTION_TENDON, t); if (r < 0) return;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++)
      d->efc_J[r * nv + m->jnt_dofadr[m->wrap_objid[w]]] = m->wrap_prm[w];
    d->efc_frictionloss[r] = m->tendon_frictionloss[t];
    o_copy(d->efc_solref + 2 * r, m->tendon_solref_fri ? m->tendon_solref_fri + 2 * t : def_ref, 2);
    o_copy(d->efc_solimp + 5 * r, m->tendon_solimp_fri ? m->tendon_solimp_fri + 5 * t : def_imp, 5);
    d->efc_diagApprox[r] = m->tendon_invweight0[t];
    d->nf++;
  }
  /* joint limits */
  for (int j = 0; j < m->njnt; j++) if (!no_limit && m->jnt_limited[j] &&
      (m->jnt_type[j] == MJPC_JNT_SLIDE || m->jnt_type[j] == MJPC_JNT_HINGE)) {
    double value = d->qpos[m->jnt_qposadr[j]], margin = m->jnt_margin[j];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->jnt_range[2 * j + (side + 1) / 2] - value);
      if (dist < margin) {
        int r = add_row(om, d, O_CNSTR_LIMIT_JOINT, j); if (r < 0) return;
        d->efc_J[r * nv + m->jnt_dofadr[j]] = -side;
        d->efc_pos[r] = dist; d->efc_margin[r] = margin;
        o_copy(d->efc_solref + 2 * r, m->jnt_solref + 2 * j, 2);
        o_copy(d->efc_solimp + 5 * r, m->jnt_solimp + 5 * j, 5);
        d->efc_diagApprox[r] = m->dof_invweight0[m->jnt_dofadr[j]];
        d->nl++;
      }
    }
  }
  /* ball-joint limits (mj_instantiateLimit): rotation angle of the joint quaternion against max(range); J = -axis at the three
   * dofs.  Rows follow the hinge / slide limits (MuJoCo interleaves them in joint order: same constraint set, the engine keeps
   * its single-entry rows together) */
  for (int j = 0; j < m->njnt; j++) if (!no_limit && m->jnt_limited[j] && m->jnt_type[j] == MJPC_JNT_BALL) {
    double axis[3], angle = oracle_ball_angle(axis, d->qpos + m->jnt_qposadr[j]);
    double margin = m->jnt_margin[j];
    double dist = fmax(m->jnt_range[2 * j], m->jnt_range[2 * j + 1]) - angle;
    if (dist < margin) {
      int r = add_row(om, d, O_CNSTR_LIMIT_JOINT, j); if (r < 0) return;
      for (int k = 0; k < 3; k++) d->efc_J[r * nv + m->jnt_dofadr[j] + k] = -axis[k];
      d->efc_pos[r] = dist; d->efc_margin[r] = margin;
      o_copy(d->efc_solref + 2 * r, m->jnt_solref + 2 * j, 2);
      o_copy(d->efc_solimp + 5 * r, m->jnt_solimp + 5 * j, 5);
      d->efc_diagApprox[r] = m->dof_invweight0[m->jnt_dofadr[j]];
      d->nl++;
    }
  }
  /* fixed-tendon limits: length = sum coef*qpos, J = coef at the joints' dofs */
  for (int t = 0; t < m->ntendon; t++) if (!no_limit && m->tendon_limited[t]) {
    double value = 0, margin = m->tendon_margin[t];
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) value += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[m->wrap_objid[w]]];
    for (int side = -1; side <= 1; side += 2) {
      double dist = side * (m->tendon_range[2 * t + (side + 1) / 2] - value);
      if (dist < margin) {
        int r = add_row(om, d, O_CNSTR_LIMIT_TENDON, t); if (r < 0) return;
        for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) d->efc_J[r * nv + m->jnt_dofadr[m->wrap_objid[w]]] = -side * m->wrap_prm[w];
        d->efc_pos[r] = dist; d->efc_margin[r] = margin;
        o_copy(d->efc_solref + 2 * r, m->tendon_solref_lim + 2 * t, 2);
        o_copy(d->efc_solimp + 5 * r, m->tendon_solimp_lim + 5 * t, 5);
        d->efc_diagApprox[r] = m->tendon_invweight0[t];
        d->nl++;
      }
    }
  }
  /* contacts */
  double *jp1 = d->work, *jp2 = jp1 + 3 * nv, *jr1 = jp2 + 3 * nv, *jr2 = jr1 + 3 * nv;
  for (int ci = 0; ci < d->ncon; ci++) {
    OContact *c = d->contact + ci;
    int b1 = m->geom_bodyid[c->geom1], b2 = m->geom_bodyid[c->geom2];
    int dim = c->dim;
    int elliptic = (dim > 1 && m->cone == MJPC_CONE_ELLIPTIC);
    jac_point(om, d, jp1, dim > 3 ? jr1 : NULL, c->pos, b1);
    jac_point(om, d, jp2, dim > 3 ? jr2 : NULL, c->pos, b2);
    double tran = m->body_invweight0[2 * b1] + m->body_invweight0[2 * b2];
    double rot = m->body_invweight0[2 * b1 + 1] + m->body_invweight0[2 * b2 + 1];
    c->efc_address = d->nefc;
    if (dim > 1 && !elliptic) {
      /* pyramidal cone: 2(dim-1) unilateral rows  Jn +- mu_k * J_k */
      for (int k = 1; k < dim; k++) for (int sgn = 1; sgn >= -1; sgn -= 2) {
        int r = add_row(om, d, O_CNSTR_CONTACT_PYRAMIDAL, ci);
        if (r < 0) { d->nefc = c->efc_address; return; }   /* buffer full: no half-built cone is left behind */
        const double *ax = c->frame + 3 * (k % 3);
        const double *ja = k < 3 ? jp1 : jr1, *jb = k < 3 ? jp2 : jr2;
        double mu = c->friction[k - 1];
        for (int i = 0; i < nv; i++) {
          double jn = c->frame[0] * (jp2[i] - jp1[i]) + c->frame[1] * (jp2[nv + i] - jp1[nv + i]) + c->frame[2] * (jp2[2 * nv + i] - jp1[2 * nv + i]);
          double jk = ax[0] * (jb[i] - ja[i]) + ax[1] * (jb[nv + i] - ja[nv + i]) + ax[2] * (jb[2 * nv + i] - ja[2 * nv + i]);
          d->efc_J[r * nv + i] = jn + sgn * mu * jk;
        }
        d->efc_pos[r] = c->dist; d->efc_margin[r] = c->includemargin;
        o_copy(d->efc_solref + 2 * r, c->solref, 2);
        o_copy(d->efc_solimp + 5 * r, c->solimp, 5);
        d->efc_diagApprox[r] = tran + mu * mu * (k < 3 ? tran : rot);
      }
      continue;
    }
    for (int k = 0; k < dim; k++) {
      int r = add_row(om, d, dim == 1 ? O_CNSTR_CONTACT_FRICTIONLESS : O_CNSTR_CONTACT_ELLIPTIC, ci);
      if (r < 0) { d->nefc = c->efc_address; return; }
      const double *ax = c->frame + 3 * (k % 3);
      const double *ja = k < 3 ? jp1 : jr1, *jb = k < 3 ? jp2 : jr2;
      for (int i = 0; i < nv; i++)
        d->efc_J[r * nv + i] = ax[0] * (jb[i] - ja[i]) + ax[1] * (jb[nv + i] - ja[nv + i]) + ax[2] * (jb[2 * nv + i] - ja[2 * nv + i]);
      d->efc_pos[r] = c->dist; d->efc_margin[r] = c->includemargin;
      o_copy(d->efc_solref + 2 * r, c->solref, 2);
      o_copy(d->efc_solimp + 5 * r, c->solimp, 5);
      d->efc_diagApprox[r] = k < 3 ? tran : rot;
    }
  }
}

/* efc_vel, KBIP, R, D, aref */
static void make_impedance(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  for (int r = 0; r < d->nefc; r++) d->efc_vel[r] = o_dot(d->efc_J + r * nv, d->qvel, nv);
  for (int r = 0; r < d->nefc; r++) {
    int dim = 1, pyramidal = d->efc_type[r] == O_CNSTR_CONTACT_PYRAMIDAL;
    if (d->efc_type[r] == O_CNSTR_CONTACT_ELLIPTIC) dim = d->contact[d->efc_id[r]].dim;
    if (pyramidal) dim = 2 * (d->contact[d->efc_id[r]].dim - 1);
    const double *solref = d->efc_solref + 2 * r, *solimp = d->efc_solimp + 5 * r;
    const int equality = d->efc_type[r] == O_CNSTR_EQUALITY;
    double imp_pos = d->efc_pos[r];
    if (equality && m->eq_type[d->efc_id[r]] == MJPC_EQ_CONNECT) { dim = 3; imp_pos = o_norm(d->efc_pos + r, 3); }    /* one impedance from |residual| */
    if (equality && m->eq_type[d->efc_id[r]] == MJPC_EQ_WELD) { dim = 6; imp_pos = o_norm(d->efc_pos + r, 6); }
    double imp = impedance(solimp, imp_pos, d->efc_margin[r]);
    double dmax = o_clip(solimp[1], 0.0001, 0.9999);
    double K, B;
    if (solref[0] > 0) {
      double tc = fmax(solref[0], 2 * m->timestep);   /* refsafe */
      double dr = solref[1];
      K = 1 / fmax(O_MINVAL, dmax * dmax * tc * tc * dr * dr);
      B = 2 / fmax(O_MINVAL, dmax * tc);
    } else {
      K = -solref[0] / fmax(O_MINVAL, dmax * dmax);
      B = -solref[1] / fmax(O_MINVAL, dmax);
    }
    for (int k = 0; k < dim; k++) {
      int q = r + k;
      int friction_row = (d->efc_type[q] == O_CNSTR_FRICTION_DOF || d->efc_type[q] == O_CNSTR_FRICTION_TENDON) || (k > 0 && !pyramidal && !equality);
      double Kq = friction_row ? 0 : K;
      d->efc_KBIP[4 * q] = Kq; d->efc_KBIP[4 * q + 1] = B; d->efc_KBIP[4 * q + 2] = imp; d->efc_KBIP[4 * q + 3] = 0;
      d->efc_R[q] = fmax(O_MINVAL, (1 - imp) / imp * d->efc_diagApprox[q]);
      d->efc_aref[q] = -B * d->efc_vel[q] - Kq * imp * (d->efc_pos[q] - d->efc_margin[q]);
    }
    if (pyramidal) {   /* all pyramid edges share Rpy = 2 mu^2 R0, mu = friction of the regularised cone */
      OContact *c = d->contact + d->efc_id[r];
      double *R = d->efc_R + r;
      double R1 = R[0] / fmax(O_MINVAL, m->impratio);
      c->mu = c->friction[0] * sqrt(R1 / R[0]);
      double Rpy = 2 * c->mu * c->mu * R[0];
      for (int k = 0; k < dim; k++) R[k] = Rpy;
    } else if (dim > 1 && !equality) {   /* elliptic cone: friction regularisation from impratio, regularised mu */
      OContact *c = d->contact + d->efc_id[r];
      double *R = d->efc_R + r;
      R[1] = R[0] / fmax(O_MINVAL, m->impratio);
      c->mu = c->friction[0] * sqrt(R[1] / R[0]);
      for (int k = 2; k < dim; k++)
        R[k] = R[1] * c->friction[0] * c->friction[0] / (c->friction[k - 1] * c->friction[k - 1]);
    }
    for (int k = 0; k < dim; k++) d->efc_D[r + k] = 1 / d->efc_R[r + k];
    r += dim - 1;
  }
}

/* ---- velocity stage ------------------------------------------------------------------ */
static void com_vel(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  o_zero(d->cvel, 6);
  for (int i = 1; i < m->nbody; i++) {
    double cvel[6];
    o_copy(cvel, d->cvel + 6 * m->body_parentid[i], 6);
    int bda = m->body_dofadr[i];
    for (int j = m->body_jntadr[i]; j < m->body_jntadr[i] + m->body_jntnum[i]; j++) {
      int type = m->jnt_type[j];
      if (type == MJPC_JNT_FREE) {
        o_zero(d->cdof_dot + 6 * bda, 18);
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (bda + k) + c] * d->qvel[bda + k];
        bda += 3;
      }
      if (type == MJPC_JNT_FREE || type == MJPC_JNT_BALL) {
        for (int k = 0; k < 3; k++) o_crossmotion(d->cdof_dot + 6 * (bda + k), cvel, d->cdof + 6 * (bda + k));
        for (int k = 0; k < 3; k++) for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * (bda + k) + c] * d->qvel[bda + k];
        bda += 3;
      } else {
        o_crossmotion(d->cdof_dot + 6 * bda, cvel, d->cdof + 6 * bda);
        for (int c = 0; c < 6; c++) cvel[c] += d->cdof[6 * bda + c] * d->qvel[bda];
        bda++;
      }
    }
    o_copy(d->cvel + 6 * i, cvel, 6);
  }
  /* subtree linear velocity (mj_subtreeVel, linear part) */
  for (int i = 0; i < m->nbody; i++) {
    double off[3], v[3];
    o_sub3(off, d->xipos + 3 * i, d->subtree_com + 3 * m->body_rootid[i]);
    o_cross(v, d->cvel + 6 * i, off);
    o_add3(v, v, d->cvel + 6 * i + 3);
    o_scl3(d->subtree_linvel + 3 * i, v, m->body_mass[i]);
  }
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    o_add3(d->subtree_linvel + 3 * p, d->subtree_linvel + 3 * p, d->subtree_linvel + 3 * i);
  }
  for (int i = 0; i < m->nbody; i++) {
    double s = 1.0 / fmax(O_MINVAL, m->body_subtreemass[i]);
    o_scl3(d->subtree_linvel + 3 * i, d->subtree_linvel + 3 * i, s);
  }
}

static void passive(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  o_zero(d->qfrc_passive, m->nv);
  for (int j = 0; j < m->njnt; j++) {
    double k = m->jnt_stiffness[j];
    if (k == 0) continue;
    if (m->jnt_type[j] == MJPC_JNT_SLIDE || m->jnt_type[j] == MJPC_JNT_HINGE) {
      int qa = m->jnt_qposadr[j];
      d->qfrc_passive[m->jnt_dofadr[j]] -= k * (d->qpos[qa] - m->qpos_spring[qa]);
    }
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_passive[i] -= m->dof_damping[i] * d->qvel[i];
  /* tendon spring (dead band lengthspring) and damper, fixed tendons: J = the wrap coefficients (mj_passive, engine_passive.c) */
  for (int t = 0; t < m->ntendon; t++) {
    double k = m->tendon_stiffness ? m->tendon_stiffness[t] : 0, b = m->tendon_damping ? m->tendon_damping[t] : 0;
    if (k == 0 && b == 0) continue;
    double length = 0, velocity = 0;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) {
      int j = m->wrap_objid[w];
      length += m->wrap_prm[w] * d->qpos[m->jnt_qposadr[j]]; velocity += m->wrap_prm[w] * d->qvel[m->jnt_dofadr[j]];
    }
    double lo = m->tendon_lengthspring ? m->tendon_lengthspring[2 * t] : 0, hi = m->tendon_lengthspring ? m->tendon_lengthspring[2 * t + 1] : 0;
    double frc = 0;
    if (length > hi) frc = k * (hi - length); else if (length < lo) frc = k * (lo - length);
    frc -= b * velocity;
    for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) d->qfrc_passive[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w] * frc;
  }
  /* fluid forces, inertia-box model (mj_inertiaBoxFluidModel): per body with mass, the equivalent inertia box, the velocity of the
   * com in the inertial frame minus the wind, viscous (sphere of the mean box size) and quadratic drag force / torque, applied at
   * the com */
  if (m->density > 0 || m->viscosity > 0)
    for (int b = 1; b < m->nbody; b++) if (m->body_mass[b] >= O_MINVAL) {
      const double *I = m->body_inertia + 3 * b, *R = d->ximat + 9 * b;
      double mass = m->body_mass[b], box[3], off[3], vw[3], lvel[6], lfrc[6] = {0, 0, 0, 0, 0, 0}, F[3], T[3];
      box[0] = sqrt(fmax(O_MINVAL, I[1] + I[2] - I[0]) / mass * 6.0);
      box[1] = sqrt(fmax(O_MINVAL, I[0] + I[2] - I[1]) / mass * 6.0);
      box[2] = sqrt(fmax(O_MINVAL, I[0] + I[1] - I[2]) / mass * 6.0);
      o_sub3(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
      o_cross(vw, d->cvel + 6 * b, off); o_add3(vw, vw, d->cvel + 6 * b + 3);          /* velocity of the com, world frame */
      o_sub3(vw, vw, m->wind);
      o_mulmattvec3(lvel, R, d->cvel + 6 * b); o_mulmattvec3(lvel + 3, R, vw);
      if (m->viscosity > 0) {
        double diam = (box[0] + box[1] + box[2]) / 3.0;
        for (int k = 0; k < 3; k++) { lfrc[k] = -O_PI * diam * diam * diam * m->viscosity * lvel[k]; lfrc[3 + k] = -3.0 * O_PI * diam * m->viscosity * lvel[3 + k]; }
      }
      if (m->density > 0) {
        lfrc[3] -= 0.5 * m->density * box[1] * box[2] * fabs(lvel[3]) * lvel[3];
        lfrc[4] -= 0.5 * m->density * box[0] * box[2] * fabs(lvel[4]) * lvel[4];
        lfrc[5] -= 0.5 * m->density * box[0] * box[1] * fabs(lvel[5]) * lvel[5];
        lfrc[0] -= m->density * box[0] * (box[1] * box[1] * box[1] * box[1] + box[2] * box[2] * box[2] * box[2]) * fabs(lvel[0]) * lvel[0] / 64.0;
        lfrc[1] -= m->density * box[1] * (box[0] * box[0] * box[0] * box[0] + box[2] * box[2] * box[2] * box[2]) * fabs(lvel[1]) * lvel[1] / 64.0;
        lfrc[2] -= m->density * box[2] * (box[0] * box[0] * box[0] * box[0] + box[1] * box[1] * box[1] * box[1]) * fabs(lvel[2]) * lvel[2] / 64.0;
      }
      o_mulmatvec3(T, R, lfrc); o_mulmatvec3(F, R, lfrc + 3);
      double *jp = d->work, *jr = d->work + 3 * m->nv;
      jac_point(om, d, jp, jr, d->xipos + 3 * b, b);
      for (int i = 0; i < m->nv; i++)
        d->qfrc_passive[i] += jp[i] * F[0] + jp[m->nv + i] * F[1] + jp[2 * m->nv + i] * F[2] + jr[i] * T[0] + jr[m->nv + i] * T[1] + jr[2 * m->nv + i] * T[2];
    }
  /* gravity compensation (mj_passive): -gravity * mass * gravcomp applied at the body's centre of mass */
  if (m->body_gravcomp)
    for (int b = 1; b < m->nbody; b++) if (m->body_gravcomp[b] != 0) {
      double *jp = d->work, s = m->body_mass[b] * m->body_gravcomp[b];
      jac_point(om, d, jp, NULL, d->xipos + 3 * b, b);
      for (int i = 0; i < m->nv; i++)
        d->qfrc_passive[i] -= s * (jp[i] * m->gravity[0] + jp[m->nv + i] * m->gravity[1] + jp[2 * m->nv + i] * m->gravity[2]);
    }
}

static void rne_bias(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  o_zero(d->cacc, 6);
  for (int k = 0; k < 3; k++) d->cacc[3 + k] = -m->gravity[k];
  o_zero(d->cfrc, 6);
  for (int i = 1; i < m->nbody; i++) {
    double *a = d->cacc + 6 * i;
    o_copy(a, d->cacc + 6 * m->body_parentid[i], 6);
    int bda = m->body_dofadr[i];
    for (int k = 0; k < m->body_dofnum[i]; k++)
      for (int c = 0; c < 6; c++) a[c] += d->cdof_dot[6 * (bda + k) + c] * d->qvel[bda + k];
    double t1[6], t2[6], t3[6];
    o_mulinertvec(t1, d->cinert + 10 * i, a);
    o_mulinertvec(t2, d->cinert + 10 * i, d->cvel + 6 * i);
    o_crossforce(t3, d->cvel + 6 * i, t2);
    for (int c = 0; c < 6; c++) d->cfrc[6 * i + c] = t1[c] + t3[c];
  }
  for (int i = m->nbody - 1; i > 0; i--) {
    int p = m->body_parentid[i];
    if (p > 0) for (int c = 0; c < 6; c++) d->cfrc[6 * p + c] += d->cfrc[6 * i + c];
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = o_dot(d->cdof + 6 * i, d->cfrc + 6 * m->dof_bodyid[i], 6);
}

/* mj_fwdActuation for joint and fixed-tendon transmissions: length = sum coef * qpos, velocity = sum coef * qvel with
 * coef = gear (joint) or gear * wrap coefficient (tendon); force = gain * ctrl + bias0 + bias1 * length + bias2 * velocity,
 * clamped to forcerange; qfrc_actuator += moment^T force.  <position kp> servos are gain = kp, bias = (0, -kp, 0). */
static void actuation(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  const int nv = m->nv;
  o_zero(d->qfrc_actuator, m->nv);
  for (int i = 0; i < m->nu; i++) {
    double ctrl = d->ctrl[i];
    if (m->actuator_ctrllimited[i]) ctrl = o_clip(ctrl, m->actuator_ctrlrange[2 * i], m->actuator_ctrlrange[2 * i + 1]);
    double gear = m->actuator_gear[i];
    int id = m->actuator_trnid[i];
    const int refsite = (m->actuator_trntype[i] == MJPC_TRN_SITE && m->actuator_refsite) ? m->actuator_refsite[i] : -1;
    if (refsite >= 0) {
      /* mjTRN_SITE with a reference site (mj_transmission): length = gear[0:3] . R_ref^T (x_site - x_ref), moment = (Jp_site - Jp_ref)^T R_ref
       * gear[0:3], cleared on the dofs the two sites' bodies share; gain / affine bias / activation as for the other transmissions */
      const double *g6 = m->actuator_gear6 + 6 * i;
      if (g6[3] != 0 || g6[4] != 0 || g6[5] != 0) { d->unsupported++; d->warning |= MJPC_WARN_UNSUPPORTED; }
      double *jp = d->work, *jq = d->work + 3 * nv, *mom = d->work + 6 * nv, vec[3], loc[3], w[3];
      int bs = m->site_bodyid[id], br = m->site_bodyid[refsite];
      jac_point(om, d, jp, NULL, d->site_xpos + 3 * id, bs); jac_point(om, d, jq, NULL, d->site_xpos + 3 * refsite, br);
      o_sub3(vec, d->site_xpos + 3 * id, d->site_xpos + 3 * refsite);
      const double *R = d->site_xmat + 9 * refsite;
      for (int k = 0; k < 3; k++) loc[k] = R[k] * vec[0] + R[3 + k] * vec[1] + R[6 + k] * vec[2];      /* R^T vec */
      double length = loc[0] * g6[0] + loc[1] * g6[1] + loc[2] * g6[2], velocity = 0;
      o_mulmatvec3(w, R, g6);
      for (int k = 0; k < nv; k++) mom[k] = (jp[k] - jq[k]) * w[0] + (jp[nv + k] - jq[nv + k]) * w[1] + (jp[2 * nv + k] - jq[2 * nv + k]) * w[2];
      {      /* dofs above both bodies: the common ancestors of the two weld bodies' last dofs */
        int b0 = m->body_weldid[bs], b1 = m->body_weldid[br];
        int d0 = m->body_dofadr[b0] + m->body_dofnum[b0] - 1, d1 = m->body_dofadr[b1] + m->body_dofnum[b1] - 1;
        if (m->body_dofnum[b0] == 0) d0 = -1;
        if (m->body_dofnum[b1] == 0) d1 = -1;
        int common = -1;
        if (d0 >= 0 && d1 >= 0) {
          while (d0 != d1) { if (d0 < d1) d1 = m->dof_parentid[d1]; else d0 = m->dof_parentid[d0]; if (d0 == -1 || d1 == -1) break; }
          if (d0 == d1) common = d0;
        }
        for (; common >= 0; common = m->dof_parentid[common]) mom[common] = 0;
      }
      for (int k = 0; k < nv; k++) velocity += mom[k] * d->qvel[k];
      double input = ctrl;
      if (m->na > 0 && m->actuator_dyntype && m->actuator_dyntype[i] != MJPC_DYN_NONE) {
        int a = m->actuator_actadr[i];
        input = d->act[a];
        d->act_dot[a] = m->actuator_dyntype[i] == MJPC_DYN_INTEGRATOR ? ctrl : (ctrl - input) / fmax(O_MINVAL, m->actuator_dynprm[i]);
      }
      double force = m->actuator_gainprm[3 * i] * input;
      if (m->actuator_biastype[i] == MJPC_BIAS_AFFINE)
        force += m->actuator_biasprm[3 * i] + m->actuator_biasprm[3 * i + 1] * length + m->actuator_biasprm[3 * i + 2] * velocity;
      if (m->actuator_forcelimited[i]) force = o_clip(force, m->actuator_forcerange[2 * i], m->actuator_forcerange[2 * i + 1]);
      d->actuator_force[i] = force;
      for (int k = 0; k < nv; k++) d->qfrc_actuator[k] += mom[k] * force;
      continue;
    }
    if (m->actuator_trntype[i] == MJPC_TRN_SITE) {
      /* mjTRN_SITE without refsite: length 0, moment = J_site^T (R_site gear[0:3]; R_site gear[3:6]); motors only */
      double *jp = d->work, *jr = d->work + 3 * nv, f[3], tq[3];
      double force = m->actuator_gainprm[3 * i] * ctrl;
      if (m->actuator_forcelimited[i]) force = o_clip(force, m->actuator_forcerange[2 * i], m->actuator_forcerange[2 * i + 1]);
      d->actuator_force[i] = force;
      jac_point(om, d, jp, jr, d->site_xpos + 3 * id, m->site_bodyid[id]);
      o_mulmatvec3(f, d->site_xmat + 9 * id, m->actuator_gear6 + 6 * i); o_mulmatvec3(tq, d->site_xmat + 9 * id, m->actuator_gear6 + 6 * i + 3);
      for (int k = 0; k < nv; k++)
        d->qfrc_actuator[k] += force * (jp[k] * f[0] + jp[nv + k] * f[1] + jp[2 * nv + k] * f[2] + jr[k] * tq[0] + jr[nv + k] * tq[1] + jr[2 * nv + k] * tq[2]);
      continue;
    }
    int tendon = m->actuator_trntype[i] == MJPC_TRN_TENDON;
    int w0 = tendon ? m->tendon_adr[id] : 0, nw = tendon ? m->tendon_num[id] : 1;
    double length = 0, velocity = 0;
    for (int w = 0; w < nw; w++) {
      int j = tendon ? m->wrap_objid[w0 + w] : id;
      double coef = tendon ? gear * m->wrap_prm[w0 + w] : gear;
      length += coef * d->qpos[m->jnt_qposadr[j]];
      velocity += coef * d->qvel[m->jnt_dofadr[j]];
    }
    double input = ctrl;
    if (m->na > 0 && m->actuator_dyntype && m->actuator_dyntype[i] != MJPC_DYN_NONE) {
      /* stateful actuator: act_dot from the (clamped) control, force from the current activation */
      int a = m->actuator_actadr[i];
      input = d->act[a];
      d->act_dot[a] = m->actuator_dyntype[i] == MJPC_DYN_INTEGRATOR ? ctrl : (ctrl - input) / fmax(O_MINVAL, m->actuator_dynprm[i]);
    }
    double force = m->actuator_gainprm[3 * i] * input;
    if (m->actuator_biastype[i] == MJPC_BIAS_AFFINE)
      force += m->actuator_biasprm[3 * i] + m->actuator_biasprm[3 * i + 1] * length + m->actuator_biasprm[3 * i + 2] * velocity;
    if (m->actuator_forcelimited[i]) force = o_clip(force, m->actuator_forcerange[2 * i], m->actuator_forcerange[2 * i + 1]);
    d->actuator_force[i] = force;
    for (int w = 0; w < nw; w++) {
      int j = tendon ? m->wrap_objid[w0 + w] : id;
      double coef = tendon ? gear * m->wrap_prm[w0 + w] : gear;
      d->qfrc_actuator[m->jnt_dofadr[j]] += coef * force;
    }
  }
  /* joint-level clamp of the total actuator force (end of mj_fwdActuation): scalar joints with jnt_actfrclimited */
  if (m->jnt_actfrclimited && m->jnt_actfrcrange)
    for (int jn = 0; jn < m->njnt; jn++)
      if (m->jnt_actfrclimited[jn] && (m->jnt_type[jn] == MJPC_JNT_HINGE || m->jnt_type[jn] == MJPC_JNT_SLIDE)) {
        int da = m->jnt_dofadr[jn];
        d->qfrc_actuator[da] = o_clip(d->qfrc_actuator[da], m->jnt_actfrcrange[2 * jn], m->jnt_actfrcrange[2 * jn + 1]);
      }
}

/* ---- primal constraint solver (Newton) ------------------------------------------------ */
/* cost of constraints at jar; fills force/state; optionally cone Hessians */
static double constraint_update(const OModel *om, OData *d, const double *jar, double *force, int *state, int hess) {
  double cost = 0;
  for (int i = 0; i < d->nefc; i++) {
    double D = d->efc_D[i], R = d->efc_R[i], x = jar[i];
    int type = d->efc_type[i];
    if (type == O_CNSTR_EQUALITY) { cost += 0.5 * D * x * x; force[i] = -D * x; state[i] = O_STATE_QUADRATIC; }
    else if (type <= O_CNSTR_FRICTION_TENDON) {
      double f = d->efc_frictionloss[i];
      if (x <= -R * f) { cost += -0.5 * R * f * f - f * x; force[i] = f; state[i] = O_STATE_LINEARNEG; }
      else if (x >= R * f) { cost += -0.5 * R * f * f + f * x; force[i] = -f; state[i] = O_STATE_LINEARPOS; }
      else { cost += 0.5 * D * x * x; force[i] = -D * x; state[i] = O_STATE_QUADRATIC; }
    } else if (type != O_CNSTR_CONTACT_ELLIPTIC) {
      if (x >= 0) { force[i] = 0; state[i] = O_STATE_SATISFIED; }
      else { cost += 0.5 * D * x * x; force[i] = -D * x; state[i] = O_STATE_QUADRATIC; }
    } else {
      OContact *c = d->contact + d->efc_id[i];
      int dim = c->dim;
      double mu = c->mu, U[6];
      U[0] = jar[i] * mu;
      for (int j = 1; j < dim; j++) U[j] = jar[i + j] * c->friction[j - 1];
      double N = U[0], T = o_norm(U + 1, dim - 1);
      if (N >= mu * T || (T <= 0 && N >= 0)) {
        for (int j = 0; j < dim; j++) force[i + j] = 0;
        state[i] = O_STATE_SATISFIED;
      } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        for (int j = 0; j < dim; j++) { cost += 0.5 * d->efc_D[i + j] * jar[i + j] * jar[i + j]; force[i + j] = -d->efc_D[i + j] * jar[i + j]; }
        state[i] = O_STATE_QUADRATIC;
      } else {
        double Dm = d->efc_D[i] / (mu * mu * (1 + mu * mu));
        double NmT = N - mu * T;
        cost += 0.5 * Dm * NmT * NmT;
        force[i] = -Dm * NmT * mu;
        for (int j = 1; j < dim; j++) force[i + j] = -force[i] / T * U[j] * c->friction[j - 1];
        state[i] = O_STATE_CONE;
        if (hess) {
          /* H = S * d2s/dU2 * S, S = diag(mu, friction);  s = 0.5*Dm*(N - mu*T)^2 */
          double g[6], S[6];
          S[0] = mu; g[0] = 1;
          for (int j = 1; j < dim; j++) { S[j] = c->friction[j - 1]; g[j] = -mu * U[j] / T; }
          for (int a = 0; a < dim; a++) for (int b = 0; b < dim; b++) {
            double h = g[a] * g[b];
            if (a > 0 && b > 0) h += NmT * (-mu) * ((a == b ? 1.0 / T : 0.0) - U[a] * U[b] / (T * T * T));
            c->H[a * 6 + b] = Dm * h * S[a] * S[b];
          }
        }
      }
      for (int j = 1; j < dim; j++) state[i + j] = state[i];
      i += dim - 1;
    }
  }
  return cost;
}

typedef struct { double cost, d1, d2; } LSPoint;

/* exact 1-D evaluation of phi(alpha) = Gauss(alpha) + sum_i s_i(jar + alpha*jv) */
static LSPoint ls_eval(const OModel *om, const OData *d, const double *jar, const double *jv, const double quad[3], double a) {
  LSPoint p;
  p.cost = quad[0] + a * quad[1] + a * a * quad[2];
  p.d1 = quad[1] + 2 * a * quad[2];
  p.d2 = 2 * quad[2];
  for (int i = 0; i < d->nefc; i++) {
    double D = d->efc_D[i], R = d->efc_R[i], x = jar[i] + a * jv[i], v = jv[i];
    int type = d->efc_type[i];
    if (type == O_CNSTR_EQUALITY) { p.cost += 0.5 * D * x * x; p.d1 += D * x * v; p.d2 += D * v * v; }
    else if (type <= O_CNSTR_FRICTION_TENDON) {
      double f = d->efc_frictionloss[i];
      if (x <= -R * f) { p.cost += -0.5 * R * f * f - f * x; p.d1 += -f * v; }
      else if (x >= R * f) { p.cost += -0.5 * R * f * f + f * x; p.d1 += f * v; }
      else { p.cost += 0.5 * D * x * x; p.d1 += D * x * v; p.d2 += D * v * v; }
    } else if (type != O_CNSTR_CONTACT_ELLIPTIC) {
      if (x < 0) { p.cost += 0.5 * D * x * x; p.d1 += D * x * v; p.d2 += D * v * v; }
    } else {
      const OContact *c = d->contact + d->efc_id[i];
      int dim = c->dim;
      double mu = c->mu, U[6], V[6];
      U[0] = x * mu; V[0] = v * mu;
      for (int j = 1; j < dim; j++) { U[j] = (jar[i + j] + a * jv[i + j]) * c->friction[j - 1]; V[j] = jv[i + j] * c->friction[j - 1]; }
      double N = U[0], T = o_norm(U + 1, dim - 1);
      if (N >= mu * T || (T <= 0 && N >= 0)) {
        /* satisfied */
      } else if (mu * N + T <= 0 || (T <= 0 && N < 0)) {
        for (int j = 0; j < dim; j++) {
          double xj = jar[i + j] + a * jv[i + j], Dj = d->efc_D[i + j];
          p.cost += 0.5 * Dj * xj * xj; p.d1 += Dj * xj * jv[i + j]; p.d2 += Dj * jv[i + j] * jv[i + j];
        }
      } else {
        double Dm = d->efc_D[i] / (mu * mu * (1 + mu * mu));
        double NmT = N - mu * T;
        double UV = o_dot(U + 1, V + 1, dim - 1), VV = o_dot(V + 1, V + 1, dim - 1);
        double T1 = UV / T;
        double T2 = (VV - UV * UV / (T * T)) / T;
        double g1 = V[0] - mu * T1;
        p.cost += 0.5 * Dm * NmT * NmT;
        p.d1 += Dm * NmT * g1;
        p.d2 += Dm * (g1 * g1 - NmT * mu * T2);
      }
      i += dim - 1;
    }
  }
  return p;
}

/* safeguarded Newton on phi'(alpha); phi convex, C1 */
static double line_search(const OModel *om, OData *d, double cost0_gauss, double cost0) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  double snorm = o_norm(d->search, nv);
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  if (snorm < O_MINVAL) return 0;
  double gtol = m->tolerance * m->ls_tolerance * snorm / scale;
  /* Mv, jv, Gauss quadratic */
  for (int i = 0; i < nv; i++) d->Mv[i] = o_dot(d->qM + i * nv, d->search, nv);
  for (int r = 0; r < d->nefc; r++) d->efc_jv[r] = o_dot(d->efc_J + r * nv, d->search, nv);
  double quad[3];
  quad[0] = cost0_gauss;
  quad[1] = 0; for (int i = 0; i < nv; i++) quad[1] += d->search[i] * (d->Ma[i] - d->qfrc_smooth[i]);
  quad[2] = 0.5 * o_dot(d->search, d->Mv, nv);
  /* the point alpha = 0 needs no evaluation: search = -H^-1 grad with H the exact Hessian there, so
   * phi(0) = cost, phi'(0) = grad.search = -phi''(0), and the Newton step from alpha = 0 is exactly 1 */
  LSPoint p0;
  p0.cost = cost0; p0.d1 = o_dot(d->grad, d->search, nv); p0.d2 = -p0.d1;
  if (p0.d1 >= 0) return 0;
  /* safeguarded Newton on phi'(alpha) (rtsafe): expand until phi' changes sign, then Newton steps that
   * stay inside the bracket and at least halve the previous step, else bisection; return the best point */
  double lo = 0, hi = -1;   /* hi < 0: no upper bracket yet */
  double a = 1.0;
  double best_a = 0, best_cost = p0.cost, dxold = a, dx = a;
  for (int it = 0; it < m->ls_iterations; it++) {
    LSPoint p = ls_eval(om, d, d->efc_jar, d->efc_jv, quad, a);
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
  if (getenv("ORACLE_DEBUG_SOLVER")) fprintf(stderr, "    ls: p0 cost %.10g d1 %.4g | best a %.6g cost %.10g lo %.4g hi %.4g\n", p0.cost, p0.d1, best_a, best_cost, lo, hi);
  return best_a;
}

static void newton_gradient(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  /* grad = Ma - qfrc_smooth - J^T force */
  for (int i = 0; i < nv; i++) d->grad[i] = d->Ma[i] - d->qfrc_smooth[i];
  for (int r = 0; r < d->nefc; r++) {
    double f = d->efc_force[r];
    if (f == 0) continue;
    for (int i = 0; i < nv; i++) d->grad[i] -= d->efc_J[r * nv + i] * f;
  }
  /* H = M + J^T diag(D*active) J + cone blocks */
  o_copy(d->qH, d->qM, nv * nv);
  for (int r = 0; r < d->nefc; r++) {
    if (d->efc_state[r] == O_STATE_QUADRATIC) {
      const double *J = d->efc_J + r * nv;
      double D = d->efc_D[r];
      for (int i = 0; i < nv; i++) { if (J[i] == 0) continue; double s = D * J[i]; for (int j = 0; j < nv; j++) d->qH[i * nv + j] += s * J[j]; }
    } else if (d->efc_state[r] == O_STATE_CONE) {
      const OContact *c = d->contact + d->efc_id[r];
      int dim = c->dim;
      for (int a = 0; a < dim; a++) for (int b = 0; b < dim; b++) {
        double h = c->H[a * 6 + b];
        const double *Ja = d->efc_J + (r + a) * nv, *Jb = d->efc_J + (r + b) * nv;
        for (int i = 0; i < nv; i++) { if (Ja[i] == 0) continue; double s = h * Ja[i]; for (int j = 0; j < nv; j++) d->qH[i * nv + j] += s * Jb[j]; }
      }
      r += dim - 1;
    }
  }
  chol_factor(d->qLD2, d->qH, nv);
  chol_solve(d->Mgrad, d->qLD2, d->grad, nv);
}

static double total_cost(const OModel *om, OData *d, const double *qacc, double *gauss_out) {
  int nv = om->m.nv;
  for (int i = 0; i < nv; i++) d->Ma[i] = o_dot(d->qM + i * nv, qacc, nv);
  for (int r = 0; r < d->nefc; r++) d->efc_jar[r] = o_dot(d->efc_J + r * nv, qacc, nv) - d->efc_aref[r];
  double gauss = 0;
  for (int i = 0; i < nv; i++) gauss += 0.5 * (d->Ma[i] - d->qfrc_smooth[i]) * (qacc[i] - d->qacc_smooth[i]);
  if (gauss_out) *gauss_out = gauss;
  return gauss + constraint_update(om, d, d->efc_jar, d->efc_force, d->efc_state, 1);
}

static void solve_constraints(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  d->solver_iter = 0;
  if (d->nefc == 0) {
    o_copy(d->qacc, d->qacc_smooth, nv);
    o_zero(d->qfrc_constraint, nv);
    return;
  }
  /* warm start: better of qacc_warmstart and qacc_smooth */
  double cost_ws = total_cost(om, d, d->qacc_warmstart, NULL);
  double cost_sm = total_cost(om, d, d->qacc_smooth, NULL);
  if (cost_ws > cost_sm) o_copy(d->qacc, d->qacc_smooth, nv); else o_copy(d->qacc, d->qacc_warmstart, nv);
  double gauss;
  double cost = total_cost(om, d, d->qacc, &gauss);
  newton_gradient(om, d);
  for (int i = 0; i < nv; i++) d->search[i] = -d->Mgrad[i];
  double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
  for (int iter = 0; iter < m->iterations; iter++) {
    double alpha = line_search(om, d, gauss, cost);
    if (alpha == 0) break;
    for (int i = 0; i < nv; i++) d->qacc[i] += alpha * d->search[i];
    double oldcost = cost;
    cost = total_cost(om, d, d->qacc, &gauss);   /* recomputes Ma, jar exactly */
    newton_gradient(om, d);
    d->solver_iter++;
    double improvement = scale * (oldcost - cost);
    double gradient = scale * o_norm(d->grad, nv);
    if (getenv("ORACLE_DEBUG_SOLVER")) fprintf(stderr, "  iter %d alpha %.6g cost %.12g improvement %.3g gradient %.3g\n", iter, alpha, cost, improvement, gradient);
    if (improvement < m->tolerance || gradient < m->tolerance) break;
    for (int i = 0; i < nv; i++) d->search[i] = -d->Mgrad[i];
  }
  o_zero(d->qfrc_constraint, nv);
  for (int r = 0; r < d->nefc; r++) {
    double f = d->efc_force[r];
    if (f == 0) continue;
    for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[r * nv + i] * f;
  }
}

/* ---- mj_solNoSlip: friction-loss rows and the friction dimensions of the contacts re-solved in the dual without the
 * regulariser R (projected Gauss-Seidel over efc_force; limits, equalities and the contacts' normal forces stay as the Newton
 * solve left them), then qacc = qacc_smooth + M^-1 J^T force ------------------------------------------------------------- */
/* min 0.5 x'Ax + x'b  s.t.  sum (x_i / d_i)^2 <= r^2  (mju_QCQP2 / QCQP3 / QCQP): scaled to a ball, Newton on the multiplier */
static int qcqp2(double *res, const double *Ain, const double *bin, const double *d, double r) {
  double b1 = bin[0] * d[0], b2 = bin[1] * d[1];
  double A11 = Ain[0] * d[0] * d[0], A22 = Ain[3] * d[1] * d[1], A12 = Ain[1] * d[0] * d[1];
  double la = 0, v1 = 0, v2 = 0;
  for (int iter = 0; iter < 20; iter++) {
    double det = (A11 + la) * (A22 + la) - A12 * A12;
    if (det < 1e-10) { res[0] = 0; res[1] = 0; return 0; }
    double detinv = 1 / det;
    double P11 = (A22 + la) * detinv, P22 = (A11 + la) * detinv, P12 = -A12 * detinv;
    v1 = -P11 * b1 - P12 * b2; v2 = -P12 * b1 - P22 * b2;
    double val = v1 * v1 + v2 * v2 - r * r;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + 2 * P12 * v1 * v2 + P22 * v2 * v2);
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d[0]; res[1] = v2 * d[1];
  return la != 0;
}
static int qcqp3(double *res, const double *Ain, const double *bin, const double *d, double r) {
  double b1 = bin[0] * d[0], b2 = bin[1] * d[1], b3 = bin[2] * d[2];
  double A11 = Ain[0] * d[0] * d[0], A22 = Ain[4] * d[1] * d[1], A33 = Ain[8] * d[2] * d[2];
  double A12 = Ain[1] * d[0] * d[1], A13 = Ain[2] * d[0] * d[2], A23 = Ain[5] * d[1] * d[2];
  double la = 0, v1 = 0, v2 = 0, v3 = 0;
  for (int iter = 0; iter < 20; iter++) {
    double P11 = (A22 + la) * (A33 + la) - A23 * A23, P22 = (A11 + la) * (A33 + la) - A13 * A13, P33 = (A11 + la) * (A22 + la) - A12 * A12;
    double P12 = A13 * A23 - A12 * (A33 + la), P13 = A12 * A23 - A13 * (A22 + la), P23 = A12 * A13 - A23 * (A11 + la);
    double det = (A11 + la) * P11 + A12 * P12 + A13 * P13;
    if (det < 1e-10) { res[0] = res[1] = res[2] = 0; return 0; }
    double detinv = 1 / det;
    P11 *= detinv; P22 *= detinv; P33 *= detinv; P12 *= detinv; P13 *= detinv; P23 *= detinv;
    v1 = -P11 * b1 - P12 * b2 - P13 * b3; v2 = -P12 * b1 - P22 * b2 - P23 * b3; v3 = -P13 * b1 - P23 * b2 - P33 * b3;
    double val = v1 * v1 + v2 * v2 + v3 * v3 - r * r;
    if (val < 1e-10) break;
    double deriv = -2 * (P11 * v1 * v1 + P22 * v2 * v2 + P33 * v3 * v3) - 4 * (P12 * v1 * v2 + P13 * v1 * v3 + P23 * v2 * v3);
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  res[0] = v1 * d[0]; res[1] = v2 * d[1]; res[2] = v3 * d[2];
  return la != 0;
}
/* mju_cholFactor with a rank threshold (diagonal below mindiag: set to it, rank - 1) / mju_cholSolve, n <= 5 */
static int small_chol(double *A, int n, double mindiag) {
  int rank = n;
  for (int j = 0; j < n; j++) {
    double t = A[j * n + j];
    for (int k = 0; k < j; k++) t -= A[j * n + k] * A[j * n + k];
    if (t < mindiag) { t = mindiag; rank--; }
    A[j * n + j] = sqrt(t);
    double inv = 1 / A[j * n + j];
    for (int i = j + 1; i < n; i++) {
      double s = A[i * n + j];
      for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s * inv;
    }
  }
  return rank;
}
static void small_chol_solve(double *x, const double *L, const double *b, int n) {
  for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
  for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k]; x[i] = s / L[i * n + i]; }
}
static int qcqpn(double *res, const double *Ain, const double *bin, const double *d, double r, int n) {
  double A[25], Ala[25], b[5], tmp[5], la = 0;
  for (int i = 0; i < n; i++) { b[i] = bin[i] * d[i]; for (int j = 0; j < n; j++) A[j + i * n] = Ain[j + i * n] * d[i] * d[j]; }
  for (int iter = 0; iter < 20; iter++) {
    for (int i = 0; i < n * n; i++) Ala[i] = A[i];
    for (int i = 0; i < n; i++) Ala[i * (n + 1)] += la;
    if (small_chol(Ala, n, 1e-10) < n) { for (int i = 0; i < n; i++) res[i] = 0; return 0; }
    small_chol_solve(res, Ala, b, n);
    for (int i = 0; i < n; i++) res[i] = -res[i];
    double val = o_dot(res, res, n) - r * r;
    if (val < 1e-10) break;
    small_chol_solve(tmp, Ala, res, n);
    double deriv = -2 * o_dot(res, tmp, n);
    double delta = -val / deriv;
    if (delta < 1e-10) break;
    la += delta;
  }
  for (int i = 0; i < n; i++) res[i] = res[i] * d[i];
  return la != 0;
}
/* cost change of a block update; an update that raises the cost is taken back (costChange) */
static double noslip_cost_change(const double *A, double *force, const double *oldforce, const double *res, int dim) {
  double delta[6], change = 0;
  for (int j = 0; j < dim; j++) delta[j] = force[j] - oldforce[j];
  for (int j = 0; j < dim; j++) { double t = 0; for (int k = 0; k < dim; k++) t += A[j * dim + k] * delta[k]; change += 0.5 * delta[j] * t; }
  for (int j = 0; j < dim; j++) change += delta[j] * res[j];
  if (change > 1e-10) { for (int j = 0; j < dim; j++) force[j] = oldforce[j]; change = 0; }
  return change;
}
static void noslip(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  const int nv = m->nv, nefc = d->nefc;
  double *AR = d->efc_AR, *b = d->efc_b, *force = d->efc_force;
  d->noslip_iter = 0;
  if (nefc == 0) return;
  /* mj_projectConstraint: AR = J M^-1 J^T + diag(R);  b = J qacc_smooth - aref */
  for (int r = 0; r < nefc; r++) chol_solve(d->efc_MiJT + r * nv, d->qL, d->efc_J + r * nv, nv);
  for (int r = 0; r < nefc; r++) {
    for (int q = 0; q < nefc; q++) AR[r * nefc + q] = o_dot(d->efc_J + r * nv, d->efc_MiJT + q * nv, nv);
    AR[r * nefc + r] += d->efc_R[r];
    b[r] = o_dot(d->efc_J + r * nv, d->qacc_smooth, nv) - d->efc_aref[r];
  }
  const double scale = 1.0 / (m->meaninertia * (nv > 1 ? nv : 1));
#define NS_RES(res_, i_, dim_) for (int j_ = 0; j_ < (dim_); j_++) (res_)[j_] = o_dot(AR + ((i_) + j_) * nefc, force, nefc) + b[(i_) + j_] - d->efc_R[(i_) + j_] * force[(i_) + j_]
#define NS_BLOCK(Ac_, i_, dim_) do { for (int j_ = 0; j_ < (dim_); j_++) for (int k_ = 0; k_ < (dim_); k_++) (Ac_)[j_ * (dim_) + k_] = AR[((i_) + j_) * nefc + (i_) + k_]; \
    for (int j_ = 0; j_ < (dim_); j_++) { (Ac_)[j_ * ((dim_) + 1)] -= d->efc_R[(i_) + j_]; (Ac_)[j_ * ((dim_) + 1)] = fmax(1e-10, (Ac_)[j_ * ((dim_) + 1)]); } } while (0)
  int iter = 0;
  while (iter < m->noslip_iterations) {
    double improvement = 0;
    if (iter == 0) for (int i = 0; i < nefc; i++) improvement += 0.5 * force[i] * force[i] * d->efc_R[i];
    /* dry friction (dof rows, then tendon rows: MuJoCo's row order) */
    for (int pass = 0; pass < 2; pass++)
      for (int i = 0; i < nefc; i++) {
        if (d->efc_type[i] != (pass == 0 ? O_CNSTR_FRICTION_DOF : O_CNSTR_FRICTION_TENDON)) continue;
        double res[1], old = force[i], arinv = 1 / (AR[i * (nefc + 1)] - d->efc_R[i]), fl = d->efc_frictionloss[i];
        NS_RES(res, i, 1);
        force[i] -= res[0] * arinv;
        if (force[i] < -fl) force[i] = -fl; else if (force[i] > fl) force[i] = fl;
        double delta = force[i] - old;
        improvement -= 0.5 * delta * delta / arinv + delta * res[0];
      }
    /* contact friction */
    for (int i = 0; i < nefc; i++) {
      if (d->efc_type[i] == O_CNSTR_CONTACT_PYRAMIDAL) {
        const OContact *con = d->contact + d->efc_id[i];
        int dim = con->dim;
        for (int j = i; j < i + 2 * (dim - 1); j += 2) {
          double res[2], old[2] = {force[j], force[j + 1]}, Ac[4];
          NS_RES(res, j, 2);
          double mid = 0.5 * (force[j] + force[j + 1]);
          NS_BLOCK(Ac, j, 2);
          /* along f = (mid + y, mid - y) the cost is 0.5 K1 y^2 + K0 y + const, with bc = res - Ac f_old the part of the residual that does not move */
          double bc0 = res[0] - Ac[0] * old[0] - Ac[1] * old[1], bc1 = res[1] - Ac[2] * old[0] - Ac[3] * old[1];
          double K1 = Ac[0] + Ac[3] - Ac[1] - Ac[2], K0 = mid * (Ac[0] - Ac[3]) + bc0 - bc1;
          if (K1 < O_MINVAL) force[j] = force[j + 1] = mid;
          else {
            double y = -K0 / K1;
            if (y < -mid) { force[j] = 0; force[j + 1] = 2 * mid; }
            else if (y > mid) { force[j] = 2 * mid; force[j + 1] = 0; }
            else { force[j] = mid + y; force[j + 1] = mid - y; }
          }
          improvement -= noslip_cost_change(Ac, force + j, old, res, 2);
        }
        i += 2 * (dim - 1) - 1;
      } else if (d->efc_type[i] == O_CNSTR_CONTACT_ELLIPTIC) {
        const OContact *con = d->contact + d->efc_id[i];
        int dim = con->dim;
        if (dim == 1) continue;
        double res[6], old[6], Ac[36];
        NS_RES(res, i, dim);
        for (int j = 0; j < dim; j++) old[j] = force[i + j];
        NS_BLOCK(Ac, i, dim);
        if (force[i] < O_MINVAL) { for (int j = 1; j < dim; j++) force[i + j] = 0; }
        else {
          double bc[5], Af[25], v[5];
          for (int j = 0; j < dim - 1; j++) {
            bc[j] = res[j + 1];
            for (int k = 0; k < dim - 1; k++) { Af[j * (dim - 1) + k] = Ac[(j + 1) * dim + (k + 1)]; bc[j] -= Ac[(j + 1) * dim + (k + 1)] * old[k + 1]; }
          }
          if (dim == 3) qcqp2(v, Af, bc, con->friction, force[i]);
          else if (dim == 4) qcqp3(v, Af, bc, con->friction, force[i]);
          else qcqpn(v, Af, bc, con->friction, force[i], dim - 1);
          for (int j = 0; j < dim - 1; j++) force[i + 1 + j] = v[j];
        }
        improvement -= noslip_cost_change(Ac, force + i, old, res, dim);
        i += dim - 1;
      }
    }
    improvement *= scale;
    iter++;
    if (improvement < m->noslip_tolerance) break;
  }
#undef NS_RES
#undef NS_BLOCK
  d->noslip_iter = iter;
  /* dualFinish: qfrc_constraint = J^T force, qacc = qacc_smooth + M^-1 qfrc_constraint */
  o_zero(d->qfrc_constraint, nv);
  for (int r = 0; r < nefc; r++) { double f = force[r]; if (f == 0) continue; for (int i = 0; i < nv; i++) d->qfrc_constraint[i] += d->efc_J[r * nv + i] * f; }
  chol_solve(d->qacc, d->qL, d->qfrc_constraint, nv);
  for (int i = 0; i < nv; i++) d->qacc[i] += d->qacc_smooth[i];
}

/* ---- pipeline ------------------------------------------------------------------------ */
static int bad(const double *x, int n) {
  for (int i = 0; i < n; i++) if (!(x[i] == x[i]) || x[i] > 1e10 || x[i] < -1e10) return 1;
  return 0;
}

void oracle_forward(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  kinematics(om, d);
  com_pos(om, d);
  crb_and_factor(om, d);
  collision(om, d);
  make_constraint(om, d);
  com_vel(om, d);
  passive(om, d);
  make_impedance(om, d);
  rne_bias(om, d);
  actuation(om, d);
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  if (d->xfrc_on) {
    /* mj_xfrcAccumulate: J^T [force; torque] with the force applied at the body's inertial frame origin (xipos) */
    for (int b = 1; b < m->nbody; b++) {
      const double *f = d->xfrc_applied + 6 * b, *tq = f + 3;
      double off[3];
      o_sub3(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
      for (int a = b; a > 0; a = m->body_parentid[a]) {
        for (int k = 0; k < m->body_dofnum[a]; k++) {
          int dof = m->body_dofadr[a] + k;
          const double *cd = d->cdof + 6 * dof;
          double t[3], jp[3];
          o_cross(t, cd, off);
          jp[0] = cd[3] + t[0]; jp[1] = cd[4] + t[1]; jp[2] = cd[5] + t[2];
          d->qfrc_smooth[dof] += jp[0] * f[0] + jp[1] * f[1] + jp[2] * f[2] + cd[0] * tq[0] + cd[1] * tq[1] + cd[2] * tq[2];
        }
      }
    }
  }
  chol_solve(d->qacc_smooth, d->qL, d->qfrc_smooth, nv);
  solve_constraints(om, d);
  o_copy(d->qacc_newton, d->qacc, nv);           /* mj_fwdConstraint saves the warm start before the noslip pass */
  if (m->noslip_iterations > 0) noslip(om, d);
  oracle_residual(om, d, d->sensordata);
}

static void integrate_pos(const OModel *om, OData *d, double h) {
  const MjpcHipModel *m = &om->m;
  for (int j = 0; j < m->njnt; j++) {
    int qa = m->jnt_qposadr[j], da = m->jnt_dofadr[j];
    switch (m->jnt_type[j]) {
      case MJPC_JNT_FREE:
        for (int k = 0; k < 3; k++) d->qpos[qa + k] += h * d->qvel[da + k];
        o_quatintegrate(d->qpos + qa + 3, d->qvel + da + 3, h);
        break;
      case MJPC_JNT_BALL:
        o_quatintegrate(d->qpos + qa, d->qvel + da, h);
        break;
      default:
        d->qpos[qa] += h * d->qvel[da];
    }
  }
}


/* ---- velocity derivatives for the implicit integrators (mjd_smooth_vel, engine_derivative.c) ---------------------------------- */
/* A += h * (-d qfrc_fluid / d qvel) for the inertia-box fluid model (mjd_inertiaBoxFluid): per body the local 6-D force is
 * component-wise  lfrc_k = -(visc_k + quad_k |lvel_k|) lvel_k,  so  d lfrc_k / d lvel_k = -(visc_k + 2 quad_k |lvel_k|);  with
 * lvel = Rt [jacr; jacp] qvel at the com the generalized derivative is  -J^T R diag(...) R^T J  (symmetric). */
static void fluid_vel_derivative(const OModel *om, OData *d, double *A, double h) {
  const MjpcHipModel *m = &om->m;
  const int nv = m->nv;
  if (!(m->density > 0 || m->viscosity > 0)) return;
  double *jp = (double *)malloc(sizeof(double) * 12 * (size_t)nv), *jr = jp + 3 * nv, *lj = jr + 3 * nv;      /* lj: local rows [6][nv] */
  for (int b = 1; b < m->nbody; b++) if (m->body_mass[b] >= O_MINVAL) {
    const double *I = m->body_inertia + 3 * b, *R = d->ximat + 9 * b;
    double mass = m->body_mass[b], box[3], off[3], vw[3], lvel[6], dl[6] = {0, 0, 0, 0, 0, 0};
    box[0] = sqrt(fmax(O_MINVAL, I[1] + I[2] - I[0]) / mass * 6.0);
    box[1] = sqrt(fmax(O_MINVAL, I[0] + I[2] - I[1]) / mass * 6.0);
    box[2] = sqrt(fmax(O_MINVAL, I[0] + I[1] - I[2]) / mass * 6.0);
    o_sub3(off, d->xipos + 3 * b, d->subtree_com + 3 * m->body_rootid[b]);
    o_cross(vw, d->cvel + 6 * b, off); o_add3(vw, vw, d->cvel + 6 * b + 3);
    o_sub3(vw, vw, m->wind);
    o_mulmattvec3(lvel, R, d->cvel + 6 * b); o_mulmattvec3(lvel + 3, R, vw);
    if (m->viscosity > 0) {
      double diam = (box[0] + box[1] + box[2]) / 3.0;
      for (int k = 0; k < 3; k++) { dl[k] += O_PI * diam * diam * diam * m->viscosity; dl[3 + k] += 3.0 * O_PI * diam * m->viscosity; }
    }
    if (m->density > 0) {
      dl[3] += 2 * 0.5 * m->density * box[1] * box[2] * fabs(lvel[3]);
      dl[4] += 2 * 0.5 * m->density * box[0] * box[2] * fabs(lvel[4]);
      dl[5] += 2 * 0.5 * m->density * box[0] * box[1] * fabs(lvel[5]);
      dl[0] += 2 * m->density * box[0] * (box[1] * box[1] * box[1] * box[1] + box[2] * box[2] * box[2] * box[2]) * fabs(lvel[0]) / 64.0;
      dl[1] += 2 * m->density * box[1] * (box[0] * box[0] * box[0] * box[0] + box[2] * box[2] * box[2] * box[2]) * fabs(lvel[1]) / 64.0;
      dl[2] += 2 * m->density * box[2] * (box[0] * box[0] * box[0] * box[0] + box[1] * box[1] * box[1] * box[1]) * fabs(lvel[2]) / 64.0;
    }
    jac_point(om, d, jp, jr, d->xipos + 3 * b, b);
    for (int i = 0; i < nv; i++) {                     /* local-frame Jacobian rows: R^T jacr (0..2), R^T jacp (3..5) */
      double wr[3] = {jr[i], jr[nv + i], jr[2 * nv + i]}, wp[3] = {jp[i], jp[nv + i], jp[2 * nv + i]}, lr[3], lp[3];
      o_mulmattvec3(lr, R, wr); o_mulmattvec3(lp, R, wp);
      for (int k = 0; k < 3; k++) { lj[k * nv + i] = lr[k]; lj[(3 + k) * nv + i] = lp[k]; }
    }
    for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) {
      double s_ = 0;
      for (int k = 0; k < 6; k++) s_ += lj[k * nv + i] * dl[k] * lj[k * nv + j];
      A[i * nv + j] += h * s_;
    }
  }
  free(jp);
}
/* A += h * d qfrc_bias / d qvel (mjd_rne_vel; qDeriv -= that, and the integrator factors M - h qDeriv).  qfrc_bias is an exact
 * quadratic polynomial of qvel (cdof_dot is linear, cacc and cvel x* I cvel quadratic, gravity constant), so the central difference
 * with step 1 IS the derivative: [bias(v + e_j) - bias(v - e_j)] / 2 has no truncation error. */
static void rne_vel_derivative(const OModel *om, OData *d, double *A, double h) {
  const MjpcHipModel *m = &om->m;
  const int nv = m->nv, nb = m->nbody;
  size_t nsave = (size_t)(6 * nb * 3 + 6 * nv + nv + 3 * nb + nv);
  double *save = (double *)malloc(sizeof(double) * (nsave + 2 * (size_t)nv)), *q = save;
  double *s_cvel = q; q += 6 * nb; double *s_cacc = q; q += 6 * nb; double *s_cfrc = q; q += 6 * nb; double *s_cdd = q; q += 6 * nv;
  double *s_bias = q; q += nv; double *s_lin = q; q += 3 * nb; double *s_qvel = q; q += nv; double *plus = q; q += nv; double *minus = q;
  o_copy(s_cvel, d->cvel, 6 * nb); o_copy(s_cacc, d->cacc, 6 * nb); o_copy(s_cfrc, d->cfrc, 6 * nb); o_copy(s_cdd, d->cdof_dot, 6 * nv);
  o_copy(s_bias, d->qfrc_bias, nv); o_copy(s_lin, d->subtree_linvel, 3 * nb); o_copy(s_qvel, d->qvel, nv);
  for (int j = 0; j < nv; j++) {
    d->qvel[j] = s_qvel[j] + 1.0; com_vel(om, d); rne_bias(om, d); o_copy(plus, d->qfrc_bias, nv);
    d->qvel[j] = s_qvel[j] - 1.0; com_vel(om, d); rne_bias(om, d); o_copy(minus, d->qfrc_bias, nv);
    d->qvel[j] = s_qvel[j];
    for (int i = 0; i < nv; i++) A[i * nv + j] += h * 0.5 * (plus[i] - minus[i]);
  }
  o_copy(d->cvel, s_cvel, 6 * nb); o_copy(d->cacc, s_cacc, 6 * nb); o_copy(d->cfrc, s_cfrc, 6 * nb); o_copy(d->cdof_dot, s_cdd, 6 * nv);
  o_copy(d->qfrc_bias, s_bias, nv); o_copy(d->subtree_linvel, s_lin, 3 * nb);
  free(save);
}
/* x = A^-1 b for a general (non-symmetric) A, LU without pivoting like mju_factorLUSparse (A = M + h (...) is diagonally dominant
 * enough: M dominates); A is overwritten */
static void lu_solve(double *A, double *x, const double *b, int n) {
  for (int k = 0; k < n; k++)
    for (int i = k + 1; i < n; i++) {
      double l = A[i * n + k] / A[k * n + k];
      A[i * n + k] = l;
      for (int j = k + 1; j < n; j++) A[i * n + j] -= l * A[k * n + j];
    }
  for (int i = 0; i < n; i++) { double s_ = b[i]; for (int j = 0; j < i; j++) s_ -= A[i * n + j] * x[j]; x[i] = s_; }
  for (int i = n - 1; i >= 0; i--) { double s_ = x[i]; for (int j = i + 1; j < n; j++) s_ -= A[i * n + j] * x[j]; x[i] = s_ / A[i * n + i]; }
}
/* debug accessor for the unit tests: d qfrc_bias / d qvel and -d qfrc_fluid / d qvel at (qpos, qvel), plus the two force vectors
 * themselves (qfrc_bias; qfrc_passive, which holds the fluid forces) so that a test can difference them */
int oracle_debug_vel_derivatives(const OModel *om, const double *qpos, const double *qvel, double *dbias, double *dfluid, double *bias,
                                 double *passive_out) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  OData *d = oracle_make_data(om);
  o_copy(d->qpos, qpos, m->nq); o_copy(d->qvel, qvel, nv);
  oracle_forward(om, d);
  int w = d->warning;
  o_zero(dbias, nv * nv); o_zero(dfluid, nv * nv);
  rne_vel_derivative(om, d, dbias, 1.0);
  fluid_vel_derivative(om, d, dfluid, 1.0);
  if (bias) o_copy(bias, d->qfrc_bias, nv);
  if (passive_out) o_copy(passive_out, d->qfrc_passive, nv);
  oracle_free_data(d);
  return w;
}

/* actuator forces and their generalized force at (qpos, qvel, ctrl, act).  Test hook (transmission lengths / moments) */
int oracle_debug_actuation(const OModel *om, const double *qpos, const double *qvel, const double *ctrl, const double *act, double *force, double *qfrc) {
  const MjpcHipModel *m = &om->m;
  OData *d = oracle_make_data(om);
  o_copy(d->qpos, qpos, m->nq); o_copy(d->qvel, qvel, m->nv); o_copy(d->ctrl, ctrl, m->nu);
  if (act && m->na > 0) o_copy(d->act, act, m->na);
  oracle_forward(om, d);
  o_copy(force, d->actuator_force, m->nu); o_copy(qfrc, d->qfrc_actuator, m->nv);
  int w = d->warning;
  oracle_free_data(d);
  return w;
}

/* the constraint rows at (qpos, qvel): efc_J [nefc x nv], efc_pos, efc_diagApprox, efc_R, efc_aref (each up to `cap` rows); returns
 * nefc.  Test hook (finite-difference checks of the row Jacobians) */
int oracle_debug_constraints(const OModel *om, const double *qpos, const double *qvel, const double *mocap, int cap, double *J, double *pos,
                             double *diag, double *R, double *aref) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  OData *d = oracle_make_data(om);
  o_copy(d->qpos, qpos, m->nq); o_copy(d->qvel, qvel, nv);
  if (mocap) for (int i = 0; i < m->nmocap; i++) { o_copy3(d->mocap_pos + 3 * i, mocap + 7 * i); o_copy(d->mocap_quat + 4 * i, mocap + 7 * i + 3, 4); }
  oracle_forward(om, d);
  int n = d->nefc < cap ? d->nefc : cap;
  o_copy(J, d->efc_J, n * nv); o_copy(pos, d->efc_pos, n); o_copy(diag, d->efc_diagApprox, n); o_copy(R, d->efc_R, n); o_copy(aref, d->efc_aref, n);
  n = d->nefc;
  oracle_free_data(d);
  return n;
}

void oracle_step(const OModel *om, OData *d) {
  const MjpcHipModel *m = &om->m;
  int nv = m->nv;
  double h = m->timestep;
  if (bad(d->qpos, m->nq)) d->warning |= MJPC_WARN_BADQPOS;
  if (bad(d->qvel, nv)) d->warning |= MJPC_WARN_BADQVEL;
  if (d->warning) return;
  oracle_forward(om, d);
  if (d->warning & (MJPC_WARN_CONTACTFULL | MJPC_WARN_CNSTRFULL)) return;      /* the overflow is the failure code; the truncated solve does not matter */
  if (bad(d->qacc, nv)) { d->warning |= MJPC_WARN_BADQACC; return; }
  /* Euler, implicit in joint damping */
  int damped = 0;
  for (int i = 0; i < nv; i++) if (m->dof_damping[i] > 0) damped = 1;
  const int full = m->integrator == MJPC_INT_IMPLICIT;
  const int fast = m->integrator == MJPC_INT_IMPLICITFAST || full;
  if (fast) {
    if (m->density > 0 || m->viscosity > 0 || full) damped = 1;
    for (int t = 0; t < m->ntendon; t++) if (m->tendon_damping && m->tendon_damping[t] != 0) damped = 1;
    for (int i = 0; i < m->nu; i++) if (m->actuator_biastype[i] == MJPC_BIAS_AFFINE && m->actuator_biasprm[3 * i + 2] != 0) damped = 1;
  }
  double *qacc = d->qacc;
  if (damped) {
    o_copy(d->qH, d->qM, nv * nv);
    for (int i = 0; i < nv; i++) d->qH[i * nv + i] += h * m->dof_damping[i];
    if (fast) {
      /* mjINT_IMPLICITFAST: M - h dF/dv with the velocity derivatives of the passive and actuator forces (mjd_smooth_vel without the
       * Coriolis term): tendon damping  -b J^T J,  affine actuator bias  prm2 moment^T moment (zero while the force is clamped) */
      double *row = d->work + 2 * nv;
      for (int t = 0; t < m->ntendon; t++) {
        double b = m->tendon_damping ? m->tendon_damping[t] : 0;
        if (b == 0) continue;
        o_zero(row, nv);
        for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) row[m->jnt_dofadr[m->wrap_objid[w]]] += m->wrap_prm[w];
        for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) d->qH[i * nv + j] += h * b * row[i] * row[j];
      }
      for (int a = 0; a < m->nu; a++) {
        double kv = m->actuator_biastype[a] == MJPC_BIAS_AFFINE ? m->actuator_biasprm[3 * a + 2] : 0;
        if (kv == 0) continue;
        if (m->actuator_trntype[a] == MJPC_TRN_SITE) { d->unsupported++; d->warning |= MJPC_WARN_UNSUPPORTED; continue; }      /* (refused at create by the engine) */
        if (m->actuator_forcelimited[a] && (d->actuator_force[a] <= m->actuator_forcerange[2 * a] || d->actuator_force[a] >= m->actuator_forcerange[2 * a + 1])) continue;
        double gear = m->actuator_gear[a];
        o_zero(row, nv);
        if (m->actuator_trntype[a] == MJPC_TRN_TENDON) {
          int t = m->actuator_trnid[a];
          for (int w = m->tendon_adr[t]; w < m->tendon_adr[t] + m->tendon_num[t]; w++) row[m->jnt_dofadr[m->wrap_objid[w]]] += gear * m->wrap_prm[w];
        } else row[m->jnt_dofadr[m->actuator_trnid[a]]] += gear;
        for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) d->qH[i * nv + j] -= h * kv * row[i] * row[j];
      }
    }
    if (fast) fluid_vel_derivative(om, d, d->qH, h);             /* mjd_passive_vel: inertia-box fluid forces */
    double *rhs = d->work, *sol = d->work + nv;
    for (int i = 0; i < nv; i++) rhs[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    if (full) {
      /* mjINT_IMPLICIT: also the (non-symmetric) velocity derivative of the bias forces, then LU */
      rne_vel_derivative(om, d, d->qH, h);
      lu_solve(d->qH, sol, rhs, nv);
    } else {
      chol_factor(d->qLD2, d->qH, nv);
      chol_solve(sol, d->qLD2, rhs, nv);
    }
    qacc = sol;
  }
  /* mj_advance: activations first (mj_nextActivation), then velocities and positions */
  if (m->na > 0 && m->actuator_dyntype)
    for (int i = 0; i < m->nu; i++) if (m->actuator_dyntype[i] != MJPC_DYN_NONE) {
      int a = m->actuator_actadr[i];
      double act = d->act[a];
      if (m->actuator_dyntype[i] == MJPC_DYN_FILTEREXACT) {
        double tau = fmax(O_MINVAL, m->actuator_dynprm[i]);
        act += d->act_dot[a] * tau * (1 - exp(-h / tau));
      } else act += h * d->act_dot[a];
      if (m->actuator_actlimited[i]) act = o_clip(act, m->actuator_actrange[2 * i], m->actuator_actrange[2 * i + 1]);
      d->act[a] = act;
    }
  for (int i = 0; i < nv; i++) d->qvel[i] += h * qacc[i];
  integrate_pos(om, d, h);
  d->time += h;
  o_copy(d->qacc_warmstart, d->qacc_newton, nv);
}

/* ---- debug accessors for unit tests --------------------------------------------------- */
int oracle_debug_forward(const OModel *om, const double *qpos, const double *qvel,
                         const double *ctrl, const double *mocap, double time,
                         double *qacc, double *qM, double *xpos, double *sensordata,
                         double *contact_dist, int *ncon, int *nefc, double *efc_force,
                         double *geom_xpos, double *extra) {
  const MjpcHipModel *m = &om->m;
  OData *d = oracle_make_data(om);
  o_copy(d->qpos, qpos, m->nq);
  if (qvel) o_copy(d->qvel, qvel, m->nv);
  if (ctrl) o_copy(d->ctrl, ctrl, m->nu);
  if (mocap) for (int i = 0; i < m->nmocap; i++) { o_copy3(d->mocap_pos + 3 * i, mocap + 7 * i); o_copy(d->mocap_quat + 4 * i, mocap + 7 * i + 3, 4); }
  d->time = time;
  oracle_forward(om, d);
  if (qacc) o_copy(qacc, d->qacc, m->nv);
  if (qM) o_copy(qM, d->qM, m->nv * m->nv);
  if (xpos) o_copy(xpos, d->xpos, 3 * m->nbody);
  if (sensordata) o_copy(sensordata, d->sensordata, om->t.num_residual);
  if (contact_dist) for (int i = 0; i < d->ncon; i++) contact_dist[i] = d->contact[i].dist;
  if (ncon) *ncon = d->ncon;
  if (nefc) *nefc = d->nefc;
  if (efc_force) o_copy(efc_force, d->efc_force, d->nefc);
  if (geom_xpos) o_copy(geom_xpos, d->geom_xpos, 3 * m->ngeom);
  if (extra) {   /* [0:nv] qacc_smooth, [nv:2nv] qfrc_bias, [2nv:3nv] qfrc_constraint, then solver_iter, unsupported, 3*nbody subtree_com, 3*nbody subtree_linvel */
    int nv = m->nv;
    o_copy(extra, d->qacc_smooth, nv); o_copy(extra + nv, d->qfrc_bias, nv); o_copy(extra + 2 * nv, d->qfrc_constraint, nv);
    extra[3 * nv] = d->solver_iter; extra[3 * nv + 1] = d->unsupported;
    o_copy(extra + 3 * nv + 2, d->subtree_com, 3 * m->nbody);
    o_copy(extra + 3 * nv + 2 + 3 * m->nbody, d->subtree_linvel, 3 * m->nbody);
  }
  int w = d->warning;
  oracle_free_data(d);
  return w;
}

int oracle_debug_step(const OModel *om, double *qpos, double *qvel, const double *ctrl,
                      const double *mocap, double *time, int nstep, double *energy) {
  const MjpcHipModel *m = &om->m;
  OData *d = oracle_make_data(om);
  o_copy(d->qpos, qpos, m->nq); o_copy(d->qvel, qvel, m->nv);
  if (ctrl) o_copy(d->ctrl, ctrl, m->nu);
  if (mocap) for (int i = 0; i < m->nmocap; i++) { o_copy3(d->mocap_pos + 3 * i, mocap + 7 * i); o_copy(d->mocap_quat + 4 * i, mocap + 7 * i + 3, 4); }
  d->time = *time;
  for (int s = 0; s < nstep && !d->warning; s++) {
    oracle_step(om, d);
    if (energy) {   /* energy of the state the forward pass saw */
      int nv = m->nv;
      double ke = 0, pe = 0;
      for (int i = 0; i < nv; i++) ke += 0.5 * d->qvel[i] * o_dot(d->qM + i * nv, d->qvel, nv);
      for (int b = 1; b < m->nbody; b++) pe -= m->body_mass[b] * o_dot3(m->gravity, d->xipos + 3 * b);
      energy[2 * s] = ke; energy[2 * s + 1] = pe;
    }
  }
  o_copy(qpos, d->qpos, m->nq); o_copy(qvel, d->qvel, m->nv);
  *time = d->time;
  int w = d->warning;
  oracle_free_data(d);
  return w;
}

/* debug: geom pairs of the contacts at a configuration (tests / analysis): out[2*k], out[2*k+1]; returns ncon */
int oracle_debug_contact_geoms(const OModel *om, const double *qpos, int *out, int cap) {
  OData *d = oracle_make_data(om);
  o_copy(d->qpos, qpos, om->m.nq);
  oracle_forward(om, d);
  int n = d->ncon;
  for (int k = 0; k < n && k < cap; k++) { out[2 * k] = d->contact[k].geom1; out[2 * k + 1] = d->contact[k].geom2; }
  oracle_free_data(d);
  return n;
}
