// kinematics.h — position stage of a step: joint / body / geom / site poses, com-based quantities, joint-space inertia (part of core.h)
// Included by core.h only, in this order: the files share one translation unit and its macros.
#pragma once
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
  int type = MIH(jnt_type)[j], qa = MIH(jnt_qposadr)[j];
  if (type == 3) {
    double ax[3];
    d_copy3(ax, MDH(jnt_axis) + 3 * j);
    d_axisangle2quat(jq + 4 * j, ax, c.qpos[qa] - MDH(qpos0)[qa]);
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
  int jntnum = MIH(body_jntnum)[i], jntadr = MIH(body_jntadr)[i];
  int mid = MIH(body_mocapid)[i];
  if (mid >= 0) {
    d_copy3(xpos, c.mocap_pos + 3 * mid);
    d_copy4(xquat, c.mocap_quat + 4 * mid);
    d_normalize4(xquat);
  } else if (jntnum == 1 && MIH(jnt_type)[jntadr] == 0) {
    int qa = MIH(jnt_qposadr)[jntadr];
    d_copy3(xpos, c.qpos + qa);
    d_copy4(xquat, c.qpos + qa + 3);              // normalised in place by kin_joint_local
    if (store) {
      d_copy3(c.xanchor + 3 * jntadr, xpos);
      d_copy3(c.xaxis + 3 * jntadr, MDH(jnt_axis) + 3 * jntadr);
    }
  } else {
    if (have_parent) {
      d_mulmatvec3(xpos, pmat, MDH(body_pos) + 3 * i);
      d_add3(xpos, xpos, ppos);
      d_mulquat(xquat, pquat, MDH(body_quat) + 4 * i);
    } else {
      d_copy3(xpos, MDH(body_pos) + 3 * i);
      d_copy4(xquat, MDH(body_quat) + 4 * i);
    }
    if (jntnum > 0) {
      double m[9];
      d_quat2mat(m, xquat);                       // rotation of the frame the next joint is expressed in
      for (int j = jntadr; j < jntadr + jntnum; j++) {
        int qa = MIH(jnt_qposadr)[j], type = MIH(jnt_type)[j];
        double vec[3], ax[3], jp[3], xax[3], xan[3];
        d_copy3(ax, MDH(jnt_axis) + 3 * j); d_copy3(jp, MDH(jnt_pos) + 3 * j);
        d_mulmatvec3(xax, m, ax);
        d_mulmatvec3(vec, m, jp);
        d_add3(xan, vec, xpos);
        if (store) { d_copy3(c.xaxis + 3 * j, xax); d_copy3(c.xanchor + 3 * j, xan); }
        if (type == 2) {
          d_addtoscl3(xpos, xax, c.qpos[qa] - MDH(qpos0)[qa]);
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
    d_copy3(ip, MDH(body_ipos) + 3 * i); d_copy4(iq, MDH(body_iquat) + 4 * i);
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
    for (int q = MIH(chain_adr)[b]; q < MIH(chain_adr)[b + 1]; q++) {
      int a = MIH(chain_list)[q];
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
    for (int k = MIH(subtree_adr)[b]; k < MIH(subtree_adr)[b + 1]; k++) {
      int cb = MIH(subtree_list)[k];
      d_addtoscl3(s, c.xipos + 3 * cb, MDH(body_mass)[cb]);
    }
    double sm = MDH(body_subtreemass)[b];
    if (sm < D_MINVAL) d_copy3(c.subtree_com + 3 * b, c.xipos + 3 * b);
    else d_scl3(c.subtree_com + 3 * b, s, 1.0 / sm);
  }
  SYNC();
  PFOR(b, M.nbody) {
    if (b == 0) { for (int k = 0; k < 10; k++) c.cinert[k] = 0; continue; }
    double off[3], ine[3], r[10];
    d_sub3(off, c.xipos + 3 * b, c.subtree_com + 3 * MIH(body_rootid)[b]);
    d_copy3(ine, MDH(body_inertia) + 3 * b);
    d_inertcom(r, ine, c.ximat + 9 * b, off, MDH(body_mass)[b]);
    for (int k = 0; k < 10; k++) c.cinert[10 * b + k] = r[k];
  }
  PFOR(j, M.njnt) {
    int b = MIH(jnt_bodyid)[j], da = MIH(jnt_dofadr)[j], type = MIH(jnt_type)[j];
    double off[3];
    d_sub3(off, c.subtree_com + 3 * MIH(body_rootid)[b], c.xanchor + 3 * j);
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
    if (b > 0) for (int q = MIH(subtree_adr)[b]; q < MIH(subtree_adr)[b + 1]; q++) s += c.cinert[10 * MIH(subtree_list)[q] + k];
    c.crb[e] = s;
  }
  SYNC();
  PROFW(c, 1);
  PFOR(p, M.nmpair) {
    int i = MI(mpair_i)[p], j = MI(mpair_j)[p];
    double buf[6];
    d_mulinertvec(buf, c.crb + 10 * MIH(dof_bodyid)[i], c.cdof + 6 * i);
    const double *cj = c.cdof + 6 * j;
    double v = cj[0]*buf[0] + cj[1]*buf[1] + cj[2]*buf[2] + cj[3]*buf[3] + cj[4]*buf[4] + cj[5]*buf[5];
    if (i == j) v += MDH(dof_armature)[i];
    c.qM[i * nvp + j] = v; c.qM[j * nvp + i] = v;
  }
  SYNC();
  PFOR(e, nv * nvp) c.qL[e] = c.qM[e];
  PROFW(c, 4);
  chol_factor<NVT>(c.qL, c.Linv, c.vtmp, nv, nvp, c.M->tree_ok);
  PROFW(c, 5);
}

