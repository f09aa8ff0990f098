// constraint.h — constraint rows: friction loss, limits, contacts, their Jacobians and impedances (part of core.h)
// Included by core.h only, in this order: the files share one translation unit and its macros.
#pragma once
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
      double value = c.qpos[MIH(jnt_qposadr)[j]], margin = MDH(jnt_margin)[j];
      for (int s = -1; s <= 1; s += 2) {
        double dd = s * (MDH(jnt_range)[2 * j + (s + 1) / 2] - value);
        if (dd < margin) { dist[cnt] = dd; side[cnt] = s; cnt++; }
      }
    }
    int tot, off = wave_excl_scan(cnt, &tot);
    if (nefc + tot > M.nefcmax) { c.warning |= WARN_CNSTRFULL; break; }
    for (int k = 0; k < cnt; k++) {
      int r = nefc + off + k;
      c.efc_type[r] = CNSTR_LIMIT_JOINT; c.efc_id[r] = j; c.efc_dof[r] = MIH(jnt_dofadr)[j];
      c.efc_floss[r] = (double)(-side[k]);      // J entry, consumed below
      c.efc_pos[r] = dist[k]; c.efc_margin[r] = MDH(jnt_margin)[j];
      c.efc_diag[r] = MD(dof_invweight0)[MIH(jnt_dofadr)[j]];
    }
    nefc += tot;
  }
  int nlim_end = nefc;
  // equality constraints (mj_instantiateEquality; MuJoCo lists them first: same constraint set): static general rows.  They ride
  // the friction-loss cost with an unreachable friction loss: |D jar| never gets there, so the row is quadratic on both sides
  if (M.neq) {
    if (nefc + M.neqrow > M.nefcmax) c.warning |= WARN_CNSTRFULL;
    else {
      PFOR(k, M.neq) {
        const int *et = MI(eq_tab) + 4 * k; const double *ep = MD(eq_prm) + 18 * k;
        int r0 = nefc + et[3], nr = 1;
        double pos[6] = {0, 0, 0, 0, 0, 0}, diag, diag_rot = 0;
        if (et[0] == 1) {            // weld: anchor offset (body 1 holds ep[3..5], body 2 the anchor ep[0..2]) and torquescale * vec(conj(q2) q1 relpose)
          int b1 = et[1], b2 = et[2];
          double p1[3], p2[3], quat[4], quat1[4], quat2[4];
          d_mulmatvec3(p1, c.xmat + 9 * b1, ep + 3); d_mulmatvec3(p2, c.xmat + 9 * b2, ep);
          for (int q = 0; q < 3; q++) pos[q] = (p1[q] + c.xpos[3 * b1 + q]) - (p2[q] + c.xpos[3 * b2 + q]);
          d_mulquat(quat, c.xquat + 4 * b1, ep + 6);
          quat1[0] = c.xquat[4 * b2]; for (int q = 1; q < 4; q++) quat1[q] = -c.xquat[4 * b2 + q];
          d_mulquat(quat2, quat1, quat);
          for (int q = 0; q < 3; q++) pos[3 + q] = ep[10] * quat2[1 + q];
          diag = MDH(body_invweight0)[2 * b1] + MDH(body_invweight0)[2 * b2];
          diag_rot = MDH(body_invweight0)[2 * b1 + 1] + MDH(body_invweight0)[2 * b2 + 1];
          nr = 6;
        } else if (et[0] == 2) {     // joint: (q1 - q1_0) - poly(q2 - q2_0)
          int q1 = MIH(jnt_qposadr)[et[1]];
          pos[0] = c.qpos[q1] - MDH(qpos0)[q1];
          diag = MD(dof_invweight0)[MIH(jnt_dofadr)[et[1]]];
          if (et[2] >= 0) {
            int q2 = MIH(jnt_qposadr)[et[2]];
            double dif = c.qpos[q2] - MDH(qpos0)[q2];
            pos[0] -= ep[0] + dif * (ep[1] + dif * (ep[2] + dif * (ep[3] + dif * ep[4])));
            diag += MD(dof_invweight0)[MIH(jnt_dofadr)[et[2]]];
          } else pos[0] -= ep[0];
        } else if (et[0] == 3) {     // tendon: (L1 - L1_0) - poly(L2 - L2_0), fixed-tendon lengths relative to qpos0
          int t1 = et[1], t2 = et[2];
          for (int w = MI(tendon_adr)[t1]; w < MI(tendon_adr)[t1] + MI(tendon_num)[t1]; w++) { int qa = MI(wrap_qposadr)[w]; pos[0] += MD(wrap_prm)[w] * (c.qpos[qa] - MDH(qpos0)[qa]); }
          diag = MD(tendon_invweight0)[t1];
          if (t2 >= 0) {
            double dif = 0;
            for (int w = MI(tendon_adr)[t2]; w < MI(tendon_adr)[t2] + MI(tendon_num)[t2]; w++) { int qa = MI(wrap_qposadr)[w]; dif += MD(wrap_prm)[w] * (c.qpos[qa] - MDH(qpos0)[qa]); }
            pos[0] -= ep[0] + dif * (ep[1] + dif * (ep[2] + dif * (ep[3] + dif * ep[4])));
            diag += MD(tendon_invweight0)[t2];
          } else pos[0] -= ep[0];
        } else {                     // connect: anchor of body 1 - anchor of body 2, world frame
          int b1 = et[1], b2 = et[2];
          double p1[3], p2[3];
          d_mulmatvec3(p1, c.xmat + 9 * b1, ep); d_mulmatvec3(p2, c.xmat + 9 * b2, ep + 3);
          for (int q = 0; q < 3; q++) pos[q] = (p1[q] + c.xpos[3 * b1 + q]) - (p2[q] + c.xpos[3 * b2 + q]);
          diag = MDH(body_invweight0)[2 * b1] + MDH(body_invweight0)[2 * b2];
          nr = 3;
        }
        for (int q = 0; q < nr; q++) {
          int r = r0 + q;
          c.efc_type[r] = CNSTR_EQUALITY; c.efc_id[r] = k; c.efc_dof[r] = r0;
          c.efc_floss[r] = 1e300; c.efc_pos[r] = pos[q]; c.efc_margin[r] = 0; c.efc_diag[r] = q > 2 ? diag_rot : diag;
        }
      }
      nefc += M.neqrow;
    }
  }
  int neq_end = nefc;
  // tendon friction loss (mjCNSTR_FRICTION_TENDON): static general rows (MuJoCo lists them before the limits: same constraint set)
  if (M.ntfric) {
    if (nefc + M.ntfric > M.nefcmax) c.warning |= WARN_CNSTRFULL;
    else {
      PFOR(k, M.ntfric) {
        int r = nefc + k, t = MI(tfric_id)[k];
        c.efc_type[r] = CNSTR_FRICTION_TENDON; c.efc_id[r] = k; c.efc_dof[r] = 0;
        c.efc_floss[r] = MD(tfric_prm)[8 * k]; c.efc_pos[r] = 0; c.efc_margin[r] = 0;
        c.efc_diag[r] = MD(tendon_invweight0)[t];
      }
      nefc += M.ntfric;
    }
  }
  int ntf_end = nefc;
  // ball-joint limits (mj_instantiateLimit): rotation angle of the joint quaternion against max(range), J = -axis at its three
  // dofs: general rows, kept behind the single-entry rows (MuJoCo interleaves them in joint order; same constraint set)
  for (int base = 0; base < M.nlimit_ball; base += NLANE) {
    int q = base + LANE, cnt = 0, j = 0;
    double dist = 0;
    if (q < M.nlimit_ball) {
      j = MI(limit_ball)[q];
      double axis[3];
      double angle = ball_angle(axis, c.qpos + MIH(jnt_qposadr)[j]);
      dist = fmax(MDH(jnt_range)[2 * j], MDH(jnt_range)[2 * j + 1]) - angle;
      if (dist < MDH(jnt_margin)[j]) cnt = 1;
    }
    int tot, off = wave_excl_scan(cnt, &tot);
    if (nefc + tot > M.nefcmax) { c.warning |= WARN_CNSTRFULL; break; }
    if (cnt) {
      int r = nefc + off;
      c.efc_type[r] = CNSTR_LIMIT_JOINT; c.efc_id[r] = j; c.efc_dof[r] = MIH(jnt_dofadr)[j];
      c.efc_floss[r] = 0;
      c.efc_pos[r] = dist; c.efc_margin[r] = MDH(jnt_margin)[j];
      c.efc_diag[r] = MD(dof_invweight0)[MIH(jnt_dofadr)[j]];
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
    c.efc_J[r * nvp + MIH(jnt_dofadr)[c.efc_id[r]]] = c.efc_floss[r]; c.efc_floss[r] = 0;
  }
  if (neq_end > nlim_end) PFOR(k, M.neq) {
    const int *et = MI(eq_tab) + 4 * k; const double *ep = MD(eq_prm) + 18 * k;
    int r0 = nlim_end + et[3];
    if (et[0] == 2) {
      c.efc_J[r0 * nvp + MIH(jnt_dofadr)[et[1]]] = 1;
      if (et[2] >= 0) {
        int q2 = MIH(jnt_qposadr)[et[2]];
        double dif = c.qpos[q2] - MDH(qpos0)[q2];
        c.efc_J[r0 * nvp + MIH(jnt_dofadr)[et[2]]] = -(ep[1] + dif * (2 * ep[2] + dif * (3 * ep[3] + dif * 4 * ep[4])));
      }
    } else if (et[0] == 3) {
      int t1 = et[1], t2 = et[2];
      for (int w = MI(tendon_adr)[t1]; w < MI(tendon_adr)[t1] + MI(tendon_num)[t1]; w++) c.efc_J[r0 * nvp + MI(wrap_dofadr)[w]] += MD(wrap_prm)[w];
      if (t2 >= 0) {
        double dif = 0;
        for (int w = MI(tendon_adr)[t2]; w < MI(tendon_adr)[t2] + MI(tendon_num)[t2]; w++) { int qa = MI(wrap_qposadr)[w]; dif += MD(wrap_prm)[w] * (c.qpos[qa] - MDH(qpos0)[qa]); }
        double deriv = ep[1] + dif * (2 * ep[2] + dif * (3 * ep[3] + dif * 4 * ep[4]));
        for (int w = MI(tendon_adr)[t2]; w < MI(tendon_adr)[t2] + MI(tendon_num)[t2]; w++) c.efc_J[r0 * nvp + MI(wrap_dofadr)[w]] -= deriv * MD(wrap_prm)[w];
      }
    } else {       // point Jacobians of the two anchors (cdof about the root's subtree com); a weld adds the rotation rows
      int b1 = et[1], b2 = et[2];
      const int weld = et[0] == 1;
      double p1[3], p2[3], quat[4], quat1[4];
      d_mulmatvec3(p1, c.xmat + 9 * b1, weld ? ep + 3 : ep); d_mulmatvec3(p2, c.xmat + 9 * b2, weld ? ep : ep + 3);
      for (int q = 0; q < 3; q++) { p1[q] += c.xpos[3 * b1 + q]; p2[q] += c.xpos[3 * b2 + q]; }
      if (weld) {
        d_mulquat(quat, c.xquat + 4 * b1, ep + 6);
        quat1[0] = c.xquat[4 * b2]; for (int q = 1; q < 4; q++) quat1[q] = -c.xquat[4 * b2 + q];
      }
      for (int d = 0; d < nv; d++) {
        unsigned long long bit = 1ull << d;
        int in1 = (MDM()[b1] & bit) != 0, in2 = (MDM()[b2] & bit) != 0;
        if (!in1 && !in2) continue;
        const double *cd = c.cdof + 6 * d;
        double jp[3] = {0, 0, 0}, off[3], t[3];
        if (in1) { d_sub3(off, p1, c.subtree_com + 3 * MIH(body_rootid)[b1]); d_cross(t, cd, off); for (int q = 0; q < 3; q++) jp[q] += cd[3 + q] + t[q]; }
        if (in2) { d_sub3(off, p2, c.subtree_com + 3 * MIH(body_rootid)[b2]); d_cross(t, cd, off); for (int q = 0; q < 3; q++) jp[q] -= cd[3 + q] + t[q]; }
        for (int q = 0; q < 3; q++) c.efc_J[(r0 + q) * nvp + d] = jp[q];
        if (weld) {          // torquescale * 0.5 * vec(conj(q2) (0, w1 - w2) q1 relpose)
          double ax[4] = {0, 0, 0, 0}, tq[4], q3[4];
          if (in1) for (int q = 0; q < 3; q++) ax[1 + q] = cd[q];
          if (in2) for (int q = 0; q < 3; q++) ax[1 + q] -= cd[q];
          d_mulquat(tq, quat1, ax); d_mulquat(q3, tq, quat);
          for (int q = 0; q < 3; q++) c.efc_J[(r0 + 3 + q) * nvp + d] = ep[10] * (0.5 * q3[1 + q]);
        }
      }
    }
  }
  PFOR(rr, ntf_end - neq_end) {
    int r = neq_end + rr, t = MI(tfric_id)[c.efc_id[r]];
    for (int w = MI(tendon_adr)[t]; w < MI(tendon_adr)[t] + MI(tendon_num)[t]; w++) c.efc_J[r * nvp + MI(wrap_dofadr)[w]] = MD(wrap_prm)[w];
  }
  PFOR(rr, nball_end - ntf_end) {
    int r = ntf_end + rr, j = c.efc_id[r], da = MIH(jnt_dofadr)[j];
    double axis[3];
    ball_angle(axis, c.qpos + MIH(jnt_qposadr)[j]);
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
      double tran = MDH(body_invweight0)[2 * b1] + MDH(body_invweight0)[2 * b2];
      double rot = MDH(body_invweight0)[2 * b1 + 1] + MDH(body_invweight0)[2 * b2 + 1];
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
      d_sub3(off, cc + CON_POS, c.subtree_com + 3 * MIH(body_rootid)[b2]);
      d_cross(t, cd, off);
      jp[0] += cd[3] + t[0]; jp[1] += cd[4] + t[1]; jp[2] += cd[5] + t[2];
      jr[0] += cd[0]; jr[1] += cd[1]; jr[2] += cd[2];
    }
    if (in1) {
      double off[3], t[3];
      d_sub3(off, cc + CON_POS, c.subtree_com + 3 * MIH(body_rootid)[b1]);
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
    } else if (type == CNSTR_EQUALITY) {
      for (int k = 0; k < 2; k++) solref[k] = MD(eq_prm)[18 * id + 11 + k];
      for (int k = 0; k < 5; k++) solimp[k] = MD(eq_prm)[18 * id + 13 + k];
    } else if (type == CNSTR_FRICTION_TENDON) {
      for (int k = 0; k < 2; k++) solref[k] = MD(tfric_prm)[8 * id + 1 + k];
      for (int k = 0; k < 5; k++) solimp[k] = MD(tfric_prm)[8 * id + 3 + k];
    } else if (type == CNSTR_LIMIT_JOINT) {
      for (int k = 0; k < 2; k++) solref[k] = MDH(jnt_solref)[2 * id + k];
      for (int k = 0; k < 5; k++) solimp[k] = MDH(jnt_solimp)[5 * id + k];
    } else if (type == CNSTR_LIMIT_TENDON) {
      for (int k = 0; k < 2; k++) solref[k] = MD(tendon_solref_lim)[2 * id + k];
      for (int k = 0; k < 5; k++) solimp[k] = MD(tendon_solimp_lim)[5 * id + k];
    } else {
      const double *cc = c.contact + EFC_CON_CI(id) * c.M->con_stride;
      for (int k = 0; k < 2; k++) solref[k] = cc[CON_SOLREF + k];
      for (int k = 0; k < 5; k++) solimp[k] = cc[CON_SOLIMP + k];
      first = (r == EFC_CON_R0(id)) || type == CNSTR_CONTACT_PYRAMIDAL;
    }
    double imp_pos = c.efc_pos[r];
    if (type == CNSTR_EQUALITY && MI(eq_tab)[4 * id] == 0) {      // connect: one impedance from the norm of its three residuals
      const double *ps = c.efc_pos + c.efc_dof[r];
      imp_pos = d_sqrt(ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2]);
    }
    if (type == CNSTR_EQUALITY && MI(eq_tab)[4 * id] == 1) {      // weld: from the norm of all six
      const double *ps = c.efc_pos + c.efc_dof[r];
      imp_pos = d_sqrt(ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2] + ps[3] * ps[3] + ps[4] * ps[4] + ps[5] * ps[5]);
    }
    double imp = impedance(solimp, imp_pos, c.efc_margin[r]);
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
    if (type == CNSTR_FRICTION_DOF || type == CNSTR_FRICTION_TENDON || !first) K = 0;
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

