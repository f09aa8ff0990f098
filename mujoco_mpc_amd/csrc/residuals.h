// residuals.h — task residuals (device restatement of the reference's ResidualFn::Residual) and the cost (part of core.h)
// Included by core.h only, in this order: the files share one translation unit and its macros.
#pragma once
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
    d_sub3(off, c.site_xpos + 3 * sid, c.subtree_com + 3 * MIH(body_rootid)[body]);
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
  d_sub3(off, c.xipos + 3 * body, c.subtree_com + 3 * MIH(body_rootid)[body]);
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
// mjpc/tasks/humanoid/interact/interact.cc:31-186.  int_data = [body torso, pelvis, foot_right, foot_left, head, shin_right, shin_left,
// has facing target, (body1, body2) x 5]; dbl_data = [facing x, y, (local_pos1[3], local_pos2[3]) x 5]; 68 residuals
DEV void residual_humanoid_interact(Ctx &c, double *residual) {
  const DevModel &M = *c.M;
  const int *I = MI(task.int_data);
  const double *P = MD(task.parameters), *D = MD(task.dbl_data);
  const int nv = M.nv, nu = M.nu;
  const int torso = I[0], pelvis = I[1], fr = I[2], fl = I[3], head = I[4], kr = I[5], kl = I[6];
  if (LANE == 0) {
    residual[0] = fabs(c.xmat[9 * torso + 8] - 1.0);
    residual[1] = fabs(c.xmat[9 * pelvis + 8] - 1.0);
    residual[2] = fabs(c.xmat[9 * fr + 8] - 1.0);
    residual[3] = fabs(c.xmat[9 * fl + 8] - 1.0);
    residual[4] = fabs(c.xipos[3 * head + 2] - P[0]);
    residual[5] = fabs(c.xipos[3 * torso + 2] - P[1]);
    const double *knee_right = c.xipos + 3 * kr, *knee_left = c.xipos + 3 * kl, *foot_right = c.xipos + 3 * fr, *foot_left = c.xipos + 3 * fl;
    double kx = (knee_left[0] + knee_right[0]) * 0.5, ky = (knee_left[1] + knee_right[1]) * 0.5;
    double fx = (foot_left[0] + foot_right[0]) * 0.5, fy = (foot_left[1] + foot_right[1]) * 0.5;
    kx -= fx; ky -= fy;
    residual[6] = sqrt(kx * kx + ky * ky);
    double cx = c.subtree_com[3 * torso] - fx, cy = c.subtree_com[3 * torso + 1] - fy;
    residual[7] = sqrt(cx * cx + cy * cy);
    if (!I[7]) residual[8] = 0;
    else {
      const double *xi = c.ximat + 9 * torso, *tp = c.xipos + 3 * torso;
      double tx = D[0] - tp[0], ty = D[1] - tp[1];
      double n = sqrt(tx * tx + ty * ty);
      if (n < D_MINVAL) { tx = 1; ty = 0; } else { tx = d_div(tx, n); ty = d_div(ty, n); }
      tx -= xi[0]; ty -= xi[3];
      residual[8] = sqrt(tx * tx + ty * ty);
    }
    double tv[3];
    body_linvel(c, torso, tv);
    residual[9] = tv[0]; residual[10] = tv[1];
  }
  PFOR(i, nv - 6) residual[11 + i] = c.qvel[6 + i];
  PFOR(i, nu) residual[11 + nv - 6 + i] = c.ctrl[i];
  const int o = 11 + nv - 6 + nu;
  PFOR(e, 15) {
    const int i = e / 3, k = e - 3 * i;
    const int b1 = I[8 + 2 * i], b2 = I[9 + 2 * i];
    double r = 0;
    if (b1 >= 0 && b2 >= 0) {
      const double *l1 = D + 2 + 6 * i, *l2 = l1 + 3;
      const double *R1 = c.xmat + 9 * b1 + 3 * k, *R2 = c.xmat + 9 * b2 + 3 * k;
      double g1 = (R1[0] * l1[0] + R1[1] * l1[1] + R1[2] * l1[2]) + c.xpos[3 * b1 + k];
      double g2 = (R2[0] * l2[0] + R2[1] * l2[1] + R2[2] * l2[2]) + c.xpos[3 * b2 + k];
      r = fabs(g1 - g2);
    }
    residual[o + e] = r;
  }
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
    d_copy4(iq, MDH(body_iquat) + 4 * goal); d_mulquat(gq, c.xquat + 4 * goal, iq);
    d_copy4(iq, MDH(body_iquat) + 4 * cube); d_mulquat(cq, c.xquat + 4 * cube, iq);
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
    if (M.na && M.task.num_residual >= M.nq + M.nv + M.na) PFOR(i, M.na) residual[M.nq + M.nv + i] = C_ACT(c)[i];
  } else if (id == 2) {
    residual_quadruped(c, residual);
  } else if (id == 4) {
    residual_humanoid_track(c, residual);
  } else if (id == 5) {
    residual_humanoid_stand(c, residual);
  } else if (id == 6) {
    residual_humanoid_walk(c, residual);
  } else if (id == 15) {
    residual_humanoid_interact(c, residual);
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
  } else if (id == 10) {  // quadruped.cc:726-768 (Quadruped Hill): height over the feet, position and orientation against the goal, control
    const int *I = MI(task.int_data);
    int b = I[0], nu = M.nu;
    if (LANE == 0) {
      double avg = 0.25 * (c.site_xpos[3 * I[1] + 2] + c.site_xpos[3 * I[2] + 2] + c.site_xpos[3 * I[3] + 2] + c.site_xpos[3 * I[4] + 2]);
      residual[0] = (c.xpos[3 * b + 2] - avg) - MD(task.parameters)[0];
      for (int k = 0; k < 3; k++) residual[1 + k] = c.xpos[3 * b + k] - c.mocap_pos[k];
      double gm[9], bm[9];
      d_quat2mat(gm, c.mocap_quat);
      d_quat2mat(bm, c.xquat + 4 * b);
      for (int k = 0; k < 9; k++) residual[4 + k] = bm[k] - gm[k];
    }
    PFOR(i, nu) residual[13 + i] = c.ctrl[i];
  } else if (id == 11 || id == 12) {   // particle.cc:30-50 / 68-73: tip - goal (Lissajous curve of the time, or the mocap body), tip velocity, control
    if (LANE == 0) {
      int s = MI(task.int_data)[0], body = MI(site_bodyid)[s];
      double goal[2] = {c.mocap_pos[0], c.mocap_pos[1]}, off[3], lin[3];
      if (id == 11) { goal[0] = 0.25 * sin(c.time); goal[1] = 0.25 * cos(c.time / 3.14159265358979323846); }
      residual[0] = c.site_xpos[3 * s] - goal[0]; residual[1] = c.site_xpos[3 * s + 1] - goal[1];
      d_sub3(off, c.site_xpos + 3 * s, c.subtree_com + 3 * MIH(body_rootid)[body]);
      d_cross(lin, c.cvel + 6 * body, off);
      d_add3(lin, lin, c.cvel + 6 * body + 3);
      residual[2] = lin[0]; residual[3] = lin[1];
      residual[4] = c.ctrl[0]; residual[5] = c.ctrl[1];
    }
  } else if (id == 14) {   // swimmer.cc:33-46: control, nose - target in the plane
    int g = MI(task.int_data)[0];
    PFOR(i, M.nu) residual[i] = c.ctrl[i];
    if (LANE == 0) { residual[M.nu] = c.geom_xpos[3 * g] - c.mocap_pos[0]; residual[M.nu + 1] = c.geom_xpos[3 * g + 1] - c.mocap_pos[1]; }
  } else if (id == 13) {   // quadrotor.cc:37-60: position - goal, linear velocity, angular velocity (world frame), control - hover thrust
    if (LANE == 0) {
      int b = MI(task.int_data)[0];
      double lin[3];
      d_sub3(residual, c.xipos + 3 * b, c.mocap_pos);
      body_linvel(c, b, lin); d_copy3(residual + 3, lin);
      d_copy3(residual + 6, c.cvel + 6 * b);
      double g = d_sqrt(M.gravity[0] * M.gravity[0] + M.gravity[1] * M.gravity[1] + M.gravity[2] * M.gravity[2]);
      double thrust = d_div((MDH(body_mass)[0] + MDH(body_mass)[1]) * g, (double)M.nu);
      for (int i = 0; i < M.nu; i++) residual[9 + i] = c.ctrl[i] - thrust;
      for (int i = 9 + M.nu; i < M.task.num_residual; i++) residual[i] = 0;
    }
  } else if (id == 16) {  // fingers.cc:31-62: finger - object (framepos of a body = its inertial frame origin), object sites - target sites, control
    const int *I = MI(task.int_data);
    if (LANE == 0) {
      d_sub3(residual, c.xipos + 3 * I[0], c.xipos + 3 * I[2]);
      d_sub3(residual + 3, c.xipos + 3 * I[1], c.xipos + 3 * I[2]);
      for (int i = 0; i < 3; i++) {
        double df[3];
        d_sub3(df, c.site_xpos + 3 * I[3 + i], c.site_xpos + 3 * I[6 + i]);
        residual[6 + i] = d_sqrt(df[0] * df[0] + df[1] * df[1] + df[2] * df[2]);
      }
    }
    PFOR(i, M.nu) residual[9 + i] = c.ctrl[i];
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

