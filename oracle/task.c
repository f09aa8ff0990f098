/*
 * oracle/task.c — TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Cost (norms, risk), task residuals and the Ground() ray cast, restated from:
 *   mjpc/norm.cc:25-210            NormParameterDimension, Norm
 *   mjpc/task.cc:71-110            BaseResidualFn::CostTerms / CostValue
 *   mjpc/tasks/cartpole/cartpole.cc:36-49
 *   mjpc/test/testdata/particle_residual.h:33-43
 *   mjpc/tasks/quadruped/quadruped.cc:33-221,602-714 (+ constants quadruped.h:68-140)
 *   mjpc/utilities.cc:538-556      Ground (mj_ray straight down, geom group 0)
 */
#include <stdio.h>
#include "oracle.h"
#include "omath.h"

int oracle_norm_parameter_dimension(int type) {   /* norm.cc:25-47 */
  switch (type) {
    case MJPC_NORM_NULL: return 0;
    case MJPC_NORM_QUADRATIC: return 0;
    case MJPC_NORM_L22: return 2;
    case MJPC_NORM_L2: return 1;
    case MJPC_NORM_COSH: return 1;
    case MJPC_NORM_POWER: return 1;
    case MJPC_NORM_SMOOTHABS: return 1;
    case MJPC_NORM_SMOOTHABS2: return 2;
    case MJPC_NORM_RECTIFY: return 1;
  }
  return 0;
}

double oracle_norm(const double *x, const double *params, int n, int type) {   /* norm.cc:50-210, value only */
  double y = 0;
  int np_ = oracle_norm_parameter_dimension(type);
  double p = (params && np_ > 0) ? params[0] : 0, q = (params && np_ > 1) ? params[1] : 0;
  switch (type) {
    case MJPC_NORM_NULL: y = x[0]; break;
    case MJPC_NORM_QUADRATIC:
      for (int i = 0; i < n; i++) y += x[i] * x[i];
      y *= 0.5;
      break;
    case MJPC_NORM_L22: {
      double c = 0;
      for (int i = 0; i < n; i++) c += x[i] * x[i];
      double a = pow(c, q / 2) + pow(p, q);
      double s = pow(a, 1 / q);
      y = s - p;
      break;
    }
    case MJPC_NORM_L2: {
      double s = sqrt(o_dot(x, x, n) + p * p);
      y = s - p;
      break;
    }
    case MJPC_NORM_COSH:
      for (int i = 0; i < n; i++) y += p * p * (cosh(x[i] / p) - 1.0);
      break;
    case MJPC_NORM_POWER:
      for (int i = 0; i < n; i++) y += pow(fabs(x[i]), p);
      break;
    case MJPC_NORM_SMOOTHABS:
      for (int i = 0; i < n; i++) { double s = sqrt(x[i] * x[i] + p * p); y += s - p; }
      break;
    case MJPC_NORM_SMOOTHABS2:
      for (int i = 0; i < n; i++) {
        double a = fabs(x[i]);
        double d = pow(a, q);
        double e = d + pow(p, q);
        double s = pow(e, 1 / q);
        y += s - p;
      }
      break;
    case MJPC_NORM_RECTIFY:
      for (int i = 0; i < n; i++) {
        if (p > 0) { double s = exp(x[i] / p); y += p * log(1 + s); }
        else y += x[i] > 0 ? x[i] : 0;
      }
      break;
    default: break;
  }
  return y;
}

double oracle_cost_value(const MjpcHipTask *t, const double *residual, double *terms_out) {   /* task.cc:71-110 */
  double terms[MJPC_MAX_COST_TERMS];
  int f_shift = 0, p_shift = 0;
  for (int k = 0; k < t->num_term; k++) {
    terms[k] = t->weight[k] * oracle_norm(residual + f_shift, t->norm_parameter + p_shift,
                                          t->dim_norm_residual[k], t->norm[k]);
    f_shift += t->dim_norm_residual[k];
    p_shift += t->num_norm_parameter[k];
  }
  double cost = 0;
  for (int i = 0; i < t->num_term; i++) cost += terms[i];
  if (terms_out) for (int i = 0; i < t->num_term; i++) terms_out[i] = terms[i];
  if (fabs(t->risk) < 1e-6) return cost;                 /* kRiskNeutralTolerance */
  return (exp(t->risk * cost) - 1.0) / t->risk;
}

/* ---- ray casting against plane / sphere / box (mju_rayGeom semantics) ----------------- */
static double ray_geom(const double *pos, const double *mat, const double *size, const double *pnt, const double *vec, int type) {
  double dif[3], lp[3], lv[3];
  o_sub3(dif, pnt, pos);
  o_mulmattvec3(lp, mat, dif);
  o_mulmattvec3(lv, mat, vec);
  if (type == MJPC_GEOM_PLANE) {
    if (lv[2] > -O_MINVAL) return -1;
    double x = -lp[2] / lv[2];
    if (x < 0) return -1;
    double p0 = lp[0] + x * lv[0], p1 = lp[1] + x * lv[1];
    if ((size[0] <= 0 || fabs(p0) <= size[0]) && (size[1] <= 0 || fabs(p1) <= size[1])) return x;
    return -1;
  }
  if (type == MJPC_GEOM_SPHERE) {
    double a = o_dot3(lv, lv), b = o_dot3(lv, lp), c = o_dot3(lp, lp) - size[0] * size[0];
    double det = b * b - a * c;
    if (det < O_MINVAL || a < O_MINVAL) return -1;
    det = sqrt(det);
    double x0 = (-b - det) / a, x1 = (-b + det) / a;
    if (x0 >= 0) return x0;
    if (x1 >= 0) return x1;
    return -1;
  }
  if (type == MJPC_GEOM_BOX) {
    double best = -1;
    for (int i = 0; i < 3; i++) {
      if (fabs(lv[i]) <= O_MINVAL) continue;
      for (int side = -1; side <= 1; side += 2) {
        double x = (side * size[i] - lp[i]) / lv[i];
        if (x < 0) continue;
        int j = (i + 1) % 3, k = (i + 2) % 3;
        double pj = lp[j] + x * lv[j], pk = lp[k] + x * lv[k];
        if (fabs(pj) <= size[j] && fabs(pk) <= size[k]) if (best < 0 || x < best) best = x;
      }
    }
    return best;
  }
  return -1;
}

/* The reference aborts the process when the ray hits nothing (mju_error, utilities.cc:549-552); a planner candidate must
 * not do that, so a miss fails the rollout instead (MJPC_WARN_RAY) — same definition on the device. */
double oracle_ray_ground(const OModel *om, OData *d, const double pos[3]) {   /* utilities.cc:538-556 */
  const MjpcHipModel *m = &om->m;
  double down[3] = {0, 0, -1};
  double query[3] = {pos[0], pos[1], pos[2] + 0.5};
  double dist = -1;
  for (int r = 0; r < om->nray; r++) {
    int g = om->ray_geom[r];
    double x = ray_geom(d->geom_xpos + 3 * g, d->geom_xmat + 9 * g, m->geom_size + 3 * g, query, down, m->geom_type[g]);
    if (x >= 0 && (dist < 0 || x < dist)) dist = x;
  }
  if (dist < 0) d->warning |= MJPC_WARN_RAY;
  return pos[2] + 0.5 - dist;
}

/* ---- quadruped ------------------------------------------------------------------------ */
enum { QI_TORSO = 0, QI_HEAD = 1, QI_GOAL = 2, QI_FOOT = 3, QI_GAIT = 7, QI_GAIT_SWITCH = 8, QI_FLIP_DIR = 9,
       QI_BIPED_TYPE = 10, QI_CADENCE = 11, QI_AMPLITUDE = 12, QI_DUTY = 13, QI_HEADING = 14, QI_HOME = 15,
       QI_CROUCH = 16, QI_MODE = 17 };
enum { QD_MODE_START = 0, QD_POSITION = 1, QD_HEADING = 4, QD_SPEED = 6, QD_ANGVEL = 7, QD_GROUND = 8,
       QD_ORIENT = 9, QD_GAIT = 13, QD_PHASE_START = 14, QD_PHASE_START_TIME = 15, QD_PHASE_VEL = 16,
       QD_GRAVITY = 17, QD_JUMP_VEL = 18, QD_FLIGHT_TIME = 19, QD_JUMP_ACC = 20, QD_CROUCH_TIME = 21,
       QD_LEAP_TIME = 22, QD_JUMP_TIME = 23, QD_CROUCH_VEL = 24, QD_LAND_TIME = 25, QD_LAND_ACC = 26,
       QD_FLIGHT_ROT_VEL = 27, QD_JUMP_ROT_VEL = 28, QD_JUMP_ROT_ACC = 29, QD_LAND_ROT_ACC = 30 };
enum { kModeQuadruped = 0, kModeBiped, kModeWalk, kModeScramble, kModeFlip };
enum { kFootFL = 0, kFootHL, kFootFR, kFootHR };
static const double kGaitPhase[5][4] = {   /* quadruped.h:77-85 */
  {0, 0, 0, 0}, {0, 0.75, 0.5, 0.25}, {0, 0.5, 0.5, 0}, {0, 0.33, 0.33, 0.66}, {0, 0.4, 0.05, 0.35}};
static const double kHeightQuadruped = 0.25, kHeightBiped = 0.6, kFootRadius = 0.02, kMinAngvel = 0.01;
static const double kJointPostureGain[3] = {2, 1, 1};
static const double kLeapHeight = 0.5;

static int reinterpret_int(double v) { int i; memcpy(&i, &v, sizeof(int)); return i; }   /* utilities.cc:100-102 */

static double q_step_height(double time, double footphase, double duty_ratio) {   /* quadruped.cc:650-659 */
  double angle = fmod(time + O_PI - footphase, 2 * O_PI) - O_PI;
  double value = 0;
  if (duty_ratio < 1) {
    angle *= 0.5 / (1 - duty_ratio);
    value = cos(o_clip(angle, -O_PI / 2, O_PI / 2));
  }
  return fabs(value) < 1e-6 ? 0.0 : value;
}
static double q_flip_height(const double *D, double time) {   /* quadruped.cc:674-690 */
  double jump_time = D[QD_JUMP_TIME], flight_time = D[QD_FLIGHT_TIME], land_time = D[QD_LAND_TIME];
  if (time >= jump_time + flight_time + land_time) return kHeightQuadruped + D[QD_GROUND];
  double h = 0;
  if (time < jump_time) h = kHeightQuadruped + time * D[QD_CROUCH_VEL] + 0.5 * time * time * D[QD_JUMP_ACC];
  else if (time >= jump_time && time < jump_time + flight_time) { time -= jump_time; h = kLeapHeight + D[QD_JUMP_VEL] * time - 0.5 * 9.81 * time * time; }
  else if (time >= jump_time + flight_time) { time -= jump_time + flight_time; h = kLeapHeight - D[QD_JUMP_VEL] * time + 0.5 * D[QD_LAND_ACC] * time * time; }
  return h + D[QD_GROUND];
}
static void q_flip_quat(const double *D, const double *params, const int *I, double quat[4], double time) {   /* quadruped.cc:695-714 */
  double angle = 0;
  double jump_time = D[QD_JUMP_TIME], flight_time = D[QD_FLIGHT_TIME], land_time = D[QD_LAND_TIME], crouch_time = D[QD_CROUCH_TIME];
  if (time >= jump_time + flight_time + land_time) angle = 2 * O_PI;
  else if (time >= crouch_time && time < jump_time) { time -= crouch_time; angle = 0.5 * D[QD_JUMP_ROT_ACC] * time * time + D[QD_JUMP_ROT_VEL] * time; }
  else if (time >= jump_time && time < jump_time + flight_time) { time -= jump_time; angle = O_PI / 2 + D[QD_FLIGHT_ROT_VEL] * time; }
  else if (time >= jump_time + flight_time) { time -= jump_time + flight_time; angle = 1.75 * O_PI + D[QD_FLIGHT_ROT_VEL] * time - 0.5 * D[QD_LAND_ROT_ACC] * time * time; }
  int flip_dir = reinterpret_int(params[I[QI_FLIP_DIR]]);
  double axis[3] = {0, flip_dir ? 1.0 : -1.0, 0}, q[4];
  o_axisangle2quat(q, axis, angle);
  o_mulquat(quat, D + QD_ORIENT, q);
}
static void q_walk(const double *D, double pos[2], double time) {   /* quadruped.cc:619-636 */
  if (fabs(D[QD_ANGVEL]) < kMinAngvel) {
    double fwd[2] = {D[QD_HEADING], D[QD_HEADING + 1]};
    o_normalize(fwd, 2);
    pos[0] = D[QD_POSITION] + D[QD_HEADING] + time * D[QD_SPEED] * fwd[0];
    pos[1] = D[QD_POSITION + 1] + D[QD_HEADING + 1] + time * D[QD_SPEED] * fwd[1];
  } else {
    double angle = time * D[QD_ANGVEL];
    double c = cos(angle), s = sin(angle);
    pos[0] = c * D[QD_HEADING] - s * D[QD_HEADING + 1] + D[QD_POSITION];
    pos[1] = s * D[QD_HEADING] + c * D[QD_HEADING + 1] + D[QD_POSITION + 1];
  }
}

static void residual_quadruped(const OModel *om, OData *d, double *residual) {   /* quadruped.cc:33-221 */
  const MjpcHipModel *m = &om->m;
  const MjpcHipTask *t = &om->t;
  const int *I = t->int_data;
  const double *D = t->dbl_data, *P = t->parameters;
  int mode = I[QI_MODE], torso = I[QI_TORSO];
  int counter = 0;
  const double *foot_pos[4];
  for (int f = 0; f < 4; f++) foot_pos[f] = d->geom_xpos + 3 * I[QI_FOOT + f];
  /* average foot position (quadruped.cc:602-616) */
  double avg[3];
  if (mode == kModeBiped) {
    int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]);
    if (handstand) o_add3(avg, foot_pos[kFootFL], foot_pos[kFootFR]);
    else o_add3(avg, foot_pos[kFootHL], foot_pos[kFootHR]);
    o_scl3(avg, avg, 0.5);
  } else {
    o_add3(avg, foot_pos[kFootHL], foot_pos[kFootHR]);
    o_add3(avg, avg, foot_pos[kFootFL]);
    o_add3(avg, avg, foot_pos[kFootFR]);
    o_scl3(avg, avg, 0.25);
  }
  const double *torso_xmat = d->xmat + 9 * torso;
  const double *goal_pos = d->mocap_pos + 3 * I[QI_GOAL];
  const double *compos = d->subtree_com + 3 * torso;
  /* Upright */
  if (mode != kModeFlip) {
    if (mode == kModeBiped) {
      int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]) ? -1 : 1;
      residual[counter++] = torso_xmat[6] - handstand;
    } else residual[counter++] = torso_xmat[8] - 1;
    residual[counter++] = 0;
    residual[counter++] = 0;
  } else {
    double quat[4];
    q_flip_quat(D, P, I, quat, d->time - D[QD_MODE_START]);
    o_subquat(residual + counter, d->xquat + 4 * torso, quat);
    counter += 3;
  }
  /* Height */
  const double *torso_pos = d->xipos + 3 * torso;
  int is_biped = mode == kModeBiped;
  double height_goal = is_biped ? kHeightBiped : kHeightQuadruped;
  if (mode == kModeScramble) residual[counter++] = 0;
  else if (mode == kModeFlip) residual[counter++] = torso_pos[2] - q_flip_height(D, d->time - D[QD_MODE_START]);
  else residual[counter++] = (torso_pos[2] - avg[2]) - height_goal;
  /* Position */
  const double *head = d->site_xpos + 3 * I[QI_HEAD];
  double target[3];
  if (mode == kModeWalk) { q_walk(D, target, d->time - D[QD_MODE_START]); target[2] = 0; }
  else o_copy3(target, goal_pos);
  residual[counter++] = head[0] - target[0];
  residual[counter++] = head[1] - target[1];
  residual[counter++] = mode == kModeScramble ? 2 * (head[2] - target[2]) : 0;
  /* Gait */
  int gait = mode == kModeBiped ? 2 : reinterpret_int(D[QD_GAIT]);
  double phase = D[QD_PHASE_START] + (d->time - D[QD_PHASE_START_TIME]) * D[QD_PHASE_VEL];
  double amplitude = P[I[QI_AMPLITUDE]], duty = P[I[QI_DUTY]];
  double step[4];
  for (int f = 0; f < 4; f++) step[f] = amplitude * q_step_height(phase, 2 * O_PI * kGaitPhase[gait][f], duty);
  for (int f = 0; f < 4; f++) {
    if (is_biped) {
      int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]) != 0;
      int front_hand = !handstand && (f == kFootFL || f == kFootFR);
      int back_hand = handstand && (f == kFootHL || f == kFootHR);
      if (front_hand || back_hand) { residual[counter++] = 0; continue; }
    }
    double query[3] = {foot_pos[f][0], foot_pos[f][1], foot_pos[f][2]};
    if (mode == kModeScramble) {
      double v[3];
      o_sub3(v, goal_pos, torso_pos); o_normalize3(v);
      o_sub3(v, goal_pos, foot_pos[f]);
      v[2] = 0; o_normalize3(v);
      o_addtoscl3(query, v, 0.15);
    }
    double ground_height = oracle_ray_ground(om, d, query);
    double height_target = ground_height + kFootRadius + step[f];
    double height_difference = foot_pos[f][2] - height_target;
    if (mode == kModeScramble) height_difference = fmin(0, height_difference);
    residual[counter++] = step[f] ? height_difference : 0;
  }
  /* Balance */
  const double *comvel = d->subtree_linvel + 3 * torso;
  double fall_time = sqrt(2 * height_goal / 9.81), capture[3];
  o_addscl3(capture, compos, comvel, fall_time);
  residual[counter++] = capture[0] - avg[0];
  residual[counter++] = capture[1] - avg[1];
  /* Effort */
  for (int i = 0; i < m->nu; i++) residual[counter + i] = d->actuator_force[i] * 2e-2;
  counter += m->nu;
  /* Posture */
  const double *home = m->key_qpos + I[QI_HOME] * m->nq;
  for (int i = 0; i < m->nu; i++) residual[counter + i] = d->qpos[7 + i] - home[7 + i];
  if (mode == kModeFlip) {
    double flip_time = d->time - D[QD_MODE_START];
    if (flip_time < D[QD_CROUCH_TIME]) {
      const double *crouch = m->key_qpos + I[QI_CROUCH] * m->nq;
      for (int i = 0; i < m->nu; i++) residual[counter + i] = d->qpos[7 + i] - crouch[7 + i];
    } else if (flip_time >= D[QD_CROUCH_TIME] && flip_time < D[QD_JUMP_TIME] + D[QD_FLIGHT_TIME]) {
      for (int i = 0; i < m->nu; i++) residual[counter + i] = 0;
    }
  }
  for (int f = 0; f < 4; f++) for (int j = 0; j < 3; j++) residual[counter + 3 * f + j] *= kJointPostureGain[j];
  if (mode == kModeBiped) {
    int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]) != 0;
    if (handstand) { residual[counter + 4] *= 0.03; residual[counter + 5] *= 0.03; residual[counter + 10] *= 0.03; residual[counter + 11] *= 0.03; }
    else { residual[counter + 1] *= 0.03; residual[counter + 2] *= 0.03; residual[counter + 7] *= 0.03; residual[counter + 8] *= 0.03; }
  }
  counter += m->nu;
  /* Yaw */
  double th[2] = {torso_xmat[0], torso_xmat[3]};
  if (mode == kModeBiped) {
    int handstand = reinterpret_int(P[I[QI_BIPED_TYPE]]) ? 1 : -1;
    th[0] = handstand * torso_xmat[2]; th[1] = handstand * torso_xmat[5];
  }
  o_normalize(th, 2);
  double heading_goal = P[I[QI_HEADING]];
  residual[counter++] = th[0] - cos(heading_goal);
  residual[counter++] = th[1] - sin(heading_goal);
  /* "Angmom": declared as a second subtreelinvel sensor (task_flat.xml:143) */
  o_copy3(residual + counter, d->subtree_linvel + 3 * torso);
  counter += 3;
}

/* ---- humanoid tracking (mjpc/tasks/humanoid/tracking/tracking.cc:94-216) --------------------------------
 * int_data: [0] motion id, [1] first key of the motion, [2] motion length, [3..18] tracking site ids,
 *           [19..34] mocap ids, both in the order of tracking.cc:59-63 body_names; dbl_data: [0] reference_time */
static void residual_humanoid_track(const OModel *om, OData *d, double *residual) {
  const MjpcHipModel *m = &om->m;
  const int *I = om->t.int_data;
  const double kFps = 30.0;
  int start = I[1], length = I[2];
  double current_index = (d->time - om->t.dbl_data[0]) * kFps + start;
  int last_key_index = start + length - 1;
  double ci = current_index < 0 ? 0 : (current_index > last_key_index ? (double)last_key_index : current_index);
  int k0 = (int)floor(ci), k1 = k0 + 1 < last_key_index ? k0 + 1 : last_key_index;
  double w1 = ci - k0, w0 = 1.0 - w1;
  int counter = 0;
  o_copy(residual + counter, d->qvel + 6, m->nv - 6); counter += m->nv - 6;
  o_copy(residual + counter, d->ctrl, m->nu); counter += m->nu;
  double avg_m[3] = {0, 0, 0}, avg_s[3] = {0, 0, 0}, mp[16][3];
  for (int b = 0; b < 16; b++) {
    int mid = I[19 + b];
    const double *p0 = m->key_mpos + m->nmocap * 3 * k0 + 3 * mid, *p1 = m->key_mpos + m->nmocap * 3 * k1 + 3 * mid;
    o_scl3(mp[b], p0, w0);
    o_addtoscl3(mp[b], p1, w1);
    o_add3(avg_m, avg_m, mp[b]);
    o_add3(avg_s, avg_s, d->site_xpos + 3 * I[3 + b]);
  }
  o_scl3(avg_m, avg_m, 1.0 / 16); o_scl3(avg_s, avg_s, 1.0 / 16);
  o_sub3(residual + counter, avg_m, avg_s); counter += 3;
  for (int b = 0; b < 16; b++) {
    double bm[3], bs[3];
    o_sub3(bm, mp[b], avg_m);
    o_sub3(bs, d->site_xpos + 3 * I[3 + b], avg_s);
    o_sub3(residual + counter, bm, bs); counter += 3;
  }
  for (int b = 0; b < 16; b++) {
    int mid = I[19 + b], sid = I[3 + b], body = m->site_bodyid[sid];
    const double *p0 = m->key_mpos + m->nmocap * 3 * k0 + 3 * mid, *p1 = m->key_mpos + m->nmocap * 3 * k1 + 3 * mid;
    double v[3], off[3], lin[3];
    o_sub3(v, p1, p0); o_scl3(v, v, kFps);
    /* framelinvel: velocity of the site point in the world frame */
    o_sub3(off, d->site_xpos + 3 * sid, d->subtree_com + 3 * m->body_rootid[body]);
    o_cross(lin, d->cvel + 6 * body, off);
    o_add3(lin, lin, d->cvel + 6 * body + 3);
    o_sub3(residual + counter, v, lin); counter += 3;
  }
}

/* velocity of a body's inertial-frame origin in the world frame (framelinvel objtype="body", mj_objectVelocity) */
static void body_linvel(const OModel *om, const OData *d, int body, double *lin) {
  double off[3];
  o_sub3(off, d->xipos + 3 * body, d->subtree_com + 3 * om->m.body_rootid[body]);
  o_cross(lin, d->cvel + 6 * body, off);
  o_add3(lin, lin, d->cvel + 6 * body + 3);
}

/* mjpc/tasks/humanoid/stand/stand.cc:41-94.  int_data = [site sp0, sp1, sp2, sp3, body head, body torso] */
static void residual_humanoid_stand(const OModel *om, OData *d, double *residual) {
  const MjpcHipModel *m = &om->m;
  const int *I = om->t.int_data;
  int counter = 0;
  const double *f1 = d->site_xpos + 3 * I[0], *f2 = d->site_xpos + 3 * I[1], *f3 = d->site_xpos + 3 * I[2], *f4 = d->site_xpos + 3 * I[3];
  const double *head = d->xipos + 3 * I[4];                                   /* framepos objtype="body": inertial frame */
  double head_feet_error = head[2] - 0.25 * (f1[2] + f2[2] + f3[2] + f4[2]);
  residual[counter++] = head_feet_error - om->t.parameters[0];
  const double *com = d->subtree_com + 3 * I[5], *comvel = d->subtree_linvel + 3 * I[5];
  double kFallTime = 0.2;
  double cp[3] = {com[0], com[1], com[2]};
  o_addtoscl3(cp, comvel, kFallTime);
  double fxy[2] = {0, 0};
  fxy[0] += f1[0]; fxy[1] += f1[1]; fxy[0] += f2[0]; fxy[1] += f2[1]; fxy[0] += f3[0]; fxy[1] += f3[1]; fxy[0] += f4[0]; fxy[1] += f4[1];
  fxy[0] *= 0.25; fxy[1] *= 0.25;
  fxy[0] -= cp[0]; fxy[1] -= cp[1];
  residual[counter++] = sqrt(fxy[0] * fxy[0] + fxy[1] * fxy[1]);
  residual[counter++] = comvel[0]; residual[counter++] = comvel[1];
  o_copy(residual + counter, d->qvel + 6, m->nv - 6); counter += m->nv - 6;
  o_copy(residual + counter, d->ctrl, m->nu); counter += m->nu;
}

/* mjpc/tasks/humanoid/interact/interact.cc:31-186 (sensors of task.xml:52-68: framezaxis of the xbody frames = column 2 of xmat,
 * framepos / framexaxis / framelinvel of a body = its inertial frame, subtreecom).  int_data = [body torso, pelvis, foot_right,
 * foot_left, head, shin_right, shin_left, has facing target, (body1, body2) x 5]; dbl_data = [facing x, y, (local1, local2) x 5] */
static void residual_humanoid_interact(const OModel *om, OData *d, double *residual) {
  const MjpcHipModel *m = &om->m;
  const int *I = om->t.int_data;
  const double *P = om->t.parameters, *D = om->t.dbl_data;
  int torso = I[0], pelvis = I[1], fr = I[2], fl = I[3], head = I[4], kr = I[5], kl = I[6];
  int counter = 0;
  residual[counter++] = fabs(d->xmat[9 * torso + 8] - 1.0);
  residual[counter++] = fabs(d->xmat[9 * pelvis + 8] - 1.0);
  residual[counter++] = fabs(d->xmat[9 * fr + 8] - 1.0);
  residual[counter++] = fabs(d->xmat[9 * fl + 8] - 1.0);
  residual[counter++] = fabs(d->xipos[3 * head + 2] - P[0]);
  residual[counter++] = fabs(d->xipos[3 * torso + 2] - P[1]);
  const double *knee_right = d->xipos + 3 * kr, *knee_left = d->xipos + 3 * kl, *foot_right = d->xipos + 3 * fr, *foot_left = d->xipos + 3 * fl;
  double knee[2] = {0, 0}, foot[2] = {0, 0};
  knee[0] += knee_left[0]; knee[1] += knee_left[1]; knee[0] += knee_right[0]; knee[1] += knee_right[1]; knee[0] *= 0.5; knee[1] *= 0.5;
  foot[0] += foot_left[0]; foot[1] += foot_left[1]; foot[0] += foot_right[0]; foot[1] += foot_right[1]; foot[0] *= 0.5; foot[1] *= 0.5;
  knee[0] -= foot[0]; knee[1] -= foot[1];
  residual[counter++] = sqrt(knee[0] * knee[0] + knee[1] * knee[1]);
  double com[2] = {d->subtree_com[3 * torso] - foot[0], d->subtree_com[3 * torso + 1] - foot[1]};
  residual[counter++] = sqrt(com[0] * com[0] + com[1] * com[1]);
  if (!I[7]) residual[counter++] = 0;
  else {
    const double *xi = d->ximat + 9 * torso, *tp = d->xipos + 3 * torso;
    double target[2] = {D[0] - tp[0], D[1] - tp[1]};
    double n = sqrt(target[0] * target[0] + target[1] * target[1]);
    if (n < O_MINVAL) { target[0] = 1; target[1] = 0; } else { target[0] /= n; target[1] /= n; }      /* mju_normalize */
    target[0] -= xi[0]; target[1] -= xi[3];
    residual[counter++] = sqrt(target[0] * target[0] + target[1] * target[1]);
  }
  double tv[3];
  body_linvel(om, d, torso, tv);                          /* the sensor named torso_subtreelinvel is a framelinvel of the torso body */
  residual[counter++] = tv[0]; residual[counter++] = tv[1];
  o_copy(residual + counter, d->qvel + 6, m->nv - 6); counter += m->nv - 6;
  o_copy(residual + counter, d->ctrl, m->nu); counter += m->nu;
  for (int i = 0; i < 5; i++) {
    int b1 = I[8 + 2 * i], b2 = I[9 + 2 * i];
    if (b1 >= 0 && b2 >= 0) {
      const double *l1 = D + 2 + 6 * i, *l2 = l1 + 3;
      double g1[3], g2[3];
      o_mulmatvec3(g1, d->xmat + 9 * b1, l1); o_add3(g1, g1, d->xpos + 3 * b1);
      o_mulmatvec3(g2, d->xmat + 9 * b2, l2); o_add3(g2, g2, d->xpos + 3 * b2);
      for (int k = 0; k < 3; k++) residual[counter++] = fabs(g1[k] - g2[k]);
    } else for (int k = 0; k < 3; k++) residual[counter++] = 0;
  }
}

/* mjpc/tasks/humanoid/walk/walk.cc:44-166.  int_data = [body torso, pelvis, foot_right, foot_left, waist_lower] */
static void residual_humanoid_walk(const OModel *om, OData *d, double *residual) {
  const MjpcHipModel *m = &om->m;
  const int *I = om->t.int_data;
  const double *P = om->t.parameters;
  int torso = I[0], pelvis = I[1], fr = I[2], fl = I[3], wl = I[4];
  int counter = 0;
  double torso_height = d->xipos[3 * torso + 2];
  residual[counter++] = torso_height - P[0];
  const double *foot_right = d->xipos + 3 * fr, *foot_left = d->xipos + 3 * fl;
  double pelvis_height = d->xipos[3 * pelvis + 2];
  residual[counter++] = 0.5 * (foot_left[2] + foot_right[2]) - pelvis_height - 0.2;
  const double *subcom = d->subtree_com + 3 * torso, *subcomvel = d->subtree_linvel + 3 * torso;
  double capture_point[3], axis[3], center[3], vec[3], pcp[3];
  for (int k = 0; k < 3; k++) capture_point[k] = subcom[k] + subcomvel[k] * 0.3;
  capture_point[2] = 1.0e-3;
  o_sub3(axis, foot_right, foot_left);
  axis[2] = 1.0e-3;
  double length = 0.5 * o_normalize3(axis) - 0.05;
  o_add3(center, foot_right, foot_left);
  o_scl3(center, center, 0.5);
  o_sub3(vec, capture_point, center);
  double t = o_dot3(vec, axis);
  t = fmax(-length, fmin(length, t));
  o_scl3(vec, axis, t);
  o_add3(pcp, vec, center);
  pcp[2] = 1.0e-3;
  double standing = torso_height / sqrt(torso_height * torso_height + 0.45 * 0.45) - 0.4;
  residual[counter] = (capture_point[0] - pcp[0]) * standing; residual[counter + 1] = (capture_point[1] - pcp[1]) * standing;
  counter += 2;
  /* framezaxis / framexaxis of the xbody frames: columns 2 / 0 of xmat */
  const double *xt = d->xmat + 9 * torso, *xp = d->xmat + 9 * pelvis, *xr = d->xmat + 9 * fr, *xl = d->xmat + 9 * fl;
  residual[counter++] = xt[8] - 1.0;
  residual[counter++] = 0.3 * (xp[8] - 1.0);
  double zref[3] = {0, 0, 1};
  for (int k = 0; k < 3; k++) residual[counter + k] = (xr[3 * k + 2] - zref[k]) * (0.1 * standing);
  counter += 3;
  for (int k = 0; k < 3; k++) residual[counter + k] = (xl[3 * k + 2] - zref[k]) * (0.1 * standing);
  counter += 3;
  o_copy(residual + counter, d->qpos + 7, m->nq - 7); counter += m->nq - 7;
  double forward[2] = {xt[0], xt[3]};
  forward[0] += xp[0]; forward[1] += xp[3]; forward[0] += xr[0]; forward[1] += xr[3]; forward[0] += xl[0]; forward[1] += xl[3];
  { double n = sqrt(forward[0] * forward[0] + forward[1] * forward[1]);
    if (n < O_MINVAL) { forward[0] = 1; forward[1] = 0; } else { double sc = 1.0 / n; forward[0] *= sc; forward[1] *= sc; } }     /* mju_normalize */
  double tv[3], rv[3], lv[3];
  body_linvel(om, d, torso, tv); body_linvel(om, d, fr, rv); body_linvel(om, d, fl, lv);
  const double *wlv = d->subtree_linvel + 3 * wl;
  double com_vel[2] = {(wlv[0] + tv[0]) * 0.5, (wlv[1] + tv[1]) * 0.5};
  residual[counter++] = standing * (com_vel[0] * forward[0] + com_vel[1] * forward[1] - P[1]);
  double mf[2] = {com_vel[0], com_vel[1]};
  mf[0] += rv[0] * -0.5; mf[1] += rv[1] * -0.5; mf[0] += lv[0] * -0.5; mf[1] += lv[1] * -0.5;
  residual[counter] = mf[0] * standing; residual[counter + 1] = mf[1] * standing;
  counter += 2;
  o_copy(residual + counter, d->ctrl, m->nu); counter += m->nu;
}

/* mjpc/tasks/shadow_reorient/hand.cc:37-84.  int_data = [palm site (framepos "palm_position", task.xml:45), cube body,
 * goal body, keyframe].  Sensors restated: framepos / framequat / framelinvel with objtype="body" read the body's INERTIAL
 * frame (xipos, xquat * body_iquat, velocity at xipos in world axes), common_assets/reorientation_cube.xml:31-36. */
static void residual_shadow(const OModel *om, OData *d, double *residual) {
  const MjpcHipModel *m = &om->m;
  const int *I = om->t.int_data;
  int palm = I[0], cube = I[1], goal = I[2], key = I[3];
  int counter = 0;
  o_sub3(residual + counter, d->xipos + 3 * cube, d->site_xpos + 3 * palm);            /* (0) cube position - palm position */
  counter += 3;
  double goal_orientation[4], orientation[4];
  o_mulquat(goal_orientation, d->xquat + 4 * goal, m->body_iquat + 4 * goal);
  o_mulquat(orientation, d->xquat + 4 * cube, m->body_iquat + 4 * cube);
  o_normalize4(goal_orientation);
  o_subquat(residual + counter, goal_orientation, orientation);                          /* (1) orientation error */
  counter += 3;
  body_linvel(om, d, cube, residual + counter);                                          /* (2) cube linear velocity */
  counter += 3;
  o_copy(residual + counter, d->actuator_force, m->nu);                                  /* (3) actuator forces */
  counter += m->nu;
  /* (4), (5): the 26-wide slices start at 7 / 6 although the first joint is the goal's ball joint: they straddle the cube's
   * free joint (hand.cc:75-80) */
  for (int i = 0; i < 26; i++) residual[counter + i] = d->qpos[7 + i] - m->key_qpos[key * m->nq + 7 + i];
  counter += 26;
  o_copy(residual + counter, d->qvel + 6, 26);
}

void oracle_residual(const OModel *om, OData *d, double *residual) {
  const MjpcHipModel *m = &om->m;
  switch (om->t.task_id) {
    case MJPC_TASK_PARTICLE:   /* particle_residual.h:33-43 */
      o_copy(residual, d->qpos, m->nq);
      residual[0] -= d->mocap_pos[0];
      residual[1] -= d->mocap_pos[1];
      o_copy(residual + 2, d->qvel, m->nv);
      break;
    case MJPC_TASK_CARTPOLE:   /* cartpole.cc:36-49 */
      residual[0] = cos(d->qpos[1]) - 1;
      residual[1] = d->qpos[0] - om->t.parameters[0];
      residual[2] = d->qvel[1];
      residual[3] = d->ctrl[0];
      break;
    case MJPC_TASK_COPYSTATE:
      o_copy(residual, d->qpos, m->nq);
      o_copy(residual + m->nq, d->qvel, m->nv);
      if (m->na > 0 && om->t.num_residual >= m->nq + m->nv + m->na) o_copy(residual + m->nq + m->nv, d->act, m->na);      /* the whole state [qpos, qvel, act] when the task has room for it */
      break;
    case MJPC_TASK_QUADRUPED:
      residual_quadruped(om, d, residual);
      break;
    case MJPC_TASK_HUMANOID_TRACK:
      residual_humanoid_track(om, d, residual);
      break;
    case MJPC_TASK_HUMANOID_STAND:
      residual_humanoid_stand(om, d, residual);
      break;
    case MJPC_TASK_HUMANOID_WALK:
      residual_humanoid_walk(om, d, residual);
      break;
    case MJPC_TASK_HUMANOID_INTERACT:
      residual_humanoid_interact(om, d, residual);
      break;
    case MJPC_TASK_SHADOW_REORIENT:
      residual_shadow(om, d, residual);
      break;
    case MJPC_TASK_WALKER: {   /* walker.cc:39-57 */
      int nu = m->nu, b = om->t.int_data[0];
      o_copy(residual, d->ctrl, nu);
      residual[nu] = d->xpos[3 * b + 2] - om->t.parameters[0];        /* framepos of xbody torso, z */
      residual[nu + 1] = d->xmat[9 * b + 8] - 1.0;                    /* framezaxis of xbody torso, z */
      residual[nu + 2] = d->subtree_linvel[3 * b] - om->t.parameters[1];
      break;
    }
    case MJPC_TASK_QUADRUPED_HILL: {   /* quadruped.cc:726-768 */
      const int *I = om->t.int_data;
      int b = I[0];
      double avg = 0.25 * (d->site_xpos[3 * I[1] + 2] + d->site_xpos[3 * I[2] + 2] + d->site_xpos[3 * I[3] + 2] + d->site_xpos[3 * I[4] + 2]);
      residual[0] = (d->xpos[3 * b + 2] - avg) - om->t.parameters[0];
      for (int k = 0; k < 3; k++) residual[1 + k] = d->xpos[3 * b + k] - d->mocap_pos[k];
      double gm[9], bm[9];
      o_quat2mat(gm, d->mocap_quat);
      o_quat2mat(bm, d->xquat + 4 * b);
      for (int k = 0; k < 9; k++) residual[4 + k] = bm[k] - gm[k];
      o_copy(residual + 13, d->ctrl, m->nu);
      break;
    }
    case MJPC_TASK_PARTICLE_TIMEVARYING:   /* particle.cc:30-50: some Lissajous curve of data->time */
    case MJPC_TASK_PARTICLE_FIXED: {       /* particle.cc:68-73: goal = the mocap body */
      int s = om->t.int_data[0], body = m->site_bodyid[s];
      double goal[2] = {d->mocap_pos[0], d->mocap_pos[1]}, off[3], lin[3];
      if (om->t.task_id == MJPC_TASK_PARTICLE_TIMEVARYING) { goal[0] = 0.25 * sin(d->time); goal[1] = 0.25 * cos(d->time / 3.14159265358979323846); }
      for (int k = 0; k < m->nq && k < 2; k++) residual[k] = d->site_xpos[3 * s + k] - goal[k];
      o_sub3(off, d->site_xpos + 3 * s, d->subtree_com + 3 * m->body_rootid[body]);      /* framelinvel of the site */
      o_cross(lin, d->cvel + 6 * body, off);
      o_add3(lin, lin, d->cvel + 6 * body + 3);
      for (int k = 0; k < m->nv && k < 2; k++) residual[2 + k] = lin[k];
      o_copy(residual + 4, d->ctrl, m->nu);
      break;
    }
    case MJPC_TASK_SWIMMER: {     /* swimmer.cc:33-46 */
      int g = om->t.int_data[0];
      o_copy(residual, d->ctrl, m->nu);
      residual[m->nu] = d->geom_xpos[3 * g] - d->mocap_pos[0];
      residual[m->nu + 1] = d->geom_xpos[3 * g + 1] - d->mocap_pos[1];
      break;
    }
    case MJPC_TASK_QUADROTOR: {   /* quadrotor.cc:37-60; the two "Orientation" residuals the XML declares are never written */
      int b = om->t.int_data[0];
      double lin[3];
      o_sub3(residual, d->xipos + 3 * b, d->mocap_pos);
      body_linvel(om, d, b, lin); o_copy3(residual + 3, lin);
      o_copy3(residual + 6, d->cvel + 6 * b);                      /* frameangvel: angular velocity in the world frame */
      double thrust = (m->body_mass[0] + m->body_mass[1]) * o_norm3(m->gravity) / m->nu;
      for (int i = 0; i < m->nu; i++) residual[9 + i] = d->ctrl[i] - thrust;
      for (int i = 9 + m->nu; i < om->t.num_residual; i++) residual[i] = 0;
      break;
    }
    case MJPC_TASK_FINGERS: {  /* fingers.cc:31-62 (framepos of a body = its inertial frame origin) */
      const int *id = om->t.int_data;
      o_sub3(residual, d->xipos + 3 * id[0], d->xipos + 3 * id[2]);
      o_sub3(residual + 3, d->xipos + 3 * id[1], d->xipos + 3 * id[2]);
      for (int i = 0; i < 3; i++) {
        double df[3];
        o_sub3(df, d->site_xpos + 3 * id[3 + i], d->site_xpos + 3 * id[6 + i]);
        residual[6 + i] = o_norm3(df);
      }
      o_copy(residual + 9, d->ctrl, m->nu);
      break;
    }
    case MJPC_TASK_ACROBOT: {  /* acrobot.cc:34-49 */
      int g = om->t.int_data[0], t = om->t.int_data[1];
      residual[0] = d->site_xpos[3 * g + 2] - d->site_xpos[3 * t + 2];
      residual[1] = d->site_xpos[3 * g] - d->site_xpos[3 * t];
      residual[2] = d->qvel[0];
      residual[3] = d->qvel[1];
      residual[4] = d->ctrl[0];
      break;
    }
    default: break;
  }
}
