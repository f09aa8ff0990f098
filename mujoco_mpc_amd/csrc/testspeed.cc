// testspeed.cc — closed-loop harness (include/mjpc_hip_testspeed.h): mjpc/testspeed.cc:44-129 with the world and the planner on
// the HIP engine; host Task::Transition for the built-in tasks.
#include "../../include/mjpc_hip_testspeed.h"
#include "../../include/mjpc_hip_planner_c.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>

namespace mjpc_hip {

static void Die(const char* msg) { std::fprintf(stderr, "mjpc_hip testspeed error: %s\n", msg); std::abort(); }

// ------------------------------------------------------------------ HostTask
HostTask::HostTask(const MjpcHipTask& t) : base(t) {
  int np = 0;
  for (int k = 0; k < t.num_term; k++) np += t.num_norm_parameter[k];
  dim_norm_residual.assign(t.dim_norm_residual, t.dim_norm_residual + t.num_term);
  norm.assign(t.norm, t.norm + t.num_term);
  num_norm_parameter.assign(t.num_norm_parameter, t.num_norm_parameter + t.num_term);
  weight.assign(t.weight, t.weight + t.num_term);
  norm_parameter.assign(t.norm_parameter, t.norm_parameter + np);
  parameters.assign(t.parameters, t.parameters + t.num_parameter);
  trace_objtype.assign(t.trace_objtype, t.trace_objtype + t.num_trace);
  trace_objid.assign(t.trace_objid, t.trace_objid + t.num_trace);
  int_data.assign(t.int_data, t.int_data + t.num_int);
  dbl_data.assign(t.dbl_data, t.dbl_data + t.num_dbl);
}
const MjpcHipTask* HostTask::view() {
  base.dim_norm_residual = dim_norm_residual.data(); base.norm = norm.data(); base.num_norm_parameter = num_norm_parameter.data();
  base.weight = weight.data(); base.norm_parameter = norm_parameter.data(); base.parameters = parameters.data();
  base.trace_objtype = trace_objtype.data(); base.trace_objid = trace_objid.data();
  base.int_data = int_data.data(); base.dbl_data = dbl_data.data();
  return &base;
}

// ------------------------------------------------------------------ Simulator
Simulator::Simulator(const MjpcHipModel* model, const MjpcHipTask* task, int device)
    : nq(model->nq), nv(model->nv), nu(model->nu), nmocap(model->nmocap), nr(task->num_residual), timestep(model->timestep) {
  engine_ = mjpc_hip_create(model, task, 1, 2, device);
  if (!engine_) Die(mjpc_hip_last_error());
  int ds = nq + nv + model->na;
  states_.assign(2 * (size_t)ds, 0.0); residual_.assign(2 * (size_t)std::max(nr, 1), 0.0); costs_.assign(2, 0.0); times_.assign(2, 0.0);
  actions_.assign(2 * (size_t)std::max(nu, 1), 0.0); trace_.assign(2 * 3 * (size_t)std::max(task->num_trace, 1), 0.0);
  frame_.xpos.assign(3 * (size_t)model->nbody, 0.0); frame_.xmat.assign(9 * (size_t)model->nbody, 0.0);
  frame_.site_xpos.assign(3 * (size_t)std::max(model->nsite, 1), 0.0);
  frame_.subtree_com.assign(3 * (size_t)model->nbody, 0.0); frame_.subtree_linvel.assign(3 * (size_t)model->nbody, 0.0);
}
void Simulator::FetchFrame() {
  if (mjpc_hip_get_frame(engine_, frame_.xpos.data(), frame_.xmat.data(), frame_.site_xpos.data(), frame_.subtree_com.data(),
                         frame_.subtree_linvel.data()) != 0) Die(mjpc_hip_last_error());
  frame_.valid = true;
}
void Simulator::Forward(const SimState& s) {
  MjpcHipPlanInput in;
  std::memset(&in, 0, sizeof(in));
  double kt = s.time;
  std::vector<double> zero(std::max(nu, 1), 0.0);
  in.state = s.state.data(); in.mocap = s.mocap.data(); in.userdata = s.userdata.data(); in.time = s.time;
  in.knot_times = &kt; in.knot_values = zero.data(); in.num_spline_points = 1; in.interpolation = 0;
  in.num_trajectory = 1; in.horizon = 1; in.num_local = 1;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  if (mjpc_hip_plan(engine_, &in, &out) != 0) Die(mjpc_hip_last_error());
  FetchFrame();
}
Simulator::~Simulator() { if (engine_) mjpc_hip_destroy(engine_); }
void Simulator::SetTask(const MjpcHipTask* task) { if (mjpc_hip_set_task(engine_, task) != 0) Die(mjpc_hip_last_error()); }

double Simulator::Step(SimState& s, const double* ctrl, double* residual) {
  MjpcHipPlanInput in;
  std::memset(&in, 0, sizeof(in));
  double kt = s.time;
  in.state = s.state.data(); in.mocap = s.mocap.data(); in.userdata = s.userdata.data(); in.time = s.time;
  in.knot_times = &kt; in.knot_values = ctrl; in.num_spline_points = 1; in.interpolation = 0;
  in.num_trajectory = 1; in.horizon = 2; in.candidate_offset = 0; in.num_local = 1;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  double ret = 0; int fail = 0;
  out.returns = &ret; out.failure = &fail; out.states = states_.data(); out.actions = actions_.data(); out.times = times_.data();
  out.residual = residual_.data(); out.costs = costs_.data(); out.trace = trace_.data();
  if (mjpc_hip_plan(engine_, &in, &out) != 0) Die(mjpc_hip_last_error());
  size_t ds = s.state.size();
  std::copy(states_.begin() + ds, states_.begin() + 2 * ds, s.state.begin());       // x_{t+1}
  s.time = times_[1];
  failure_ = failure_ || fail != 0;
  if (residual) std::copy(residual_.begin(), residual_.begin() + nr, residual);
  FetchFrame();                                                                      // poses / sensors at x_t, like mjData after mj_step
  return costs_[0];                                                                  // CostValue(sensordata) at (x_t, u_t)
}

// ------------------------------------------------------------------ Task::Transition on the host
static void TrackingTransition(const MjpcHipModel& m, SimState& s, HostTask& t, const SimFrame&) {    // tracking.cc:223-267
  // int_data: [mode, motion start key, motion length, ...]; dbl_data[0]: reference_time_
  const double kFps = 30.0;
  int start = t.int_data[1], length = t.int_data[2];
  if (s.time == 0.0) t.dbl_data[0] = s.time;            // motion (re)start; the caller provides the motion's first key as the state
  double current_index = (s.time - t.dbl_data[0]) * kFps + start;
  int last_key_index = start + length - 1;
  double ci = std::min(std::max(current_index, 0.0), (double)last_key_index);
  int k0 = (int)std::floor(ci), k1 = std::min(k0 + 1, last_key_index);
  double w1 = ci - k0, w0 = 1.0 - w1;
  int n3 = 3 * m.nmocap;
  for (int b = 0; b < m.nmocap; b++)
    for (int k = 0; k < 3; k++) {
      double p0 = m.key_mpos[(size_t)n3 * k0 + 3 * b + k] * w0, p1 = m.key_mpos[(size_t)n3 * k1 + 3 * b + k] * w1;   // mju_scl, mju_scl, add
      s.mocap[7 * b + k] = p0 + p1;
    }
}
// ---- host side of the quadruped task between plan steps (what QuadrupedFlat::TransitionLocked does, quadruped.cc:224-390)
// Organised as a small state machine over tables instead of one long routine:
//   Restart        simulation time went backwards (reset)            -> phase bookkeeping restarts, exotic modes fall back
//   Admit          which requested mode changes are legal            -> kModeAdmits[from][to]
//   Rephase        cadence slider moved                              -> keep the gait phase continuous
//   PickGait       automatic gait by filtered com speed              -> speed bands kGaitBand[]
//   ApplyGait      gait changed (slider or automatic)                -> one row of kGaitRow[] into parameters / weights
//   SteerWalk      Walk mode: goal rides on a line or a circle       -> same closed form the device residual evaluates
//   Flip           entry snapshot / exit restore of the flip stunt   -> kFlipWeights[]
namespace {
// layouts of modelgen/tasks.py quadruped() == the enums in csrc/core.h
enum { QI_TORSO = 0, QI_HEAD = 1, QI_GOAL = 2, QI_MODE = 17 };
enum { QD_MODE_START = 0, QD_POSITION = 1, QD_HEADING = 4, QD_SPEED = 6, QD_ANGVEL = 7, QD_GROUND = 8, QD_ORIENT = 9, QD_GAIT = 13,
       QD_PHASE_START = 14, QD_PHASE_START_TIME = 15, QD_PHASE_VEL = 16, QD_FLIGHT_TIME = 19, QD_JUMP_TIME = 23, QD_LAND_TIME = 25 };
enum { P_GAIT = 0, P_SWITCH = 1, P_CADENCE = 2, P_AMPLITUDE = 3, P_DUTY = 4, P_WALK_SPEED = 5, P_WALK_TURN = 6 };
enum { W_UPRIGHT = 0, W_HEIGHT = 1, W_POSITION = 2, W_GAIT = 3, W_BALANCE = 4, W_EFFORT = 5, W_POSTURE = 6, W_COUNT = 7 };
enum Mode { kQuadruped = 0, kBiped, kWalk, kScramble, kFlip, kNumMode };
enum Gait { kStand = 0, kGaitWalk, kTrot, kCanter, kGallop, kNumGait };

// a requested mode is admitted from the current one?  Walk and Flip can only be entered from plain Quadruped (quadruped.cc:240-246)
const bool kModeAdmits[kNumMode][kNumMode] = {
    /* from Quadruped */ {true, true, true, true, true},
    /* from Biped     */ {true, true, false, true, false},
    /* from Walk      */ {true, true, true, true, false},
    /* from Scramble  */ {true, true, false, true, false},
    /* from Flip      */ {true, true, false, true, true}};
// duty ratio, cadence, amplitude | weights balance, upright, height          (quadruped.h:88-97)
struct GaitRow { double duty, cadence, amplitude, balance, upright, height; };
const GaitRow kGaitRow[kNumGait] = {{1, 1, 0, 0, 1, 1}, {0.75, 1, 0.03, 0, 1, 1}, {0.45, 2, 0.03, 0.2, 1, 1},
                                    {0.4, 4, 0.05, 0.03, 0.5, 0.2}, {0.3, 3.5, 0.10, 0.03, 0.2, 0.1}};
// automatic gait: com speed in (lo, hi] selects the gait (thresholds of quadruped.h:100-107; the Walk band is empty by design)
struct SpeedBand { double lo, hi; };
const SpeedBand kGaitBand[kNumGait] = {{0, 0.02}, {0.02, 0.02}, {0.02, 0.6}, {0.6, 2}, {2, 1e300}};
const double kSpeedFilterTime = 0.2, kGaitDwellTime = 1, kStraightAngvel = 0.01;
// cost weights during the flip, by term (Upright .. Posture)                (quadruped.cc:361-367)
const double kFlipWeights[W_COUNT] = {0.2, 5, 0, 0, 0, 0.005, 0.1};

long long BitsOf(double v) { long long i; std::memcpy(&i, &v, 8); return i; }       // select parameters travel as int64 bit patterns (utilities.cc:207-211)
double FromBits(long long i) { double v; std::memcpy(&v, &i, 8); return v; }

void Unit2(double v[2]) {                                  // mju_normalize semantics: a null vector becomes (1, 0)
  double n = std::sqrt(v[0] * v[0] + v[1] * v[1]);
  if (n < 1e-15) { v[0] = 1; v[1] = 0; } else { v[0] /= n; v[1] /= n; }
}

void QuatOfRotation(const double* R, double q[4]) {        // row-major rotation matrix -> unit quaternion (largest pivot branch)
  double tr = R[0] + R[4] + R[8];
  if (tr > 0) { double k = std::sqrt(tr + 1.0) * 2; q[0] = 0.25 * k; q[1] = (R[7] - R[5]) / k; q[2] = (R[2] - R[6]) / k; q[3] = (R[3] - R[1]) / k; }
  else if (R[0] > R[4] && R[0] > R[8]) { double k = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / k; q[1] = 0.25 * k; q[2] = (R[1] + R[3]) / k; q[3] = (R[2] + R[6]) / k; }
  else if (R[4] > R[8]) { double k = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / k; q[1] = (R[1] + R[3]) / k; q[2] = 0.25 * k; q[3] = (R[5] + R[7]) / k; }
  else { double k = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / k; q[1] = (R[2] + R[6]) / k; q[2] = (R[5] + R[7]) / k; q[3] = 0.25 * k; }
}

// Height of the nearest group-0 geom under `pos`: a ray straight down from 0.5 above it (Ground(), utilities.cc:538-556) against
// the planes, boxes and spheres of the model, posed from the simulator's kinematic frame.  No hit: the height of the query point.
double GroundHeight(const MjpcHipModel& m, const SimFrame& f, const double pos[3]) {
  const double start[3] = {pos[0], pos[1], pos[2] + 0.5};
  double best = -1;
  for (int g = 0; g < m.ngeom; g++) {
    if (m.geom_group[g] != 0) continue;
    int b = m.geom_bodyid[g];
    const double* X = f.xmat.data() + 9 * b; const double* gp = m.geom_pos + 3 * g; const double* gq = m.geom_quat + 4 * g;
    double w = gq[0], x = gq[1], y = gq[2], z = gq[3];
    double L[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                   2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    double R[9], c[3];
    for (int i = 0; i < 3; i++) {
      c[i] = f.xpos[3 * b + i] + X[3 * i] * gp[0] + X[3 * i + 1] * gp[1] + X[3 * i + 2] * gp[2];
      for (int j = 0; j < 3; j++) R[3 * i + j] = X[3 * i] * L[j] + X[3 * i + 1] * L[3 + j] + X[3 * i + 2] * L[6 + j];
    }
    // ray in the geom's frame: origin o, direction d = R^T (0, 0, -1)
    double rel[3] = {start[0] - c[0], start[1] - c[1], start[2] - c[2]}, o[3], d[3];
    for (int j = 0; j < 3; j++) { o[j] = R[j] * rel[0] + R[3 + j] * rel[1] + R[6 + j] * rel[2]; d[j] = -R[6 + j]; }
    const double* sz = m.geom_size + 3 * g;
    double hit = -1;
    if (m.geom_type[g] == MJPC_GEOM_PLANE) {
      if (d[2] < -1e-15) {
        double t = -o[2] / d[2];
        double px = o[0] + t * d[0], py = o[1] + t * d[1];
        if (t >= 0 && (sz[0] <= 0 || std::fabs(px) <= sz[0]) && (sz[1] <= 0 || std::fabs(py) <= sz[1])) hit = t;
      }
    } else if (m.geom_type[g] == MJPC_GEOM_SPHERE) {
      double A = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], B = d[0] * o[0] + d[1] * o[1] + d[2] * o[2];
      double C = o[0] * o[0] + o[1] * o[1] + o[2] * o[2] - sz[0] * sz[0], det = B * B - A * C;
      if (det >= 1e-15 && A >= 1e-15) { double r = std::sqrt(det), t0 = (-B - r) / A, t1 = (-B + r) / A; hit = t0 >= 0 ? t0 : (t1 >= 0 ? t1 : -1); }
    } else if (m.geom_type[g] == MJPC_GEOM_BOX) {
      for (int i = 0; i < 3; i++) {
        if (std::fabs(d[i]) <= 1e-15) continue;
        int j = (i + 1) % 3, k = (i + 2) % 3;
        for (int side = -1; side <= 1; side += 2) {
          double t = (side * sz[i] - o[i]) / d[i];
          if (t < 0) continue;
          if (std::fabs(o[j] + t * d[j]) <= sz[j] && std::fabs(o[k] + t * d[k]) <= sz[k] && (hit < 0 || t < hit)) hit = t;
        }
      }
    }
    if (hit >= 0 && (best < 0 || hit < best)) best = hit;
  }
  return best < 0 ? start[2] : start[2] - best;
}
}  // namespace

void QuadrupedTransition::operator()(const MjpcHipModel& model, SimState& s, HostTask& t, const SimFrame& f) {
  std::vector<double>& P = t.parameters; std::vector<double>& W = t.weight; std::vector<double>& D = t.dbl_data;
  const double now = s.time;
  const int torso = t.int_data[QI_TORSO];
  double* goal = s.mocap.data() + 7 * t.int_data[QI_GOAL];

  // Restart
  if (last_transition_time == -1 || now < last_transition_time) {
    if (mode != kQuadruped && mode != kBiped) mode = kQuadruped;
    D[QD_PHASE_START] = D[QD_PHASE_START_TIME] = last_transition_time = now;
  }
  // Admit
  if (mode != current_mode && !kModeAdmits[current_mode][mode]) mode = kQuadruped;
  // Rephase: the phase accumulated so far becomes the new origin, then the new angular rate applies
  const double rate = 2 * 3.14159265358979323846 * P[P_CADENCE];
  if (rate != D[QD_PHASE_VEL]) {
    D[QD_PHASE_START] += (now - D[QD_PHASE_START_TIME]) * D[QD_PHASE_VEL];
    D[QD_PHASE_START_TIME] = now;
    D[QD_PHASE_VEL] = rate;
  }
  // PickGait: first-order low-pass of the horizontal com velocity, then the speed band it falls in
  {
    const double* v = f.subtree_linvel.data() + 3 * torso;
    const double keep = std::exp(-(now - last_transition_time) / kSpeedFilterTime);
    com_vel[0] = keep * com_vel[0] + (1 - keep) * v[0];
    com_vel[1] = keep * com_vel[1] + (1 - keep) * v[1];
    if (mode == kBiped) P[P_GAIT] = FromBits(kTrot);
    else if (BitsOf(P[P_SWITCH]) != 0) {
      const double speed = std::sqrt(com_vel[0] * com_vel[0] + com_vel[1] * com_vel[1]);
      for (int g = (mode == kScramble ? kGaitWalk : kStand); g < kNumGait; g++) {
        if (speed > kGaitBand[g].lo && speed <= kGaitBand[g].hi && std::fabs(gait_switch_time - now) > kGaitDwellTime) {
          P[P_GAIT] = FromBits(g); gait_switch_time = now;
          break;                                       // the bands are disjoint
        }
      }
    }
  }
  // ApplyGait (bit patterns are compared: the selector is an integer carried in a double)
  if (BitsOf(P[P_GAIT]) != BitsOf(D[QD_GAIT])) {
    D[QD_GAIT] = P[P_GAIT];
    const GaitRow& row = kGaitRow[current_mode == kBiped ? (int)kTrot : (int)BitsOf(D[QD_GAIT])];
    P[P_DUTY] = row.duty; P[P_CADENCE] = row.cadence; P[P_AMPLITUDE] = row.amplitude;
    W[W_BALANCE] = row.balance; W[W_UPRIGHT] = row.upright; W[W_HEIGHT] = row.height;
  }
  // SteerWalk
  if (mode == kWalk) {
    const double turn = P[P_WALK_TURN], speed = P[P_WALK_SPEED];
    if (mode != current_mode || D[QD_ANGVEL] != turn || D[QD_SPEED] != speed) {
      // new trajectory: centre = torso (straight line) or the point speed / turn to the robot's left (circle); heading = goal - centre
      const double* X = f.xmat.data() + 9 * torso;
      double ahead[2] = {X[0], X[3]};
      Unit2(ahead);
      double centre[2] = {f.xpos[3 * torso], f.xpos[3 * torso + 1]};
      if (std::fabs(turn) > kStraightAngvel) { const double radius = speed / turn; centre[0] += radius * -ahead[1]; centre[1] += radius * ahead[0]; }
      D[QD_MODE_START] = now; D[QD_SPEED] = speed; D[QD_ANGVEL] = turn;
      D[QD_POSITION] = centre[0]; D[QD_POSITION + 1] = centre[1];
      D[QD_HEADING] = goal[0] - centre[0]; D[QD_HEADING + 1] = goal[1] - centre[1];
    }
    const double along = now - D[QD_MODE_START];
    if (std::fabs(D[QD_ANGVEL]) < kStraightAngvel) {
      double dir[2] = {D[QD_HEADING], D[QD_HEADING + 1]};
      Unit2(dir);
      goal[0] = D[QD_POSITION] + D[QD_HEADING] + along * D[QD_SPEED] * dir[0];
      goal[1] = D[QD_POSITION + 1] + D[QD_HEADING + 1] + along * D[QD_SPEED] * dir[1];
    } else {
      const double a = along * D[QD_ANGVEL], ca = std::cos(a), sa = std::sin(a);
      goal[0] = ca * D[QD_HEADING] - sa * D[QD_HEADING + 1] + D[QD_POSITION];
      goal[1] = sa * D[QD_HEADING] + ca * D[QD_HEADING + 1] + D[QD_POSITION + 1];
    }
  }
  // Flip
  if (mode == kFlip) {
    if (mode != current_mode) {                        // entry: snapshot pose, ground height under the com, weights; stunt weights on
      D[QD_MODE_START] = now;
      QuatOfRotation(f.xmat.data() + 9 * torso, D.data() + QD_ORIENT);
      D[QD_GROUND] = GroundHeight(model, f, f.subtree_com.data() + 3 * torso);
      save_weight = W; save_gait_switch = P[P_SWITCH];
      for (int k = 0; k < W_COUNT; k++) W[k] = kFlipWeights[k];
      P[P_SWITCH] = FromBits(0);
    }
    if (now - D[QD_MODE_START] >= D[QD_JUMP_TIME] + D[QD_FLIGHT_TIME] + D[QD_LAND_TIME]) {      // exit: back on the feet
      mode = kQuadruped;
      W = save_weight; P[P_SWITCH] = save_gait_switch;
      goal[0] = f.site_xpos[3 * t.int_data[QI_HEAD]]; goal[1] = f.site_xpos[3 * t.int_data[QI_HEAD] + 1];
    }
  }
  current_mode = mode;
  t.int_data[QI_MODE] = mode;
  last_transition_time = now;
}

// ---- ShadowReorient::TransitionLocked (hand.cc:90-119): a cube lying still on the floor is put back over the hand.
// The reference reads mjData.contact for a cube-floor contact; here the same fact comes from the pose: a box touches a plane
// exactly when its lowest corner is at or below it (the plane-box collider's own criterion, margin 0).
// int_data = [palm site, cube body, goal body, key]
static void HandTransition(const MjpcHipModel& m, SimState& s, HostTask& t, const SimFrame& f) {
  const int cube = t.int_data[1];
  int cube_geom = -1, floor_geom = -1;
  for (int g = 0; g < m.ngeom; g++) {
    if (m.geom_bodyid[g] == cube && m.geom_type[g] == MJPC_GEOM_BOX && cube_geom < 0) cube_geom = g;
    if (m.geom_bodyid[g] == 0 && m.geom_type[g] == MJPC_GEOM_PLANE && floor_geom < 0) floor_geom = g;
  }
  if (cube_geom < 0 || floor_geom < 0 || m.body_jntnum[cube] != 1) return;
  const int jnt = m.body_jntadr[cube];
  if (m.jnt_type[jnt] != MJPC_JNT_FREE) return;
  const int qa = m.jnt_qposadr[jnt], da = m.jnt_dofadr[jnt];
  // lowest corner of the cube along the floor normal (the floor is a world-fixed plane: normal = z axis of its quaternion)
  const double* fq = m.geom_quat + 4 * floor_geom;
  const double n[3] = {2 * (fq[1] * fq[3] + fq[0] * fq[2]), 2 * (fq[2] * fq[3] - fq[0] * fq[1]), 1 - 2 * (fq[1] * fq[1] + fq[2] * fq[2])};
  const double* X = f.xmat.data() + 9 * cube; const double* c = f.xpos.data() + 3 * cube; const double* h = m.geom_size + 3 * cube_geom;
  double reach = 0, centre = 0;
  for (int k = 0; k < 3; k++) {
    reach += h[k] * std::fabs(n[0] * X[k] + n[1] * X[3 + k] + n[2] * X[6 + k]);      // box half-extent along the normal
    centre += n[k] * (c[k] - m.geom_pos[3 * floor_geom + k]);
  }
  const bool on_floor = centre - reach <= 0.0;
  const double* v = s.state.data() + m.nq + da;                                      // free joint: linear velocity in the world frame
  const double speed = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  if (on_floor && speed < 0.001) {
    for (int k = 0; k < 7; k++) s.state[qa + k] = m.qpos0[qa + k];
    for (int k = 0; k < 6; k++) s.state[m.nq + da + k] = 0.0;
  }
  (void)t;
}

// ---- QuadrupedHill::TransitionLocked (quadruped.cc:776-812): when the trunk is within 0.15 of the goal pose (position norm and
// 1 - |<q_goal, q_trunk>|) the goal moves on to the next stage key, wrapping at the end.
// int_data = [trunk body, sites FR FL RR RL, stage]; dbl_data = stage goals [nstage][7]
static void HillTransition(const MjpcHipModel& m, SimState& s, HostTask& t, const SimFrame& f) {
  const int trunk = t.int_data[0], nstage = (int)t.dbl_data.size() / 7;
  if (nstage < 1 || m.nmocap < 1) return;
  int stage = t.int_data[5];
  const double* goal = s.mocap.data();
  double q[4], err = 0, dot = 0;
  QuatOfRotation(f.xmat.data() + 9 * trunk, q);
  for (int k = 0; k < 3; k++) { double e = f.xpos[3 * trunk + k] - goal[k]; err += e * e; }
  for (int k = 0; k < 4; k++) dot += goal[3 + k] * q[k];
  const double tolerance = 1.5e-1;
  if (std::sqrt(err) <= tolerance && 1.0 - std::fabs(dot) <= tolerance) stage = (stage + 1) % nstage;
  t.int_data[5] = stage;
  for (int k = 0; k < 7; k++) s.mocap[k] = t.dbl_data[7 * stage + k];
}

// Particle::TransitionLocked (particle.cc:52-60): the mocap goal follows the Lissajous curve the residual tracks
static void ParticleTransition(const MjpcHipModel&, SimState& s, HostTask&, const SimFrame&) {
  if (s.mocap.size() >= 2) { s.mocap[0] = 0.25 * std::sin(s.time); s.mocap[1] = 0.25 * std::cos(s.time / 3.14159265358979323846); }
}

// Quadrotor::TransitionLocked (quadrotor.cc:63-95), mode 0 ("Loop"): within 0.5 m of the goal -> next keyframe position
static void QuadrotorTransition(const MjpcHipModel& m, SimState& s, HostTask& t, const SimFrame& f) {
  const int body = t.int_data[0], nstage = (int)t.dbl_data.size() / 7;
  int stage = t.int_data[1];
  double err = 0;
  for (int k = 0; k < 3; k++) { double e = f.subtree_com[3 * body + k] - s.mocap[k]; err += e * e; }      // one free body: its subtree com is xipos
  if (std::sqrt(err) <= 5.0e-1) stage = (stage + 1) % nstage;
  t.int_data[1] = stage;
  for (int k = 0; k < 7; k++) s.mocap[k] = t.dbl_data[7 * stage + k];
  (void)m;
}

// Swimmer::TransitionLocked (swimmer.cc:52-61): nose within 4 cm of the target -> a new target in [-0.8, 0.8]^2 (the reference draws
// it from absl::BitGen; here a counter-based hash, so that runs are reproducible)
static void SwimmerTransition(const MjpcHipModel& m, SimState& s, HostTask& t, const SimFrame& f) {
  const int g = t.int_data[0];
  (void)g; (void)m;
  // the harness frame carries body / site poses, not geoms: the nose geom sits at (0, -0.06, 0) of the head (body 1)
  const double *R = f.xmat.data() + 9 * 1, *p = f.xpos.data() + 3 * 1;
  double nose[2] = {p[0] + R[1] * -0.06, p[1] + R[4] * -0.06};
  double dx = s.mocap[0] - nose[0], dy = s.mocap[1] - nose[1];
  if (std::sqrt(dx * dx + dy * dy) < 0.04) {
    unsigned k = (unsigned)++t.int_data[1];
    auto u = [&](unsigned salt) { unsigned h = (k * 2654435761u) ^ (salt * 40503u); h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; return -0.8 + 1.6 * (h & 0xFFFFFF) / 16777216.0; };
    s.mocap[0] = u(1); s.mocap[1] = u(2);
  }
}

// Interact::TransitionLocked (interact.cc:191-197): a mode change installs that mode's row of default_weights (interact.h:40-45)
static const double kInteractWeights[4][13] = {{10, 10, 5, 5, 0, 20, 30, 0, 0, 0, 0.01, .1, 80.}, {10, 0, 1, 1, 80, 0, 0, 100, 0, 0, 0.01, 0.025, 0.},
                                               {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.01, .8, 80.}, {0, 0, 0, 0, 0, 0, 0, 0, 0, 50, 20, .025, 80.}};

TransitionFn TransitionForTask(int task_id, int mode, double mode_time) {
  if (task_id == MJPC_TASK_HUMANOID_INTERACT) {
    auto current = std::make_shared<int>(0);                 // residual_.current_task_mode_ starts as kSitting
    return [current, mode, mode_time](const MjpcHipModel&, SimState& s, HostTask& t, const SimFrame&) {
      int want = s.time >= mode_time ? mode : 0;
      if (want < 0 || want > 3) want = 0;
      if (*current != want) { *current = want; for (size_t k = 0; k < t.weight.size() && k < 13; k++) t.weight[k] = kInteractWeights[want][k]; }
    };
  }
  if (task_id == MJPC_TASK_SWIMMER) return SwimmerTransition;
  if (task_id == MJPC_TASK_QUADROTOR) return QuadrotorTransition;
  if (task_id == MJPC_TASK_PARTICLE_TIMEVARYING) return ParticleTransition;
  if (task_id == MJPC_TASK_HUMANOID_TRACK) return TrackingTransition;
  if (task_id == MJPC_TASK_QUADRUPED_HILL) return HillTransition;
  if (task_id == MJPC_TASK_SHADOW_REORIENT) return HandTransition;
  if (task_id == MJPC_TASK_QUADRUPED) {
    auto q = std::make_shared<QuadrupedTransition>();
    auto switched = std::make_shared<bool>(false);
    return [q, switched, mode, mode_time](const MjpcHipModel& m, SimState& s, HostTask& t, const SimFrame& f) {
      if (!*switched && s.time >= mode_time && (mode_time > 0 || mode == 0 || mode == 1)) { q->mode = mode; *switched = true; }   // the GUI's mode selector
      (*q)(m, s, t, f);
    };
  }
  return [](const MjpcHipModel&, SimState&, HostTask&, const SimFrame&) {};
}

// ------------------------------------------------------------------ planner adapters
PlannerOps Ops(SamplingPlanner& p) {
  PlannerOps o;
  o.SetState = [&p](const SimState& s) { p.SetState(s.state.data(), s.mocap.data(), s.userdata.data(), s.time); };
  o.OptimizePolicy = [&p](int h) { p.OptimizePolicy(h); };
  o.ActionFromPolicy = [&p](double* a, double t) { p.ActionFromPolicy(a, nullptr, t, false); };
  o.SetTask = [&p](const MjpcHipTask* t) { p.SetTask(t); };
  return o;
}
PlannerOps Ops(CrossEntropyPlanner& p) {
  PlannerOps o;
  o.SetState = [&p](const SimState& s) { p.SetState(s.state.data(), s.mocap.data(), s.userdata.data(), s.time); };
  o.OptimizePolicy = [&p](int h) { p.OptimizePolicy(h); };
  o.ActionFromPolicy = [&p](double* a, double t) { p.ActionFromPolicy(a, nullptr, t, false); };
  o.SetTask = [&p](const MjpcHipTask* t) { p.SetTask(t); };
  return o;
}

// ------------------------------------------------------------------ the loop (testspeed.cc:97-116)
TestspeedResult SynchronousPlanningCost(const MjpcHipModel& model, HostTask& task, PlannerOps planner, Simulator& sim, SimState& s,
                                        int horizon, int steps_per_planning_iteration, double total_time, const TransitionFn& transition) {
  TestspeedResult r;
  r.total_steps = (int)std::ceil(total_time / model.timestep);
  std::vector<double> ctrl(std::max(model.nu, 1), 0.0);
  SimState planning_state;
  if (!sim.frame().valid) sim.Forward(s);                 // testspeed.cc:75: mjData is forwarded once before the loop
  auto loop_start = std::chrono::steady_clock::now();
  for (int i = 0; i < r.total_steps; i++) {
    transition(model, s, task, sim.frame());              // agent.ActiveTask()->Transition(model, data)
    planning_state = s;                                   // agent.state.Set(model, data)
    planner.ActionFromPolicy(ctrl.data(), s.time);
    double cost = sim.Step(s, ctrl.data());               // mj_step; CostValue(data->sensordata)
    r.total_cost += cost;
    r.cost_per_step.push_back(cost);
    if (i % steps_per_planning_iteration == 0) {          // agent.PlanIteration: fresh ResidualFn copy, state from before the step
      auto t0 = std::chrono::steady_clock::now();
      const MjpcHipTask* v = task.view();
      planner.SetTask(v); sim.SetTask(v);
      planner.SetState(planning_state);
      planner.OptimizePolicy(horizon);
      r.plan_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      r.plan_steps++;
    }
  }
  r.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - loop_start).count();
  r.realtime_factor = r.wall_seconds > 0 ? total_time / r.wall_seconds : 0;
  r.average_cost = r.total_steps ? r.total_cost / r.total_steps : 0;
  r.failure = sim.failed();
  return r;
}

}  // namespace mjpc_hip

// ====================================================================== flat C view (ctypes tests / Python front end)
extern "C" {
// planner_kind: 0 = SamplingPlanner handle (mjpc_planner_create), 1 = CrossEntropyPlanner handle (mjpc_cem_create).
// state/mocap are in-out (final simulator state); cost_per_step[total_steps] optional.  Returns the total cost
// (testspeed.cc:128) and fills out[6] = {average_cost, wall_seconds, realtime_factor, plan_seconds, plan_steps, failure}.
double mjpc_testspeed_run(const MjpcHipModel* model, const MjpcHipTask* task, void* planner, int planner_kind, double* state, double* mocap,
                          double time0, int horizon, int steps_per_planning_iteration, double total_time, int device,
                          double* cost_per_step, double* out, int mode, double mode_time, double* task_parameters_io) {
  using namespace mjpc_hip;
  HostTask ht(*task);
  Simulator sim(model, ht.view(), device);
  SimState s;
  s.state.assign(state, state + model->nq + model->nv + model->na);
  s.mocap.assign(7 * (size_t)model->nmocap, 0.0);
  if (mocap) std::copy(mocap, mocap + 7 * model->nmocap, s.mocap.begin());
  s.userdata.assign((size_t)std::max(model->nuserdata, 1), 0.0);
  s.time = time0;
  PlannerOps ops = planner_kind == 0 ? Ops(*(SamplingPlanner*)planner) : Ops(*(CrossEntropyPlanner*)planner);
  TestspeedResult r = SynchronousPlanningCost(*model, ht, ops, sim, s, horizon, steps_per_planning_iteration, total_time,
                                              TransitionForTask(task->task_id, mode, mode_time));
  if (task_parameters_io) std::copy(ht.parameters.begin(), ht.parameters.end(), task_parameters_io);   // what Transition left behind
  std::copy(s.state.begin(), s.state.end(), state);
  if (mocap) std::copy(s.mocap.begin(), s.mocap.begin() + 7 * model->nmocap, mocap);
  if (cost_per_step) std::copy(r.cost_per_step.begin(), r.cost_per_step.end(), cost_per_step);
  if (out) { out[0] = r.average_cost; out[1] = r.wall_seconds; out[2] = r.realtime_factor; out[3] = r.plan_seconds; out[4] = r.plan_steps; out[5] = r.failure ? 1 : 0; }
  return r.total_cost;
}
}  // extern "C"
