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
// ---- QuadrupedFlat::TransitionLocked (quadruped.cc:224-390)
namespace {
// layouts of modelgen/tasks.py quadruped() == the enums in csrc/core.h
enum { QI_TORSO = 0, QI_HEAD = 1, QI_GOAL = 2, QI_MODE = 17 };
enum { QD_MODE_START = 0, QD_POSITION = 1, QD_HEADING = 4, QD_SPEED = 6, QD_ANGVEL = 7, QD_GROUND = 8, QD_ORIENT = 9, QD_GAIT = 13,
       QD_PHASE_START = 14, QD_PHASE_START_TIME = 15, QD_PHASE_VEL = 16, QD_FLIGHT_TIME = 19, QD_JUMP_TIME = 23, QD_LAND_TIME = 25 };
enum { P_GAIT = 0, P_SWITCH = 1, P_CADENCE = 2, P_AMPLITUDE = 3, P_DUTY = 4, P_WALK_SPEED = 5, P_WALK_TURN = 6 };
enum { W_UPRIGHT = 0, W_HEIGHT = 1, W_POSITION = 2, W_GAIT = 3, W_BALANCE = 4, W_EFFORT = 5, W_POSTURE = 6 };
enum { kModeQuadruped = 0, kModeBiped, kModeWalk, kModeScramble, kModeFlip };
enum { kGaitStand = 0, kGaitWalk, kGaitTrot, kGaitCanter, kGaitGallop, kNumGait };
const double kGaitParam[kNumGait][6] = {{1, 1, 0, 0, 1, 1}, {0.75, 1, 0.03, 0, 1, 1}, {0.45, 2, 0.03, 0.2, 1, 1},
                                        {0.4, 4, 0.05, 0.03, 0.5, 0.2}, {0.3, 3.5, 0.10, 0.03, 0.2, 0.1}};      // quadruped.h:88-97
const double kGaitAuto[kNumGait] = {0, 0.02, 0.02, 0.6, 2};                                                       // quadruped.h:100-107
const double kAutoGaitFilter = 0.2, kAutoGaitMinTime = 1, kMinAngvel = 0.01;
long long AsInt(double v) { long long i; std::memcpy(&i, &v, 8); return i; }        // ReinterpretAsInt (utilities.cc:207-211)
double AsDouble(long long i) { double v; std::memcpy(&v, &i, 8); return v; }
}  // namespace

void QuadrupedTransition::operator()(const MjpcHipModel& model, SimState& s, HostTask& t, const SimFrame& f) {
  std::vector<double>& P = t.parameters; std::vector<double>& W = t.weight; std::vector<double>& D = t.dbl_data;
  double time = s.time;
  int torso = t.int_data[QI_TORSO];
  // ---------- handle mjData reset ----------
  if (time < last_transition_time || last_transition_time == -1) {
    if (mode != kModeQuadruped && mode != kModeBiped) mode = kModeQuadruped;
    last_transition_time = D[QD_PHASE_START_TIME] = D[QD_PHASE_START] = time;
  }
  // ---------- prevent forbidden mode transitions ----------
  if (mode != current_mode && current_mode != kModeQuadruped) {
    if (mode == kModeWalk || mode == kModeFlip) mode = kModeQuadruped;
  }
  // ---------- handle phase velocity change ----------
  double phase_velocity = 2 * 3.14159265358979323846 * P[P_CADENCE];
  if (phase_velocity != D[QD_PHASE_VEL]) {
    D[QD_PHASE_START] = D[QD_PHASE_START] + (time - D[QD_PHASE_START_TIME]) * D[QD_PHASE_VEL];      // GetPhase(time)
    D[QD_PHASE_START_TIME] = time;
    D[QD_PHASE_VEL] = phase_velocity;
  }
  // ---------- automatic gait switching ----------
  const double* comvel = f.subtree_linvel.data() + 3 * torso;
  double beta = std::exp(-(time - last_transition_time) / kAutoGaitFilter);
  com_vel[0] = beta * com_vel[0] + (1 - beta) * comvel[0];
  com_vel[1] = beta * com_vel[1] + (1 - beta) * comvel[1];
  int auto_switch = (int)AsInt(P[P_SWITCH]);
  if (mode == kModeBiped) {
    P[P_GAIT] = AsDouble(kGaitTrot);
  } else if (auto_switch) {
    double com_speed = std::sqrt(com_vel[0] * com_vel[0] + com_vel[1] * com_vel[1]);
    for (int gait = 0; gait < kNumGait; gait++) {
      if (mode == kModeScramble && gait == kGaitStand) continue;
      bool lower = com_speed > kGaitAuto[gait];
      bool upper = gait == kGaitGallop || com_speed <= kGaitAuto[gait + 1];
      bool wait = std::fabs(gait_switch_time - time) > kAutoGaitMinTime;
      if (lower && upper && wait) { P[P_GAIT] = AsDouble(gait); gait_switch_time = time; }
    }
  }
  // ---------- handle gait switch, manual or auto ----------
  double gait_selection = P[P_GAIT];
  if (AsInt(gait_selection) != AsInt(D[QD_GAIT])) {          // compared as doubles in the reference; bit patterns here (denormals)
    D[QD_GAIT] = gait_selection;
    int gait = current_mode == kModeBiped ? kGaitTrot : (int)AsInt(D[QD_GAIT]);                     // GetGait()
    P[P_DUTY] = kGaitParam[gait][0]; P[P_CADENCE] = kGaitParam[gait][1]; P[P_AMPLITUDE] = kGaitParam[gait][2];
    W[W_BALANCE] = kGaitParam[gait][3]; W[W_UPRIGHT] = kGaitParam[gait][4]; W[W_HEIGHT] = kGaitParam[gait][5];
  }
  // ---------- Walk ----------
  double* goal_pos = s.mocap.data() + 7 * t.int_data[QI_GOAL];
  if (mode == kModeWalk) {
    double angvel = P[P_WALK_TURN], speed = P[P_WALK_SPEED];
    const double* torso_xmat = f.xmat.data() + 9 * torso;
    double forward[2] = {torso_xmat[0], torso_xmat[3]};
    { double n = std::sqrt(forward[0] * forward[0] + forward[1] * forward[1]);
      if (n < 1e-15) { forward[0] = 1; forward[1] = 0; } else { forward[0] /= n; forward[1] /= n; } }
    double leftward[2] = {-forward[1], forward[0]};
    if (mode != current_mode || D[QD_ANGVEL] != angvel || D[QD_SPEED] != speed) {
      D[QD_MODE_START] = time;
      D[QD_SPEED] = speed; D[QD_ANGVEL] = angvel;
      double axis[2] = {f.xpos[3 * torso], f.xpos[3 * torso + 1]};
      if (std::fabs(angvel) > kMinAngvel) {
        double d = speed / angvel;
        axis[0] += d * leftward[0]; axis[1] += d * leftward[1];
      }
      D[QD_POSITION] = axis[0]; D[QD_POSITION + 1] = axis[1];
      D[QD_HEADING] = goal_pos[0] - axis[0]; D[QD_HEADING + 1] = goal_pos[1] - axis[1];
    }
    // move goal: ResidualFn::Walk (quadruped.cc:627-643)
    double wt = time - D[QD_MODE_START];
    if (std::fabs(D[QD_ANGVEL]) < kMinAngvel) {
      double fw[2] = {D[QD_HEADING], D[QD_HEADING + 1]};
      double n = std::sqrt(fw[0] * fw[0] + fw[1] * fw[1]);
      if (n < 1e-15) { fw[0] = 1; fw[1] = 0; } else { fw[0] /= n; fw[1] /= n; }
      goal_pos[0] = D[QD_POSITION] + D[QD_HEADING] + wt * D[QD_SPEED] * fw[0];
      goal_pos[1] = D[QD_POSITION + 1] + D[QD_HEADING + 1] + wt * D[QD_SPEED] * fw[1];
    } else {
      double angle = wt * D[QD_ANGVEL], cs = std::cos(angle), sn = std::sin(angle);
      goal_pos[0] = cs * D[QD_HEADING] - sn * D[QD_HEADING + 1] + D[QD_POSITION];
      goal_pos[1] = sn * D[QD_HEADING] + cs * D[QD_HEADING + 1] + D[QD_POSITION + 1];
    }
  }
  // ---------- Flip ----------
  if (mode == kModeFlip) {
    if (mode != current_mode) {
      D[QD_MODE_START] = time;
      // torso orientation from its rotation matrix (mjData.xquat in the reference)
      const double* R = f.xmat.data() + 9 * torso;
      double q[4]; double tr = R[0] + R[4] + R[8];
      if (tr > 0) { double sq = std::sqrt(tr + 1.0) * 2; q[0] = 0.25 * sq; q[1] = (R[7] - R[5]) / sq; q[2] = (R[2] - R[6]) / sq; q[3] = (R[3] - R[1]) / sq; }
      else if (R[0] > R[4] && R[0] > R[8]) { double sq = std::sqrt(1.0 + R[0] - R[4] - R[8]) * 2; q[0] = (R[7] - R[5]) / sq; q[1] = 0.25 * sq; q[2] = (R[1] + R[3]) / sq; q[3] = (R[2] + R[6]) / sq; }
      else if (R[4] > R[8]) { double sq = std::sqrt(1.0 + R[4] - R[0] - R[8]) * 2; q[0] = (R[2] - R[6]) / sq; q[1] = (R[1] + R[3]) / sq; q[2] = 0.25 * sq; q[3] = (R[5] + R[7]) / sq; }
      else { double sq = std::sqrt(1.0 + R[8] - R[0] - R[4]) * 2; q[0] = (R[3] - R[1]) / sq; q[1] = (R[2] + R[6]) / sq; q[2] = (R[5] + R[7]) / sq; q[3] = 0.25 * sq; }
      for (int k = 0; k < 4; k++) D[QD_ORIENT + k] = q[k];
      D[QD_GROUND] = 0.0;                       // Ground() ray cast: the flat task's floor plane is z = 0
      save_weight = W; save_gait_switch = P[P_SWITCH];
      W[W_UPRIGHT] = 0.2; W[W_HEIGHT] = 5; W[W_POSITION] = 0; W[W_GAIT] = 0; W[W_BALANCE] = 0; W[W_EFFORT] = 0.005; W[W_POSTURE] = 0.1;
      P[P_SWITCH] = AsDouble(0);
    }
    double flip_time = time - D[QD_MODE_START];
    if (flip_time >= D[QD_JUMP_TIME] + D[QD_FLIGHT_TIME] + D[QD_LAND_TIME]) {
      mode = kModeQuadruped;
      W = save_weight; P[P_SWITCH] = save_gait_switch;
      goal_pos[0] = f.site_xpos[3 * t.int_data[QI_HEAD]]; goal_pos[1] = f.site_xpos[3 * t.int_data[QI_HEAD] + 1];
    }
  }
  current_mode = mode;
  t.int_data[QI_MODE] = mode;
  last_transition_time = time;
  (void)model;
}

TransitionFn TransitionForTask(int task_id, int mode, double mode_time) {
  if (task_id == MJPC_TASK_HUMANOID_TRACK) return TrackingTransition;
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
