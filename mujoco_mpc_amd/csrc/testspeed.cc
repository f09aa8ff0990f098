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
  return costs_[0];                                                                  // CostValue(sensordata) at (x_t, u_t)
}

// ------------------------------------------------------------------ Task::Transition on the host
static void TrackingTransition(const MjpcHipModel& m, SimState& s, HostTask& t) {    // tracking.cc:223-267
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
TransitionFn TransitionForTask(int task_id) {
  if (task_id == MJPC_TASK_HUMANOID_TRACK) return TrackingTransition;
  return [](const MjpcHipModel&, SimState&, HostTask&) {};
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
  auto loop_start = std::chrono::steady_clock::now();
  for (int i = 0; i < r.total_steps; i++) {
    transition(model, s, task);                           // agent.ActiveTask()->Transition(model, data)
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
                          double* cost_per_step, double* out) {
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
                                              TransitionForTask(task->task_id));
  std::copy(s.state.begin(), s.state.end(), state);
  if (mocap) std::copy(s.mocap.begin(), s.mocap.begin() + 7 * model->nmocap, mocap);
  if (cost_per_step) std::copy(r.cost_per_step.begin(), r.cost_per_step.end(), cost_per_step);
  if (out) { out[0] = r.average_cost; out[1] = r.wall_seconds; out[2] = r.realtime_factor; out[3] = r.plan_seconds; out[4] = r.plan_steps; out[5] = r.failure ? 1 : 0; }
  return r.total_cost;
}
}  // extern "C"
