// planner.cc — C++ host side above the C ABI (include/mjpc_hip_planner.h) + a flat C wrapper for the tests.
// Restates, with the rollouts forwarded to the HIP engine:
//   mjpc/spline/spline.cc:103-277            TimeSpline::Sample / DiscardBefore / AddNode / Slope
//   mjpc/planners/sampling/policy.cc:30-78   SamplingPolicy
//   mjpc/planners/sampling/planner.cc:40-310,525-534   SamplingPlanner host logic
#include "../../include/mjpc_hip_planner.h"
#include "../../include/mjpc_hip_planner_c.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <limits>
#include <numeric>

namespace mjpc_hip {

// ------------------------------------------------------------------ error convention
// The reference aborts through mju_error* on configuration errors (planner.cc:69-72); same here, unless a handler
// is installed (the Python tests install one that records the message instead of killing the interpreter).
static void (*g_error_handler)(const char*) = nullptr;
static void Fatal(const char* msg) {
  if (g_error_handler) { g_error_handler(msg); return; }
  std::fprintf(stderr, "mjpc_hip planner error: %s\n", msg);
  std::abort();
}

// ------------------------------------------------------------------ TimeSpline
TimeSpline::TimeSpline(int dim, SplineInterpolation interpolation, int) : interpolation_(interpolation), dim_(dim) {}
void TimeSpline::Reserve(int) {}
void TimeSpline::Clear() { times_.clear(); values_.clear(); }

double* TimeSpline::AddNode(double time, const double* new_values) {
  // spline.cc:203-238: only before the first or after the last node
  if (!(times_.empty() || time > times_.back() || time < times_.front())) {
    Fatal("Adding nodes to the middle of the spline isn't supported.");
    return nullptr;
  }
  std::vector<double> v(dim_, 0.0);
  if (new_values) std::copy(new_values, new_values + dim_, v.begin());
  if (times_.empty() || time > times_.back()) {
    times_.push_back(time); values_.push_back(std::move(v));
    return values_.back().data();
  }
  times_.push_front(time); values_.push_front(std::move(v));
  return values_.front().data();
}

int TimeSpline::DiscardBefore(double time) {   // spline.cc:164-188
  auto last_node = std::upper_bound(times_.begin(), times_.end(), time);
  if (last_node == times_.begin()) return 0;
  int keep_nodes = interpolation_ == kCubicSpline ? 1 : 0;
  last_node--;
  while (last_node != times_.begin() && keep_nodes) { last_node--; keep_nodes--; }
  int nodes_to_remove = (int)(last_node - times_.begin());
  times_.erase(times_.begin(), last_node);
  values_.erase(values_.begin(), values_.begin() + nodes_to_remove);
  return nodes_to_remove;
}

double TimeSpline::Slope(int node_index, int value_index) const {   // spline.cc:259-277
  if (node_index == 0)
    return (values_[1][value_index] - values_[0][value_index]) / (times_[1] - times_[0]);
  if (node_index == (int)times_.size() - 1)
    return (values_[node_index][value_index] - values_[node_index - 1][value_index]) / (times_[node_index] - times_[node_index - 1]);
  return 0.5 * (values_[node_index + 1][value_index] - values_[node_index][value_index]) / (times_[node_index + 1] - times_[node_index]) +
         0.5 * (values_[node_index][value_index] - values_[node_index - 1][value_index]) / (times_[node_index] - times_[node_index - 1]);
}

void TimeSpline::Sample(double time, double* values) const {   // spline.cc:103-156
  if (times_.empty()) { std::fill(values, values + dim_, 0.0); return; }
  auto upper = std::upper_bound(times_.begin(), times_.end(), time);
  if (upper == times_.end()) { const auto& n = values_[times_.size() - 1]; std::copy(n.begin(), n.end(), values); return; }
  if (upper == times_.begin()) { const auto& n = values_[0]; std::copy(n.begin(), n.end(), values); return; }
  int iu = (int)(upper - times_.begin()), il = iu - 1;
  double lo = times_[il], up = times_[iu];
  double t = (time - lo) / (up - lo);
  const auto& lower_node = values_[il];
  const auto& upper_node = values_[iu];
  switch (interpolation_) {
    case kZeroSpline:
      std::copy(lower_node.begin(), lower_node.end(), values);
      return;
    case kLinearSpline:
      for (int i = 0; i < dim_; i++) values[i] = lower_node[i] * (1 - t) + upper_node[i] * t;
      return;
    case kCubicSpline: {
      double c0 = 2.0 * t*t*t - 3.0 * t*t + 1.0;
      double c1 = (t*t*t - 2.0 * t*t + t) * (up - lo);
      double c2 = -2.0 * t*t*t + 3 * t*t;
      double c3 = (t*t*t - t*t) * (up - lo);
      for (int i = 0; i < dim_; i++) {
        double p0 = lower_node[i], m0 = Slope(il, i), m1 = Slope(iu, i), p1 = upper_node[i];
        values[i] = c0 * p0 + c1 * m0 + c2 * p1 + c3 * m1;
      }
      return;
    }
    default:
      Fatal("Unknown interpolation");
  }
}
std::vector<double> TimeSpline::Sample(double time) const {
  std::vector<double> v(dim_);
  Sample(time, v.data());
  return v;
}

// ------------------------------------------------------------------ SamplingPolicy
void SamplingPolicy::Allocate(const MjpcHipModel* model, int nsp) {
  nu = model->nu;
  ctrlrange.assign(model->actuator_ctrlrange, model->actuator_ctrlrange + 2 * nu);
  num_spline_points = nsp;
  plan = TimeSpline(nu);
}
void SamplingPolicy::Reset(int, const double* initial_repeated_action) {
  plan.Clear();
  if (initial_repeated_action != nullptr) plan.AddNode(0, initial_repeated_action);
}
void SamplingPolicy::Action(double* action, const double*, double time) const {
  if (action == nullptr) { Fatal("SamplingPolicy::Action: action == nullptr"); return; }
  plan.Sample(time, action);
  for (int i = 0; i < nu; i++) action[i] = std::max(ctrlrange[2 * i], std::min(ctrlrange[2 * i + 1], action[i]));   // Clamp
}
void SamplingPolicy::CopyFrom(const SamplingPolicy& p, int) {
  plan = p.plan; num_spline_points = p.num_spline_points; nu = p.nu; ctrlrange = p.ctrlrange;
}

// ------------------------------------------------------------------ SamplingPlanner
static double Micros(std::chrono::steady_clock::time_point t0) {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
}

SamplingPlanner::~SamplingPlanner() {
  if (engine_) mjpc_hip_multi_destroy(engine_);
  if (nominal_engine_) mjpc_hip_destroy(nominal_engine_);
}

// returns sorted ascending with every non-finite value treated as +inf: a strict weak order even when a rollout produced NaN
// (the engine's argmin skips NaN the same way)
static inline bool ReturnLess(double a, double b) {
  const double inf = std::numeric_limits<double>::infinity();
  if (!(a == a)) a = inf;
  if (!(b == b)) b = inf;
  return a < b;
}

void SamplingPlanner::Initialize(const MjpcHipModel* model, const MjpcHipTask* task, const Numerics& numerics) {
  numerics_ = numerics;
  nq_ = model->nq; nv_ = model->nv; na_ = model->na; ns_ = nq_ + nv_ + na_; nu_ = model->nu; nmocap_ = model->nmocap; nuserdata_ = model->nuserdata;
  nr_ = task->num_residual; ntrace_ = task->num_trace; timestep_ = model->timestep;
  ctrlrange_.assign(model->actuator_ctrlrange, model->actuator_ctrlrange + 2 * nu_);
  noise_exploration[0] = numerics.sampling_exploration[0];
  noise_exploration[1] = numerics.sampling_exploration[1];
  num_trajectory_ = numerics.sampling_trajectories;
  interpolation_ = numerics.sampling_representation;
  sliding_plan_ = numerics.sampling_sliding_plan;
  if (num_trajectory_ > numerics.max_samples) {
    char msg[128]; std::snprintf(msg, sizeof(msg), "Too many trajectories, %d is the maximum allowed.", numerics.max_samples);
    Fatal(msg);
    return;
  }
  if (engine_) { mjpc_hip_multi_destroy(engine_); engine_ = nullptr; }     // the model might have changed
  // one engine per GPU (Numerics::n_devices, ordinals in Numerics::devices or device, device+1, ...): the candidate batch of a
  // plan step is block-partitioned over them (include/mjpc_hip.h, mjpc_hip_multi_plan)
  {
    int G = std::max(1, numerics.n_devices);
    std::vector<int> devs(G);
    for (int k = 0; k < G; k++) devs[k] = k < (int)numerics.devices.size() ? numerics.devices[k] : numerics.device + k;
    engine_ = mjpc_hip_multi_create(model, task, numerics.max_samples, numerics.max_horizon, G, devs.data());
  }
  if (!engine_) { Fatal(mjpc_hip_last_error()); return; }
  // NominalTrajectory() rolls the nominal policy out on its own one-candidate engine, so that the candidates of the last
  // plan step (returns, order, trajectories on the device) stay available to the RankedPlanner calls afterwards
  if (nominal_engine_) { mjpc_hip_destroy(nominal_engine_); nominal_engine_ = nullptr; }
  nominal_engine_ = mjpc_hip_create(model, task, 1, numerics.max_horizon, numerics.device);
  if (!nominal_engine_) { Fatal(mjpc_hip_last_error()); return; }
  policy.Allocate(model, numerics.sampling_spline_points);
  previous_policy.Allocate(model, numerics.sampling_spline_points);
  winner_policy_.Allocate(model, numerics.sampling_spline_points);
  winner = 0;
}

void SamplingPlanner::Allocate() {
  state.assign(ns_, 0.0); mocap.assign(7 * nmocap_, 0.0); userdata.assign(nuserdata_, 0.0);
  plan_scratch_ = TimeSpline(nu_);
  trajectory_winner.dim_state = ns_; trajectory_winner.dim_action = nu_;
  trajectory_winner.dim_residual = nr_; trajectory_winner.dim_trace = 3 * ntrace_;
  trajectory_winner.states.assign((size_t)numerics_.max_horizon * ns_, 0.0);
  trajectory_winner.actions.assign((size_t)numerics_.max_horizon * nu_, 0.0);
  trajectory_winner.times.assign(numerics_.max_horizon, 0.0);
  trajectory_winner.residual.assign((size_t)numerics_.max_horizon * nr_, 0.0);
  trajectory_winner.costs.assign(numerics_.max_horizon, 0.0);
  trajectory_winner.trace.assign((size_t)numerics_.max_horizon * 3 * std::max(ntrace_, 1), 0.0);
  returns.assign(numerics_.max_samples, 0.0); failures.assign(numerics_.max_samples, 0);
  winner = -1;
}

void SamplingPlanner::Reset(int horizon, const double* initial_repeated_action) {
  std::fill(state.begin(), state.end(), 0.0); std::fill(mocap.begin(), mocap.end(), 0.0);
  std::fill(userdata.begin(), userdata.end(), 0.0);
  time = 0.0;
  policy.Reset(horizon, initial_repeated_action);
  previous_policy.Reset(horizon, initial_repeated_action);
  winner_policy_.Reset(horizon, initial_repeated_action);
  plan_scratch_.Clear();
  improvement = 0.0;
  winner = 0;
  fetched_ = -1;
}

void SamplingPlanner::SetState(const double* s, const double* m, const double* u, double t) {
  std::copy(s, s + ns_, state.begin());
  if (m) std::copy(m, m + 7 * nmocap_, mocap.begin());
  if (u) std::copy(u, u + nuserdata_, userdata.begin());
  time = t;
}

void SamplingPlanner::SetTask(const MjpcHipTask* task) {
  if (mjpc_hip_multi_set_task(engine_, task) != 0) Fatal(mjpc_hip_last_error());
  if (nominal_engine_ && mjpc_hip_set_task(nominal_engine_, task) != 0) Fatal(mjpc_hip_last_error());
}

void SamplingPlanner::UpdateNominalPolicy(int horizon) {   // planner.cc:236-310
  int num_spline_points = winner_policy_.num_spline_points;
  double nominal_time = time;
  double time_horizon = (horizon - 1) * timestep_;
  if (sliding_plan_) {
    int extra_points = interpolation_ == kZeroSpline ? 1 : (interpolation_ == kLinearSpline ? 2 : 4);
    double time_shift;
    if (num_spline_points > extra_points) time_shift = std::max(time_horizon / (num_spline_points - extra_points), 1.0e-5);
    else time_shift = time_horizon;
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    policy.plan.DiscardBefore(nominal_time);
    if (policy.plan.Size() == 0) policy.plan.AddNode(time);
    while ((int)policy.plan.Size() < num_spline_points) {
      int last = (int)policy.plan.Size() - 1;
      double new_node_time = policy.plan.NodeTime(last) + time_shift;
      std::vector<double> copy(policy.plan.NodeValues(last), policy.plan.NodeValues(last) + nu_);
      policy.plan.AddNode(new_node_time, copy.data());
    }
  } else {
    double time_shift;
    if (interpolation_ == kZeroSpline) time_shift = std::max(time_horizon / num_spline_points, 1.0e-5);
    else time_shift = std::max(time_horizon / (num_spline_points - 1), 1.0e-5);
    plan_scratch_.Clear();
    plan_scratch_.SetInterpolation((SplineInterpolation)interpolation_);
    for (int t = 0; t < num_spline_points; t++) {
      double* node = plan_scratch_.AddNode(nominal_time);
      winner_policy_.Action(node, nullptr, nominal_time);
      nominal_time += time_shift;                         // repeated addition, like the reference
    }
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    policy.plan = plan_scratch_;
  }
}

int SamplingPlanner::OptimizePolicyCandidates(int ncandidates, int horizon) {   // planner.cc:151-187
  int num_trajectory = num_trajectory_;
  ncandidates = std::min(ncandidates, num_trajectory);
  auto rollouts_start = std::chrono::steady_clock::now();
  policy.plan.SetInterpolation((SplineInterpolation)interpolation_);
  int P = (int)policy.plan.Size();
  knot_times_.resize(std::max(P, 1)); knot_values_.resize((size_t)std::max(P, 1) * nu_);
  for (int p = 0; p < P; p++) {
    knot_times_[p] = policy.plan.NodeTime(p);
    std::copy(policy.plan.NodeValues(p), policy.plan.NodeValues(p) + nu_, knot_values_.begin() + (size_t)p * nu_);
  }
  if (P == 0) { P = 1; knot_times_[0] = time; std::fill(knot_values_.begin(), knot_values_.end(), 0.0); }   // empty plan samples zeros
  winner_knots_.assign((size_t)P * nu_, 0.0);
  MjpcHipPlanInput in;
  std::memset(&in, 0, sizeof(in));
  in.state = state.data(); in.mocap = mocap.data(); in.userdata = userdata.data(); in.time = time;
  in.knot_times = knot_times_.data(); in.knot_values = knot_values_.data(); in.num_spline_points = P;
  in.interpolation = interpolation_; in.num_trajectory = num_trajectory; in.horizon = horizon;
  in.candidate_offset = 0; in.num_local = num_trajectory;
  in.noise_exploration[0] = noise_exploration[0]; in.noise_exploration[1] = noise_exploration[1];
  in.noise_eps = injected_noise_eps; in.noise_sel = injected_noise_sel; in.seed = seed; in.stream = plan_iter++;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  out.returns = returns.data(); out.failure = failures.data();
  out.states = trajectory_winner.states.data(); out.actions = trajectory_winner.actions.data();
  out.times = trajectory_winner.times.data(); out.residual = trajectory_winner.residual.data();
  out.costs = trajectory_winner.costs.data(); out.trace = trajectory_winner.trace.data(); out.winner_knots = winner_knots_.data();
  if (mjpc_hip_multi_plan(engine_, &in, &out) != 0) { Fatal(mjpc_hip_last_error()); return 0; }
  last_horizon_ = horizon; fetched_ = out.winner;
  noise_compute_time = out.noise_compute_time_us;
  // order so that the first ncandidates are the best (ties: lowest index, like the engine's argmin)
  trajectory_order.resize(num_trajectory);
  std::iota(trajectory_order.begin(), trajectory_order.end(), 0);
  std::stable_sort(trajectory_order.begin(), trajectory_order.end(), [this](int a, int b) { return ReturnLess(returns[a], returns[b]); });
  rollouts_compute_time = Micros(rollouts_start);
  return ncandidates;
}

void SamplingPlanner::FetchCandidate(int global_index) {
  if (fetched_ == global_index) return;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  out.states = trajectory_winner.states.data(); out.actions = trajectory_winner.actions.data();
  out.times = trajectory_winner.times.data(); out.residual = trajectory_winner.residual.data();
  out.costs = trajectory_winner.costs.data(); out.trace = trajectory_winner.trace.data(); out.winner_knots = winner_knots_.data();
  if (mjpc_hip_multi_get_candidate(engine_, global_index, &out) != 0) { Fatal(mjpc_hip_last_error()); return; }
  fetched_ = global_index;
}

void SamplingPlanner::CopyCandidateToPolicy(int candidate) {   // planner.cc:525-534 (with a UNIQUE lock)
  winner = trajectory_order[candidate];
  FetchCandidate(winner);
  trajectory_winner.horizon = last_horizon_;
  trajectory_winner.total_return = returns[winner];
  trajectory_winner.failure = failures[winner] != 0;
  int P = (int)knot_times_.size();
  winner_policy_.plan = TimeSpline(nu_, (SplineInterpolation)interpolation_);
  for (int p = 0; p < P; p++) winner_policy_.plan.AddNode(knot_times_[p], winner_knots_.data() + (size_t)p * nu_);
  winner_policy_.num_spline_points = policy.num_spline_points;
  const std::unique_lock<std::shared_mutex> lock(mtx_);
  previous_policy.CopyFrom(policy, 0);
  policy.CopyFrom(winner_policy_, 0);
}

void SamplingPlanner::OptimizePolicy(int horizon) {   // planner.cc:190-208
  UpdateNominalPolicy(horizon);
  OptimizePolicyCandidates(1, horizon);
  auto policy_update_start = std::chrono::steady_clock::now();
  CopyCandidateToPolicy(0);
  double best_return = returns[0];
  improvement = std::max(best_return - returns[winner], 0.0);
  policy_update_compute_time = Micros(policy_update_start);
}

void SamplingPlanner::NominalTrajectory(int horizon) {   // planner.cc:211-222: rolls out `policy` into trajectory[0] only
  // one un-noised candidate on the dedicated engine: returns / failures / trajectory_order / candidate knots of the last
  // OptimizePolicyCandidates() are not touched (the reference writes nothing but trajectory[0] here either)
  policy.plan.SetInterpolation((SplineInterpolation)interpolation_);
  int P = (int)policy.plan.Size();
  std::vector<double> kt(std::max(P, 1)), kv((size_t)std::max(P, 1) * nu_, 0.0);
  for (int p = 0; p < P; p++) {
    kt[p] = policy.plan.NodeTime(p);
    std::copy(policy.plan.NodeValues(p), policy.plan.NodeValues(p) + nu_, kv.begin() + (size_t)p * nu_);
  }
  if (P == 0) { P = 1; kt[0] = time; }
  MjpcHipPlanInput in;
  std::memset(&in, 0, sizeof(in));
  in.state = state.data(); in.mocap = mocap.data(); in.userdata = userdata.data(); in.time = time;
  in.knot_times = kt.data(); in.knot_values = kv.data(); in.num_spline_points = P;
  in.interpolation = interpolation_; in.num_trajectory = 1; in.horizon = horizon; in.candidate_offset = 0; in.num_local = 1;
  in.noise_exploration[0] = noise_exploration[0]; in.noise_exploration[1] = noise_exploration[1];
  in.seed = seed; in.stream = plan_iter;
  double ret = 0; int fail = 0;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  out.returns = &ret; out.failure = &fail;
  out.states = trajectory_winner.states.data(); out.actions = trajectory_winner.actions.data();
  out.times = trajectory_winner.times.data(); out.residual = trajectory_winner.residual.data();
  out.costs = trajectory_winner.costs.data(); out.trace = trajectory_winner.trace.data();
  if (mjpc_hip_plan(nominal_engine_, &in, &out) != 0) { Fatal(mjpc_hip_last_error()); return; }
  fetched_ = -1;                 // trajectory_winner no longer mirrors a candidate of the last plan: re-fetch on demand
  nominal_horizon_ = horizon;
  trajectory_winner.horizon = horizon; trajectory_winner.total_return = ret; trajectory_winner.failure = fail != 0;
  if (winner < 0) winner = 0;
}

void SamplingPlanner::ActionFromPolicy(double* action, const double* s, double t, bool use_previous) {   // planner.cc:225-233
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  if (use_previous) previous_policy.Action(action, s, t);
  else policy.Action(action, s, t);
}

const Trajectory* SamplingPlanner::BestTrajectory() { return winner >= 0 ? &trajectory_winner : nullptr; }

double SamplingPlanner::CandidateScore(int candidate) const { return returns[trajectory_order[candidate]]; }

void SamplingPlanner::ActionFromCandidatePolicy(double* action, int candidate, const double* s, double t) {
  FetchCandidate(trajectory_order[candidate]);
  TimeSpline sp(nu_, (SplineInterpolation)interpolation_);
  for (size_t p = 0; p < knot_times_.size(); p++) sp.AddNode(knot_times_[p], winner_knots_.data() + p * nu_);
  sp.Sample(t, action);
  for (int i = 0; i < nu_; i++) action[i] = std::max(ctrlrange_[2 * i], std::min(ctrlrange_[2 * i + 1], action[i]));
  (void)s;
}

// ------------------------------------------------------------------ CrossEntropyPlanner
CrossEntropyPlanner::~CrossEntropyPlanner() { if (engine_) mjpc_hip_destroy(engine_); }

void CrossEntropyPlanner::Initialize(const MjpcHipModel* model, const MjpcHipTask* task, const Numerics& numerics) {   // planner.cc:41-72
  numerics_ = numerics;
  nq_ = model->nq; nv_ = model->nv; na_ = model->na; ns_ = nq_ + nv_ + na_; nu_ = model->nu; nmocap_ = model->nmocap;
  nuserdata_ = model->nuserdata; nr_ = task->num_residual; ntrace_ = task->num_trace; timestep_ = model->timestep;
  ctrlrange_.assign(model->actuator_ctrlrange, model->actuator_ctrlrange + 2 * nu_);
  std_initial_ = numerics.sampling_exploration[0];
  std_min_ = numerics.std_min;
  num_trajectory_ = numerics.sampling_trajectories;
  n_elite_ = numerics.n_elite > 0 ? numerics.n_elite : std::max(num_trajectory_ / 10, 2);
  interpolation_ = numerics.sampling_representation;
  if (num_trajectory_ > numerics.max_samples) {
    char msg[128]; std::snprintf(msg, sizeof(msg), "Too many trajectories, %d is the maximum allowed.", numerics.max_samples);
    Fatal(msg);
    return;
  }
  if (engine_) { mjpc_hip_destroy(engine_); engine_ = nullptr; }
  engine_ = mjpc_hip_create(model, task, numerics.max_samples + 1, numerics.max_horizon, numerics.device);   // + the nominal rollout
  if (!engine_) { Fatal(mjpc_hip_last_error()); return; }
  policy.Allocate(model, numerics.sampling_spline_points);
  resampled_policy.Allocate(model, numerics.sampling_spline_points);
  previous_policy.Allocate(model, numerics.sampling_spline_points);
}

void CrossEntropyPlanner::Allocate() {   // planner.cc:75-117
  state.assign(ns_, 0.0); mocap.assign(7 * nmocap_, 0.0); userdata.assign(nuserdata_, 0.0);
  int P = policy.num_spline_points;
  parameters_scratch.assign((size_t)P * nu_, 0.0); times_scratch.assign(P, 0.0);
  variance.assign((size_t)P * nu_, 0.0); noise_std_.assign((size_t)P * nu_, 0.0); knot_values_.assign((size_t)P * nu_, 0.0);
  trajectory_order.resize(numerics_.max_samples);
  std::iota(trajectory_order.begin(), trajectory_order.end(), 0);
  returns.assign(numerics_.max_samples + 1, 0.0); failures.assign(numerics_.max_samples + 1, 0);
  size_t Hm = (size_t)numerics_.max_horizon;
  nominal_trajectory.dim_state = ns_; nominal_trajectory.dim_action = nu_; nominal_trajectory.dim_residual = nr_;
  nominal_trajectory.dim_trace = 3 * ntrace_;
  nominal_trajectory.states.assign(Hm * ns_, 0.0); nominal_trajectory.actions.assign(Hm * nu_, 0.0);
  nominal_trajectory.times.assign(Hm, 0.0); nominal_trajectory.residual.assign(Hm * nr_, 0.0);
  nominal_trajectory.costs.assign(Hm, 0.0); nominal_trajectory.trace.assign(Hm * 3 * std::max(ntrace_, 1), 0.0);
}

void CrossEntropyPlanner::Reset(int horizon, const double* initial_repeated_action) {   // planner.cc:120-155
  std::fill(state.begin(), state.end(), 0.0); std::fill(mocap.begin(), mocap.end(), 0.0);
  std::fill(userdata.begin(), userdata.end(), 0.0);
  time = 0.0;
  policy.Reset(horizon, initial_repeated_action);
  resampled_policy.Reset(horizon, initial_repeated_action);
  previous_policy.Reset(horizon, initial_repeated_action);
  std::fill(parameters_scratch.begin(), parameters_scratch.end(), 0.0);
  std::fill(times_scratch.begin(), times_scratch.end(), 0.0);
  double var = std_initial_ * std_initial_;
  std::fill(variance.begin(), variance.end(), var);
  improvement = 0.0;
}

void CrossEntropyPlanner::SetState(const double* s, const double* m, const double* u, double t) {
  std::copy(s, s + ns_, state.begin());
  if (m) std::copy(m, m + 7 * nmocap_, mocap.begin());
  if (u) std::copy(u, u + nuserdata_, userdata.begin());
  time = t;
}

void CrossEntropyPlanner::SetTask(const MjpcHipTask* task) {
  if (mjpc_hip_set_task(engine_, task) != 0) Fatal(mjpc_hip_last_error());
}

void CrossEntropyPlanner::ResamplePolicy(int horizon) {   // planner.cc:313-338
  int num_spline_points = resampled_policy.num_spline_points;
  double nominal_time = time;
  double time_shift = std::max((horizon - 1) * timestep_ / (num_spline_points - 1), 1.0e-5);
  for (int t = 0; t < num_spline_points; t++) {
    times_scratch[t] = nominal_time;
    resampled_policy.Action(parameters_scratch.data() + (size_t)t * nu_, nullptr, nominal_time);
    nominal_time += time_shift;
  }
  SplineInterpolation keep = policy.plan.Interpolation();
  resampled_policy.plan.Clear();
  for (int t = 0; t < num_spline_points; t++) resampled_policy.plan.AddNode(times_scratch[t], parameters_scratch.data() + (size_t)t * nu_);
  resampled_policy.plan.SetInterpolation(keep);
}

void CrossEntropyPlanner::OptimizePolicy(int horizon) {   // planner.cc:164-283
  resampled_policy.plan.SetInterpolation((SplineInterpolation)interpolation_);
  int num_trajectory = num_trajectory_;
  n_elite_ = std::min(n_elite_, num_trajectory);
  int n_elite = std::min(n_elite_, num_trajectory);
  {
    const std::shared_lock<std::shared_mutex> lock(mtx_);
    resampled_policy.CopyFrom(policy, policy.num_spline_points);
  }
  ResamplePolicy(horizon);

  // ----- rollouts (planner.cc:377-415): N perturbed candidates + the nominal as candidate N, one launch
  auto rollouts_start = std::chrono::steady_clock::now();
  int P = resampled_policy.num_spline_points;
  for (int t = 0; t < P; t++) std::copy(resampled_policy.plan.NodeValues(t), resampled_policy.plan.NodeValues(t) + nu_, knot_values_.begin() + (size_t)t * nu_);
  for (int k = 0; k < P * nu_; k++) noise_std_[k] = std::max(std::sqrt(variance[k]), std_min_);   // AddNoiseToPolicy, planner.cc:359-362
  MjpcHipPlanInput in;
  std::memset(&in, 0, sizeof(in));
  in.state = state.data(); in.mocap = mocap.data(); in.userdata = userdata.data(); in.time = time;
  in.knot_times = times_scratch.data(); in.knot_values = knot_values_.data(); in.num_spline_points = P;
  in.interpolation = (int)resampled_policy.plan.Interpolation();
  in.num_trajectory = num_trajectory + 1; in.horizon = horizon; in.candidate_offset = 0; in.num_local = num_trajectory + 1;
  in.noise_eps = injected_noise_eps; in.seed = seed; in.stream = plan_iter++;
  in.noise_std = noise_std_.data(); in.nominal_index = num_trajectory;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  out.returns = returns.data(); out.failure = failures.data();
  if (mjpc_hip_plan(engine_, &in, &out) != 0) { Fatal(mjpc_hip_last_error()); return; }
  noise_compute_time = out.noise_compute_time_us;
  last_horizon_ = horizon;
  // nominal trajectory = candidate N
  MjpcHipPlanOutput nom;
  std::memset(&nom, 0, sizeof(nom));
  nom.states = nominal_trajectory.states.data(); nom.actions = nominal_trajectory.actions.data(); nom.times = nominal_trajectory.times.data();
  nom.residual = nominal_trajectory.residual.data(); nom.costs = nominal_trajectory.costs.data(); nom.trace = nominal_trajectory.trace.data();
  if (mjpc_hip_get_candidate(engine_, num_trajectory, &nom) != 0) { Fatal(mjpc_hip_last_error()); return; }
  nominal_trajectory.horizon = horizon; nominal_trajectory.total_return = returns[num_trajectory];
  nominal_trajectory.failure = failures[num_trajectory] != 0;
  all_knots_.resize((size_t)(num_trajectory + 1) * P * nu_);
  if (mjpc_hip_get_knots(engine_, all_knots_.data()) != 0) { Fatal(mjpc_hip_last_error()); return; }
  trajectory_order.resize(std::max((int)trajectory_order.size(), num_trajectory));
  for (int i = 0; i < num_trajectory; i++) trajectory_order[i] = i;
  std::stable_sort(trajectory_order.begin(), trajectory_order.begin() + num_trajectory,
                   [this](int a, int b) { return ReturnLess(returns[a], returns[b]); });
  rollouts_compute_time = Micros(rollouts_start);

  // ----- update policy (planner.cc:205-283)
  auto policy_update_start = std::chrono::steady_clock::now();
  int num_parameters = P * nu_;
  double avg_return = 0.0;
  std::fill(parameters_scratch.begin(), parameters_scratch.end(), 0.0);
  for (int i = 0; i < n_elite; i++) {
    int idx = trajectory_order[i];
    const double* kn = all_knots_.data() + (size_t)idx * num_parameters;
    for (int k = 0; k < num_parameters; k++) parameters_scratch[k] += kn[k];
    avg_return += returns[idx];
  }
  for (int k = 0; k < num_parameters; k++) parameters_scratch[k] *= 1.0 / n_elite;     // mju_scl
  avg_return /= n_elite;
  std::fill(variance.begin(), variance.end(), 0.0);
  {
    const double* best = all_knots_.data() + (size_t)trajectory_order[0] * num_parameters;   // the reference reads elite 0 for every i
    for (int k = 0; k < num_parameters; k++) {
      double p_avg = parameters_scratch[k];
      for (int i = 0; i < n_elite; i++) {
        double diff = best[k] - p_avg;
        variance[k] += diff * diff / (n_elite - 1);
      }
    }
  }
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    policy.plan.Clear();
    policy.plan.SetInterpolation((SplineInterpolation)interpolation_);
    for (int t = 0; t < P; t++) policy.plan.AddNode(times_scratch[t], parameters_scratch.data() + (size_t)t * nu_);
  }
  improvement = std::max(avg_return - returns[trajectory_order[0]], 0.0);
  policy_update_compute_time = Micros(policy_update_start);
}

void CrossEntropyPlanner::NominalTrajectory(int horizon) {   // planner.cc:286-297: rollout of resampled_policy
  int P = (int)resampled_policy.plan.Size();
  std::vector<double> kt(std::max(P, 1), time), kv((size_t)std::max(P, 1) * nu_, 0.0);
  for (int p = 0; p < P; p++) {
    kt[p] = resampled_policy.plan.NodeTime(p);
    std::copy(resampled_policy.plan.NodeValues(p), resampled_policy.plan.NodeValues(p) + nu_, kv.begin() + (size_t)p * nu_);
  }
  MjpcHipPlanInput in;
  std::memset(&in, 0, sizeof(in));
  in.state = state.data(); in.mocap = mocap.data(); in.userdata = userdata.data(); in.time = time;
  in.knot_times = kt.data(); in.knot_values = kv.data(); in.num_spline_points = std::max(P, 1);
  in.interpolation = (int)resampled_policy.plan.Interpolation(); in.num_trajectory = 1; in.horizon = horizon; in.num_local = 1;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  double ret = 0; int fail = 0;
  out.returns = &ret; out.failure = &fail;
  out.states = nominal_trajectory.states.data(); out.actions = nominal_trajectory.actions.data(); out.times = nominal_trajectory.times.data();
  out.residual = nominal_trajectory.residual.data(); out.costs = nominal_trajectory.costs.data(); out.trace = nominal_trajectory.trace.data();
  if (mjpc_hip_plan(engine_, &in, &out) != 0) { Fatal(mjpc_hip_last_error()); return; }
  nominal_trajectory.horizon = horizon; nominal_trajectory.total_return = ret; nominal_trajectory.failure = fail != 0;
}

void CrossEntropyPlanner::ActionFromPolicy(double* action, const double* s, double t, bool use_previous) {   // planner.cc:302-310
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  if (use_previous) previous_policy.Action(action, s, t);
  else policy.Action(action, s, t);
}

const Trajectory* CrossEntropyPlanner::BestTrajectory() { return &nominal_trajectory; }

void SamplingPlanner::CandidateKnots(int candidate, double* out) {
  FetchCandidate(trajectory_order[candidate]);
  std::copy(winner_knots_.begin(), winner_knots_.end(), out);
}

// trajectory[i] / candidate_policy[i] of the reference by BATCH index i (not ranked): what iLQS reads (ilqs/planner.cc:98-198)
void SamplingPlanner::FetchCandidateUnranked(int index) { FetchCandidate(index); trajectory_winner.horizon = last_horizon_;
  trajectory_winner.total_return = returns[index]; trajectory_winner.failure = failures[index] != 0; }
void SamplingPlanner::CandidateKnotsUnranked(int index, double* out) {
  FetchCandidate(index);
  std::copy(winner_knots_.begin(), winner_knots_.end(), out);
}
// every candidate's trace rows of the last plan step, [num_trajectory][horizon][3 * num_trace]: SamplingPlanner::Traces
void SamplingPlanner::AllTraces(double* out) {
  if (mjpc_hip_multi_get_traces(engine_, out) != 0) Fatal(mjpc_hip_last_error());
}

void SetErrorHandler(void (*handler)(const char*)) { g_error_handler = handler; }

// ------------------------------------------------------------------ RobustPlanner
RobustPlanner::~RobustPlanner() { if (engine_) mjpc_hip_destroy(engine_); }

void RobustPlanner::Initialize(const MjpcHipModel* model, const MjpcHipTask* task, const Numerics& numerics) {   // robust_planner.cc:30-58
  numerics_ = numerics;
  delegate.Initialize(model, task, numerics);
  nu_ = model->nu; ns_ = model->nq + model->nv + model->na; nmocap_ = model->nmocap; nuserdata_ = model->nuserdata;
  nrepetitions_ = numerics.robust_repetitions;
  ncandidates_ = numerics.robust_candidates;
  if (ncandidates_ == -1) ncandidates_ = numerics.sampling_trajectories / nrepetitions_;
  xfrc_std_ = numerics.robust_xfrc; xfrc_rate_ = numerics.robust_xfrc_rate;
  if (engine_) { mjpc_hip_destroy(engine_); engine_ = nullptr; }
  int cap = std::max(1, std::max(ncandidates_, 1) * std::max(nrepetitions_, 1));
  engine_ = mjpc_hip_create(model, task, cap, numerics.max_horizon, numerics.device);
  if (!engine_) Fatal(mjpc_hip_last_error());
}
void RobustPlanner::Allocate() {
  delegate.Allocate();
  state_.assign(ns_, 0.0); mocap_.assign(7 * nmocap_, 0.0); userdata_.assign(std::max(nuserdata_, 1), 0.0);
}
void RobustPlanner::Reset(int horizon, const double* initial_repeated_action) {
  delegate.Reset(horizon, initial_repeated_action);
  std::fill(state_.begin(), state_.end(), 0.0); std::fill(mocap_.begin(), mocap_.end(), 0.0);
  std::fill(userdata_.begin(), userdata_.end(), 0.0);
  time_ = 0.0;
}
void RobustPlanner::SetState(const double* s, const double* m, const double* u, double t) {
  delegate.SetState(s, m, u, t);
  std::copy(s, s + ns_, state_.begin());
  if (m) std::copy(m, m + 7 * nmocap_, mocap_.begin());
  if (u) std::copy(u, u + nuserdata_, userdata_.begin());
  time_ = t;
}
void RobustPlanner::SetTask(const MjpcHipTask* task) {
  delegate.SetTask(task);
  if (mjpc_hip_set_task(engine_, task) != 0) Fatal(mjpc_hip_last_error());
}

void RobustPlanner::OptimizePolicy(int horizon) {   // robust_planner.cc:91-157
  // Conscious fix of a reference quirk: at this snapshot RobustPlanner calls the delegate's OptimizePolicyCandidates directly
  // (robust_planner.cc:93) and UpdateNominalPolicy only runs inside SamplingPlanner::OptimizePolicy (planner.cc:192), so the
  // wrapped planner never re-times its knots to the current time.  Resample first, as OptimizePolicy does.
  delegate.UpdateNominalPolicy(horizon);
  int ncandidates = delegate.OptimizePolicyCandidates(ncandidates_, horizon);
  best_candidate = -1;
  if (!ncandidates) return;
  if (ncandidates == 1) { delegate.CopyCandidateToPolicy(0); best_candidate = 0; return; }
  int repetitions = nrepetitions_;
  const std::vector<double>& kt = delegate.KnotTimes();
  int P = (int)kt.size();
  size_t row = (size_t)P * nu_;
  cand_knots_.resize((size_t)ncandidates * repetitions * row);
  for (int i = 0; i < ncandidates; i++) {
    delegate.CandidateKnots(i, cand_knots_.data() + (size_t)i * repetitions * row);
    for (int j = 1; j < repetitions; j++)
      std::copy(cand_knots_.begin() + (size_t)i * repetitions * row, cand_knots_.begin() + ((size_t)i * repetitions + 1) * row,
                cand_knots_.begin() + ((size_t)i * repetitions + j) * row);
  }
  int total = ncandidates * repetitions;
  noisy_returns.assign(total, 0.0); noisy_failures.assign(total, 0);
  std::vector<double> zeros(row, 0.0);
  MjpcHipPlanInput in;
  std::memset(&in, 0, sizeof(in));
  in.state = state_.data(); in.mocap = mocap_.data(); in.userdata = userdata_.data(); in.time = time_;
  in.knot_times = kt.data(); in.knot_values = zeros.data(); in.num_spline_points = P;
  in.interpolation = delegate.interpolation_; in.num_trajectory = total; in.horizon = horizon; in.num_local = total;
  in.candidate_knots = cand_knots_.data(); in.xfrc_std = xfrc_std_; in.xfrc_rate = xfrc_rate_;
  in.seed = seed; in.stream = plan_iter++;
  MjpcHipPlanOutput out;
  std::memset(&out, 0, sizeof(out));
  out.returns = noisy_returns.data(); out.failure = noisy_failures.data();
  if (mjpc_hip_plan(engine_, &in, &out) != 0) { Fatal(mjpc_hip_last_error()); return; }
  // mean over the delegate's score and the valid noisy rollouts; the best mean wins (robust_planner.cc:128-151)
  candidate_scores.assign(ncandidates, 0.0);
  double best_score = 0;
  for (int candidate = 0; candidate < ncandidates; candidate++) {
    double mean_return = delegate.CandidateScore(candidate);
    int valid_rollouts = 0;
    for (int j = 0; j < repetitions; j++) {
      if (noisy_failures[repetitions * candidate + j]) continue;
      double total_return = noisy_returns[repetitions * candidate + j];
      mean_return = (valid_rollouts * mean_return + total_return) / (valid_rollouts + 1);
      valid_rollouts++;
    }
    candidate_scores[candidate] = mean_return;
    if (best_candidate == -1 || mean_return < best_score) { best_candidate = candidate; best_score = mean_return; }
  }
  delegate.CopyCandidateToPolicy(best_candidate);
}

}  // namespace mjpc_hip

// ====================================================================== flat C wrapper (tests / ctypes)
using mjpc_hip::SamplingPlanner;
using mjpc_hip::TimeSpline;
extern "C" {

void mjpc_planner_set_error_handler(void (*h)(const char*)) { mjpc_hip::g_error_handler = h; }

// ---- TimeSpline (so that the reference's spline goldens can be run against the C++ class)
void* mjpc_spline_create(int dim, int interpolation) { return new TimeSpline(dim, (mjpc_hip::SplineInterpolation)interpolation); }
void mjpc_spline_destroy(void* s) { delete (TimeSpline*)s; }
int mjpc_spline_size(void* s) { return (int)((TimeSpline*)s)->Size(); }
void mjpc_spline_add_node(void* s, double time, const double* values) { ((TimeSpline*)s)->AddNode(time, values); }
void mjpc_spline_sample(void* s, double time, double* out) { ((TimeSpline*)s)->Sample(time, out); }
int mjpc_spline_discard_before(void* s, double time) { return ((TimeSpline*)s)->DiscardBefore(time); }
void mjpc_spline_clear(void* s) { ((TimeSpline*)s)->Clear(); }
void mjpc_spline_set_interpolation(void* s, int i) { ((TimeSpline*)s)->SetInterpolation((mjpc_hip::SplineInterpolation)i); }

// ---- SamplingPlanner
void* mjpc_planner_create(const MjpcHipModel* model, const MjpcHipTask* task, const double* exploration, int trajectories,
                          int representation, int sliding_plan, int spline_points, int max_samples, int max_horizon, int device) {
  auto* p = new SamplingPlanner();
  mjpc_hip::Numerics n;
  n.sampling_exploration[0] = exploration[0]; n.sampling_exploration[1] = exploration[1];
  n.sampling_trajectories = trajectories; n.sampling_representation = representation; n.sampling_sliding_plan = sliding_plan;
  n.sampling_spline_points = spline_points; n.max_samples = max_samples; n.max_horizon = max_horizon; n.device = device;
  p->Initialize(model, task, n);
  p->Allocate();
  return p;
}
// the same planner with its candidate batch sharded over n_devices GPUs (devices[k]: HIP ordinals, repeats allowed)
void* mjpc_planner_create_sharded(const MjpcHipModel* model, const MjpcHipTask* task, const double* exploration, int trajectories,
                                  int representation, int sliding_plan, int spline_points, int max_samples, int max_horizon,
                                  int n_devices, const int* devices) {
  auto* p = new SamplingPlanner();
  mjpc_hip::Numerics n;
  n.sampling_exploration[0] = exploration[0]; n.sampling_exploration[1] = exploration[1];
  n.sampling_trajectories = trajectories; n.sampling_representation = representation; n.sampling_sliding_plan = sliding_plan;
  n.sampling_spline_points = spline_points; n.max_samples = max_samples; n.max_horizon = max_horizon;
  n.n_devices = n_devices; n.device = (devices && n_devices > 0) ? devices[0] : 0;
  if (devices) n.devices.assign(devices, devices + n_devices);
  p->Initialize(model, task, n);
  p->Allocate();
  return p;
}
void mjpc_planner_destroy(void* p) { delete (SamplingPlanner*)p; }
void mjpc_planner_reset(void* p, int horizon, const double* initial_repeated_action) { ((SamplingPlanner*)p)->Reset(horizon, initial_repeated_action); }
void mjpc_planner_set_state(void* p, const double* s, const double* m, const double* u, double t) { ((SamplingPlanner*)p)->SetState(s, m, u, t); }
void mjpc_planner_set_task(void* p, const MjpcHipTask* task) { ((SamplingPlanner*)p)->SetTask(task); }
void mjpc_planner_timings(void* p, double* a, double* b, double* c) { auto* q = (SamplingPlanner*)p; *a = q->noise_compute_time; *b = q->rollouts_compute_time; *c = q->policy_update_compute_time; }
void mjpc_planner_optimize_policy(void* p, int horizon) { ((SamplingPlanner*)p)->OptimizePolicy(horizon); }
void mjpc_planner_nominal_trajectory(void* p, int horizon) { ((SamplingPlanner*)p)->NominalTrajectory(horizon); }
void mjpc_planner_action_from_policy(void* p, double* action, double time, int use_previous) { ((SamplingPlanner*)p)->ActionFromPolicy(action, nullptr, time, use_previous != 0); }
int mjpc_planner_optimize_policy_candidates(void* p, int ncandidates, int horizon) { auto* q = (SamplingPlanner*)p; q->UpdateNominalPolicy(horizon); return q->OptimizePolicyCandidates(ncandidates, horizon); }
double mjpc_planner_candidate_score(void* p, int candidate) { return ((SamplingPlanner*)p)->CandidateScore(candidate); }
void mjpc_planner_action_from_candidate_policy(void* p, double* action, int candidate, double time) { ((SamplingPlanner*)p)->ActionFromCandidatePolicy(action, candidate, nullptr, time); }
void mjpc_planner_copy_candidate_to_policy(void* p, int candidate) { ((SamplingPlanner*)p)->CopyCandidateToPolicy(candidate); }
int mjpc_planner_winner(void* p) { return ((SamplingPlanner*)p)->winner; }
double mjpc_planner_improvement(void* p) { return ((SamplingPlanner*)p)->improvement; }
int mjpc_planner_num_parameters(void* p) { return ((SamplingPlanner*)p)->NumParameters(); }
void mjpc_planner_set_seed(void* p, unsigned long long seed, unsigned long long plan_iter) { auto* q = (SamplingPlanner*)p; q->seed = seed; q->plan_iter = plan_iter; }
void mjpc_planner_set_num_trajectory(void* p, int n) { ((SamplingPlanner*)p)->num_trajectory_ = n; }
void mjpc_planner_set_noise(void* p, const double* eps, const int* sel) { auto* q = (SamplingPlanner*)p; q->injected_noise_eps = eps; q->injected_noise_sel = sel; }
void mjpc_planner_returns(void* p, double* out, int n) { auto* q = (SamplingPlanner*)p; std::copy(q->returns.begin(), q->returns.begin() + n, out); }
// policy knots: returns P; fills times[P] and values[P*nu] when non-null
int mjpc_planner_policy(void* p, int which, double* times, double* values) {
  auto* q = (SamplingPlanner*)p;
  const mjpc_hip::SamplingPolicy& pol = which ? q->previous_policy : q->policy;
  int P = (int)pol.plan.Size();
  for (int i = 0; i < P; i++) {
    if (times) times[i] = pol.plan.NodeTime(i);
    if (values) std::copy(pol.plan.NodeValues(i), pol.plan.NodeValues(i) + pol.nu, values + (size_t)i * pol.nu);
  }
  return P;
}
// best trajectory: returns horizon (0 if none); copies the arrays that are non-null
int mjpc_planner_best_trajectory(void* p, double* states, double* actions, double* times, double* residual, double* costs,
                                 double* trace, double* total_return, int* failure) {
  const mjpc_hip::Trajectory* t = ((SamplingPlanner*)p)->BestTrajectory();
  if (!t) return 0;
  size_t H = (size_t)t->horizon;
  if (states) std::copy(t->states.begin(), t->states.begin() + H * t->dim_state, states);
  if (actions) std::copy(t->actions.begin(), t->actions.begin() + H * t->dim_action, actions);
  if (times) std::copy(t->times.begin(), t->times.begin() + H, times);
  if (residual) std::copy(t->residual.begin(), t->residual.begin() + H * t->dim_residual, residual);
  if (costs) std::copy(t->costs.begin(), t->costs.begin() + H, costs);
  if (trace) std::copy(t->trace.begin(), t->trace.begin() + H * t->dim_trace, trace);
  if (total_return) *total_return = t->total_return;
  if (failure) *failure = t->failure ? 1 : 0;
  return t->horizon;
}

// ---- CrossEntropyPlanner
void* mjpc_cem_create(const MjpcHipModel* model, const MjpcHipTask* task, double std_initial, double std_min, int trajectories,
                      int n_elite, int representation, int spline_points, int max_samples, int max_horizon, int device) {
  auto* p = new mjpc_hip::CrossEntropyPlanner();
  mjpc_hip::Numerics n;
  n.sampling_exploration[0] = std_initial; n.std_min = std_min; n.sampling_trajectories = trajectories; n.n_elite = n_elite;
  n.sampling_representation = representation; n.sampling_spline_points = spline_points; n.max_samples = max_samples;
  n.max_horizon = max_horizon; n.device = device;
  p->Initialize(model, task, n);
  p->Allocate();
  return p;
}
void mjpc_cem_destroy(void* p) { delete (mjpc_hip::CrossEntropyPlanner*)p; }
void mjpc_cem_reset(void* p, int horizon, const double* a) { ((mjpc_hip::CrossEntropyPlanner*)p)->Reset(horizon, a); }
void mjpc_cem_set_state(void* p, const double* s, const double* m, const double* u, double t) { ((mjpc_hip::CrossEntropyPlanner*)p)->SetState(s, m, u, t); }
void mjpc_cem_set_seed(void* p, unsigned long long seed, unsigned long long it) { auto* q = (mjpc_hip::CrossEntropyPlanner*)p; q->seed = seed; q->plan_iter = it; }
void mjpc_cem_set_noise(void* p, const double* eps) { ((mjpc_hip::CrossEntropyPlanner*)p)->injected_noise_eps = eps; }
void mjpc_cem_optimize_policy(void* p, int horizon) { ((mjpc_hip::CrossEntropyPlanner*)p)->OptimizePolicy(horizon); }
void mjpc_cem_nominal_trajectory(void* p, int horizon) { ((mjpc_hip::CrossEntropyPlanner*)p)->NominalTrajectory(horizon); }
void mjpc_cem_action_from_policy(void* p, double* a, double t, int prev) { ((mjpc_hip::CrossEntropyPlanner*)p)->ActionFromPolicy(a, nullptr, t, prev != 0); }
double mjpc_cem_improvement(void* p) { return ((mjpc_hip::CrossEntropyPlanner*)p)->improvement; }
void mjpc_cem_returns(void* p, double* out, int n) { auto* q = (mjpc_hip::CrossEntropyPlanner*)p; std::copy(q->returns.begin(), q->returns.begin() + n, out); }
void mjpc_cem_variance(void* p, double* out, int n) { auto* q = (mjpc_hip::CrossEntropyPlanner*)p; std::copy(q->variance.begin(), q->variance.begin() + n, out); }
int mjpc_cem_policy(void* p, double* times, double* values) {
  auto* q = (mjpc_hip::CrossEntropyPlanner*)p;
  int P = (int)q->policy.plan.Size();
  for (int i = 0; i < P; i++) {
    if (times) times[i] = q->policy.plan.NodeTime(i);
    if (values) std::copy(q->policy.plan.NodeValues(i), q->policy.plan.NodeValues(i) + q->policy.nu, values + (size_t)i * q->policy.nu);
  }
  return P;
}
int mjpc_cem_best_trajectory(void* p, double* states, double* actions, double* costs, double* total_return) {
  const mjpc_hip::Trajectory* t = ((mjpc_hip::CrossEntropyPlanner*)p)->BestTrajectory();
  size_t H = (size_t)t->horizon;
  if (states) std::copy(t->states.begin(), t->states.begin() + H * t->dim_state, states);
  if (actions) std::copy(t->actions.begin(), t->actions.begin() + H * t->dim_action, actions);
  if (costs) std::copy(t->costs.begin(), t->costs.begin() + H, costs);
  if (total_return) *total_return = t->total_return;
  return t->horizon;
}

// ---- RobustPlanner
void* mjpc_robust_create(const MjpcHipModel* model, const MjpcHipTask* task, const double* exploration, int trajectories, int representation,
                         int spline_points, int repetitions, int candidates, double xfrc_std, double xfrc_rate, int max_samples, int max_horizon,
                         int device) {
  auto* p = new mjpc_hip::RobustPlanner();
  mjpc_hip::Numerics n;
  n.sampling_exploration[0] = exploration[0]; n.sampling_exploration[1] = exploration[1];
  n.sampling_trajectories = trajectories; n.sampling_representation = representation; n.sampling_spline_points = spline_points;
  n.robust_repetitions = repetitions; n.robust_candidates = candidates; n.robust_xfrc = xfrc_std; n.robust_xfrc_rate = xfrc_rate;
  n.max_samples = max_samples; n.max_horizon = max_horizon; n.device = device;
  p->Initialize(model, task, n);
  p->Allocate();
  return p;
}
void mjpc_robust_destroy(void* p) { delete (mjpc_hip::RobustPlanner*)p; }
void mjpc_robust_reset(void* p, int horizon) { ((mjpc_hip::RobustPlanner*)p)->Reset(horizon, nullptr); }
void mjpc_robust_set_state(void* p, const double* s, const double* m, const double* u, double t) { ((mjpc_hip::RobustPlanner*)p)->SetState(s, m, u, t); }
void mjpc_robust_set_seed(void* p, unsigned long long delegate_seed, unsigned long long robust_seed, unsigned long long plan_iter) {
  auto* q = (mjpc_hip::RobustPlanner*)p; q->delegate.seed = delegate_seed; q->delegate.plan_iter = plan_iter; q->seed = robust_seed; q->plan_iter = plan_iter;
}
void mjpc_robust_optimize_policy(void* p, int horizon) { ((mjpc_hip::RobustPlanner*)p)->OptimizePolicy(horizon); }
void mjpc_robust_action_from_policy(void* p, double* a, double t) { ((mjpc_hip::RobustPlanner*)p)->ActionFromPolicy(a, nullptr, t, false); }
// out[0] = best candidate, out[1] = candidates scored, out[2] = repetitions; scores[ncand], noisy_returns[ncand*rep] optional
void mjpc_robust_last(void* p, int* out, double* scores, double* noisy_returns) {
  auto* q = (mjpc_hip::RobustPlanner*)p;
  out[0] = q->best_candidate; out[1] = (int)q->candidate_scores.size(); out[2] = q->nrepetitions_;
  if (scores) std::copy(q->candidate_scores.begin(), q->candidate_scores.end(), scores);
  if (noisy_returns) std::copy(q->noisy_returns.begin(), q->noisy_returns.end(), noisy_returns);
}
void* mjpc_robust_delegate(void* p) { return &((mjpc_hip::RobustPlanner*)p)->delegate; }

}  // extern "C"
