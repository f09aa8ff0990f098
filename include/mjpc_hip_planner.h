// mjpc_hip_planner.h — C++ host side above the C ABI: the reference's SamplingPlanner surface, with the
// rollouts forwarded to the HIP engine (include/mjpc_hip.h).
//
// Mirrors (names, argument meaning, error behaviour):
//   mjpc/spline/spline.h:41-276              TimeSpline
//   mjpc/planners/sampling/policy.h,.cc      SamplingPolicy
//   mjpc/trajectory.h:74-86                  Trajectory (public arrays)
//   mjpc/planners/sampling/planner.h:51-162  SamplingPlanner (+ RankedPlanner virtuals, planners/planner.h:84-101)
//   mjpc/planners/cross_entropy/planner.h:32-147  CrossEntropyPlanner (same rollout engine, elite mean/variance update)
//   mjpc/planners/robust/robust_planner.h:31-80   RobustPlanner (top-k candidates x R noisy rollouts on a second engine)
// Differences forced by the boundary: `mjModel*` / `const Task&` become the ABI's MjpcHipModel / MjpcHipTask views
// plus the planner's <custom><numeric> settings (Numerics); `ThreadPool&` arguments are gone (the GPU is the pool);
// `State` is passed as its raw arrays (State::CopyTo, mjpc/states/state.cc:128-135).
#ifndef MJPC_HIP_PLANNER_H_
#define MJPC_HIP_PLANNER_H_

#include <array>
#include <deque>
#include <shared_mutex>
#include <vector>

#include "mjpc_hip.h"

namespace mjpc_hip {

inline constexpr int kMaxTrajectoryHorizon = 512;   // mjpc/trajectory.h:27
inline constexpr int MinSamplingSplinePoints = 1;    // mjpc/planners/sampling/planner.h:35-36
inline constexpr int MaxSamplingSplinePoints = 36;

enum SplineInterpolation : int { kZeroSpline = 0, kLinearSpline = 1, kCubicSpline = 2 };

// time-indexed spline of `dim`-vectors: zero / linear / cubic-Hermite sampling (spline.cc:103-156,240-277)
class TimeSpline {
 public:
  explicit TimeSpline(int dim = 0, SplineInterpolation interpolation = kZeroSpline, int initial_capacity = 1);
  std::size_t Size() const { return times_.size(); }
  int Dim() const { return dim_; }
  SplineInterpolation Interpolation() const { return interpolation_; }
  void SetInterpolation(SplineInterpolation interpolation) { interpolation_ = interpolation; }
  void Reserve(int num_nodes);
  void Sample(double time, double* values) const;             // values[dim]
  std::vector<double> Sample(double time) const;
  int DiscardBefore(double time);
  void Clear();
  double* AddNode(double time, const double* values = nullptr);   // returns the node's values; zeros when values == nullptr
  double NodeTime(int index) const { return times_[index]; }
  const double* NodeValues(int index) const { return values_[index].data(); }
  double* NodeValues(int index) { return values_[index].data(); }

 private:
  double Slope(int node_index, int value_index) const;
  SplineInterpolation interpolation_;
  int dim_;
  std::deque<double> times_;
  std::deque<std::vector<double>> values_;
};

// mjpc/planners/sampling/policy.cc:30-78
class SamplingPolicy {
 public:
  void Allocate(const MjpcHipModel* model, int num_spline_points);
  void Reset(int horizon, const double* initial_repeated_action = nullptr);
  void Action(double* action, const double* state, double time) const;     // spline sample, clamped to ctrlrange
  void CopyFrom(const SamplingPolicy& policy, int horizon);
  TimeSpline plan;
  int num_spline_points = 0;
  int nu = 0;
  std::vector<double> ctrlrange;
};

// mjpc/trajectory.h:74-86
struct Trajectory {
  int horizon = 0, dim_state = 0, dim_action = 0, dim_residual = 0, dim_trace = 0;
  std::vector<double> states, actions, times, residual, costs, trace;
  double total_return = 0;
  bool failure = false;
};

// configuration errors abort like mju_error (planner.cc:69-72) unless a handler is installed
void SetErrorHandler(void (*handler)(const char*));

struct Numerics {                       // the planner's <custom><numeric> entries (planner.cc:53-67, policy.cc:36-37)
  double sampling_exploration[2] = {0.1, 0.0};
  int sampling_trajectories = 10;
  int sampling_representation = kCubicSpline;
  int sampling_sliding_plan = 0;
  int sampling_spline_points = kMaxTrajectoryHorizon;
  int max_samples = 4096;              // kMaxTrajectory is 128 in the reference (planners/planner.h:28); lifted here
  double std_min = 0.1;                // cross-entropy: minimum std (cross_entropy/planner.cc:58)
  int robust_repetitions = 5;          // robust planner (robust_planner.cc:45-57)
  int robust_candidates = -1;          // default sampling_trajectories / robust_repetitions
  double robust_xfrc = 0.1, robust_xfrc_rate = 0.1;
  int n_elite = -1;                    // cross-entropy: default max(sampling_trajectories / 10, 2) (planner.cc:63-64)
  int max_horizon = kMaxTrajectoryHorizon;   // device trajectory buffers are sized max_samples x max_horizon
  int device = 0;                      // first HIP device ordinal
  int n_devices = 1;                   // SamplingPlanner: GPUs the candidate batch is sharded over (one engine each)
  std::vector<int> devices;            // optional explicit ordinals (repeats allowed); default device, device+1, ...
};

class SamplingPlanner {
 public:
  SamplingPlanner() = default;
  ~SamplingPlanner();
  SamplingPlanner(const SamplingPlanner&) = delete;
  SamplingPlanner& operator=(const SamplingPlanner&) = delete;

  // ---- Planner virtuals (planners/planner.h:38-80); errors abort like mju_error unless a handler is installed
  void Initialize(const MjpcHipModel* model, const MjpcHipTask* task, const Numerics& numerics);
  void Allocate();
  void Reset(int horizon, const double* initial_repeated_action = nullptr);
  void SetState(const double* state, const double* mocap, const double* userdata, double time);
  void OptimizePolicy(int horizon);
  void NominalTrajectory(int horizon);
  void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false);
  const Trajectory* BestTrajectory();
  int NumParameters() { return policy.num_spline_points * nu_; }
  // ---- RankedPlanner virtuals (planners/planner.h:84-101)
  int OptimizePolicyCandidates(int ncandidates, int horizon);
  double CandidateScore(int candidate) const;
  void ActionFromCandidatePolicy(double* action, int candidate, const double* state, double time);
  void CopyCandidateToPolicy(int candidate);
  // ---- sampling-specific (planner.h:95-112)
  void UpdateNominalPolicy(int horizon);
  void SetTask(const MjpcHipTask* task);               // fresh ResidualFn copy per plan step (agent.cc:290)

  // knot values of candidate `candidate` (ranked order) of the last OptimizePolicyCandidates, [P*nu]; P via KnotTimes()
  void CandidateKnots(int candidate, double* out);
  // by batch index (unranked): the reference's trajectory[i] / candidate_policy[i] (ilqs/planner.cc:98-198); the rows land in
  // trajectory_winner
  void FetchCandidateUnranked(int index);
  void CandidateKnotsUnranked(int index, double* out);
  // every candidate's trace rows of the last plan step [num_trajectory][horizon][3*num_trace] in one copy (Traces, planner.cc:388-434)
  void AllTraces(double* out);
  const std::vector<double>& KnotTimes() const { return knot_times_; }
  // ---- public members other code reads/writes in the reference (planner.h:115-162)
  SamplingPolicy policy, previous_policy;
  std::vector<double> state, mocap, userdata;
  double time = 0;
  Trajectory trajectory_winner;                        // trajectory[winner]; other candidates stay on the device
  std::vector<double> returns;                         // trajectory[i].total_return
  std::vector<int> failures;
  std::vector<int> trajectory_order;
  int winner = 0;
  double noise_exploration[2] = {0.1, 0.0};
  int num_trajectory_ = 10;
  int interpolation_ = kCubicSpline;
  int sliding_plan_ = 0;
  double improvement = 0;
  double noise_compute_time = 0, rollouts_compute_time = 0, policy_update_compute_time = 0;   // microseconds
  unsigned long long seed = 0x5EED;
  unsigned long long plan_iter = 0;
  // reproducible-noise hook (the reference's absl::BitGen is unseedable): optional injected tensors
  const double* injected_noise_eps = nullptr;   // [num_trajectory * P * nu]
  const int* injected_noise_sel = nullptr;      // [num_trajectory]

 private:
  void FetchCandidate(int global_index);
  MjpcHipMulti* engine_ = nullptr;             // one rollout engine per GPU
  Numerics numerics_;
  int nq_ = 0, nv_ = 0, na_ = 0, ns_ = 0, nu_ = 0, nmocap_ = 0, nuserdata_ = 0, nr_ = 0, ntrace_ = 0;
  double timestep_ = 0;
  std::vector<double> ctrlrange_;
  SamplingPolicy winner_policy_;               // candidate_policy[winner]
  TimeSpline plan_scratch_;
  std::vector<double> knot_times_, knot_values_, winner_knots_;
  int last_horizon_ = 0, fetched_ = -1, nominal_horizon_ = 0;
  MjpcHipEngine* nominal_engine_ = nullptr;            // one-candidate engine of NominalTrajectory()
  mutable std::shared_mutex mtx_;
};

// mjpc/planners/cross_entropy/planner.{h,cc}: all N candidates are perturbed with a per-parameter std (elite variance,
// floored at std_min), the nominal (resampled) policy is rolled out as one extra candidate, the new policy is the mean
// of the n_elite best candidates.  Quirks kept: the variance loop reads the best elite's parameters for every elite
// (planner.cc:240-253); previous_policy is never refreshed by OptimizePolicy.
class CrossEntropyPlanner {
 public:
  CrossEntropyPlanner() = default;
  ~CrossEntropyPlanner();
  CrossEntropyPlanner(const CrossEntropyPlanner&) = delete;
  CrossEntropyPlanner& operator=(const CrossEntropyPlanner&) = delete;

  void Initialize(const MjpcHipModel* model, const MjpcHipTask* task, const Numerics& numerics);
  void Allocate();
  void Reset(int horizon, const double* initial_repeated_action = nullptr);
  void SetState(const double* state, const double* mocap, const double* userdata, double time);
  void OptimizePolicy(int horizon);
  void NominalTrajectory(int horizon);
  void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false);
  void ResamplePolicy(int horizon);
  const Trajectory* BestTrajectory();                  // the nominal trajectory (planner.cc:418-420)
  int NumParameters() { return policy.num_spline_points * nu_; }
  void SetTask(const MjpcHipTask* task);

  SamplingPolicy policy, resampled_policy, previous_policy;
  std::vector<double> state, mocap, userdata;
  double time = 0;
  Trajectory nominal_trajectory;
  std::vector<double> returns;                         // trajectory[i].total_return, i < num_trajectory
  std::vector<int> failures;
  std::vector<int> trajectory_order;
  std::vector<double> parameters_scratch, times_scratch, variance;
  double std_initial_ = 0.1, std_min_ = 0.1;
  int n_elite_ = 2;
  double improvement = 0;
  double noise_compute_time = 0, rollouts_compute_time = 0, policy_update_compute_time = 0;   // microseconds
  int interpolation_ = kZeroSpline;
  int num_trajectory_ = 10;
  unsigned long long seed = 0x5EED;
  unsigned long long plan_iter = 0;
  const double* injected_noise_eps = nullptr;          // [(num_trajectory + 1) * P * nu], standard normal (tests)

 private:
  MjpcHipEngine* engine_ = nullptr;
  Numerics numerics_;
  int nq_ = 0, nv_ = 0, na_ = 0, ns_ = 0, nu_ = 0, nmocap_ = 0, nuserdata_ = 0, nr_ = 0, ntrace_ = 0;
  double timestep_ = 0;
  std::vector<double> ctrlrange_, knot_values_, noise_std_, all_knots_;
  int last_horizon_ = 0;
  mutable std::shared_mutex mtx_;
};

// mjpc/planners/robust/robust_planner.{h,cc}: the delegate ranks its candidates; the best `ncandidates_` are rolled out
// `nrepetitions_` times each with Ornstein-Uhlenbeck force noise on every body (Trajectory::NoisyRollout) in ONE launch of a
// second engine (explicit candidate policies); the candidate with the best mean return over the delegate's own score and
// its valid noisy rollouts is adopted.
class RobustPlanner {
 public:
  RobustPlanner() = default;
  ~RobustPlanner();
  RobustPlanner(const RobustPlanner&) = delete;
  RobustPlanner& operator=(const RobustPlanner&) = delete;

  void Initialize(const MjpcHipModel* model, const MjpcHipTask* task, const Numerics& numerics);
  void Allocate();
  void Reset(int horizon, const double* initial_repeated_action = nullptr);
  void SetState(const double* state, const double* mocap, const double* userdata, double time);
  void OptimizePolicy(int horizon);
  void NominalTrajectory(int horizon) { delegate.NominalTrajectory(horizon); }
  void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false) { delegate.ActionFromPolicy(action, state, time, use_previous); }
  const Trajectory* BestTrajectory() { return delegate.BestTrajectory(); }
  int NumParameters() { return delegate.NumParameters(); }
  void SetTask(const MjpcHipTask* task);

  SamplingPlanner delegate;                            // delegate_ (a RankedPlanner)
  int ncandidates_ = 12, nrepetitions_ = 5;
  double xfrc_std_ = 0.1, xfrc_rate_ = 0.1;
  std::vector<double> noisy_returns;                   // [ncandidates * nrepetitions] of the last OptimizePolicy
  std::vector<int> noisy_failures;
  std::vector<double> candidate_scores;                // mean score per candidate (robust_planner.cc:131-146)
  int best_candidate = -1;
  unsigned long long seed = 0x0B057;
  unsigned long long plan_iter = 0;

 private:
  MjpcHipEngine* engine_ = nullptr;                    // noisy rollouts (the delegate's engine keeps its plan for CopyCandidateToPolicy)
  Numerics numerics_;
  int nu_ = 0, ns_ = 0, nmocap_ = 0, nuserdata_ = 0;
  std::vector<double> state_, mocap_, userdata_, cand_knots_;
  double time_ = 0;
};

}  // namespace mjpc_hip
#endif  // MJPC_HIP_PLANNER_H_
