// integration/hip_sampling_planner.h — the reference-side binding: an mjpc::RankedPlanner that runs its rollouts on MI355X.
//
// Drop this file and hip_sampling_planner.cc into an MJPC checkout as mjpc/planners/sampling_hip/, add them to
// mjpc/CMakeLists.txt, link libmjpc_hip.so (+ include/ of this repository) and create the class in
// mjpc/planners/include.cc:44 instead of SamplingPlanner (README.md in this directory).  It implements every pure virtual
// of mjpc/planners/planner.h:32-102 and keeps the public members other reference code reads
// (mjpc/planners/sampling/planner.h:115-162): policy, previous_policy, candidate_policy[], trajectory[], trajectory_order,
// winner, time, state / mocap / userdata, model, task, noise_exploration, num_trajectory_, interpolation_, sliding_plan_,
// timing fields.
//
// NOT compiled in this repository: it needs <mujoco/mujoco.h>, abseil and the MJPC headers, none of which exist in the build
// image.  Everything below it — include/mjpc_hip.h (C ABI), include/mjpc_hip_planner.h (C++ planner
// with the reference's semantics) — is compiled and tested here.
#ifndef MJPC_PLANNERS_SAMPLING_HIP_PLANNER_H_
#define MJPC_PLANNERS_SAMPLING_HIP_PLANNER_H_

#include <mujoco/mujoco.h>

#include <atomic>
#include <shared_mutex>
#include <vector>

#include "mjpc/planners/planner.h"
#include "mjpc/planners/sampling/policy.h"
#include "mjpc/spline/spline.h"
#include "mjpc/states/state.h"
#include "mjpc/task.h"
#include "mjpc/threadpool.h"
#include "mjpc/trajectory.h"
#include "mjpc_hip.h"
#include "mjpc_hip_planner.h"

namespace mjpc {

// the engine lifts kMaxTrajectory (128): BASELINE runs 256 .. 4096 candidates per plan step
inline constexpr int kMaxTrajectoryHip = 4096;

class HipSamplingPlanner : public RankedPlanner {
 public:
  HipSamplingPlanner() = default;
  ~HipSamplingPlanner() override = default;

  // ---- Planner (mjpc/planners/planner.h:38-80)
  void Initialize(mjModel* model, const Task& task) override;
  void Allocate() override;
  void Reset(int horizon, const double* initial_repeated_action = nullptr) override;
  void SetState(const State& state) override;
  void OptimizePolicy(int horizon, ThreadPool& pool) override;
  void NominalTrajectory(int horizon, ThreadPool& pool) override;
  void ActionFromPolicy(double* action, const double* state, double time, bool use_previous = false) override;
  const Trajectory* BestTrajectory() override;
  void Traces(mjvScene* scn) override;
  void GUI(mjUI& ui) override;
  void Plots(mjvFigure* fig_planner, mjvFigure* fig_timer, int planner_shift, int timer_shift, int planning,
             int* shift) override;
  int NumParameters() override { return policy.num_spline_points * model->nu; }
  // ---- RankedPlanner (mjpc/planners/planner.h:84-101)
  int OptimizePolicyCandidates(int ncandidates, int horizon, ThreadPool& pool) override;
  double CandidateScore(int candidate) const override;
  void ActionFromCandidatePolicy(double* action, int candidate, const double* state, double time) override;
  void CopyCandidateToPolicy(int candidate) override;

  // true when the active task has a built-in device residual; otherwise the caller keeps the stock SamplingPlanner
  // (a GPU cannot call back into a host ResidualFn)
  static bool Supports(const Task& task);

  // ---- members with the reference's names (sampling/planner.h:115-162)
  mjModel* model = nullptr;
  const Task* task = nullptr;
  std::vector<double> state;
  double time = 0;
  std::vector<double> mocap;
  std::vector<double> userdata;
  SamplingPolicy policy;                 // (guarded by mtx_)
  SamplingPolicy previous_policy;
  // candidate i of the last plan step, materialised on demand from the device (i is an index into the last batch)
  const SamplingPolicy& candidate_policy(int i);
  const Trajectory& trajectory(int i);
  std::vector<int> trajectory_order;
  double noise_exploration[2] = {0};
  mjpc::spline::SplineInterpolation interpolation_ = mjpc::spline::SplineInterpolation::kZeroSpline;
  int winner = 0;
  double improvement = 0;
  std::atomic<double> noise_compute_time{0};
  double rollouts_compute_time = 0;
  double policy_update_compute_time = 0;
  std::uint8_t sliding_plan_ = false;
  int num_trajectory_ = 10;
  int n_devices = 1;                     // GPUs the candidate batch is sharded over (numeric "sampling_devices", default 1)

 private:
  void SyncFromImpl();                   // impl_ -> policy / previous_policy / winner / improvement / timings
  void RefreshTask();                    // fresh frozen ResidualFn state + cost weights for this plan step (agent.cc:290)
  mjpc_hip::SamplingPlanner impl_;       // the reference's planner logic over the C ABI (include/mjpc_hip_planner.h)
  // views handed to the engine; the vectors own what mjModel stores in another width / stride
  MjpcHipModel model_view_{};
  MjpcHipTask task_view_{};
  std::vector<int> jnt_limited_, ctrllimited_, forcelimited_, biastype_, trntype_, trnid_, tendon_limited_, wrap_objid_, trace_type_, trace_id_,
      norm_, task_int_, act_i_, eq_active_, actfrclimited_;
  std::vector<double> gainprm_, biasprm_, gear_, wrap_prm_, mesh_vert_, hfield_size_, hfield_data_, task_dbl_, dynprm_;
  Trajectory best_, scratch_trajectory_;
  SamplingPolicy scratch_policy_;
  std::vector<double> traces_;           // [N][H][3 * num_trace] of the last plan (Traces)
  int last_horizon_ = 0;
  mutable std::shared_mutex mtx_;
};

}  // namespace mjpc

#endif  // MJPC_PLANNERS_SAMPLING_HIP_PLANNER_H_
