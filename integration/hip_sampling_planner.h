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
// Not linked in this repository: it needs MuJoCo and abseil, neither of which exists in the build image (SURVEY.md section
// 8c).  What IS checked here: tests/test_integration_syntax.py runs `g++ -std=c++20 -fsyntax-only` over integration/*.cc and
// over the reference's own ilqs/planner.cc with SamplingPlanner swapped for this class, against the reference's real mjpc/
// headers and declaration-only stand-ins for <mujoco/*.h> / <absl/*> (tests/stubs/) — a syntax and type check, nothing more.
// Everything below this class — include/mjpc_hip.h (C ABI), include/mjpc_hip_planner.h (C++ planner with the reference's
// semantics) — is compiled, linked and tested here.
#ifndef MJPC_PLANNERS_SAMPLING_HIP_PLANNER_H_
#define MJPC_PLANNERS_SAMPLING_HIP_PLANNER_H_

#include <mujoco/mujoco.h>

#include <atomic>
#include <cstdint>
#include <memory>
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
  // trajectory[i] / candidate_policy[i] of the reference are arrays of kMaxTrajectory objects (sampling/planner.h:126,133);
  // here the candidates live on the device, so the two members are indexable views: operator[] copies candidate i of the
  // last plan step from its GPU the first time it is asked for and hands out the cached copy until the next plan step
  // (ilqs/planner.cc:177-198 compiles unchanged: `sampling.trajectory[sampling.winner].total_return`).
  template <class T>
  class CandidateArray {
   public:
    const T& operator[](int i) const { return (owner_->*fetch_)(i); }
   private:
    friend class HipSamplingPlanner;
    CandidateArray(HipSamplingPlanner* owner, const T& (HipSamplingPlanner::*fetch)(int)) : owner_(owner), fetch_(fetch) {}
    HipSamplingPlanner* owner_;
    const T& (HipSamplingPlanner::*fetch_)(int);
  };
  CandidateArray<SamplingPolicy> candidate_policy{this, &HipSamplingPlanner::FetchCandidatePolicy};
  CandidateArray<Trajectory> trajectory{this, &HipSamplingPlanner::FetchTrajectory};
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
  void PushPolicyToImpl();               // policy (iLQS and the GUI write it, ilqs/planner.cc:162-169) -> impl_
  const Trajectory& FetchTrajectory(int i);
  const SamplingPolicy& FetchCandidatePolicy(int i);
  void RefreshTask();                    // fresh frozen ResidualFn state + cost weights for this plan step (agent.cc:290)
  mjpc_hip::SamplingPlanner impl_;       // the reference's planner logic over the C ABI (include/mjpc_hip_planner.h)
  // views handed to the engine; the vectors own what mjModel stores in another width / stride
  MjpcHipModel model_view_{};
  MjpcHipTask task_view_{};
  std::vector<int> jnt_limited_, ctrllimited_, forcelimited_, biastype_, trntype_, trnid_, tendon_limited_, wrap_objid_, trace_type_, trace_id_,
      norm_, task_int_, act_i_, eq_active_, actfrclimited_;
  std::vector<double> gainprm_, biasprm_, gear_, wrap_prm_, mesh_vert_, hfield_size_, hfield_data_, task_dbl_, dynprm_;
  Trajectory best_;
  // lazily filled per-candidate copies behind trajectory[] / candidate_policy[]; an entry is valid while its stamp equals plan_stamp_
  std::vector<std::unique_ptr<Trajectory>> trajectory_cache_;
  std::vector<std::unique_ptr<SamplingPolicy>> policy_cache_;
  std::vector<std::uint64_t> trajectory_stamp_, policy_stamp_;
  std::uint64_t plan_stamp_ = 1;
  std::vector<double> traces_;           // [N][H][3 * num_trace] of the last plan (Traces)
  int last_horizon_ = 0;
  mutable std::shared_mutex mtx_;
};

}  // namespace mjpc

#endif  // MJPC_PLANNERS_SAMPLING_HIP_PLANNER_H_
