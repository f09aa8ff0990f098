// mjpc_hip_testspeed.h — closed-loop harness: the reference's headless benchmark loop (mjpc/testspeed.cc:44-129,
// `SynchronousPlanningCost`) with both the planner's rollouts AND the simulated world on the HIP engine.
//
//   for every simulation step:  Task::Transition -> state.Set -> ActionFromPolicy -> mj_step -> CostValue(sensordata)
//                               -> every k-th step: PlanIteration from the state captured BEFORE the step
//
// `Simulator` is the world: one mj_step = one engine launch with a single candidate, horizon 2 and a zero-order knot holding
// the control; row 0 of that rollout carries the residual / cost at (x_t, u_t) exactly like `data->sensordata` after
// mj_step.  `Transition` restates Task::Transition for the built-in tasks on the host (mjpc/task.cc:141-145):
//   humanoid tracking: mocap targets interpolated between key frames, reference time on motion start (tracking.cc:223-267);
//   particle / cartpole: none;  quadruped in its default mode (Quadruped, manual gait): a no-op after the first call
//   (quadruped.cc:224-390 only acts on mode / gait / parameter changes), so the frozen task block stays valid.
#ifndef MJPC_HIP_TESTSPEED_H_
#define MJPC_HIP_TESTSPEED_H_

#include <functional>
#include <vector>

#include "mjpc_hip.h"
#include "mjpc_hip_planner.h"

namespace mjpc_hip {

// mutable host copy of a task block (weights, parameters, frozen ResidualFn state) with an ABI view onto it
struct HostTask {
  explicit HostTask(const MjpcHipTask& t);
  const MjpcHipTask* view();
  MjpcHipTask base;
  std::vector<int> dim_norm_residual, norm, num_norm_parameter, trace_objtype, trace_objid, int_data;
  std::vector<double> weight, norm_parameter, parameters, dbl_data;
};

struct SimState {
  std::vector<double> state, mocap, userdata;
  double time = 0;
};

class Simulator {
 public:
  Simulator(const MjpcHipModel* model, const MjpcHipTask* task, int device = 0);
  ~Simulator();
  Simulator(const Simulator&) = delete;
  Simulator& operator=(const Simulator&) = delete;
  // mj_step with data->ctrl = ctrl: advances s, returns CostValue(sensordata) of (x_t, u_t); residual[nr] optional
  double Step(SimState& s, const double* ctrl, double* residual = nullptr);
  void SetTask(const MjpcHipTask* task);
  bool failed() const { return failure_; }
  int nq, nv, nu, nmocap, nr;
  double timestep;

 private:
  MjpcHipEngine* engine_ = nullptr;
  std::vector<double> states_, residual_, costs_, times_, actions_, trace_;
  bool failure_ = false;
};

// Task::Transition on the host; may edit the state (mocap targets), the task block, or both
using TransitionFn = std::function<void(const MjpcHipModel&, SimState&, HostTask&)>;
TransitionFn TransitionForTask(int task_id);      // MJPC_TASK_* -> built-in transition (no-op where the reference has none)

struct PlannerOps {                               // the four Planner calls the loop needs (planners/planner.h:38-80)
  std::function<void(const SimState&)> SetState;
  std::function<void(int horizon)> OptimizePolicy;
  std::function<void(double* action, double time)> ActionFromPolicy;
  std::function<void(const MjpcHipTask*)> SetTask;
};
PlannerOps Ops(SamplingPlanner& p);
PlannerOps Ops(CrossEntropyPlanner& p);

struct TestspeedResult {
  int total_steps = 0, plan_steps = 0;
  double total_cost = 0, average_cost = 0, wall_seconds = 0, realtime_factor = 0, plan_seconds = 0;
  std::vector<double> cost_per_step;
  bool failure = false;
};

// testspeed.cc:44-129.  `horizon` = Agent::steps_ (agent.cc:107: agent_horizon / agent_timestep + 1).
TestspeedResult SynchronousPlanningCost(const MjpcHipModel& model, HostTask& task, PlannerOps planner, Simulator& sim, SimState& s,
                                        int horizon, int steps_per_planning_iteration, double total_time,
                                        const TransitionFn& transition);

}  // namespace mjpc_hip
#endif  // MJPC_HIP_TESTSPEED_H_
