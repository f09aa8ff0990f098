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
//   quadruped: the mode / gait state machine of QuadrupedFlat::TransitionLocked (quadruped.cc:224-390): phase bookkeeping,
//   automatic gait switching on the filtered com speed, gait parameter / weight tables, the Walk goal trajectory, Flip
//   entry / exit — reading the simulator's kinematic frame of the previous step like the reference reads mjData;
//   particle / cartpole / humanoid stand / walk: none.
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

// what Task::Transition reads from mjData after a simulation step (poses / sensors of the state the step started from)
struct SimFrame {
  std::vector<double> xpos, xmat, site_xpos, subtree_com, subtree_linvel;
  bool valid = false;
};

class Simulator {
 public:
  Simulator(const MjpcHipModel* model, const MjpcHipTask* task, int device = 0);
  ~Simulator();
  Simulator(const Simulator&) = delete;
  Simulator& operator=(const Simulator&) = delete;
  // mj_step with data->ctrl = ctrl: advances s, returns CostValue(sensordata) of (x_t, u_t); residual[nr] optional
  double Step(SimState& s, const double* ctrl, double* residual = nullptr);
  // mj_forward at s without advancing (testspeed.cc:75: data is forwarded once before the loop); fills frame()
  void Forward(const SimState& s);
  const SimFrame& frame() const { return frame_; }
  void SetTask(const MjpcHipTask* task);
  bool failed() const { return failure_; }
  int nq, nv, nu, nmocap, nr;
  double timestep;

 private:
  MjpcHipEngine* engine_ = nullptr;
  std::vector<double> states_, residual_, costs_, times_, actions_, trace_;
  SimFrame frame_;
  void FetchFrame();
  bool failure_ = false;
};

// Task::Transition on the host; may edit the state (mocap targets), the task block, or both
using TransitionFn = std::function<void(const MjpcHipModel&, SimState&, HostTask&, const SimFrame&)>;
// MJPC_TASK_* -> built-in transition (no-op where the reference has none); `mode` is Task::mode, the mode the user asks for
// (quadruped: 0 Quadruped, 1 Biped, 2 Walk, 3 Scramble, 4 Flip, quadruped.h:40-47)
// `mode_time`: simulation time at which the user switches to `mode` (the reference resets stateful modes to Quadruped on the
// first Transition after a data reset, quadruped.cc:226-232, so Walk / Flip can only be entered later)
TransitionFn TransitionForTask(int task_id, int mode = 0, double mode_time = 0.0);

// QuadrupedFlat::TransitionLocked (quadruped.cc:224-390) with its ResidualFn state split between the task block the kernel
// reads (HostTask int/dbl data, parameters, weights; layout of modelgen/tasks.py quadruped()) and the host-only members here
struct QuadrupedTransition {
  int mode = 0;                       // Task::mode (requested)
  int current_mode = 0;               // residual_.current_mode_
  double last_transition_time = -1, com_vel[2] = {0, 0}, gait_switch_time = 0;
  std::vector<double> save_weight; double save_gait_switch = 0;
  void operator()(const MjpcHipModel& model, SimState& s, HostTask& t, const SimFrame& f);
};

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
