/* mjpc_hip_planner_c.h — flat C view of the C++ host planner (include/mjpc_hip_planner.h), exported by
 * libmjpc_hip.so for bindings that cannot take C++ classes (ctypes tests, the Python front end).
 * Each function forwards to the method of the same name; the reference interface each one stands for:
 *   mjpc_spline_*   -> mjpc::spline::TimeSpline              (mjpc/spline/spline.h:41-276)
 *   mjpc_planner_*  -> mjpc::SamplingPlanner / RankedPlanner (mjpc/planners/sampling/planner.h:51-162,
 *                                                             mjpc/planners/planner.h:38-101)
 * Handles are opaque; errors go through the installed handler (default: print + abort, like mju_error).
 */
#ifndef MJPC_HIP_PLANNER_C_H_
#define MJPC_HIP_PLANNER_C_H_
#include "mjpc_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

void mjpc_planner_set_error_handler(void (*handler)(const char *message));

/* TimeSpline */
void *mjpc_spline_create(int dim, int interpolation);
void mjpc_spline_destroy(void *spline);
int mjpc_spline_size(void *spline);
void mjpc_spline_add_node(void *spline, double time, const double *values /* NULL = zeros */);
void mjpc_spline_sample(void *spline, double time, double *out /* [dim] */);
int mjpc_spline_discard_before(void *spline, double time);
void mjpc_spline_clear(void *spline);
void mjpc_spline_set_interpolation(void *spline, int interpolation);

/* SamplingPlanner: create = Initialize + Allocate (planner.cc:40-133) */
void *mjpc_planner_create(const MjpcHipModel *model, const MjpcHipTask *task, const double *exploration /* [2] */,
                          int trajectories, int representation, int sliding_plan, int spline_points,
                          int max_samples, int max_horizon, int device);
/* the same planner with the candidate batch of every plan step sharded over n_devices GPUs (one engine each, elite picked
 * across them; devices[k] = HIP ordinals, repeats allowed for a 1-GPU rehearsal) */
void *mjpc_planner_create_sharded(const MjpcHipModel *model, const MjpcHipTask *task, const double *exploration /* [2] */,
                                  int trajectories, int representation, int sliding_plan, int spline_points,
                                  int max_samples, int max_horizon, int n_devices, const int *devices);
void mjpc_planner_destroy(void *planner);
void mjpc_planner_reset(void *planner, int horizon, const double *initial_repeated_action);
void mjpc_planner_set_state(void *planner, const double *state, const double *mocap, const double *userdata, double time);
void mjpc_planner_set_task(void *planner, const MjpcHipTask *task);
void mjpc_planner_optimize_policy(void *planner, int horizon);
void mjpc_planner_nominal_trajectory(void *planner, int horizon);
void mjpc_planner_action_from_policy(void *planner, double *action, double time, int use_previous);
int mjpc_planner_optimize_policy_candidates(void *planner, int ncandidates, int horizon);
double mjpc_planner_candidate_score(void *planner, int candidate);
void mjpc_planner_action_from_candidate_policy(void *planner, double *action, int candidate, double time);
void mjpc_planner_copy_candidate_to_policy(void *planner, int candidate);
int mjpc_planner_winner(void *planner);
double mjpc_planner_improvement(void *planner);
int mjpc_planner_num_parameters(void *planner);
void mjpc_planner_set_seed(void *planner, unsigned long long seed, unsigned long long plan_iter);
void mjpc_planner_set_num_trajectory(void *planner, int num_trajectory);
void mjpc_planner_set_noise(void *planner, const double *eps, const int *sel);   /* borrowed until the next call */
void mjpc_planner_returns(void *planner, double *out, int n);
int mjpc_planner_policy(void *planner, int previous, double *times, double *values);          /* returns P */
int mjpc_planner_best_trajectory(void *planner, double *states, double *actions, double *times, double *residual,
                                 double *costs, double *trace, double *total_return, int *failure); /* returns H */
void mjpc_planner_timings(void *planner, double *noise_us, double *rollouts_us, double *policy_update_us);

/* CrossEntropyPlanner (mjpc/planners/cross_entropy/planner.h:32-147): create = Initialize + Allocate */
void *mjpc_cem_create(const MjpcHipModel *model, const MjpcHipTask *task, double std_initial, double std_min, int trajectories,
                      int n_elite, int representation, int spline_points, int max_samples, int max_horizon, int device);
void mjpc_cem_destroy(void *planner);
void mjpc_cem_reset(void *planner, int horizon, const double *initial_repeated_action);
void mjpc_cem_set_state(void *planner, const double *state, const double *mocap, const double *userdata, double time);
void mjpc_cem_set_seed(void *planner, unsigned long long seed, unsigned long long plan_iter);
void mjpc_cem_set_noise(void *planner, const double *eps);     /* borrowed until the next call */
void mjpc_cem_optimize_policy(void *planner, int horizon);
void mjpc_cem_nominal_trajectory(void *planner, int horizon);
void mjpc_cem_action_from_policy(void *planner, double *action, double time, int use_previous);
double mjpc_cem_improvement(void *planner);
void mjpc_cem_returns(void *planner, double *out, int n);
void mjpc_cem_variance(void *planner, double *out, int n);
int mjpc_cem_policy(void *planner, double *times, double *values);                               /* returns P */
int mjpc_cem_best_trajectory(void *planner, double *states, double *actions, double *costs, double *total_return);   /* returns H */

/* RobustPlanner (mjpc/planners/robust/robust_planner.h:31-80) over a SamplingPlanner delegate */
void *mjpc_robust_create(const MjpcHipModel *model, const MjpcHipTask *task, const double *exploration, int trajectories, int representation,
                         int spline_points, int repetitions, int candidates, double xfrc_std, double xfrc_rate, int max_samples,
                         int max_horizon, int device);
void mjpc_robust_destroy(void *planner);
void mjpc_robust_reset(void *planner, int horizon);
void mjpc_robust_set_state(void *planner, const double *state, const double *mocap, const double *userdata, double time);
void mjpc_robust_set_seed(void *planner, unsigned long long delegate_seed, unsigned long long robust_seed, unsigned long long plan_iter);
void mjpc_robust_optimize_policy(void *planner, int horizon);
void mjpc_robust_action_from_policy(void *planner, double *action, double time);
void mjpc_robust_last(void *planner, int *out /* [3] best, ncand, rep */, double *scores, double *noisy_returns);
void *mjpc_robust_delegate(void *planner);     /* the SamplingPlanner handle (mjpc_planner_* calls), owned by the robust planner */

/* Closed-loop harness (include/mjpc_hip_testspeed.h; mjpc/testspeed.cc:44-129 `SynchronousPlanningCost`): world and planner on the
 * HIP engine.  planner_kind 0 = handle from mjpc_planner_create, 1 = handle from mjpc_cem_create.  state / mocap are in-out;
 * cost_per_step[ceil(total_time/timestep)] optional; out[6] = {average_cost, wall_seconds, realtime_factor, plan_seconds,
 * plan_steps, failure}.  Returns the total cost. */
double mjpc_testspeed_run(const MjpcHipModel *model, const MjpcHipTask *task, void *planner, int planner_kind, double *state,
                          double *mocap, double time0, int horizon, int steps_per_planning_iteration, double total_time, int device,
                          double *cost_per_step, double *out, int mode /* Task::mode */, double mode_time /* when the user selects it */,
                          double *task_parameters_out /* [num_parameter] or NULL */);

#ifdef __cplusplus
}
#endif
#endif /* MJPC_HIP_PLANNER_C_H_ */
