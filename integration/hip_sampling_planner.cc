// integration/hip_sampling_planner.cc — see hip_sampling_planner.h.  Host-side glue only: field-by-field views of mjModel /
// Task for the C ABI and forwarding of the Planner / RankedPlanner virtuals to mjpc_hip::SamplingPlanner.
#include "hip_sampling_planner.h"

#include <algorithm>
#include <cstring>
#include <memory>
#include <string>
#include <utility>

#include "mjpc/array_safety.h"
#include "mjpc/planners/sampling/planner.h"   // MinSamplingSplinePoints, MinNoiseStdDev ...
#include "mjpc/utilities.h"

namespace mjpc {

namespace mju = ::mujoco::util_mjpc;
using mjpc::spline::SplineInterpolation;

// frozen_state.cc
void FillFrozenState(const Task& task, const ResidualFn* residual, const mjModel* m, int task_id, std::vector<int>& ints,
                     std::vector<double>& dbls);

namespace {

// Task::Name() -> built-in device residual (include/mjpc_hip.h MJPC_TASK_*); -1: none
int DeviceResidual(const Task& task) {
  const std::string name = task.Name();
  if (name == "Cartpole") return MJPC_TASK_CARTPOLE;
  if (name == "Quadruped Flat") return MJPC_TASK_QUADRUPED;
  if (name == "Quadruped Hill") return MJPC_TASK_QUADRUPED_HILL;
  if (name == "Humanoid Track") return MJPC_TASK_HUMANOID_TRACK;
  if (name == "Humanoid Stand") return MJPC_TASK_HUMANOID_STAND;
  if (name == "Humanoid Walk") return MJPC_TASK_HUMANOID_WALK;
  if (name == "Humanoid Interact") return MJPC_TASK_HUMANOID_INTERACT;
  if (name == "Shadow") return MJPC_TASK_SHADOW_REORIENT;
  if (name == "Particle") return MJPC_TASK_PARTICLE_TIMEVARYING;
  if (name == "ParticleFixed") return MJPC_TASK_PARTICLE_FIXED;
  if (name == "Walker") return MJPC_TASK_WALKER;
  if (name == "Quadrotor") return MJPC_TASK_QUADROTOR;
  if (name == "Swimmer") return MJPC_TASK_SWIMMER;
  if (name == "Acrobot") return MJPC_TASK_ACROBOT;
  if (name == "FreeFingers") return MJPC_TASK_FINGERS;
  return -1;
}

template <class T>
std::vector<int> Widen(const T* src, int n) { return std::vector<int>(src, src + n); }

}  // namespace

bool HipSamplingPlanner::Supports(const Task& task) { return DeviceResidual(task) >= 0; }

// ---- views ------------------------------------------------------------------------------------------------------------
// MjpcHipModel mirrors mjModel's names: pointers are copied where layout and width agree, small vectors bridge the rest
// (mjtByte flags -> int32, actuator_gainprm[mjNGAIN] -> first 3, actuator_trnid[2] -> first, actuator_gear[6] -> first)
static void FillModelView(const mjModel* m, MjpcHipModel& v, std::vector<int>& jnt_limited, std::vector<int>& ctrllimited,
                          std::vector<int>& forcelimited, std::vector<int>& biastype, std::vector<int>& trntype,
                          std::vector<int>& trnid, std::vector<int>& tendon_limited, std::vector<int>& wrap_objid,
                          std::vector<double>& gainprm, std::vector<double>& biasprm, std::vector<double>& gear,
                          std::vector<double>& wrap_prm, std::vector<double>& mesh_vert, std::vector<double>& hfield_size,
                          std::vector<double>& hfield_data, std::vector<int>& act_i, std::vector<double>& dynprm, std::vector<int>& eq_active, std::vector<int>& actfrclimited) {
  std::memset(&v, 0, sizeof(v));
  v.struct_size = sizeof(MjpcHipModel);                  // checked by mjpc_hip_create against the library's own layout
  v.nq = m->nq; v.nv = m->nv; v.nu = m->nu; v.na = m->na; v.nbody = m->nbody; v.njnt = m->njnt; v.ngeom = m->ngeom;
  v.nsite = m->nsite; v.nmocap = m->nmocap; v.nuserdata = m->nuserdata; v.nkey = m->nkey; v.nexclude = m->nexclude;
  v.ntendon = m->ntendon; v.nwrap = m->nwrap;
  v.timestep = m->opt.timestep; mju_copy3(v.gravity, m->opt.gravity);
  v.impratio = m->opt.impratio; v.tolerance = m->opt.tolerance; v.ls_tolerance = m->opt.ls_tolerance;
  v.cone = m->opt.cone; v.iterations = m->opt.iterations; v.ls_iterations = m->opt.ls_iterations;
  v.disableflags = m->opt.disableflags; v.enableflags = m->opt.enableflags; v.solver = m->opt.solver; v.integrator = m->opt.integrator;
  v.noslip_iterations = m->opt.noslip_iterations; v.noslip_tolerance = m->opt.noslip_tolerance; v.neq = m->neq; v.meaninertia = m->stat.meaninertia;
  // what this view does not carry: found here, refused by mjpc_hip_create
  v.unsupported = 0;
  v.density = m->opt.density; v.viscosity = m->opt.viscosity; mju_copy3(v.wind, m->opt.wind);
  if (m->opt.density != 0 || m->opt.viscosity != 0)      // the inertia-box model travels in the view; the ellipsoid model does not
    for (int g = 0; g < m->ngeom; g++) if (m->geom_fluid[mjNFLUID * g] != 0) v.unsupported |= MJPC_UNSUP_FLUID;
  for (int i = 0; i < m->nu; i++) {
    if (m->actuator_gaintype[i] != mjGAIN_FIXED || (m->actuator_biastype[i] != mjBIAS_NONE && m->actuator_biastype[i] != mjBIAS_AFFINE)) v.unsupported |= MJPC_UNSUP_ACTUATOR_GAIN;
    // stateful actuators: integrator / filter / filterexact with one activation each travel in the view (actuator_dyntype ...)
    int dyn = m->actuator_dyntype[i];
    if (!(dyn == mjDYN_NONE || dyn == mjDYN_INTEGRATOR || dyn == mjDYN_FILTER || dyn == mjDYN_FILTEREXACT) ||
        (dyn != mjDYN_NONE && m->actuator_actnum[i] != 1) || m->actuator_actearly[i]) v.unsupported |= MJPC_UNSUP_ACTUATOR_DYN;
  }
  for (int w = 0; w < m->nwrap; w++) if (m->wrap_type[w] != mjWRAP_JOINT) v.unsupported |= MJPC_UNSUP_SPATIAL_TENDON;
  if (m->nflex > 0 || m->nplugin > 0) v.unsupported |= MJPC_UNSUP_FLEX_SKIN_PLUGIN;
  v.nconmax = 0; v.nefcmax = 0;      // engine defaults (32 contacts, 128 rows per candidate)
  v.body_parentid = m->body_parentid; v.body_rootid = m->body_rootid; v.body_weldid = m->body_weldid;
  v.body_mocapid = m->body_mocapid; v.body_jntnum = m->body_jntnum; v.body_jntadr = m->body_jntadr;
  v.body_dofnum = m->body_dofnum; v.body_dofadr = m->body_dofadr;
  v.body_pos = m->body_pos; v.body_quat = m->body_quat; v.body_ipos = m->body_ipos; v.body_iquat = m->body_iquat;
  v.body_mass = m->body_mass; v.body_subtreemass = m->body_subtreemass; v.body_inertia = m->body_inertia;
  v.body_invweight0 = m->body_invweight0; v.body_gravcomp = m->body_gravcomp;
  actfrclimited = Widen(m->jnt_actfrclimited, m->njnt); v.jnt_actfrclimited = actfrclimited.data(); v.jnt_actfrcrange = m->jnt_actfrcrange;
  v.jnt_type = m->jnt_type; v.jnt_qposadr = m->jnt_qposadr; v.jnt_dofadr = m->jnt_dofadr; v.jnt_bodyid = m->jnt_bodyid;
  jnt_limited = Widen(m->jnt_limited, m->njnt); v.jnt_limited = jnt_limited.data();
  v.jnt_pos = m->jnt_pos; v.jnt_axis = m->jnt_axis; v.jnt_stiffness = m->jnt_stiffness; v.jnt_range = m->jnt_range;
  v.jnt_margin = m->jnt_margin; v.jnt_solref = m->jnt_solref; v.jnt_solimp = m->jnt_solimp;
  v.qpos0 = m->qpos0; v.qpos_spring = m->qpos_spring;
  v.dof_bodyid = m->dof_bodyid; v.dof_jntid = m->dof_jntid; v.dof_parentid = m->dof_parentid;
  v.dof_armature = m->dof_armature; v.dof_damping = m->dof_damping; v.dof_frictionloss = m->dof_frictionloss;
  v.dof_invweight0 = m->dof_invweight0; v.dof_solref = m->dof_solref; v.dof_solimp = m->dof_solimp;
  v.geom_type = m->geom_type; v.geom_contype = m->geom_contype; v.geom_conaffinity = m->geom_conaffinity;
  v.geom_condim = m->geom_condim; v.geom_bodyid = m->geom_bodyid; v.geom_group = m->geom_group;
  v.geom_priority = m->geom_priority; v.geom_size = m->geom_size; v.geom_pos = m->geom_pos; v.geom_quat = m->geom_quat;
  v.geom_friction = m->geom_friction; v.geom_solmix = m->geom_solmix; v.geom_solref = m->geom_solref;
  v.geom_solimp = m->geom_solimp; v.geom_margin = m->geom_margin; v.geom_gap = m->geom_gap; v.geom_rbound = m->geom_rbound;
  v.exclude_signature = m->exclude_signature;
  // equality constraints: same layout (mjNEQDATA = 11, mjNREF = 2, mjNIMP = 5); eq_active0 is mjtByte
  eq_active = Widen(m->eq_active0, m->neq);
  v.eq_type = m->eq_type; v.eq_obj1id = m->eq_obj1id; v.eq_obj2id = m->eq_obj2id; v.eq_active0 = eq_active.data();
  v.eq_data = m->eq_data; v.eq_solref = m->eq_solref; v.eq_solimp = m->eq_solimp;
  v.site_bodyid = m->site_bodyid; v.site_pos = m->site_pos; v.site_quat = m->site_quat;
  trntype.clear(); trnid.clear(); gainprm.clear(); biasprm.clear(); gear.clear();
  for (int i = 0; i < m->nu; i++) {
    trntype.push_back(m->actuator_trntype[i]);
    trnid.push_back(m->actuator_trnid[2 * i]);
    for (int k = 0; k < 3; k++) {
      gainprm.push_back(m->actuator_gainprm[mjNGAIN * i + k]);
      biasprm.push_back(m->actuator_biasprm[mjNBIAS * i + k]);
    }
    gear.push_back(m->actuator_gear[6 * i]);
  }
  ctrllimited = Widen(m->actuator_ctrllimited, m->nu); forcelimited = Widen(m->actuator_forcelimited, m->nu);
  biastype = Widen(m->actuator_biastype, m->nu);
  v.actuator_trntype = trntype.data(); v.actuator_trnid = trnid.data();
  v.actuator_ctrllimited = ctrllimited.data(); v.actuator_forcelimited = forcelimited.data(); v.actuator_biastype = biastype.data();
  v.actuator_gainprm = gainprm.data(); v.actuator_biasprm = biasprm.data(); v.actuator_gear = gear.data(); v.actuator_gear6 = m->actuator_gear;
  v.actuator_ctrlrange = m->actuator_ctrlrange; v.actuator_forcerange = m->actuator_forcerange;
  act_i.assign(4 * m->nu, 0); dynprm.assign(m->nu, 0.0);
  for (int i = 0; i < m->nu; i++) {
    act_i[i] = m->actuator_dyntype[i]; act_i[m->nu + i] = m->actuator_actadr[i]; act_i[2 * m->nu + i] = m->actuator_actlimited[i];
    dynprm[i] = m->actuator_dynprm[mjNDYN * i];
    act_i[3 * m->nu + i] = m->actuator_trntype[i] == mjTRN_SITE ? m->actuator_trnid[2 * i + 1] : -1;      // reference site
  }
  v.actuator_dyntype = act_i.data(); v.actuator_actadr = act_i.data() + m->nu; v.actuator_actlimited = act_i.data() + 2 * m->nu;
  v.actuator_dynprm = dynprm.data(); v.actuator_actrange = m->actuator_actrange; v.actuator_refsite = act_i.data() + 3 * m->nu;
  // fixed tendons: wrap objects are joints, wrap_prm the coefficient (spatial tendons are refused by mjpc_hip_create)
  tendon_limited = Widen(m->tendon_limited, m->ntendon);
  wrap_objid.assign(m->wrap_objid, m->wrap_objid + m->nwrap); wrap_prm.assign(m->wrap_prm, m->wrap_prm + m->nwrap);
  v.tendon_adr = m->tendon_adr; v.tendon_num = m->tendon_num; v.tendon_limited = tendon_limited.data();
  v.wrap_objid = wrap_objid.data(); v.wrap_prm = wrap_prm.data(); v.tendon_range = m->tendon_range;
  v.tendon_margin = m->tendon_margin; v.tendon_solref_lim = m->tendon_solref_lim; v.tendon_solimp_lim = m->tendon_solimp_lim;
  v.tendon_invweight0 = m->tendon_invweight0;
  v.tendon_stiffness = m->tendon_stiffness; v.tendon_damping = m->tendon_damping; v.tendon_lengthspring = m->tendon_lengthspring;
  v.tendon_frictionloss = m->tendon_frictionloss; v.tendon_solref_fri = m->tendon_solref_fri; v.tendon_solimp_fri = m->tendon_solimp_fri;
  // convex meshes: mjModel.mesh_vert is float -> widened copy (mesh_vert_ member); hull = all vertices
  v.nmesh = m->nmesh; v.nmeshvert = m->nmeshvert; v.geom_dataid = m->geom_dataid; v.mesh_vertadr = m->mesh_vertadr; v.mesh_vertnum = m->mesh_vertnum;
  mesh_vert.assign(m->mesh_vert, m->mesh_vert + 3 * m->nmeshvert); v.mesh_vert = mesh_vert.data();
  // height fields: hfield_data is float as well (appended to the same widened pool)
  v.nhfield = m->nhfield; v.nhfielddata = m->nhfielddata; v.hfield_nrow = m->hfield_nrow; v.hfield_ncol = m->hfield_ncol; v.hfield_adr = m->hfield_adr;
  hfield_size.assign(m->hfield_size, m->hfield_size + 4 * m->nhfield); v.hfield_size = hfield_size.data();
  hfield_data.assign(m->hfield_data, m->hfield_data + m->nhfielddata); v.hfield_data = hfield_data.data();
  v.key_qpos = m->key_qpos; v.key_mpos = m->key_mpos;
}

// cost table of Task::Reset (task.cc:147-245) + the trace sensors (task.cc:193-198, utilities.cc:250-267)
static void FillTaskView(const Task& task, const mjModel* m, MjpcHipTask& t, std::vector<int>& norm, std::vector<int>& trace_type,
                         std::vector<int>& trace_id, std::vector<int>& ints, std::vector<double>& dbls) {
  std::memset(&t, 0, sizeof(t));
  t.struct_size = sizeof(MjpcHipTask);
  t.task_id = DeviceResidual(task);
  t.num_residual = task.num_residual; t.num_term = task.num_term; t.num_trace = task.num_trace;
  t.dim_norm_residual = task.dim_norm_residual.data(); t.num_norm_parameter = task.num_norm_parameter.data();
  norm.assign(task.norm.begin(), task.norm.end()); t.norm = norm.data();
  t.weight = task.weight.data(); t.norm_parameter = task.norm_parameter.data(); t.risk = task.risk;
  t.num_parameter = static_cast<int>(task.parameters.size()); t.parameters = task.parameters.data();
  trace_type.clear(); trace_id.clear();
  for (int i = 0; i < task.num_trace; i++) {
    char name[32];
    mju::sprintf_arr(name, "trace%i", i);
    int id = mj_name2id(m, mjOBJ_SENSOR, name);
    trace_type.push_back(id >= 0 ? m->sensor_objtype[id] : MJPC_OBJ_XBODY);
    trace_id.push_back(id >= 0 ? m->sensor_objid[id] : 0);
  }
  t.trace_objtype = trace_type.data(); t.trace_objid = trace_id.data();
  // frozen ResidualFn members (frozen_state.cc): ids resolved at Reset and the mode / gait state Transition wrote
  std::unique_ptr<ResidualFn> frozen = task.Residual();       // the copy every planner takes per plan step (agent.cc:290)
  FillFrozenState(task, frozen.get(), m, t.task_id, ints, dbls);
  t.num_int = static_cast<int>(ints.size()); t.int_data = ints.data();
  t.num_dbl = static_cast<int>(dbls.size()); t.dbl_data = dbls.data();
}

// ---- Planner virtuals ---------------------------------------------------------------------------------------------------
void HipSamplingPlanner::Initialize(mjModel* model, const Task& task) {
  this->model = model;
  this->task = &task;
  mjpc_hip::Numerics n;                                     // exactly the numerics of sampling/planner.cc:53-67
  n.sampling_exploration[0] = GetNumberOrDefault(0.1, model, "sampling_exploration");
  int se_id = mj_name2id(model, mjOBJ_NUMERIC, "sampling_exploration");
  if (se_id >= 0 && model->numeric_size[se_id] > 1) n.sampling_exploration[1] = model->numeric_data[model->numeric_adr[se_id] + 1];
  n.sampling_trajectories = GetNumberOrDefault(10, model, "sampling_trajectories");
  n.sampling_representation = GetNumberOrDefault(static_cast<int>(SplineInterpolation::kCubicSpline), model, "sampling_representation");
  n.sampling_sliding_plan = GetNumberOrDefault(0, model, "sampling_sliding_plan");
  n.sampling_spline_points = GetNumberOrDefault(kMaxTrajectoryHorizon, model, "sampling_spline_points");
  n.max_samples = kMaxTrajectoryHip;
  n.max_horizon = kMaxTrajectoryHorizon;
  n.n_devices = n_devices = GetNumberOrDefault(1, model, "sampling_devices");
  noise_exploration[0] = n.sampling_exploration[0]; noise_exploration[1] = n.sampling_exploration[1];
  num_trajectory_ = n.sampling_trajectories;
  interpolation_ = static_cast<SplineInterpolation>(n.sampling_representation);
  sliding_plan_ = n.sampling_sliding_plan;
  if (num_trajectory_ > kMaxTrajectoryHip) mju_error_i("Too many trajectories, %d is the maximum allowed.", kMaxTrajectoryHip);
  FillModelView(model, model_view_, jnt_limited_, ctrllimited_, forcelimited_, biastype_, trntype_, trnid_, tendon_limited_,
                wrap_objid_, gainprm_, biasprm_, gear_, wrap_prm_, mesh_vert_, hfield_size_, hfield_data_, act_i_, dynprm_, eq_active_, actfrclimited_);
  FillTaskView(task, model, task_view_, norm_, trace_type_, trace_id_, task_int_, task_dbl_);
  mjpc_hip::SetErrorHandler([](const char* msg) { mju_error("HipSamplingPlanner: %s", msg); });
  impl_.Initialize(&model_view_, &task_view_, n);           // creates the engines (model may have changed: old ones dropped)
  winner = 0;
}

void HipSamplingPlanner::Allocate() {
  int num_state = model->nq + model->nv + model->na;
  state.resize(num_state); mocap.resize(7 * model->nmocap); userdata.resize(model->nuserdata);
  policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  previous_policy.Allocate(model, *task, kMaxTrajectoryHorizon);
  best_.Initialize(num_state, model->nu, task->num_residual, task->num_trace, kMaxTrajectoryHorizon);
  best_.Allocate(kMaxTrajectoryHorizon);
  trajectory_cache_.clear(); policy_cache_.clear();
  trajectory_cache_.resize(kMaxTrajectoryHip); policy_cache_.resize(kMaxTrajectoryHip);      // entries are made on first use
  trajectory_stamp_.assign(kMaxTrajectoryHip, 0); policy_stamp_.assign(kMaxTrajectoryHip, 0);
  impl_.Allocate();
  winner = -1;
}

void HipSamplingPlanner::Reset(int horizon, const double* initial_repeated_action) {
  std::fill(state.begin(), state.end(), 0.0); std::fill(mocap.begin(), mocap.end(), 0.0);
  std::fill(userdata.begin(), userdata.end(), 0.0);
  time = 0.0;
  policy.Reset(horizon, initial_repeated_action);
  previous_policy.Reset(horizon, initial_repeated_action);
  best_.Reset(kMaxTrajectoryHorizon);
  impl_.Reset(horizon, initial_repeated_action);
  improvement = 0.0;
  winner = 0;
}

void HipSamplingPlanner::SetState(const State& s) {           // sampling/planner.cc:146-149
  s.CopyTo(state.data(), mocap.data(), userdata.data(), &time);
  impl_.SetState(state.data(), mocap.data(), userdata.data(), time);
}

void HipSamplingPlanner::RefreshTask() {
  // GUI-written settings are snapshotted at the top of a plan step (sampling/planner.cc:153-156)
  impl_.num_trajectory_ = num_trajectory_; impl_.interpolation_ = static_cast<int>(interpolation_);
  impl_.sliding_plan_ = sliding_plan_;
  impl_.noise_exploration[0] = noise_exploration[0]; impl_.noise_exploration[1] = noise_exploration[1];
  FillTaskView(*task, model, task_view_, norm_, trace_type_, trace_id_, task_int_, task_dbl_);
  impl_.SetTask(&task_view_);                                 // only the task block travels, asynchronously
}

// iLQS rewrites sampling.policy.plan before it calls OptimizePolicy (ilqs/planner.cc:162-169) and the GUI moves
// policy.num_spline_points: what the plan step starts from is this object's policy, as in the reference
void HipSamplingPlanner::PushPolicyToImpl() {
  const std::shared_lock<std::shared_mutex> lock(mtx_);
  impl_.policy.num_spline_points = policy.num_spline_points;
  impl_.policy.plan.Clear();
  impl_.policy.plan.SetInterpolation(static_cast<mjpc_hip::SplineInterpolation>(policy.plan.Interpolation()));
  for (int p = 0; p < static_cast<int>(policy.plan.Size()); p++) {
    mjpc::spline::TimeSpline::ConstNode node = std::as_const(policy.plan).NodeAt(p);
    impl_.policy.plan.AddNode(node.time(), node.values().data());
  }
}

void HipSamplingPlanner::SyncFromImpl() {
  plan_stamp_++;                                              // candidates cached from the previous plan step are stale
  {
    const std::unique_lock<std::shared_mutex> lock(mtx_);
    for (int which = 0; which < 2; which++) {
      const mjpc_hip::SamplingPolicy& src = which ? impl_.previous_policy : impl_.policy;
      SamplingPolicy& dst = which ? previous_policy : policy;
      dst.plan.Clear();
      dst.plan.SetInterpolation(interpolation_);
      for (int p = 0; p < static_cast<int>(src.plan.Size()); p++)
        dst.plan.AddNode(src.plan.NodeTime(p), absl::MakeConstSpan(src.plan.NodeValues(p), model->nu));
      dst.num_spline_points = src.num_spline_points;
    }
  }
  winner = impl_.winner; improvement = impl_.improvement; trajectory_order = impl_.trajectory_order;
  noise_compute_time = impl_.noise_compute_time; rollouts_compute_time = impl_.rollouts_compute_time;
  policy_update_compute_time = impl_.policy_update_compute_time;
}

static void CopyTrajectory(const mjpc_hip::Trajectory& src, Trajectory& dst) {
  dst.horizon = src.horizon; dst.total_return = src.total_return; dst.failure = src.failure;
  std::copy(src.states.begin(), src.states.begin() + static_cast<size_t>(src.horizon) * src.dim_state, dst.states.begin());
  std::copy(src.actions.begin(), src.actions.begin() + static_cast<size_t>(src.horizon) * src.dim_action, dst.actions.begin());
  std::copy(src.times.begin(), src.times.begin() + src.horizon, dst.times.begin());
  std::copy(src.residual.begin(), src.residual.begin() + static_cast<size_t>(src.horizon) * src.dim_residual, dst.residual.begin());
  std::copy(src.costs.begin(), src.costs.begin() + src.horizon, dst.costs.begin());
  std::copy(src.trace.begin(), src.trace.begin() + static_cast<size_t>(src.horizon) * src.dim_trace, dst.trace.begin());
}

void HipSamplingPlanner::OptimizePolicy(int horizon, ThreadPool& /*pool: the GPU replaces the worker threads*/) {
  RefreshTask();
  PushPolicyToImpl();
  impl_.OptimizePolicy(horizon);                              // UpdateNominalPolicy -> rollouts on the GPU(s) -> winner adoption
  last_horizon_ = horizon;
  SyncFromImpl();
  CopyTrajectory(impl_.trajectory_winner, best_);
}

void HipSamplingPlanner::NominalTrajectory(int horizon, ThreadPool&) {
  RefreshTask();
  impl_.NominalTrajectory(horizon);
  CopyTrajectory(impl_.trajectory_winner, best_);
}

void HipSamplingPlanner::ActionFromPolicy(double* action, const double* s, double t, bool use_previous) {
  impl_.ActionFromPolicy(action, s, t, use_previous);         // any thread, concurrently with planning (shared lock inside)
}

const Trajectory* HipSamplingPlanner::BestTrajectory() { return winner >= 0 ? &best_ : nullptr; }

int HipSamplingPlanner::OptimizePolicyCandidates(int ncandidates, int horizon, ThreadPool&) {
  RefreshTask();
  PushPolicyToImpl();
  impl_.UpdateNominalPolicy(horizon);
  int n = impl_.OptimizePolicyCandidates(ncandidates, horizon);
  last_horizon_ = horizon;
  plan_stamp_++;
  trajectory_order = impl_.trajectory_order;
  return n;
}
double HipSamplingPlanner::CandidateScore(int candidate) const { return impl_.CandidateScore(candidate); }
void HipSamplingPlanner::ActionFromCandidatePolicy(double* action, int candidate, const double* s, double t) {
  impl_.ActionFromCandidatePolicy(action, candidate, s, t);
}
void HipSamplingPlanner::CopyCandidateToPolicy(int candidate) {
  impl_.CopyCandidateToPolicy(candidate);
  SyncFromImpl();
  CopyTrajectory(impl_.trajectory_winner, best_);
}

// trajectory[i] / candidate_policy[i] of the reference (ilqs/planner.cc:177-198 reads them): copied from the owning device on
// first use after a plan step, then served from the cache (two indices in one expression must not share storage)
const Trajectory& HipSamplingPlanner::FetchTrajectory(int i) {
  if (i < 0 || i >= kMaxTrajectoryHip) mju_error_i("HipSamplingPlanner: trajectory[%d] out of range", i);
  if (i == winner && winner >= 0) return best_;
  std::unique_ptr<Trajectory>& slot = trajectory_cache_[i];
  if (!slot) {
    slot = std::make_unique<Trajectory>();
    slot->Initialize(model->nq + model->nv + model->na, model->nu, task->num_residual, task->num_trace, kMaxTrajectoryHorizon);
    slot->Allocate(kMaxTrajectoryHorizon);
  }
  if (trajectory_stamp_[i] != plan_stamp_) {
    impl_.FetchCandidateUnranked(i);
    CopyTrajectory(impl_.trajectory_winner, *slot);
    trajectory_stamp_[i] = plan_stamp_;
  }
  return *slot;
}
const SamplingPolicy& HipSamplingPlanner::FetchCandidatePolicy(int i) {
  if (i < 0 || i >= kMaxTrajectoryHip) mju_error_i("HipSamplingPlanner: candidate_policy[%d] out of range", i);
  std::unique_ptr<SamplingPolicy>& slot = policy_cache_[i];
  if (!slot) {
    slot = std::make_unique<SamplingPolicy>();
    slot->Allocate(model, *task, kMaxTrajectoryHorizon);
  }
  if (policy_stamp_[i] != plan_stamp_) {
    std::vector<double> knots(static_cast<size_t>(impl_.KnotTimes().size()) * model->nu);
    impl_.CandidateKnotsUnranked(i, knots.data());
    slot->plan.Clear();
    slot->plan.SetInterpolation(interpolation_);
    for (size_t p = 0; p < impl_.KnotTimes().size(); p++)
      slot->plan.AddNode(impl_.KnotTimes()[p], absl::MakeConstSpan(knots.data() + p * model->nu, model->nu));
    slot->num_spline_points = static_cast<int>(impl_.KnotTimes().size());
    policy_stamp_[i] = plan_stamp_;
  }
  return *slot;
}

// ---- GUI side (render thread, unsynchronised reads like the reference) ----------------------------------------------------
void HipSamplingPlanner::Traces(mjvScene* scn) {              // sampling/planner.cc:388-434
  float color[4] = {1.0, 1.0, 1.0, 1.0};
  double width = GetNumberOrDefault(3, model, "agent_sample_width");          // pixels
  double zero3[3] = {0};
  double zero9[9] = {0};
  const Trajectory* best = BestTrajectory();
  int N = num_trajectory_, ntr = 3 * task->num_trace;
  if (!best || last_horizon_ < 2 || ntr == 0) return;         // nothing planned yet (the reference would dereference a null best)
  int H = last_horizon_;
  traces_.resize(static_cast<size_t>(N) * H * ntr);
  impl_.AllTraces(traces_.data());                            // one D2H copy instead of N trajectory reads: [N][H][3 * num_trace]
  for (int k = 0; k < N; k++) {
    if (k == winner) continue;                                // the winner is drawn by the agent itself
    for (int i = 0; i < best->horizon - 1; i++) {
      if (scn->ngeom + task->num_trace > scn->maxgeom) break;
      for (int j = 0; j < task->num_trace; j++) {
        const double* a = traces_.data() + (static_cast<size_t>(k) * H + i) * ntr + 3 * j;
        const double* b = a + ntr;
        mjv_initGeom(&scn->geoms[scn->ngeom], mjGEOM_LINE, zero3, zero3, zero9, color);
        mjv_makeConnector(&scn->geoms[scn->ngeom], mjGEOM_LINE, width, a[0], a[1], a[2], b[0], b[1], b[2]);
        scn->ngeom += 1;
      }
    }
  }
}

void HipSamplingPlanner::GUI(mjUI& ui) {                      // the widgets of sampling/planner.cc:437-461 on this object's members
  mjuiDef defSampling[] = {{mjITEM_SLIDERINT, "Rollouts", 2, &num_trajectory_, "0 1"},
                           {mjITEM_SELECT, "Spline", 2, &interpolation_, "Zero\nLinear\nCubic"},
                           {mjITEM_SLIDERINT, "Spline Pts", 2, &policy.num_spline_points, "0 1"},
                           {mjITEM_SLIDERNUM, "Noise Std", 2, noise_exploration, "0 1"},
                           {mjITEM_SLIDERNUM, "Noise Std2", 2, noise_exploration + 1, "0 1"},
                           {mjITEM_CHECKBYTE, "Sliding plan", 2, &sliding_plan_, ""},
                           {mjITEM_END}};
  mju::sprintf_arr(defSampling[0].other, "%i %i", 1, kMaxTrajectoryHip);      // the engine lifts kMaxTrajectory
  mju::sprintf_arr(defSampling[2].other, "%i %i", MinSamplingSplinePoints, MaxSamplingSplinePoints);
  mju::sprintf_arr(defSampling[3].other, "%f %f", MinNoiseStdDev, MaxNoiseStdDev);
  mjui_add(&ui, defSampling);
}

void HipSamplingPlanner::Plots(mjvFigure* fig_planner, mjvFigure* fig_timer, int planner_shift, int timer_shift, int planning,
                               int* shift) {                  // same lines, names and ranges as sampling/planner.cc:464-512
  double planner_bounds[2] = {-6.0, 6.0};
  double timer_bounds[2] = {0.0, 1.0};
  const int pl = planner_shift;
  PlotUpdateData(fig_planner, planner_bounds, fig_planner->linedata[pl][0] + 1, mju_log10(mju_max(improvement, 1.0e-6)), 100, pl, 0, 1, -100);
  mju::strcpy_arr(fig_planner->linename[pl], "Improvement");
  fig_planner->range[1][0] = planner_bounds[0];
  fig_planner->range[1][1] = planner_bounds[1];
  const struct { const char* name; double microseconds; } timers[3] = {
      {"Noise", noise_compute_time}, {"Rollout", rollouts_compute_time}, {"Policy Update", policy_update_compute_time}};
  for (int k = 0; k < 3; k++) {
    const int line = k + timer_shift;
    PlotUpdateData(fig_timer, timer_bounds, fig_timer->linedata[line][0] + 1, 1.0e-3 * timers[k].microseconds * planning, 100, line, 0, 1, -100);
    mju::strcpy_arr(fig_timer->linename[line], timers[k].name);
  }
  shift[0] += 1;       // planner figure: one line
  shift[1] += 3;       // timer figure: three
}

}  // namespace mjpc
