/*
 * mjpc_hip.h — C ABI of the MI355X-native Predictive-Sampling rollout engine.
 *
 * This is the drop-in boundary for ONE path of hartikainen/mujoco_mpc:
 *   mjpc/planners/sampling/planner.cc:342-380  (SamplingPlanner::Rollouts, the ThreadPool fan-out)
 *   mjpc/trajectory.cc:100-210                 (Trajectory::NoisyRollout, the horizon loop)
 *   mjpc/trajectory.cc:312-326                 (Trajectory::UpdateReturn)
 *   mjpc/planners/sampling/planner.cc:168-181  (partial_sort -> winner)
 *
 * Plain C, plain pointers and sizes, no C++/torch types.  Every array is fp64
 * (mjtNum == double in the reference) or int32.
 *
 * MjpcHipModel mirrors the fields of MuJoCo's `mjModel` the path reads (same names, same
 * layout, same units) so that the reference-side shim is a field-by-field pointer copy
 * from the `mjModel*` handed to `SamplingPlanner::Initialize`
 * (mjpc/planners/sampling/planner.cc:40-76).  See INTEGRATION.md.
 */
#ifndef MJPC_HIP_H_
#define MJPC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (numeric values identical to MuJoCo 3.1.4 / MJPC) ------------------------- */
enum { MJPC_JNT_FREE = 0, MJPC_JNT_BALL = 1, MJPC_JNT_SLIDE = 2, MJPC_JNT_HINGE = 3 };
enum {
  MJPC_GEOM_PLANE = 0, MJPC_GEOM_HFIELD = 1, MJPC_GEOM_SPHERE = 2, MJPC_GEOM_CAPSULE = 3,
  MJPC_GEOM_ELLIPSOID = 4, MJPC_GEOM_CYLINDER = 5, MJPC_GEOM_BOX = 6, MJPC_GEOM_MESH = 7
};
enum { MJPC_CONE_PYRAMIDAL = 0, MJPC_CONE_ELLIPTIC = 1 };
enum { MJPC_SOL_NEWTON = 2, MJPC_INT_EULER = 0, MJPC_INT_IMPLICIT = 2, MJPC_INT_IMPLICITFAST = 3 };
/* model features outside this view (MjpcHipModel.unsupported) */
enum { MJPC_EQ_CONNECT = 0, MJPC_EQ_WELD = 1, MJPC_EQ_JOINT = 2, MJPC_EQ_TENDON = 3 };      /* mjtEq */
enum { MJPC_DYN_NONE = 0, MJPC_DYN_INTEGRATOR = 1, MJPC_DYN_FILTER = 2, MJPC_DYN_FILTEREXACT = 3 };   /* mjtDyn */
enum { MJPC_UNSUP_FLUID = 1,          /* fluid forces of the ellipsoid model (a geom with fluidshape="ellipsoid") */
       MJPC_UNSUP_GRAVCOMP = 2,       /* (no longer set: body_gravcomp travels in the view) */
       MJPC_UNSUP_ACTUATOR_GAIN = 4,  /* gaintype other than fixed, biastype other than none / affine */
       MJPC_UNSUP_ACTUATOR_DYN = 8,   /* dyntype muscle / user, actnum != 1, actearly */
       MJPC_UNSUP_SPATIAL_TENDON = 16,/* wrap objects other than joints */
       MJPC_UNSUP_JNT_ACTFRC = 32,    /* (no longer set: jnt_actfrclimited / jnt_actfrcrange travel in the view) */
       MJPC_UNSUP_FLEX_SKIN_PLUGIN = 64 /* flexes, plugins, user callbacks other than the residual */ };
enum { MJPC_DSBL_CONSTRAINT = 1 << 0, MJPC_DSBL_EQUALITY = 1 << 1, MJPC_DSBL_FRICTIONLOSS = 1 << 2, MJPC_DSBL_LIMIT = 1 << 3,
       MJPC_DSBL_SENSOR = 1 << 12, MJPC_DSBL_MIDPHASE = 1 << 13, MJPC_ENBL_OVERRIDE = 1 << 0, MJPC_ENBL_MULTICCD = 1 << 4 };
enum { MJPC_DSBL_CONTACT = 1 << 4 };            /* mjDSBL_CONTACT */
enum { MJPC_BIAS_NONE = 0, MJPC_BIAS_AFFINE = 1 };
/* spline interpolation, mjpc/spline/spline.h:30-34 */
enum { MJPC_SPLINE_ZERO = 0, MJPC_SPLINE_LINEAR = 1, MJPC_SPLINE_CUBIC = 2 };
/* norm types, mjpc/norm.h:27-38 */
enum {
  MJPC_NORM_NULL = -1, MJPC_NORM_QUADRATIC = 0, MJPC_NORM_L22 = 1, MJPC_NORM_L2 = 2,
  MJPC_NORM_COSH = 3, MJPC_NORM_POWER = 5, MJPC_NORM_SMOOTHABS = 6, MJPC_NORM_SMOOTHABS2 = 7,
  MJPC_NORM_RECTIFY = 8
};
/* built-in device residuals (the reference dispatches through the global mjcb_sensor
 * callback, mjpc/app.cc:110-126; a GPU cannot call back into host C++) */
enum {
  MJPC_TASK_PARTICLE = 0,   /* mjpc/test/testdata/particle_residual.h:33-43 */
  MJPC_TASK_CARTPOLE = 1,   /* mjpc/tasks/cartpole/cartpole.cc:36-49 */
  MJPC_TASK_QUADRUPED = 2,  /* mjpc/tasks/quadruped/quadruped.cc:33-221 */
  MJPC_TASK_COPYSTATE = 3,  /* residual = [qpos,qvel] (mjpc/test/agent/rollout_test.cc:40-60) */
  MJPC_TASK_HUMANOID_TRACK = 4, /* mjpc/tasks/humanoid/tracking/tracking.cc:94-216 */
  MJPC_TASK_HUMANOID_STAND = 5, /* mjpc/tasks/humanoid/stand/stand.cc:41-94 */
  MJPC_TASK_HUMANOID_WALK = 6,  /* mjpc/tasks/humanoid/walk/walk.cc:44-166 */
  MJPC_TASK_SHADOW_REORIENT = 7, /* mjpc/tasks/shadow_reorient/hand.cc:37-84; int_data = [palm site, cube body, goal body, key] */
  MJPC_TASK_WALKER = 8,     /* mjpc/tasks/walker/walker.cc:39-57; int_data = [torso body]; parameters = [height goal, speed goal] */
  MJPC_TASK_ACROBOT = 9,    /* mjpc/tasks/acrobot/acrobot.cc:34-49; int_data = [goal site, tip site] */
  MJPC_TASK_QUADRUPED_HILL = 10, /* mjpc/tasks/quadruped/quadruped.cc:726-768; int_data = [trunk body, sites FR FL RR RL, stage];
                                 * dbl_data = the stage goals [nstage][7] (mpos, mquat of task_hill.xml:82-101) for the host Transition */
  MJPC_TASK_PARTICLE_TIMEVARYING = 11, /* mjpc/tasks/particle/particle.cc:30-50 ("Particle"): tip position - Lissajous goal of data->time,
                                 * tip velocity, control (6 residuals); int_data = [tip site] */
  MJPC_TASK_PARTICLE_FIXED = 12, /* particle.cc:68-73 ("ParticleFixed"): the same with goal = mocap_pos[0..1] */
  MJPC_TASK_SWIMMER = 14,        /* mjpc/tasks/swimmer/swimmer.cc:33-46: control (nu), nose - target in the plane (2); int_data = [nose geom, targets reached] */
  MJPC_TASK_HUMANOID_INTERACT = 15, /* mjpc/tasks/humanoid/interact/interact.cc:31-186: int_data = [body torso, pelvis, foot_right, foot_left, head,
                                  * shin_right, shin_left, has facing target, (body1, body2) x 5 (-1: pair not selected)]; dbl_data = [facing target
                                  * x, y, (local_pos1[3], local_pos2[3]) x 5]; parameters = [head height goal, torso height goal] */
  MJPC_TASK_FINGERS = 16,        /* mjpc/tasks/fingers/fingers.cc:31-62: finger_a - object, finger_b - object (framepos of the bodies), distance of the three
                                  * object sites to their targets, control; int_data = [body finger_a, finger_b, object, sites 0 1 2, sites 0t 1t 2t] */
  MJPC_TASK_QUADROTOR = 13       /* mjpc/tasks/quadrotor/quadrotor.cc:37-60: position - goal, linear / angular velocity, control - hover thrust (13 of
                                 * the 15 declared residuals are written); int_data = [body, stage]; dbl_data = stage goals [nstage][7] */
};
enum { MJPC_TRN_JOINT = 0, MJPC_TRN_TENDON = 3, MJPC_TRN_SITE = 4 };   /* mjtTrn values of the supported actuator transmissions */
enum { MJPC_OBJ_BODY = 1, MJPC_OBJ_XBODY = 2, MJPC_OBJ_GEOM = 5, MJPC_OBJ_SITE = 6 };

/* failure[] bits: why a candidate's rollout stopped (any bit => total_return = MJPC_MAX_RETURN, like
 * CheckWarnings -> failure, mjpc/utilities.cc:787-799 + trajectory.cc:169-173) */
enum {
  MJPC_WARN_BADQPOS = 1, MJPC_WARN_BADQVEL = 2, MJPC_WARN_BADQACC = 4,   /* mjWARN_BADQPOS / BADQVEL / BADQACC */
  MJPC_WARN_CONTACTFULL = 8, MJPC_WARN_CNSTRFULL = 16,                    /* mjWARN_CONTACTFULL / CNSTRFULL (nconmax / nefcmax) */
  MJPC_WARN_RAY = 32,                                                     /* Ground() ray hit nothing (utilities.cc:549-552) */
  MJPC_WARN_SYNC = 64,                                                    /* engine-internal: a wave hand-shake timed out */
  MJPC_WARN_UNSUPPORTED = 128   /* a geom pair without a collider came within reach of contact (none left among the pairs create() accepts) */
};
#define MJPC_MINVAL 1e-15           /* mjMINVAL */
#define MJPC_MAX_RETURN 1.0e6       /* kMaxReturnValue, mjpc/trajectory.cc:29 */
#define MJPC_MAX_COST_TERMS 128     /* kMaxCostTerms, mjpc/task.h */
#define MJPC_MAX_HORIZON 512        /* kMaxTrajectoryHorizon, mjpc/trajectory.h:27 */

/* ---- model: the subset of mjModel the path reads ------------------------------------- */
/* ABI revision of this header.  mjpc_hip_version() returns the revision the library was built from; both view structs start
   with `struct_size` (= sizeof of the struct as the CALLER compiled it): mjpc_hip_create / mjpc_hip_set_task refuse a view whose
   size differs from the library's, so a stale .so paired with a newer header (or ctypes layout) fails loudly instead of reading
   garbage pointers.  Bindings without the header: mjpc_hip_sizeof_model() / _task() / _plan_input() / _plan_output(). */
#define MJPC_HIP_ABI_VERSION 4

typedef struct MjpcHipModel {
  int struct_size;                  /* sizeof(MjpcHipModel) */
  /* sizes */
  int nq, nv, nu, na, nbody, njnt, ngeom, nsite, nmocap, nuserdata, nkey, nexclude, ntendon, nwrap, nmesh, nmeshvert, nhfield, nhfielddata;
  /* mjOption */
  double timestep;
  double gravity[3];
  double impratio;
  double tolerance;        /* solver tolerance (1e-8) */
  double ls_tolerance;     /* line-search tolerance (0.01) */
  int cone;                /* MJPC_CONE_* */
  int iterations;          /* max Newton iterations (100) */
  int ls_iterations;       /* max line-search iterations (50) */
  int disableflags;        /* mjDSBL_* bits: constraint, frictionloss, limit, contact are honoured; equality / sensor / midphase have
                            * nothing to act on; any other bit (passive, gravity, clampctrl, warmstart, filterparent, actuation,
                            * refsafe, eulerdamp) is refused at create */
  int enableflags;         /* mjENBL_* bits: override and multiccd are refused, the rest has nothing to act on */
  int solver;              /* mjtSolver: only MJPC_SOL_NEWTON (2, MuJoCo's default and what every MJPC task uses) */
  int integrator;          /* mjtIntegrator: MJPC_INT_EULER (0; with implicit joint damping, mj_Euler), MJPC_INT_IMPLICITFAST (3: also implicit in
                            * tendon damping, the velocity terms of affine actuator biases - zero while the force sits on its forcerange - and
                            * the inertia-box fluid forces; no Coriolis derivative; symmetric factorisation) or MJPC_INT_IMPLICIT (2: the same
                            * plus the velocity derivative of the bias forces, mjd_rne_vel, and an LU factorisation of the non-symmetric
                            * M - h dF/dv).  RK4 is refused */
  int noslip_iterations;   /* mjOption.noslip_iterations: > 0 runs mj_solNoSlip after the Newton solve (friction-loss rows and the friction
                            * dimensions of the contacts re-solved without regularisation; noslip_tolerance at the end of this struct) */
  int neq;                 /* number of equality constraints (eq_* below): connect, weld, joint and (fixed-)tendon equalities; flex is refused */
  int unsupported;         /* MJPC_UNSUP_* bits found by whoever fills this view in parts of mjModel the view does not carry
                            * (integration/hip_sampling_planner.cc: FillModelView); non-zero is refused at create */
  /* mjStatistic */
  double meaninertia;
  /* fluid forces of the inertia-box model (mj_passive): medium density / viscosity and wind (mjOption); all zero = none.  With the
   * implicitfast integrator they are refused (their velocity derivative is not implemented); the ellipsoid model (geom fluidshape)
   * is MJPC_UNSUP_FLUID */
  double density, viscosity, wind[3];
  /* engine capacities (0 = default); overflow => candidate failure, like MuJoCo's
   * "contact/constraint buffer full" warnings -> mjpc CheckWarnings (utilities.cc:787-799) */
  int nconmax, nefcmax;
  /* bodies */
  const int *body_parentid, *body_rootid, *body_weldid, *body_mocapid;
  const int *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr;
  const double *body_pos, *body_quat, *body_ipos, *body_iquat;
  const double *body_mass, *body_subtreemass, *body_inertia, *body_invweight0;
  /* joint-level clamp of the total actuator force on a hinge / slide joint (mjModel.jnt_actfrclimited / jnt_actfrcrange, applied to
   * qfrc_actuator at the end of mj_fwdActuation); NULL = none */
  const int *jnt_actfrclimited; const double *jnt_actfrcrange;
  const double *body_gravcomp;      /* [nbody] gravity compensation (mj_passive: force -gravity * mass * gravcomp at the body's com), NULL = none */
  /* joints */
  const int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid, *jnt_limited;
  const double *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_margin;
  const double *jnt_solref, *jnt_solimp;      /* limit solver params: 2 / 5 per joint */
  const double *qpos0, *qpos_spring;
  /* dofs */
  const int *dof_bodyid, *dof_jntid, *dof_parentid;
  const double *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0;
  const double *dof_solref, *dof_solimp;      /* friction-loss solver params */
  /* geoms */
  const int *geom_type, *geom_contype, *geom_conaffinity, *geom_condim, *geom_bodyid;
  const int *geom_group, *geom_priority;
  const double *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solmix;
  const double *geom_solref, *geom_solimp, *geom_margin, *geom_gap, *geom_rbound;
  /* <contact><exclude>: (body1 << 16) + body2, as mjModel.exclude_signature */
  const int *exclude_signature;
  /* equality constraints [neq] (mj_instantiateEquality): MJPC_EQ_CONNECT obj = the two bodies (obj2 may be the world 0), eq_data[0..2]
   * / [3..5] = the anchor in either body frame; MJPC_EQ_WELD obj = the two bodies, eq_data[0..2] = anchor in body 2, [3..5] = the same point in
   * body 1, [6..9] = quaternion of body 2 in body 1's frame (mj_setConst has filled both), [10] = torquescale; MJPC_EQ_JOINT obj = joint1 and joint2 (or -1), eq_data[0..4] = polycoef of
   * q1 - q1_0 = poly(q2 - q2_0); MJPC_EQ_TENDON the same with the lengths of two fixed tendons (relative to their length at qpos0).  eq_active0 = 0 rows are left out (no run-time activation).  All NULL when neq = 0 */
  const int *eq_type, *eq_obj1id, *eq_obj2id, *eq_active0;
  const double *eq_data;            /* 11 per equality (mjNEQDATA) */
  const double *eq_solref, *eq_solimp;   /* 2 / 5 per equality */
  /* sites */
  const int *site_bodyid;
  const double *site_pos, *site_quat;
  /* actuators (joint or fixed-tendon transmission; gain fixed; bias none/affine: motor, general, position servos) */
  const int *actuator_trntype;      /* MJPC_TRN_JOINT / MJPC_TRN_TENDON / MJPC_TRN_SITE (without a reference site: the wrench actuator_gear6 in the
                                     * site frame, motors only - biastype none; with one: actuator_refsite at the end of this struct) */
  const int *actuator_trnid;        /* joint, tendon or site id (first of the 2 mjModel ints) */
  const int *actuator_ctrllimited, *actuator_forcelimited, *actuator_biastype;
  const double *actuator_gainprm;   /* 3 per actuator (first 3 of mjNGAIN) */
  const double *actuator_biasprm;   /* 3 per actuator (first 3 of mjNBIAS) */
  const double *actuator_gear;      /* 1 per actuator (first of 6) */
  const double *actuator_gear6;     /* 6 per actuator (mjModel.actuator_gear as it is): read for site transmissions only; may be NULL without them */
  const double *actuator_ctrlrange, *actuator_forcerange;
  /* activation states (na > 0): one state per stateful actuator (actnum 1).  dyntype MJPC_DYN_*: integrator act_dot = ctrl; filter /
   * filterexact act_dot = (ctrl - act) / max(mjMINVAL, dynprm[0]); the force of a stateful actuator is gain * act + bias.  Euler
   * advance act += h * act_dot (filterexact: act_dot * tau * (1 - exp(-h / tau))), clamped to actrange when actlimited.  All NULL
   * (na = 0): no stateful actuator.  Muscles, user dynamics and actearly are MJPC_UNSUP_ACTUATOR_DYN. */
  const int *actuator_dyntype, *actuator_actadr, *actuator_actlimited;
  const double *actuator_dynprm;    /* 1 per actuator (first of mjNDYN): the time constant */
  const double *actuator_actrange;  /* 2 per actuator */
  /* fixed tendons (wrap objects are joints; wrap_prm = coefficient) */
  const int *tendon_adr, *tendon_num, *tendon_limited, *wrap_objid;
  const double *wrap_prm, *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_invweight0;
  /* passive tendon forces (mj_passive): spring with a dead band [lengthspring[2t], lengthspring[2t+1]], damper; [ntendon] each,
   * NULL = none.  tendon_frictionloss > 0 makes a friction-loss row along the tendon (mjCNSTR_FRICTION_TENDON) with the solver
   * parameters tendon_solref_fri [2 per tendon] / tendon_solimp_fri [5 per tendon] (NULL = MuJoCo's defaults). */
  const double *tendon_stiffness, *tendon_damping, *tendon_lengthspring, *tendon_frictionloss;
  const double *tendon_solref_fri, *tendon_solimp_fri;
  /* convex meshes (collision = convex hull of the vertices, in the geom frame; mjModel.mesh_vert is float: widen it).  All NULL /
   * nmesh = 0: no meshes.  geom_dataid[g] = mesh of a MJPC_GEOM_MESH geom, -1 otherwise. */
  const int *geom_dataid, *mesh_vertadr, *mesh_vertnum;
  const double *mesh_vert;          /* [3 * nmeshvert] */
  /* height fields (geom_dataid[g] = height field of a MJPC_GEOM_HFIELD geom): nrow x ncol samples in [0, 1] (mjModel.hfield_data
   * is float: widen it), row-major with x along the columns; size = (radius_x, radius_y, elevation_z, base_z) */
  const int *hfield_nrow, *hfield_ncol, *hfield_adr;
  const double *hfield_size;        /* [4 * nhfield] */
  const double *hfield_data;        /* [nhfielddata] */
  /* keyframes */
  const double *key_qpos;           /* nkey * nq */
  const double *key_mpos;           /* nkey * 3*nmocap */
  /* added with ABI revision 4 */
  const int *actuator_refsite;      /* second mjModel.actuator_trnid int of a MJPC_TRN_SITE actuator: the reference site, -1 = none; NULL = none at
                                     * all.  With a reference site the transmission has a length (site position in the reference site's frame
                                     * dotted with gear[0:3]) and the moment (Jp_site - Jp_ref)^T R_ref gear[0:3], zero on the dofs both sites
                                     * share (mj_transmission); any gain / affine bias / activation state may ride it; a rotational gear
                                     * (gear[3:6] != 0) with a reference site is refused */
  double noslip_tolerance;          /* mjOption.noslip_tolerance (1e-6) */
} MjpcHipModel;

/* ---- task: cost table (mjpc/task.cc:147-245) + frozen ResidualFn state --------------- */
typedef struct MjpcHipTask {
  int struct_size;                  /* sizeof(MjpcHipTask) */
  int task_id;                      /* MJPC_TASK_* */
  int num_residual, num_term, num_trace;
  const int *dim_norm_residual;     /* [num_term] */
  const int *norm;                  /* [num_term] MJPC_NORM_* */
  const int *num_norm_parameter;    /* [num_term] */
  const double *weight;             /* [num_term] */
  const double *norm_parameter;     /* [sum num_norm_parameter] */
  double risk;
  int num_parameter;
  const double *parameters;         /* residual_* numerics, select values bit-cast (task.cc:38-64) */
  const int *trace_objtype;         /* [num_trace] MJPC_OBJ_SITE / BODY / GEOM (framepos sensors "trace%i") */
  const int *trace_objid;           /* [num_trace] */
  int num_int;  const int *int_data;      /* task-specific frozen ids  (layout: mjpc_hip_tasks.md / DESIGN.md) */
  int num_dbl;  const double *dbl_data;   /* task-specific frozen state */
} MjpcHipTask;

/* ---- one plan step ------------------------------------------------------------------- */
typedef struct MjpcHipPlanInput {
  /* State snapshot, mjpc/planners/sampling/planner.cc:146-149 */
  const double *state;      /* [nq+nv+na] */
  const double *mocap;      /* [7*nmocap] pos3+quat4 per mocap body */
  const double *userdata;   /* [nuserdata] */
  double time;
  /* nominal spline policy after UpdateNominalPolicy (planner.cc:236-310) */
  const double *knot_times;   /* [num_spline_points] */
  const double *knot_values;  /* [num_spline_points*nu] */
  int num_spline_points;
  int interpolation;          /* MJPC_SPLINE_* */
  /* sampling */
  int num_trajectory;         /* N: GLOBAL candidate count */
  int horizon;                /* H (steps, <= MJPC_MAX_HORIZON) */
  int candidate_offset;       /* first global candidate index owned by this engine (multi-GPU shard) */
  int num_local;              /* candidates rolled out by this engine: [offset, offset+num_local) */
  double noise_exploration[2];/* sigma; [1] used with prob. 0.2 when > 0 (planner.cc:320-325) */
  /* noise: explicit standard-normal tensor, or NULL -> Philox4x32-10(seed, stream) on device */
  const double *noise_eps;    /* [num_trajectory*P*nu] (global indexing) or NULL */
  const int *noise_sel;       /* [num_trajectory] 1 = use noise_exploration[1]; or NULL */
  uint64_t seed;
  uint64_t stream;            /* plan iteration counter */
  /* Cross-Entropy sampling (mjpc/planners/cross_entropy/planner.cc:340-375): when non-NULL, candidate knots are
   * nominal + noise_std[p*nu+k] * eps (absolute per-parameter std, no ctrlrange scaling), then clamped */
  const double *noise_std;    /* [num_spline_points*nu] or NULL (= SamplingPlanner noise above) */
  /* global index of the candidate that gets no noise: 0 for the SamplingPlanner (planner.cc:361); the Cross-Entropy
   * planner perturbs all N candidates and rolls the nominal out as candidate N of num_trajectory = N+1
   * (cross_entropy/planner.cc:377-415) */
  int nominal_index;
  /* Robust planner (mjpc/planners/robust/robust_planner.cc:91-157, Trajectory::NoisyRollout trajectory.cc:100-210):
   * explicit candidate policies and Ornstein-Uhlenbeck external-force noise on every body.
   * candidate_knots != NULL: candidate i uses candidate_knots[i] verbatim (no sampling noise at all).
   * xfrc_std > 0: before every step xfrc_applied = rate * xfrc_applied + N(0, xfrc_std * sqrt(1 - rate^2)),
   * rate = exp(-timestep / xfrc_rate) (trajectory.cc:147-155); normals from Philox(seed, stream ^ "XFRC", candidate i,
   * element t*6*nbody + k). */
  const double *candidate_knots;   /* [num_trajectory][num_spline_points][nu] (global indexing) or NULL */
  double xfrc_std, xfrc_rate;
} MjpcHipPlanInput;

typedef struct MjpcHipPlanOutput {
  /* all host buffers, caller-owned; any pointer may be NULL to skip the copy */
  double *returns;        /* [num_local] total_return per local candidate */
  int *failure;           /* [num_local] */
  int winner;             /* OUT: global index of local argmin (lowest index on ties) */
  double winner_return;   /* OUT */
  /* winner trajectory, same layout as mjpc::Trajectory (trajectory.h:74-86) */
  double *states;         /* [H*(nq+nv+na)] */
  double *actions;        /* [H*nu] */
  double *times;          /* [H] */
  double *residual;       /* [H*num_residual] */
  double *costs;          /* [H] */
  double *trace;          /* [H*3*num_trace] */
  double *winner_knots;   /* [P*nu] candidate_policy[winner].plan values */
  /* device timings of the last plan step, microseconds (fields mirror
   * noise_compute_time / rollouts_compute_time, planner.h:149-152) */
  double noise_compute_time_us;
  double rollouts_compute_time_us;
} MjpcHipPlanOutput;

typedef struct MjpcHipEngine MjpcHipEngine;

/* Create an engine on HIP device `device`.  Copies model+task to HBM.  max_local = largest
 * num_local that will be planned on this device.  Returns NULL on error (see last_error).
 * Models the engine cannot roll out faithfully are REFUSED here (never silently approximated): geom pairs without a
 * collider (height field against plane / height field; meshes / height fields without data), group-0 geoms the quadruped task's ground ray cannot hit, actuator transmissions other than joint /
 * fixed tendon, nconmax > 64, nefcmax > 192, iterations > 250, num_spline_points capacity 36, LDS footprint > 160 KiB. */
MjpcHipEngine *mjpc_hip_create(const MjpcHipModel *model, const MjpcHipTask *task,
                               int max_local, int max_horizon, int device);
void mjpc_hip_destroy(MjpcHipEngine *e);
/* Re-upload cost weights / norm params / residual parameters / frozen task state
 * (Agent::PlanIteration takes a fresh ResidualFn copy every plan step, agent.cc:290): only the task block travels,
 * as one stream-ordered asynchronous copy ahead of the next plan's kernels.  Error while a plan is in flight. */
int mjpc_hip_set_task(MjpcHipEngine *e, const MjpcHipTask *task);
/* One plan step: noise -> N rollouts -> costs -> argmin; blocking. 0 on success. */
int mjpc_hip_plan(MjpcHipEngine *e, const MjpcHipPlanInput *in, MjpcHipPlanOutput *out);
/* Split form used by bench.py / multi-GPU: enqueue on the engine's stream, then fetch.  One plan in flight per engine:
 * a second mjpc_hip_plan_async before mjpc_hip_plan_fetch is an error. */
int mjpc_hip_plan_async(MjpcHipEngine *e, const MjpcHipPlanInput *in);
int mjpc_hip_plan_fetch(MjpcHipEngine *e, MjpcHipPlanOutput *out);
/* What mjpc_hip_plan_fetch brings to the host.  MJPC_FETCH_WINNER_ROWS (default): returns, failure flags, and the local
 * winner's trajectory rows + knots in one packed copy.  MJPC_FETCH_SUMMARY: returns, failure flags, local winner index /
 * return / knots only — for the shards of a multi-device plan, where only the owner of the GLOBAL winner then copies its
 * trajectory with mjpc_hip_get_candidate. */
enum { MJPC_FETCH_WINNER_ROWS = 0, MJPC_FETCH_SUMMARY = 1 };
int mjpc_hip_set_fetch_mode(MjpcHipEngine *e, int mode);
/* Copy any local candidate's trajectory / knots to host (GUI traces, RankedPlanner). */
int mjpc_hip_get_candidate(MjpcHipEngine *e, int local_index, MjpcHipPlanOutput *out);
/* Copy every local candidate's knot values [num_local][P][nu] of the last plan to host (elite statistics of the
 * Cross-Entropy planner, cross_entropy/planner.cc:226-262). */
int mjpc_hip_get_knots(MjpcHipEngine *e, double *knots);
/* Every local candidate's trace rows [num_local][H][3*num_trace] of the last plan in one copy: what
 * SamplingPlanner::Traces draws (mjpc/planners/sampling/planner.cc:388-434). */
int mjpc_hip_get_traces(MjpcHipEngine *e, double *traces);
/* Every local candidate's Trajectory arrays of the last plan, [num_local][H][...] each (any pointer may be NULL): the
 * `trajectory[]` members other planners read (mjpc/planners/ilqs/planner.cc:98-198).  diag: [num_local][4] = Newton
 * iterations summed over the steps, max contacts, max constraint rows, MJPC_WARN_* bits. */
int mjpc_hip_get_all_candidates(MjpcHipEngine *e, double *states, double *actions, double *times, double *residual,
                                double *costs, double *trace, double *knots, int *diag);
/* Bytes of LDS one candidate's workgroup occupies (its whole mjData-equivalent). */
int mjpc_hip_lds_bytes(MjpcHipEngine *e);
/* Capacity tiers: when a shard holds more candidates than the GPU has CUs and the model allows it, the engine first runs a
 * flavour that fits TWO candidates per CU (<= 80 KiB of LDS each, smaller contact / row capacity) and re-runs the rare
 * candidate that overflowed it at full capacity - same results, ~1.6x the throughput.  Returns the dense tier's LDS bytes
 * (0: none for this model); *used_last = 1 when the last plan used it. */
int mjpc_hip_dense_tier(MjpcHipEngine *e, int *used_last);
/* The same figure for a model without creating an engine (host-only, no GPU needed): with (1) / without (0) the LDS copy of
 * the model tables; use_cache | 2: the dense tier's lean layout at the model's nefcmax / nconmax.  Negative: mjpc_hip_create
 * would refuse the model (see mjpc_hip_last_error). */
int mjpc_hip_layout_bytes(const MjpcHipModel *model, const MjpcHipTask *task, int use_cache);
/* Kinematic frame of local candidate 0 at the first step of the last plan (the state handed in): what a host-side
 * Task::Transition reads from mjData after a simulation step (mjpc/tasks/quadruped/quadruped.cc:254,290-330: body poses, site
 * positions, subtree com / linear velocity sensors).  Any pointer may be NULL.  Sizes: xpos 3*nbody, xmat 9*nbody,
 * site_xpos 3*nsite, subtree_com 3*nbody, subtree_linvel 3*nbody. */
int mjpc_hip_get_frame(MjpcHipEngine *e, double *xpos, double *xmat, double *site_xpos, double *subtree_com, double *subtree_linvel);
/* Average device time of the rollout kernel over the launches since the last call,
 * measured with hipEvents on the engine's stream; returns launches counted. */
int mjpc_hip_kernel_time(MjpcHipEngine *e, double *avg_rollout_us, double *avg_total_us);
/* Device pointers of the last plan's result arrays (for zero-copy consumers / tests). */
int mjpc_hip_device_ptrs(MjpcHipEngine *e, void **returns, void **states, void **residual);
/* ---- one planner, several GPUs (north_star: "candidate batches shard across the 8 GPUs of one node") -------------------
 * A MjpcHipMulti owns one engine per device in ONE host process (the planner lives in one process, planners/include.cc:44).
 * mjpc_hip_multi_plan block-partitions the global batch [0, num_trajectory) over the engines (candidate 0, the un-noised
 * nominal, on the first), enqueues every shard asynchronously on its device's stream, fetches the per-shard summaries
 * (returns + local elite, MJPC_FETCH_SUMMARY), takes the lexicographic minimum (return, global index) on the host - G x 16
 * bytes that are on the host anyway with returns[]; an RCCL collective would add a launch + sync per device for nothing -
 * and copies the winner's trajectory from its owner only.  Results are bit-identical to one engine planning the whole batch
 * (noise is indexed by the global candidate id).  `devices`: n_devices HIP device ordinals (repeats allowed: several
 * engines on one GPU, used by the 1-GPU rehearsal test); NULL = 0 .. n_devices-1.  in->candidate_offset / num_local are
 * ignored (the whole batch is planned); out->returns / failure are [num_trajectory]. */
typedef struct MjpcHipMulti MjpcHipMulti;
MjpcHipMulti *mjpc_hip_multi_create(const MjpcHipModel *model, const MjpcHipTask *task, int max_samples, int max_horizon,
                                    int n_devices, const int *devices);
void mjpc_hip_multi_destroy(MjpcHipMulti *m);
int mjpc_hip_multi_set_task(MjpcHipMulti *m, const MjpcHipTask *task);
int mjpc_hip_multi_plan(MjpcHipMulti *m, const MjpcHipPlanInput *in, MjpcHipPlanOutput *out);
/* trajectory / knots of GLOBAL candidate `index` of the last plan, from the engine that owns it */
int mjpc_hip_multi_get_candidate(MjpcHipMulti *m, int index, MjpcHipPlanOutput *out);
/* knots [num_trajectory][P][nu] / traces [num_trajectory][H][3*num_trace] of every candidate of the last plan */
int mjpc_hip_multi_get_knots(MjpcHipMulti *m, double *knots);
int mjpc_hip_multi_get_traces(MjpcHipMulti *m, double *traces);
int mjpc_hip_multi_num_devices(const MjpcHipMulti *m);
/* engine k (0 .. n_devices-1), e.g. for mjpc_hip_get_frame / mjpc_hip_kernel_time */
MjpcHipEngine *mjpc_hip_multi_engine(MjpcHipMulti *m, int k);
const char *mjpc_hip_last_error(void);
int mjpc_hip_version(void);                 /* MJPC_HIP_ABI_VERSION of the library */
int mjpc_hip_sizeof_model(void);
int mjpc_hip_sizeof_task(void);
int mjpc_hip_sizeof_plan_input(void);
int mjpc_hip_sizeof_plan_output(void);

#ifdef __cplusplus
}
#endif
#endif  /* MJPC_HIP_H_ */
