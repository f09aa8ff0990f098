/*
 * oracle/oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp64, sequential) of the reference's Predictive-Sampling rollout
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (mujoco_mpc_amd/csrc, libmjpc_hip.so) never includes, links or calls it.
 *
 * Pinned against the reference's own fixtures (tests/test_oracle_*.py):
 *   spline goldens            mjpc/test/spline/spline_test.cc:40-231
 *   cost / risk formulae      mjpc/test/tasks/task_test.cc:58-95
 *   rollout alignment         mjpc/test/agent/rollout_test.cc:137-145
 *   planner convergence       mjpc/test/sampling_planner/sampling_planner_test.cc:89-108
 * PARITY UNPINNED at the mj_step boundary: MuJoCo 3.1.4 (CMakeLists.txt:58-61) is not in
 * /root/reference nor in this image, and none of the reference's tests pins mj_step numerics
 * (SURVEY.md §8c).  The physics below restates MuJoCo's documented computation pipeline.
 */
#ifndef ORACLE_H_
#define ORACLE_H_
#include "../include/mjpc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OContact {
  double dist, pos[3], frame[9], includemargin, friction[5], solref[2], solimp[5], mu;
  double H[36];
  int dim, geom1, geom2, efc_address;
} OContact;

/* constraint row types / states (values follow mjtConstraint / mjtConstraintState) */
enum { O_CNSTR_EQUALITY = 0, O_CNSTR_FRICTION_DOF = 1, O_CNSTR_FRICTION_TENDON = 2, O_CNSTR_LIMIT_JOINT = 3, O_CNSTR_LIMIT_TENDON = 4, O_CNSTR_CONTACT_FRICTIONLESS = 5,
       O_CNSTR_CONTACT_PYRAMIDAL = 6, O_CNSTR_CONTACT_ELLIPTIC = 7 };
enum { O_STATE_SATISFIED = 0, O_STATE_QUADRATIC = 1, O_STATE_LINEARNEG = 2, O_STATE_LINEARPOS = 3, O_STATE_CONE = 4 };

typedef struct OModel {
  MjpcHipModel m;       /* deep copy, pointers owned */
  MjpcHipTask t;        /* deep copy */
  int npair, *pair_g1, *pair_g2;   /* statically filtered collision pairs */
  int nconmax, nefcmax;
  int nray, *ray_geom;  /* group-0 geoms for Ground() ray casts */
  void *blocks[4096]; int nblocks;
} OModel;

typedef struct OData {
  /* state */
  double *qpos, *qvel, *ctrl, *mocap_pos, *mocap_quat, *userdata, time;
  double *qacc, *qacc_warmstart, *qacc_smooth, *qfrc_smooth, *qfrc_bias, *qfrc_passive,
         *qfrc_actuator, *qfrc_constraint, *actuator_force, *act, *act_dot;
  /* kinematics */
  double *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *geom_xpos, *geom_xmat,
         *site_xpos, *site_xmat, *subtree_com, *cinert, *cdof, *cvel, *cdof_dot, *crb,
         *subtree_linvel, *cacc, *cfrc;
  double *xfrc_applied; int xfrc_on;   /* [6*nbody] force(3)+torque(3) per body (mjData.xfrc_applied); used when xfrc_on */
  double *qM, *qL, *qH, *qLD2;   /* dense nv*nv */
  /* contacts + constraints */
  int ncon, nefc, nf, nl, unsupported;
  OContact *contact;
  int *efc_type, *efc_id, *efc_state;
  double *efc_J, *efc_pos, *efc_margin, *efc_frictionloss, *efc_diagApprox, *efc_D, *efc_R,
         *efc_vel, *efc_aref, *efc_force, *efc_jar, *efc_jv, *efc_KBIP;
  double *efc_AR, *efc_b, *efc_MiJT;     /* noslip only: J M^-1 J^T + diag(R), J qacc_smooth - aref, M^-1 J^T (row r = M^-1 J_r^T) */
  int noslip_iter;
  double *qacc_newton;                   /* the Newton solve's qacc (the next step's warm start; mj_solNoSlip runs after it is saved) */
  double *efc_solref, *efc_solimp;
  /* solver scratch */
  double *Ma, *grad, *Mgrad, *search, *Mv, *work;
  int solver_iter;
  int warning;          /* sticky MJPC_WARN_* bits: bad qpos/qvel/qacc, buffer overflow, ray miss, unsupported pair (mjNWARNING analogue) */
  double *sensordata;   /* residual[num_residual] */
  void *blocks[256]; int nblocks;
} OData;

/* per-candidate full trajectory record (all candidates; tests compare against GPU) */
typedef struct OPlanOutput {
  double *returns;      /* [N] */
  int *failure;         /* [N] */
  double *states;       /* [N*H*dim_state] */
  double *actions;      /* [N*H*nu] */
  double *times;        /* [N*H] */
  double *residual;     /* [N*H*nr] */
  double *costs;        /* [N*H] */
  double *trace;        /* [N*H*3*ntrace] */
  double *knots;        /* [N*P*nu] */
  int winner;
  int unsupported;      /* number of unsupported geom-pair proximity events (diagnostic) */
  int solver_iter_total;
} OPlanOutput;

OModel *oracle_create(const MjpcHipModel *model, const MjpcHipTask *task);
void oracle_destroy(OModel *om);
int oracle_set_task(OModel *om, const MjpcHipTask *task);
OData *oracle_make_data(const OModel *om);
void oracle_free_data(OData *d);

/* physics */
void oracle_forward(const OModel *om, OData *d);              /* mj_forward + residual */
void oracle_step(const OModel *om, OData *d);                 /* mj_step */
void oracle_residual(const OModel *om, OData *d, double *residual);
double oracle_ray_ground(const OModel *om, OData *d, const double pos[3]);

/* MJPC side */
void oracle_spline_sample(const double *times, const double *values, int P, int dim,
                          int interp, double t, double *out);
double oracle_norm(const double *x, const double *params, int n, int type);
int oracle_norm_parameter_dimension(int type);
double oracle_cost_value(const MjpcHipTask *t, const double *residual, double *terms);
void oracle_noise(uint64_t seed, uint64_t stream, int i0, int n, int P, int nu,
                  double sigma2, double *eps, int *sel);
void oracle_philox(uint64_t seed, uint64_t stream, uint32_t c0, uint32_t c1, uint32_t out[4]);

/* one rollout (mjpc/trajectory.cc:100-210) into row `row` of out; d is reused like the
 * per-thread mjData of the reference, but qacc_warmstart is zeroed first */
void oracle_rollout(const OModel *om, OData *d, const MjpcHipPlanInput *in,
                    const double *knots, int row, OPlanOutput *out);
/* persistent FIFO worker pool (mjpc/threadpool.cc:30-85): threads and their per-worker OData arenas live across plan
 * steps, like the reference's ThreadPool + Planner::data_ (planners/planner.cc:23-33); nthreads == 1 runs on the caller */
typedef struct OPool OPool;
OPool *oracle_pool_create(const OModel *om, int nthreads);
void oracle_pool_destroy(OPool *pool);
int oracle_pool_threads(const OPool *pool);
int oracle_pool_plan(OPool *pool, const MjpcHipPlanInput *in, OPlanOutput *out);
/* one plan step on a pool that lives for this call only (tests) */
int oracle_plan(const OModel *om, const MjpcHipPlanInput *in, OPlanOutput *out, int nthreads);

/* debug accessors for physics unit tests */
int oracle_debug_forward(const OModel *om, const double *qpos, const double *qvel,
                         const double *ctrl, const double *mocap, double time,
                         double *qacc, double *qM, double *xpos, double *sensordata,
                         double *contact_dist, int *ncon, int *nefc, double *efc_force,
                         double *geom_xpos, double *extra);
int oracle_debug_step(const OModel *om, double *qpos, double *qvel, const double *ctrl,
                      const double *mocap, double *time, int nstep, double *energy);
int oracle_debug_vel_derivatives(const OModel *om, const double *qpos, const double *qvel, double *dbias, double *dfluid, double *bias,
                                 double *passive_out);
int oracle_debug_actuation(const OModel *om, const double *qpos, const double *qvel, const double *ctrl, const double *act, double *force, double *qfrc);
int oracle_debug_constraints(const OModel *om, const double *qpos, const double *qvel, const double *mocap, int cap, double *J, double *pos,
                             double *diag, double *R, double *aref);

#ifdef __cplusplus
}
#endif
#endif
