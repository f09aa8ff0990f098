// tests/stubs/mujoco/mujoco.h — DECLARATION-ONLY stand-in for MuJoCo's public header, for ONE purpose: letting
// tests/test_integration_syntax.py run `g++ -fsyntax-only` over integration/*.cc (the reference-side adapter) against the
// reference's real mjpc/ headers in an image that has no MuJoCo.  It declares the slice of the public API (names, types,
// struct fields as documented for MuJoCo 3.1) that the adapter, the mjpc headers it includes and ilqs/planner.cc touch.
// Nothing here has a body, nothing is linked, no product or oracle code includes it, and it proves nothing about numerics.
#ifndef MJPC_TEST_STUB_MUJOCO_H_
#define MJPC_TEST_STUB_MUJOCO_H_

#include <cstddef>
#include <cstdint>

typedef double mjtNum;
typedef unsigned char mjtByte;

#define mjNEQDATA 11
#define mjNDYN 10
#define mjNGAIN 10
#define mjNBIAS 10
#define mjNFLUID 12
#define mjNREF 2
#define mjNIMP 5
#define mjMINVAL 1E-15
#define mjPI 3.14159265358979323846
#define mjMAXLINE 100
#define mjMAXLINEPNT 1000
#define mjMAXUINAME 40
#define mjMAXUITEXT 300
#define mjMAXUIEDIT 7
#define mjMAXUIMULTI 35
#define mjNGROUP 6

typedef enum { mjOBJ_UNKNOWN = 0, mjOBJ_BODY, mjOBJ_XBODY, mjOBJ_JOINT, mjOBJ_DOF, mjOBJ_GEOM, mjOBJ_SITE, mjOBJ_CAMERA, mjOBJ_LIGHT,
               mjOBJ_FLEX, mjOBJ_MESH, mjOBJ_SKIN, mjOBJ_HFIELD, mjOBJ_TEXTURE, mjOBJ_MATERIAL, mjOBJ_PAIR, mjOBJ_EXCLUDE,
               mjOBJ_EQUALITY, mjOBJ_TENDON, mjOBJ_ACTUATOR, mjOBJ_SENSOR, mjOBJ_NUMERIC, mjOBJ_TEXT, mjOBJ_TUPLE, mjOBJ_KEY,
               mjOBJ_PLUGIN } mjtObj;
typedef enum { mjGAIN_FIXED = 0, mjGAIN_AFFINE, mjGAIN_MUSCLE, mjGAIN_USER } mjtGain;
typedef enum { mjBIAS_NONE = 0, mjBIAS_AFFINE, mjBIAS_MUSCLE, mjBIAS_USER } mjtBias;
typedef enum { mjDYN_NONE = 0, mjDYN_INTEGRATOR, mjDYN_FILTER, mjDYN_FILTEREXACT, mjDYN_MUSCLE, mjDYN_USER } mjtDyn;
typedef enum { mjWRAP_NONE = 0, mjWRAP_JOINT, mjWRAP_PULLEY, mjWRAP_SITE, mjWRAP_SPHERE, mjWRAP_CYLINDER } mjtWrap;
typedef enum { mjTRN_JOINT = 0, mjTRN_JOINTINPARENT, mjTRN_SLIDERCRANK, mjTRN_TENDON, mjTRN_SITE, mjTRN_BODY } mjtTrn;
typedef enum { mjGEOM_PLANE = 0, mjGEOM_HFIELD, mjGEOM_SPHERE, mjGEOM_CAPSULE, mjGEOM_ELLIPSOID, mjGEOM_CYLINDER, mjGEOM_BOX,
               mjGEOM_MESH, mjGEOM_SDF, mjNGEOMTYPES, mjGEOM_ARROW = 100, mjGEOM_ARROW1, mjGEOM_ARROW2, mjGEOM_LINE, mjGEOM_SKIN,
               mjGEOM_LABEL, mjGEOM_TRIANGLE, mjGEOM_NONE = 1001 } mjtGeom;
typedef enum { mjITEM_END = -2, mjITEM_SECTION = -1, mjITEM_SEPARATOR = 0, mjITEM_STATIC, mjITEM_BUTTON, mjITEM_CHECKINT,
               mjITEM_CHECKBYTE, mjITEM_RADIO, mjITEM_RADIOLINE, mjITEM_SELECT, mjITEM_SLIDERINT, mjITEM_SLIDERNUM, mjITEM_EDITINT,
               mjITEM_EDITNUM, mjITEM_EDITFLOAT, mjITEM_EDITTXT, mjNITEM } mjtItem;
typedef enum { mjSENS_USER = 47 } mjtSensor;
typedef enum { mjJNT_FREE = 0, mjJNT_BALL, mjJNT_SLIDE, mjJNT_HINGE } mjtJoint;

struct mjOption {
  mjtNum timestep, apirate, impratio, tolerance, ls_tolerance, noslip_tolerance, mpr_tolerance;
  mjtNum gravity[3], wind[3], magnetic[3], density, viscosity;
  mjtNum o_margin, o_solref[mjNREF], o_solimp[mjNIMP], o_friction[5];
  int integrator, cone, jacobian, solver, iterations, ls_iterations, noslip_iterations, mpr_iterations, disableflags, enableflags,
      disableactuator, sdf_initpoints, sdf_iterations;
};
struct mjStatistic { mjtNum meaninertia, meanmass, meansize, extent, center[3]; };

struct mjModel {
  int nq, nv, nu, na, nbody, nbvh, njnt, ngeom, nsite, ncam, nlight, nflex, nmesh, nmeshvert, nhfield, nhfielddata, ntex, nmat, npair,
      nexclude, neq, ntendon, nwrap, nsensor, nnumeric, nnumericdata, ntext, ntuple, nkey, nmocap, nplugin, nuserdata, nsensordata,
      nuser_sensor, nM, nD, nB;
  mjOption opt;
  mjStatistic stat;
  mjtNum *qpos0, *qpos_spring;
  int *body_parentid, *body_rootid, *body_weldid, *body_mocapid, *body_jntnum, *body_jntadr, *body_dofnum, *body_dofadr;
  mjtNum *body_pos, *body_quat, *body_ipos, *body_iquat, *body_mass, *body_subtreemass, *body_inertia, *body_invweight0, *body_gravcomp;
  int *jnt_type, *jnt_qposadr, *jnt_dofadr, *jnt_bodyid;
  mjtByte *jnt_limited, *jnt_actfrclimited;
  mjtNum *jnt_pos, *jnt_axis, *jnt_stiffness, *jnt_range, *jnt_actfrcrange, *jnt_margin, *jnt_solref, *jnt_solimp;
  int *dof_bodyid, *dof_jntid, *dof_parentid;
  mjtNum *dof_armature, *dof_damping, *dof_frictionloss, *dof_invweight0, *dof_solref, *dof_solimp;
  int *geom_type, *geom_contype, *geom_conaffinity, *geom_condim, *geom_bodyid, *geom_dataid, *geom_group, *geom_priority;
  mjtNum *geom_size, *geom_pos, *geom_quat, *geom_friction, *geom_solmix, *geom_solref, *geom_solimp, *geom_margin, *geom_gap,
      *geom_rbound, *geom_fluid;
  int *site_bodyid;
  mjtNum *site_pos, *site_quat;
  int *mesh_vertadr, *mesh_vertnum;
  float* mesh_vert;
  mjtNum* hfield_size;
  int *hfield_nrow, *hfield_ncol, *hfield_adr;
  float* hfield_data;
  int* exclude_signature;
  int *eq_type, *eq_obj1id, *eq_obj2id;
  mjtByte* eq_active0;
  mjtNum *eq_data, *eq_solref, *eq_solimp;
  int *tendon_adr, *tendon_num;
  mjtByte* tendon_limited;
  mjtNum *tendon_range, *tendon_margin, *tendon_solref_lim, *tendon_solimp_lim, *tendon_solref_fri, *tendon_solimp_fri, *tendon_invweight0,
      *tendon_stiffness, *tendon_damping, *tendon_lengthspring, *tendon_frictionloss, *tendon_length0;
  int *wrap_type, *wrap_objid;
  mjtNum* wrap_prm;
  int *actuator_trntype, *actuator_dyntype, *actuator_gaintype, *actuator_biastype, *actuator_trnid, *actuator_actadr, *actuator_actnum;
  mjtByte *actuator_ctrllimited, *actuator_forcelimited, *actuator_actlimited, *actuator_actearly;
  mjtNum *actuator_dynprm, *actuator_gainprm, *actuator_biasprm, *actuator_ctrlrange, *actuator_forcerange, *actuator_actrange, *actuator_gear;
  int *sensor_type, *sensor_datatype, *sensor_needstage, *sensor_objtype, *sensor_objid, *sensor_reftype, *sensor_refid, *sensor_dim, *sensor_adr;
  mjtNum* sensor_user;
  int *numeric_adr, *numeric_size;
  mjtNum* numeric_data;
  int *text_adr, *text_size;
  char* text_data;
  mjtNum *key_time, *key_qpos, *key_qvel, *key_act, *key_mpos, *key_mquat, *key_ctrl;
  int *name_bodyadr, *name_jntadr, *name_geomadr, *name_siteadr, *name_sensoradr, *name_numericadr, *name_textadr, *name_keyadr;
  char* names;
};

struct mjContact { mjtNum dist, pos[3], frame[9]; int geom1, geom2, geom[2]; };
struct mjData {
  int ncon, nefc;
  mjtNum time;
  mjtNum *qpos, *qvel, *act, *qacc_warmstart, *ctrl, *qfrc_applied, *xfrc_applied, *mocap_pos, *mocap_quat, *qacc, *act_dot, *userdata,
      *sensordata, *xpos, *xquat, *xmat, *xipos, *ximat, *geom_xpos, *geom_xmat, *site_xpos, *site_xmat, *subtree_com, *cvel,
      *qfrc_actuator, *actuator_force, *qM, *qLD;
  mjContact* contact;
  int warning_number_placeholder;
};

struct mjvGeom { int type; float size[3], pos[3], mat[9], rgba[4]; char label[100]; };
struct mjvScene { int maxgeom, ngeom; mjvGeom* geoms; };
struct mjvFigure {
  int flg_legend, flg_ticklabel[2], flg_extend, flg_barplot, flg_selection, flg_symmetric;
  float linewidth, gridwidth; int gridsize[2]; float gridrgb[3], figurergba[4], panergba[4], legendrgba[4], textrgb[3], linergb[mjMAXLINE][3],
      range[2][2];
  char xformat[20], yformat[20], minwidth[20], title[1000], xlabel[100], linename[mjMAXLINE][100];
  int legendoffset, subplot, highlight[2], highlightid; float selection;
  int linepnt[mjMAXLINE]; float linedata[mjMAXLINE][2 * mjMAXLINEPNT];
};
struct mjuiDef { int type; char name[mjMAXUINAME]; int state; void* pdata; char other[mjMAXUITEXT]; };
#define mjMAXUISECT 10
#define mjMAXUIITEM 200
struct mjuiItem { int type; char name[mjMAXUINAME]; int state; void* pdata; int sectionid, itemid; };
struct mjuiSection { char name[mjMAXUINAME]; int state, modifier, shortcut, nitem; mjuiItem item[mjMAXUIITEM]; };
struct mjUI { int nsect; mjuiSection sect[mjMAXUISECT]; };
struct mjrRect { int left, bottom, width, height; };

typedef void (*mjfSensor)(const mjModel* m, mjData* d, int stage);
extern mjfSensor mjcb_sensor;

extern "C" {
int mj_name2id(const mjModel* m, int type, const char* name);
const char* mj_id2name(const mjModel* m, int type, int id);
mjData* mj_makeData(const mjModel* m);
void mj_deleteData(mjData* d);
mjModel* mj_copyModel(mjModel* dest, const mjModel* src);
void mj_deleteModel(mjModel* m);
void mj_step(const mjModel* m, mjData* d);
void mj_forward(const mjModel* m, mjData* d);
void mj_kinematics(const mjModel* m, mjData* d);
void mju_error(const char* msg, ...);
void mju_error_i(const char* msg, int i);
void mju_error_s(const char* msg, const char* text);
void mju_warning(const char* msg, ...);
void mju_copy(mjtNum* res, const mjtNum* vec, int n);
void mju_copy3(mjtNum res[3], const mjtNum data[3]);
void mju_copy4(mjtNum res[4], const mjtNum data[4]);
void mju_zero(mjtNum* res, int n);
void mju_fill(mjtNum* res, mjtNum val, int n);
void mju_add(mjtNum* res, const mjtNum* a, const mjtNum* b, int n);
void mju_addTo(mjtNum* res, const mjtNum* vec, int n);
void mju_addToScl(mjtNum* res, const mjtNum* vec, mjtNum scl, int n);
void mju_sub(mjtNum* res, const mjtNum* a, const mjtNum* b, int n);
void mju_scl(mjtNum* res, const mjtNum* vec, mjtNum scl, int n);
mjtNum mju_dot(const mjtNum* a, const mjtNum* b, int n);
mjtNum mju_norm(const mjtNum* res, int n);
mjtNum mju_max(mjtNum a, mjtNum b);
mjtNum mju_min(mjtNum a, mjtNum b);
mjtNum mju_clip(mjtNum x, mjtNum min, mjtNum max);
mjtNum mju_log10(mjtNum x);
mjtNum mju_log(mjtNum x);
mjtNum mju_exp(mjtNum x);
mjtNum mju_sqrt(mjtNum x);
mjtNum mju_abs(mjtNum x);
mjtNum mju_pow(mjtNum x, mjtNum y);
void mju_mulMatVec(mjtNum* res, const mjtNum* mat, const mjtNum* vec, int nr, int nc);
void mju_mulMatTVec(mjtNum* res, const mjtNum* mat, const mjtNum* vec, int nr, int nc);
void mju_mulMatMat(mjtNum* res, const mjtNum* a, const mjtNum* b, int r1, int c1, int c2);
void mju_mulMatTMat(mjtNum* res, const mjtNum* a, const mjtNum* b, int r1, int c1, int c2);
void mju_mulMatMatT(mjtNum* res, const mjtNum* a, const mjtNum* b, int r1, int c1, int r2);
void mju_transpose(mjtNum* res, const mjtNum* mat, int nr, int nc);
int mju_cholFactor(mjtNum* mat, int n, mjtNum mindiag);
void mju_cholSolve(mjtNum* res, const mjtNum* mat, const mjtNum* vec, int n);
void mjv_initGeom(mjvGeom* geom, int type, const mjtNum size[3], const mjtNum pos[3], const mjtNum mat[9], const float rgba[4]);
void mjv_makeConnector(mjvGeom* geom, int type, mjtNum width, mjtNum a0, mjtNum a1, mjtNum a2, mjtNum b0, mjtNum b1, mjtNum b2);
void mjui_add(mjUI* ui, const mjuiDef* def);
}

#endif  // MJPC_TEST_STUB_MUJOCO_H_
