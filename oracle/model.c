/*
 * oracle/model.c — TEST INFRASTRUCTURE ONLY (CPU oracle).
 * Deep copy of the MjpcHipModel / MjpcHipTask inputs, plus the static part of MuJoCo's
 * collision filtering (same weld body, parent-child, <exclude>, contype/conaffinity) and the
 * geom-group-0 list used by Ground() (mjpc/utilities.cc:538-556).
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

static void *mdup(OModel *om, const void *src, size_t bytes) {
  void *p = calloc(1, bytes ? bytes : 8);
  if (src && bytes) memcpy(p, src, bytes);
  om->blocks[om->nblocks++] = p;
  return p;
}
#define CD(f, n) om->m.f = (const double *)mdup(om, src->f, sizeof(double) * (size_t)(n))
#define CI(f, n) om->m.f = (const int *)mdup(om, src->f, sizeof(int) * (size_t)(n))

static void copy_task(OModel *om, const MjpcHipTask *src) {
  MjpcHipTask *t = &om->t;
  *t = *src;
  int np = 0;
  for (int k = 0; k < src->num_term; k++) np += src->num_norm_parameter[k];
  t->dim_norm_residual = (const int *)mdup(om, src->dim_norm_residual, sizeof(int) * (size_t)src->num_term);
  t->norm = (const int *)mdup(om, src->norm, sizeof(int) * (size_t)src->num_term);
  t->num_norm_parameter = (const int *)mdup(om, src->num_norm_parameter, sizeof(int) * (size_t)src->num_term);
  t->weight = (const double *)mdup(om, src->weight, sizeof(double) * (size_t)src->num_term);
  t->norm_parameter = (const double *)mdup(om, src->norm_parameter, sizeof(double) * (size_t)np);
  t->parameters = (const double *)mdup(om, src->parameters, sizeof(double) * (size_t)src->num_parameter);
  t->trace_objtype = (const int *)mdup(om, src->trace_objtype, sizeof(int) * (size_t)src->num_trace);
  t->trace_objid = (const int *)mdup(om, src->trace_objid, sizeof(int) * (size_t)src->num_trace);
  t->int_data = (const int *)mdup(om, src->int_data, sizeof(int) * (size_t)src->num_int);
  t->dbl_data = (const double *)mdup(om, src->dbl_data, sizeof(double) * (size_t)src->num_dbl);
}

OModel *oracle_create(const MjpcHipModel *src, const MjpcHipTask *task) {
  OModel *om = (OModel *)calloc(1, sizeof(OModel));
  om->m = *src;
  int nb = src->nbody, nj = src->njnt, nv = src->nv, ng = src->ngeom, ns = src->nsite, nu = src->nu;
  CI(body_parentid, nb); CI(body_rootid, nb); CI(body_weldid, nb); CI(body_mocapid, nb);
  CI(body_jntnum, nb); CI(body_jntadr, nb); CI(body_dofnum, nb); CI(body_dofadr, nb);
  CD(body_pos, 3 * nb); CD(body_quat, 4 * nb); CD(body_ipos, 3 * nb); CD(body_iquat, 4 * nb);
  CD(body_mass, nb); CD(body_subtreemass, nb); CD(body_inertia, 3 * nb); CD(body_invweight0, 2 * nb);
  if (src->body_gravcomp) CD(body_gravcomp, nb);
  if (src->jnt_actfrclimited && src->jnt_actfrcrange) { CI(jnt_actfrclimited, nj); CD(jnt_actfrcrange, 2 * nj); }
  CI(jnt_type, nj); CI(jnt_qposadr, nj); CI(jnt_dofadr, nj); CI(jnt_bodyid, nj); CI(jnt_limited, nj);
  CD(jnt_pos, 3 * nj); CD(jnt_axis, 3 * nj); CD(jnt_stiffness, nj); CD(jnt_range, 2 * nj); CD(jnt_margin, nj);
  CD(jnt_solref, 2 * nj); CD(jnt_solimp, 5 * nj); CD(qpos0, src->nq); CD(qpos_spring, src->nq);
  CI(dof_bodyid, nv); CI(dof_jntid, nv); CI(dof_parentid, nv);
  CD(dof_armature, nv); CD(dof_damping, nv); CD(dof_frictionloss, nv); CD(dof_invweight0, nv);
  CD(dof_solref, 2 * nv); CD(dof_solimp, 5 * nv);
  CI(geom_type, ng); CI(geom_contype, ng); CI(geom_conaffinity, ng); CI(geom_condim, ng); CI(geom_bodyid, ng);
  CI(geom_group, ng); CI(geom_priority, ng);
  CD(geom_size, 3 * ng); CD(geom_pos, 3 * ng); CD(geom_quat, 4 * ng); CD(geom_friction, 3 * ng); CD(geom_solmix, ng);
  CD(geom_solref, 2 * ng); CD(geom_solimp, 5 * ng); CD(geom_margin, ng); CD(geom_gap, ng); CD(geom_rbound, ng);
  CI(exclude_signature, src->nexclude);
  if (src->neq > 0) { CI(eq_type, src->neq); CI(eq_obj1id, src->neq); CI(eq_obj2id, src->neq); CI(eq_active0, src->neq); CD(eq_data, 11 * src->neq); CD(eq_solref, 2 * src->neq); CD(eq_solimp, 5 * src->neq); }
  CI(site_bodyid, ns); CD(site_pos, 3 * ns); CD(site_quat, 4 * ns);
  CI(actuator_trntype, nu); CI(actuator_trnid, nu); CI(actuator_ctrllimited, nu); CI(actuator_forcelimited, nu); CI(actuator_biastype, nu);
  CD(actuator_gainprm, 3 * nu); CD(actuator_biasprm, 3 * nu); CD(actuator_gear, nu);
  if (src->actuator_gear6) CD(actuator_gear6, 6 * nu);
  if (src->actuator_refsite) CI(actuator_refsite, nu);
  CD(actuator_ctrlrange, 2 * nu); CD(actuator_forcerange, 2 * nu);
  if (src->actuator_dyntype) { CI(actuator_dyntype, nu); CI(actuator_actadr, nu); CI(actuator_actlimited, nu); CD(actuator_dynprm, nu); CD(actuator_actrange, 2 * nu); }
  CI(tendon_adr, src->ntendon); CI(tendon_num, src->ntendon); CI(tendon_limited, src->ntendon); CI(wrap_objid, src->nwrap);
  CD(wrap_prm, src->nwrap); CD(tendon_range, 2 * src->ntendon); CD(tendon_margin, src->ntendon);
  CD(tendon_solref_lim, 2 * src->ntendon); CD(tendon_solimp_lim, 5 * src->ntendon); CD(tendon_invweight0, src->ntendon);
  if (src->tendon_stiffness) CD(tendon_stiffness, src->ntendon);
  if (src->tendon_damping) CD(tendon_damping, src->ntendon);
  if (src->tendon_lengthspring) CD(tendon_lengthspring, 2 * src->ntendon);
  if (src->tendon_frictionloss) CD(tendon_frictionloss, src->ntendon);
  if (src->tendon_solref_fri) CD(tendon_solref_fri, 2 * src->ntendon);
  if (src->tendon_solimp_fri) CD(tendon_solimp_fri, 5 * src->ntendon);
  if (src->geom_dataid) CI(geom_dataid, ng);
  if (src->nmesh > 0) { CI(mesh_vertadr, src->nmesh); CI(mesh_vertnum, src->nmesh); CD(mesh_vert, 3 * src->nmeshvert); }
  if (src->nhfield > 0) { CI(hfield_nrow, src->nhfield); CI(hfield_ncol, src->nhfield); CI(hfield_adr, src->nhfield); CD(hfield_size, 4 * src->nhfield); CD(hfield_data, src->nhfielddata); }
  CD(key_qpos, src->nkey * src->nq); CD(key_mpos, src->nkey * 3 * src->nmocap);
  copy_task(om, task);

  /* static collision filtering */
  const MjpcHipModel *m = &om->m;
  int cap = ng * (ng - 1) / 2 + 1;
  om->pair_g1 = (int *)mdup(om, NULL, sizeof(int) * (size_t)cap);
  om->pair_g2 = (int *)mdup(om, NULL, sizeof(int) * (size_t)cap);
  om->npair = 0;
  for (int a = 0; a < ng; a++) for (int b = a + 1; b < ng; b++) {
    int g1 = a, g2 = b;
    if (m->geom_type[g1] > m->geom_type[g2]) { g1 = b; g2 = a; }   /* collider table is upper-triangular in type */
    int b1 = m->geom_bodyid[g1], b2 = m->geom_bodyid[g2];
    int w1 = m->body_weldid[b1], w2 = m->body_weldid[b2];
    if (w1 == w2) continue;
    int pw1 = m->body_weldid[m->body_parentid[w1]], pw2 = m->body_weldid[m->body_parentid[w2]];
    if (w1 != 0 && w2 != 0 && (w1 == pw2 || w2 == pw1)) continue;
    int excl = 0;
    for (int e = 0; e < m->nexclude; e++)
      if (m->exclude_signature[e] == (b1 << 16) + b2 || m->exclude_signature[e] == (b2 << 16) + b1) excl = 1;
    if (excl) continue;
    if (!((m->geom_contype[g1] & m->geom_conaffinity[g2]) || (m->geom_contype[g2] & m->geom_conaffinity[g1]))) continue;
    if (m->geom_type[g1] == MJPC_GEOM_PLANE && m->geom_type[g2] == MJPC_GEOM_PLANE) continue;
    om->pair_g1[om->npair] = g1; om->pair_g2[om->npair] = g2; om->npair++;
  }
  om->nconmax = src->nconmax > 0 ? src->nconmax : 32;
  om->nefcmax = src->nefcmax > 0 ? src->nefcmax : 128;
  /* ray targets: geom group 0 */
  om->ray_geom = (int *)mdup(om, NULL, sizeof(int) * (size_t)(ng + 1));
  om->nray = 0;
  for (int g = 0; g < ng; g++) if (m->geom_group[g] == 0) om->ray_geom[om->nray++] = g;
  return om;
}

int oracle_set_task(OModel *om, const MjpcHipTask *task) { copy_task(om, task); return 0; }

void oracle_destroy(OModel *om) {
  if (!om) return;
  for (int i = 0; i < om->nblocks; i++) free(om->blocks[i]);
  free(om);
}
