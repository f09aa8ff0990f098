/*
 * oracle/collide.c — TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Analytic narrow-phase for geom primitives, restating the contact conventions of MuJoCo 3.1.4's
 * primitive colliders (third-party; reached from the reference through mj_step,
 * mjpc/trajectory.cc:158): normal points from geom1 to geom2, dist < 0 is penetration,
 * contact position is the midpoint between the two surfaces, contacts are created while
 * dist < margin.  Supported pairs: plane-{sphere,capsule,box,cylinder}, sphere-sphere,
 * sphere-capsule, capsule-capsule, sphere-box.  Other pairs (capsule-box, box-box and the
 * convex-fallback cylinder pairs) are counted in `unsupported` when their bounding spheres
 * overlap and produce no contact (documented gap).
 * PARITY UNPINNED (no MuJoCo in this image); analytic checks in tests/test_oracle_physics.py.
 */
#include "oracle.h"
#include "omath.h"

static int sphere_sphere_raw(OContact *con, double margin, const double *p1, double r1, const double *p2, double r2) {
  double dif[3];
  o_sub3(dif, p2, p1);
  double cdist = o_norm3(dif);
  double dist = cdist - r1 - r2;
  if (dist > margin) return 0;
  o_zero(con->frame, 9);
  if (cdist < O_MINVAL) { con->frame[0] = 1; }
  else o_scl3(con->frame, dif, 1.0 / cdist);
  con->dist = dist;
  o_addscl3(con->pos, p1, con->frame, r1 + 0.5 * dist);
  return 1;
}

static int plane_sphere_raw(OContact *con, double margin, const double *pp, const double *n, const double *c, double r) {
  double dif[3];
  o_sub3(dif, c, pp);
  double dist = o_dot3(dif, n) - r;
  if (dist > margin) return 0;
  o_zero(con->frame, 9);
  o_copy3(con->frame, n);
  con->dist = dist;
  o_addscl3(con->pos, c, n, -(r + 0.5 * dist));
  return 1;
}

static int plane_capsule(OContact *con, double margin, const double *pp, const double *pm, const double *cp, const double *cm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, axis[3] = {cm[2], cm[5], cm[8]}, seg[3], e[3];
  int cnt = 0;
  o_scl3(seg, axis, size[1]);
  o_add3(e, cp, seg);
  if (plane_sphere_raw(con + cnt, margin, pp, n, e, size[0])) { o_copy3(con[cnt].frame + 3, axis); cnt++; }
  o_sub3(e, cp, seg);
  if (plane_sphere_raw(con + cnt, margin, pp, n, e, size[0])) { o_copy3(con[cnt].frame + 3, axis); cnt++; }
  return cnt;
}

static int plane_box(OContact *con, double margin, const double *pp, const double *pm, const double *bp, const double *bm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, dif[3];
  o_sub3(dif, bp, pp);
  double dist = o_dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double vec[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]};
    double corner[3];
    o_mulmatvec3(corner, bm, vec);
    double ldist = o_dot3(n, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    OContact *c = con + cnt;
    c->dist = dist + ldist;
    o_zero(c->frame, 9); o_copy3(c->frame, n);
    o_add3(corner, corner, bp);
    o_addscl3(c->pos, corner, n, -0.5 * c->dist);
    if (++cnt >= 4) return 4;
  }
  return cnt;
}

static int plane_cylinder(OContact *con, double margin, const double *pp, const double *pm, const double *cp, const double *cm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, axis[3] = {cm[2], cm[5], cm[8]};
  double prjaxis = o_dot3(n, axis);
  if (prjaxis > 0) { o_scl3(axis, axis, -1); prjaxis = -prjaxis; }
  double vec[3];
  o_sub3(vec, cp, pp);
  double dist0 = o_dot3(vec, n);
  /* direction on the disk pointing most towards the plane */
  o_scl3(vec, axis, prjaxis); o_sub3(vec, vec, n);
  double len2 = o_dot3(vec, vec);
  if (len2 >= O_MINVAL) o_scl3(vec, vec, size[0] / sqrt(len2));
  else { vec[0] = cm[0] * size[0]; vec[1] = cm[3] * size[0]; vec[2] = cm[6] * size[0]; }
  double prjvec = o_dot3(vec, n);
  o_scl3(axis, axis, size[1]); prjaxis *= size[1];
  int cnt = 0;
  if (dist0 + prjaxis + prjvec <= margin) {
    OContact *c = con + cnt++;
    c->dist = dist0 + prjaxis + prjvec;
    o_add3(c->pos, cp, vec); o_add3(c->pos, c->pos, axis); o_addtoscl3(c->pos, n, -0.5 * c->dist);
    o_zero(c->frame, 9); o_copy3(c->frame, n);
  } else return 0;
  if (dist0 - prjaxis + prjvec <= margin) {
    OContact *c = con + cnt++;
    c->dist = dist0 - prjaxis + prjvec;
    o_add3(c->pos, cp, vec); o_sub3(c->pos, c->pos, axis); o_addtoscl3(c->pos, n, -0.5 * c->dist);
    o_zero(c->frame, 9); o_copy3(c->frame, n);
  }
  /* two more points of an inscribed triangle on the near disk */
  double prjvec1 = -0.5 * prjvec;
  if (dist0 + prjaxis + prjvec1 <= margin) {
    double vec1[3];
    o_cross(vec1, vec, axis);
    o_normalize3(vec1);
    o_scl3(vec1, vec1, size[0] * sqrt(3.0) / 2);
    for (int s = -1; s <= 1; s += 2) {
      OContact *c = con + cnt++;
      c->dist = dist0 + prjaxis + prjvec1;
      o_add3(c->pos, cp, axis); o_addtoscl3(c->pos, vec, -0.5); o_addtoscl3(c->pos, vec1, (double)s);
      o_addtoscl3(c->pos, n, -0.5 * c->dist);
      o_zero(c->frame, 9); o_copy3(c->frame, n);
    }
  }
  return cnt;
}

static int sphere_capsule(OContact *con, double margin, const double *sp, double sr, const double *cp, const double *cm, const double *csize) {
  double axis[3] = {cm[2], cm[5], cm[8]}, vec[3], pt[3];
  o_sub3(vec, sp, cp);
  double x = o_clip(o_dot3(axis, vec), -csize[1], csize[1]);
  o_addscl3(pt, cp, axis, x);
  return sphere_sphere_raw(con, margin, sp, sr, pt, csize[0]);
}

static int capsule_capsule(OContact *con, double margin, const double *p1, const double *m1, const double *s1,
                           const double *p2, const double *m2, const double *s2) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3];
  o_sub3(dif, p1, p2);
  double len1 = s1[1], len2 = s2[1];
  double ma = o_dot3(a1, a1), mb = -o_dot3(a1, a2), mc = o_dot3(a2, a2);
  double u = -o_dot3(a1, dif), v = o_dot3(a2, dif);
  double det = ma * mc - mb * mb;
  if (fabs(det) >= O_MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > len1) { x1 = len1; x2 = (v - mb * len1) / mc; }
    else if (x1 < -len1) { x1 = -len1; x2 = (v + mb * len1) / mc; }
    if (x2 > len2) { x2 = len2; x1 = o_clip((u - mb * len2) / ma, -len1, len1); }
    else if (x2 < -len2) { x2 = -len2; x1 = o_clip((u + mb * len2) / ma, -len1, len1); }
    double v1[3], v2[3];
    o_addscl3(v1, p1, a1, x1); o_addscl3(v2, p2, a2, x2);
    return sphere_sphere_raw(con, margin, v1, s1[0], v2, s2[0]);
  }
  /* parallel axes: test both ends of capsule 1 against segment 2 */
  int cnt = 0;
  for (int s = -1; s <= 1 && cnt < 2; s += 2) {
    double e[3], w[3], pt[3];
    o_addscl3(e, p1, a1, s * len1);
    o_sub3(w, e, p2);
    double x = o_clip(o_dot3(a2, w), -len2, len2);
    o_addscl3(pt, p2, a2, x);
    cnt += sphere_sphere_raw(con + cnt, margin, e, s1[0], pt, s2[0]);
  }
  return cnt;
}

static int sphere_box(OContact *con, double margin, const double *sp, double sr, const double *bp, const double *bm, const double *bs) {
  double dif[3], c[3], clamped[3];
  o_sub3(dif, sp, bp);
  o_mulmattvec3(c, bm, dif);              /* sphere centre in box frame */
  int inside = 1;
  for (int i = 0; i < 3; i++) {
    clamped[i] = o_clip(c[i], -bs[i], bs[i]);
    if (clamped[i] != c[i]) inside = 0;
  }
  double nloc[3], dist;
  if (!inside) {
    double d[3]; o_sub3(d, c, clamped);
    double len = o_norm3(d);
    dist = len - sr;
    if (dist > margin) return 0;
    o_scl3(nloc, d, 1.0 / len);          /* from box to sphere */
  } else {
    /* centre inside: push out through the nearest face */
    int k = 0; double best = 1e300;
    for (int i = 0; i < 3; i++) { double pen = bs[i] - fabs(c[i]); if (pen < best) { best = pen; k = i; } }
    nloc[0] = nloc[1] = nloc[2] = 0;
    nloc[k] = c[k] >= 0 ? 1 : -1;
    o_copy3(clamped, c); clamped[k] = nloc[k] * bs[k];
    dist = -best - sr;
  }
  /* geom1 = sphere, geom2 = box: normal from sphere to box */
  double nw[3], surf[3];
  o_mulmatvec3(nw, bm, nloc);
  o_zero(con->frame, 9);
  o_scl3(con->frame, nw, -1);
  con->dist = dist;
  o_mulmatvec3(surf, bm, clamped); o_add3(surf, surf, bp);   /* box surface point */
  o_addscl3(con->pos, surf, nw, 0.5 * dist);
  return 1;
}

int oracle_collide_pair(const OModel *om, OData *d, int g1, int g2, double margin, OContact *con, int *unsupported) {
  const MjpcHipModel *m = &om->m;
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
  const double *m1 = d->geom_xmat + 9 * g1, *m2 = d->geom_xmat + 9 * g2;
  const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
  /* pairs are stored with type1 <= type2 */
  if (t1 == MJPC_GEOM_PLANE) {
    double n[3] = {m1[2], m1[5], m1[8]};
    switch (t2) {
      case MJPC_GEOM_SPHERE: return plane_sphere_raw(con, margin, p1, n, p2, s2[0]);
      case MJPC_GEOM_CAPSULE: return plane_capsule(con, margin, p1, m1, p2, m2, s2);
      case MJPC_GEOM_BOX: return plane_box(con, margin, p1, m1, p2, m2, s2);
      case MJPC_GEOM_CYLINDER: return plane_cylinder(con, margin, p1, m1, p2, m2, s2);
      default: break;
    }
  } else if (t1 == MJPC_GEOM_SPHERE) {
    switch (t2) {
      case MJPC_GEOM_SPHERE: return sphere_sphere_raw(con, margin, p1, s1[0], p2, s2[0]);
      case MJPC_GEOM_CAPSULE: return sphere_capsule(con, margin, p1, s1[0], p2, m2, s2);
      case MJPC_GEOM_BOX: return sphere_box(con, margin, p1, s1[0], p2, m2, s2);
      default: break;
    }
  } else if (t1 == MJPC_GEOM_CAPSULE && t2 == MJPC_GEOM_CAPSULE) {
    return capsule_capsule(con, margin, p1, m1, s1, p2, m2, s2);
  }
  (*unsupported)++;
  return 0;
}
