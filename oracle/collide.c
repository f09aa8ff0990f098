/*
 * oracle/collide.c — TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * Analytic narrow-phase for geom primitives, restating the contact conventions of MuJoCo 3.1.4's
 * primitive colliders (third-party; reached from the reference through mj_step,
 * mjpc/trajectory.cc:158): normal points from geom1 to geom2, dist < 0 is penetration,
 * contact position is the midpoint between the two surfaces, contacts are created while
 * dist < margin.  Supported pairs: plane-{sphere,capsule,box,cylinder}, sphere-sphere,
 * sphere-capsule, capsule-capsule, sphere-box, capsule-box, box-box, sphere-cylinder, capsule-cylinder.  MuJoCo's own capsule-box
 * (mjraw_CapsuleBox) and box-box (mjc_BoxBox) routines are long case analyses that cannot be restated
 * from the reference tree; the two colliders here are this build's own constructions with the same
 * conventions and contact budgets (capsule-box <= 2 contacts: closest segment point + far end cap;
 * box-box <= 4: separating-axis test, then reference-face clipping or one edge-edge contact).
 * cylinder-cylinder, cylinder-box and every ellipsoid pair (plane-ellipsoid: closed form) go through a portal-refinement (MPR)
 * collider, below; so do convex meshes (support = hull vertex farthest along the direction).  Height fields meet convex geoms prism by prism through the same
 * collider (hfield_convex).  Pairs beyond its cell / contact caps are counted in `unsupported` and produce no contact; the engine refuses such models at create().
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

/* ---- capsule (geom1) vs box (geom2) ------------------------------------------------------------
 * The closest point of the capsule's segment to the box is found in the box frame: f(s) = dist^2(p0 + s h a, box)
 * is convex and C1 in s, g(s) = f'(s) / (2h) = sum_i excess_i * sign(q_i) * a_i is monotone and piecewise linear; its root is
 * bracketed on [-1, 1], the bracket narrowed at g's breakpoints, the root of the remaining linear piece taken exactly (the same
 * IEEE operations on host and device).
 * Contact 1 = sphere-box at that point; contact 2 = sphere-box at the end cap farther from it (a capsule lying on a
 * face gets two contacts, a poking capsule one). */
static double capsule_box_g(const double *p0, const double *a, double h, const double *b, double s) {
  double g = 0;
  for (int i = 0; i < 3; i++) {
    double q = p0[i] + (s * h) * a[i];
    double e = fabs(q) - b[i];
    if (e > 0) g += (q > 0 ? e : -e) * a[i];
  }
  return g;
}
static int capsule_box(OContact *con, double margin, const double *cp, const double *cm, const double *cs,
                       const double *bp, const double *bm, const double *bs) {
  double axis[3] = {cm[2], cm[5], cm[8]}, dif[3], p0[3], a[3];
  o_sub3(dif, cp, bp);
  o_mulmattvec3(p0, bm, dif);
  o_mulmattvec3(a, bm, axis);
  double h = cs[1], sstar;
  if (capsule_box_g(p0, a, h, bs, -1.0) >= 0) sstar = -1.0;
  else if (capsule_box_g(p0, a, h, bs, 1.0) <= 0) sstar = 1.0;
  else {
    /* g is piecewise linear with breakpoints where a coordinate crosses a face (|q_i| = b_i): the bracket is narrowed at the
     * (at most six) breakpoints inside it, then g is linear on what is left and its root is exact */
    double lo = -1.0, hi = 1.0, glo = capsule_box_g(p0, a, h, bs, -1.0), ghi = capsule_box_g(p0, a, h, bs, 1.0);
    for (int i = 0; i < 3; i++) for (int sg = -1; sg <= 1; sg += 2) {
      double den = h * a[i];
      if (fabs(den) < O_MINVAL) continue;
      double t = (sg * bs[i] - p0[i]) / den;
      if (!(t > lo && t < hi)) continue;
      double gt = capsule_box_g(p0, a, h, bs, t);
      if (gt < 0) { lo = t; glo = gt; } else { hi = t; ghi = gt; }
    }
    /* (a numerically flat piece is the zero-distance stretch of a segment that passes through the box: its left end, like the
     * leftmost point with g >= 0 everywhere else) */
    double dg = ghi - glo;
    sstar = dg > 1e-15 ? lo - glo * (hi - lo) / dg : lo;
  }
  int cnt = 0;
  double pt[3];
  o_addscl3(pt, cp, axis, sstar * h);
  cnt += sphere_box(con + cnt, margin, pt, cs[0], bp, bm, bs);
  double s2 = sstar <= 0 ? 1.0 : -1.0;
  o_addscl3(pt, cp, axis, s2 * h);
  cnt += sphere_box(con + cnt, margin, pt, cs[0], bp, bm, bs);
  return cnt;
}

/* ---- sphere (geom1) vs cylinder (geom2), capsule (geom1) vs cylinder (geom2) -------------------------
 * MuJoCo sends every cylinder pair except plane-cylinder to its general convex solver; a solid cylinder is simple enough for
 * closed forms with the same contact conventions: the distance from a point to the cylinder is
 * hypot(max(rho - R, 0), max(|z| - h, 0)) in the cylinder's frame, the closest point clamps rho and z.  The capsule case
 * bisects the monotone derivative of that (convex) distance along the capsule's segment, exactly like capsule-box. */
static int sphere_cylinder(OContact *con, double margin, const double *sp, double sr, const double *cp, const double *cm, const double *cs) {
  double dif[3], p[3];
  o_sub3(dif, sp, cp);
  o_mulmattvec3(p, cm, dif);                    /* sphere centre in the cylinder's frame (axis = z) */
  double R = cs[0], h = cs[1];
  double rho = sqrt(p[0] * p[0] + p[1] * p[1]);
  double ux = rho > O_MINVAL ? p[0] / rho : 1.0, uy = rho > O_MINVAL ? p[1] / rho : 0.0;      /* radial direction */
  double er = rho - R, ez = fabs(p[2]) - h, sz = p[2] >= 0 ? 1.0 : -1.0;
  double q[3], nl[3], dist;
  if (er <= 0 && ez <= 0) {                     /* centre inside: leave through the nearer of the lateral surface and the cap */
    if (-er < -ez) { q[0] = ux * R; q[1] = uy * R; q[2] = p[2]; nl[0] = ux; nl[1] = uy; nl[2] = 0; dist = er - sr; }
    else { q[0] = p[0]; q[1] = p[1]; q[2] = sz * h; nl[0] = 0; nl[1] = 0; nl[2] = sz; dist = ez - sr; }
  } else {
    double cr = er > 0 ? R : rho;               /* clamped radius */
    q[0] = ux * cr; q[1] = uy * cr; q[2] = ez > 0 ? sz * h : p[2];
    double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    double len = o_norm3(d);
    dist = len - sr;
    if (dist > margin) return 0;
    o_scl3(nl, d, 1.0 / len);                   /* from the cylinder to the sphere */
  }
  if (dist > margin) return 0;
  double nw[3], surf[3];
  o_mulmatvec3(nw, cm, nl);
  o_zero(con->frame, 9);
  o_scl3(con->frame, nw, -1);                   /* geom1 = sphere, geom2 = cylinder */
  con->dist = dist;
  o_mulmatvec3(surf, cm, q); o_add3(surf, surf, cp);
  o_addscl3(con->pos, surf, nw, 0.5 * dist);
  return 1;
}
static double capsule_cylinder_g(const double *p0, const double *a, double hc, double R, double h, double s) {
  double q[3] = {p0[0] + (s * hc) * a[0], p0[1] + (s * hc) * a[1], p0[2] + (s * hc) * a[2]};
  double rho = sqrt(q[0] * q[0] + q[1] * q[1]);
  double g = 0, er = rho - R, ez = fabs(q[2]) - h;
  if (er > 0) g += er * (q[0] * a[0] + q[1] * a[1]) / rho;
  if (ez > 0) g += (q[2] > 0 ? ez : -ez) * a[2];
  return g;
}
static int capsule_cylinder(OContact *con, double margin, const double *kp, const double *km, const double *ks,
                            const double *cp, const double *cm, const double *cs) {
  double axis[3] = {km[2], km[5], km[8]}, dif[3], p0[3], a[3];
  o_sub3(dif, kp, cp);
  o_mulmattvec3(p0, cm, dif);
  o_mulmattvec3(a, cm, axis);
  double hc = ks[1], sstar;
  if (capsule_cylinder_g(p0, a, hc, cs[0], cs[1], -1.0) >= 0) sstar = -1.0;
  else if (capsule_cylinder_g(p0, a, hc, cs[0], cs[1], 1.0) <= 0) sstar = 1.0;
  else {
    double lo = -1.0, hi = 1.0;
    for (int it = 0; it < 48; it++) {
      double mid = 0.5 * (lo + hi);
      if (capsule_cylinder_g(p0, a, hc, cs[0], cs[1], mid) < 0) lo = mid; else hi = mid;
    }
    sstar = 0.5 * (lo + hi);
  }
  int cnt = 0;
  double pt[3];
  o_addscl3(pt, kp, axis, sstar * hc);
  cnt += sphere_cylinder(con + cnt, margin, pt, ks[0], cp, cm, cs);
  double s2 = sstar <= 0 ? 1.0 : -1.0;
  o_addscl3(pt, kp, axis, s2 * hc);
  cnt += sphere_cylinder(con + cnt, margin, pt, ks[0], cp, cm, cs);
  return cnt;
}

/* ---- box (geom1 = A) vs box (geom2 = B) ----------------------------------------------------------
 * Separating-axis test over the 15 axes (faces of A, faces of B, edge x edge) in A's frame; the axis of largest
 * separation (least penetration) decides the case, edge axes only when clearly better than the best face.
 *   face case: the incident face of the other box is projected on the reference face; contact candidates are the
 *     incident vertices inside the reference rectangle, the rectangle corners inside the incident parallelogram and
 *     the edge crossings (24 slots, at most 8 filled); up to four are kept: deepest, farthest from it, and the two
 *     extreme points on either side of that line.  dist = signed height over the reference face, pos = half way.
 *   edge case: one contact at the closest points of the two supporting edges.
 * Normal always points from A to B. */
typedef struct { double x, y, d; int ok; } BBCand;
#define BB_TOL 1e-9
/* a later candidate replaces an earlier one only if it is better by more than this: near-ties (a cube lying flat: several corners
 * at the same depth to rounding) resolve to the lowest candidate index instead of flipping with the last bit of the pose */
#define BB_TIE_D 1e-10
#define BB_TIE_A 1e-12
static int box_box(OContact *con, double margin, const double *pa, const double *ma, const double *sa,
                   const double *pb, const double *mb, const double *sb) {
  double R[9], AR[9], t[3], tb[3], dif[3];
  o_sub3(dif, pb, pa);
  o_mulmattvec3(t, ma, dif);                       /* centre of B in A's frame */
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    double r = ma[i] * mb[j] + ma[3 + i] * mb[3 + j] + ma[6 + i] * mb[6 + j];     /* a_i . b_j (axes are matrix columns) */
    R[3 * i + j] = r; AR[3 * i + j] = fabs(r);
  }
  for (int j = 0; j < 3; j++) tb[j] = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
  double best = -1e300; int code = -1;
  for (int i = 0; i < 3; i++) {
    double s = fabs(t[i]) - (sa[i] + sb[0] * AR[3 * i] + sb[1] * AR[3 * i + 1] + sb[2] * AR[3 * i + 2]);
    if (s > margin) return 0;
    if (s > best) { best = s; code = i; }
  }
  for (int j = 0; j < 3; j++) {
    double s = fabs(tb[j]) - (sb[j] + sa[0] * AR[j] + sa[1] * AR[3 + j] + sa[2] * AR[6 + j]);
    if (s > margin) return 0;
    if (s > best) { best = s; code = 3 + j; }
  }
  double ebest = -1e300; int ecode = -1;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    double l2 = 1.0 - R[3 * i + j] * R[3 * i + j];
    if (l2 < 1e-6) continue;                       /* parallel edges: covered by the face axes */
    double proj = t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j];
    double ra = sa[i1] * AR[3 * i2 + j] + sa[i2] * AR[3 * i1 + j];
    double rb = sb[j1] * AR[3 * i + j2] + sb[j2] * AR[3 * i + j1];
    double s = (fabs(proj) - (ra + rb)) / sqrt(l2);
    if (s > margin) return 0;
    if (s > ebest) { ebest = s; ecode = 3 * i + j; }
  }
  if (ecode >= 0 && ebest > best + 0.05 * fabs(best) + BB_TOL) {
    /* ---- edge-edge */
    int i = ecode / 3, j = ecode % 3;
    double ai[3] = {ma[i], ma[3 + i], ma[6 + i]}, bj[3] = {mb[j], mb[3 + j], mb[6 + j]}, n[3];
    o_cross(n, ai, bj);
    o_normalize3(n);
    if (o_dot3(n, dif) < 0) o_scl3(n, n, -1);      /* from A to B */
    double ea[3], eb[3];
    o_copy3(ea, pa); o_copy3(eb, pb);
    for (int k = 0; k < 3; k++) {
      double ak[3] = {ma[k], ma[3 + k], ma[6 + k]}, bk[3] = {mb[k], mb[3 + k], mb[6 + k]};
      if (k != i) o_addtoscl3(ea, ak, o_dot3(n, ak) > 0 ? sa[k] : -sa[k]);
      if (k != j) o_addtoscl3(eb, bk, o_dot3(n, bk) > 0 ? -sb[k] : sb[k]);
    }
    /* closest points of the lines ea + u ai, eb + v bj */
    double w[3]; o_sub3(w, eb, ea);
    double c = R[3 * i + j], d1 = o_dot3(w, ai), d2 = o_dot3(w, bj), den = 1.0 - c * c;
    double u = o_clip((d1 - c * d2) / den, -sa[i], sa[i]);
    double v = o_clip((c * d1 - d2) / den, -sb[j], sb[j]);
    double qa[3], qb[3];
    o_addscl3(qa, ea, ai, u); o_addscl3(qb, eb, bj, v);
    o_sub3(w, qb, qa);
    double dist = o_dot3(w, n);
    if (dist > margin) return 0;
    o_zero(con->frame, 9); o_copy3(con->frame, n);
    con->dist = dist;
    con->pos[0] = 0.5 * (qa[0] + qb[0]); con->pos[1] = 0.5 * (qa[1] + qb[1]); con->pos[2] = 0.5 * (qa[2] + qb[2]);
    return 1;
  }
  /* ---- face case: reference box r, incident box q */
  int refA = code < 3, ax = refA ? code : code - 3;
  const double *pr = refA ? pa : pb, *mr = refA ? ma : mb, *sr = refA ? sa : sb;
  const double *pq = refA ? pb : pa, *mq = refA ? mb : ma, *sq = refA ? sb : sa;
  double sgn = (refA ? t[ax] : -tb[ax]) >= 0 ? 1.0 : -1.0;       /* towards the incident box */
  int u1 = (ax + 1) % 3, u2 = (ax + 2) % 3;
  double n[3] = {sgn * mr[ax], sgn * mr[3 + ax], sgn * mr[6 + ax]};
  double ru[3] = {mr[u1], mr[3 + u1], mr[6 + u1]}, rv[3] = {mr[u2], mr[3 + u2], mr[6 + u2]};
  double hu = sr[u1], hv = sr[u2], hn = sr[ax];
  /* incident face: the face of q whose outward normal is most opposed to n */
  double nl[3];
  o_mulmattvec3(nl, mq, n);
  int k = 0; double amax = fabs(nl[0]);
  if (fabs(nl[1]) > amax) { amax = fabs(nl[1]); k = 1; }
  if (fabs(nl[2]) > amax) { amax = fabs(nl[2]); k = 2; }
  int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
  double qk[3] = {mq[k], mq[3 + k], mq[6 + k]}, q1[3] = {mq[k1], mq[3 + k1], mq[6 + k1]}, q2[3] = {mq[k2], mq[3 + k2], mq[6 + k2]};
  double cen[3], rel[3];
  o_addscl3(cen, pq, qk, nl[k] > 0 ? -sq[k] : sq[k]);
  o_sub3(rel, cen, pr);
  /* incident face centre and half edges in the reference face frame (x, y, height over the face) */
  double c0[3] = {o_dot3(rel, ru), o_dot3(rel, rv), o_dot3(rel, n) - hn};
  double e1[3] = {sq[k1] * o_dot3(q1, ru), sq[k1] * o_dot3(q1, rv), sq[k1] * o_dot3(q1, n)};
  double e2[3] = {sq[k2] * o_dot3(q2, ru), sq[k2] * o_dot3(q2, rv), sq[k2] * o_dot3(q2, n)};
  BBCand cand[24];
  for (int q = 0; q < 24; q++) cand[q].ok = 0;
  double vx[4], vy[4], vd[4];
  for (int q = 0; q < 4; q++) {                    /* (a) incident vertices, counter-clockwise in (e1, e2) */
    double s1 = (q == 0 || q == 3) ? -1.0 : 1.0, s2 = (q < 2) ? -1.0 : 1.0;
    vx[q] = c0[0] + s1 * e1[0] + s2 * e2[0]; vy[q] = c0[1] + s1 * e1[1] + s2 * e2[1]; vd[q] = c0[2] + s1 * e1[2] + s2 * e2[2];
    if (fabs(vx[q]) <= hu + BB_TOL && fabs(vy[q]) <= hv + BB_TOL) { cand[q].ok = 1; cand[q].x = vx[q]; cand[q].y = vy[q]; cand[q].d = vd[q]; }
  }
  double det = e1[0] * e2[1] - e1[1] * e2[0];
  if (fabs(det) > 1e-14) {                         /* (b) reference corners inside the incident parallelogram */
    for (int q = 0; q < 4; q++) {
      double cx = (q == 0 || q == 3) ? -hu : hu, cy = (q < 2) ? -hv : hv;
      double dx = cx - c0[0], dy = cy - c0[1];
      double al = (dx * e2[1] - dy * e2[0]) / det, be = (e1[0] * dy - e1[1] * dx) / det;
      if (fabs(al) <= 1.0 + BB_TOL && fabs(be) <= 1.0 + BB_TOL) {
        BBCand *cd = cand + 4 + q;
        cd->ok = 1; cd->x = cx; cd->y = cy; cd->d = c0[2] + al * e1[2] + be * e2[2];
      }
    }
  }
  for (int q = 0; q < 4; q++) {                    /* (c) incident edges against the four rectangle lines */
    int q2i = (q + 1) & 3;
    double px = vx[q], py = vy[q], pd = vd[q], dx = vx[q2i] - px, dy = vy[q2i] - py, dd = vd[q2i] - pd;
    for (int e = 0; e < 4; e++) {
      int xline = e < 2;
      double lim = (e & 1) ? 1.0 : -1.0;
      double num = xline ? lim * hu - px : lim * hv - py, den = xline ? dx : dy;
      if (fabs(den) < 1e-14) continue;
      double s = num / den;
      if (s <= 0.0 || s >= 1.0) continue;
      double ox = xline ? lim * hu : px + s * dx, oy = xline ? py + s * dy : lim * hv;
      if ((xline ? fabs(oy) - hv : fabs(ox) - hu) > BB_TOL) continue;
      BBCand *cd = cand + 8 + 4 * q + e;
      cd->ok = 1; cd->x = ox; cd->y = oy; cd->d = pd + s * dd;
    }
  }
  /* selection: deepest, farthest from it, then the extreme points on either side of that line */
  int sel[4] = {-1, -1, -1, -1};
  double bd = 1e300;
  for (int q = 0; q < 24; q++) if (cand[q].ok && cand[q].d <= margin && cand[q].d < bd - BB_TIE_D) { bd = cand[q].d; sel[0] = q; }
  if (sel[0] < 0) return 0;
  double x0 = cand[sel[0]].x, y0 = cand[sel[0]].y, far = 1e-16;
  for (int q = 0; q < 24; q++) if (cand[q].ok && cand[q].d <= margin) {
    double r2 = (cand[q].x - x0) * (cand[q].x - x0) + (cand[q].y - y0) * (cand[q].y - y0);
    if (r2 > far + BB_TIE_A) { far = r2; sel[1] = q; }
  }
  if (sel[1] >= 0) {
    double lx = cand[sel[1]].x - x0, ly = cand[sel[1]].y - y0, amx = 1e-12, amn = -1e-12;
    for (int q = 0; q < 24; q++) if (cand[q].ok && cand[q].d <= margin) {
      double ar = lx * (cand[q].y - y0) - ly * (cand[q].x - x0);
      if (ar > amx + BB_TIE_A) { amx = ar; sel[2] = q; }
      if (ar < amn - BB_TIE_A) { amn = ar; sel[3] = q; }
    }
  }
  int cnt = 0;
  for (int q = 0; q < 4; q++) if (sel[q] >= 0) {
    const BBCand *cd = cand + sel[q];
    OContact *c = con + cnt++;
    /* the point on the incident face, moved half way down to the reference face */
    double hgt = hn + cd->d - 0.5 * cd->d;
    c->pos[0] = pr[0] + cd->x * ru[0] + cd->y * rv[0] + hgt * n[0];
    c->pos[1] = pr[1] + cd->x * ru[1] + cd->y * rv[1] + hgt * n[1];
    c->pos[2] = pr[2] + cd->x * ru[2] + cd->y * rv[2] + hgt * n[2];
    c->dist = cd->d;
    o_zero(c->frame, 9);
    if (refA) o_copy3(c->frame, n); else o_scl3(c->frame, n, -1);
  }
  return cnt;
}

/* squared distance from point q to the segment p +- h a */
static double seg_point_dist2(const double *p, const double *a, double h, const double *q) {
  double w[3];
  o_sub3(w, q, p);
  double x = o_clip(o_dot3(a, w), -h, h);
  o_addtoscl3(w, a, -x);
  return o_dot3(w, w);
}

/* ---- convex pairs without an analytic collider (cylinder-cylinder, cylinder-box): Minkowski portal refinement ------------
 * MuJoCo sends these pairs to libccd's MPR (mjc_Convex; opt.mpr_tolerance 1e-6, opt.mpr_iterations 50), a third-party
 * library that is not in the reference tree.  This is the published algorithm (G. Snethen, "XenoCollide", Game Programming
 * Gems 7): portal discovery, refinement until the portal reaches the surface of the Minkowski difference within the
 * tolerance, penetration = the portal point closest to the origin, contact position = half way between the two witness points.  Both geoms are inflated by margin / 2 so that proximity
 * within the margin is a (positive-distance) contact; ONE contact per pair, like MuJoCo without multiccd.  The engine carries
 * the same operations in the same order (csrc/collide.h: np_convex). */
#define MPR_TOLERANCE 1e-6
#define MPR_ITERATIONS 50
typedef struct { int type; const double *pos, *mat, *size; double margin; const double *vert; int nvert; } MShape;
typedef struct { double v[3], v1[3], v2[3]; } MSup;

static void mpr_support1(const MShape *s, const double *dir, double *out) {
  double l[3], v[3];
  o_mulmattvec3(l, s->mat, dir);
  if (s->type == MJPC_GEOM_BOX) {
    for (int k = 0; k < 3; k++) v[k] = l[k] >= 0 ? s->size[k] : -s->size[k];
  } else if (s->type == MJPC_GEOM_CYLINDER) {
    double n = sqrt(l[0] * l[0] + l[1] * l[1]);
    if (n > O_MINVAL) { v[0] = s->size[0] * l[0] / n; v[1] = s->size[0] * l[1] / n; } else { v[0] = 0; v[1] = 0; }
    v[2] = l[2] >= 0 ? s->size[1] : -s->size[1];
  } else if (s->type == MJPC_GEOM_MESH) {       /* hull vertex farthest along l (first of equals) */
    double best = -1e300; int bi = 0;
    for (int i = 0; i < s->nvert; i++) {
      double t = s->vert[3 * i] * l[0] + s->vert[3 * i + 1] * l[1] + s->vert[3 * i + 2] * l[2];
      if (t > best) { best = t; bi = i; }
    }
    v[0] = s->vert[3 * bi]; v[1] = s->vert[3 * bi + 1]; v[2] = s->vert[3 * bi + 2];
  } else if (s->type == MJPC_GEOM_ELLIPSOID) {
    double a = s->size[0] * s->size[0] * l[0], b = s->size[1] * s->size[1] * l[1], c = s->size[2] * s->size[2] * l[2];
    double n = sqrt(a * l[0] + b * l[1] + c * l[2]);
    if (n > O_MINVAL) { v[0] = a / n; v[1] = b / n; v[2] = c / n; } else { v[0] = 0; v[1] = 0; v[2] = 0; }
  } else if (s->type == MJPC_GEOM_CAPSULE) {
    v[0] = s->size[0] * l[0]; v[1] = s->size[0] * l[1]; v[2] = s->size[0] * l[2] + (l[2] >= 0 ? s->size[1] : -s->size[1]);
  } else {   /* sphere */
    v[0] = s->size[0] * l[0]; v[1] = s->size[0] * l[1]; v[2] = s->size[0] * l[2];
  }
  o_mulmatvec3(out, s->mat, v);
  o_add3(out, out, s->pos);
  o_addtoscl3(out, dir, s->margin);
}
/* support point of A - B in the unit direction dir */
static void mpr_support(const MShape *a, const MShape *b, const double *dir, MSup *s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  mpr_support1(a, dir, s->v1);
  mpr_support1(b, nd, s->v2);
  o_sub3(s->v, s->v1, s->v2);
}
static void mpr_portal_dir(const MSup *p, double *dir) {    /* outward normal of the portal triangle (v1, v2, v3) */
  double a[3], b[3];
  o_sub3(a, p[2].v, p[1].v);
  o_sub3(b, p[3].v, p[1].v);
  o_cross(dir, a, b);
  o_normalize3(dir);
}
static int mpr_reach_tolerance(const MSup *p, const MSup *v4, const double *dir) {
  double dv4 = o_dot3(v4->v, dir);
  double d1 = dv4 - o_dot3(p[1].v, dir), d2 = dv4 - o_dot3(p[2].v, dir), d3 = dv4 - o_dot3(p[3].v, dir);
  double dm = fmin(d1, fmin(d2, d3));
  return dm <= MPR_TOLERANCE;
}
static void mpr_expand_portal(MSup *p, const MSup *v4) {
  double v4v0[3];
  o_cross(v4v0, v4->v, p[0].v);
  if (o_dot3(p[1].v, v4v0) > 0) {
    if (o_dot3(p[2].v, v4v0) > 0) p[1] = *v4; else p[3] = *v4;
  } else {
    if (o_dot3(p[3].v, v4v0) > 0) p[2] = *v4; else p[1] = *v4;
  }
}
/* closest point of the triangle (a, b, c) to the origin (Ericson, Real-Time Collision Detection 5.1.5) */
static void mpr_closest_on_triangle(const double *a, const double *b, const double *c, double *w, double *bw) {
  double ab[3], ac[3];
  o_sub3(ab, b, a); o_sub3(ac, c, a);
  double d1 = -o_dot3(ab, a), d2 = -o_dot3(ac, a);
  if (d1 <= 0 && d2 <= 0) { o_copy3(w, a); bw[0] = 1; bw[1] = 0; bw[2] = 0; return; }
  double d3 = -o_dot3(ab, b), d4 = -o_dot3(ac, b);
  if (d3 >= 0 && d4 <= d3) { o_copy3(w, b); bw[0] = 0; bw[1] = 1; bw[2] = 0; return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); o_addscl3(w, a, ab, v); bw[0] = 1 - v; bw[1] = v; bw[2] = 0; return; }
  double d5 = -o_dot3(ab, c), d6 = -o_dot3(ac, c);
  if (d6 >= 0 && d5 <= d6) { o_copy3(w, c); bw[0] = 0; bw[1] = 0; bw[2] = 1; return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double v = d2 / (d2 - d6); o_addscl3(w, a, ac, v); bw[0] = 1 - v; bw[1] = 0; bw[2] = v; return; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double bc[3]; o_sub3(bc, c, b);
    double v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    o_addscl3(w, b, bc, v); bw[0] = 0; bw[1] = 1 - v; bw[2] = v; return;
  }
  double den = 1.0 / (va + vb + vc);
  o_addscl3(w, a, ab, vb * den);
  o_addtoscl3(w, ac, vc * den);
  bw[1] = vb * den; bw[2] = vc * den; bw[0] = 1 - bw[1] - bw[2];
}
/* 1 contact (frame[0..2] = normal from geom 1 to geom 2) or 0 */
static int convex_mpr_shapes(OContact *con, double margin, MShape A, MShape B);
static int convex_mpr(OContact *con, double margin, int t1, const double *p1, const double *m1, const double *s1,
                      int t2, const double *p2, const double *m2, const double *s2) {
  MShape A = {t1, p1, m1, s1, 0.5 * margin, 0, 0}, B = {t2, p2, m2, s2, 0.5 * margin, 0, 0};
  return convex_mpr_shapes(con, margin, A, B);
}
static int convex_mpr_shapes(OContact *con, double margin, MShape A, MShape B) {
  const double *p1 = A.pos, *p2 = B.pos;
  MSup p[4], v4;
  double dir[3], va[3], vb[3], depth, nrm[3], pos[3];
  /* portal discovery */
  o_sub3(p[0].v, p1, p2); o_copy3(p[0].v1, p1); o_copy3(p[0].v2, p2);
  if (o_dot3(p[0].v, p[0].v) < O_MINVAL * O_MINVAL) p[0].v[0] += 1e-9;
  o_scl3(dir, p[0].v, -1); o_normalize3(dir);
  mpr_support(&A, &B, dir, &p[1]);
  if (o_dot3(p[1].v, dir) <= 0) return 0;
  o_cross(dir, p[0].v, p[1].v);
  int found = 0;     /* 0: portal, 2: the origin lies on the segment v0-v1 */
  if (o_dot3(dir, dir) < O_MINVAL * O_MINVAL) found = 2;
  else {
    o_normalize3(dir);
    mpr_support(&A, &B, dir, &p[2]);
    if (o_dot3(p[2].v, dir) <= 0) return 0;
    o_sub3(va, p[1].v, p[0].v); o_sub3(vb, p[2].v, p[0].v);
    o_cross(dir, va, vb); o_normalize3(dir);
    if (o_dot3(dir, p[0].v) > 0) { MSup t = p[1]; p[1] = p[2]; p[2] = t; o_scl3(dir, dir, -1); }
    int ok = 0;
    for (int it = 0; it < MPR_ITERATIONS; it++) {
      mpr_support(&A, &B, dir, &v4);
      if (o_dot3(v4.v, dir) <= 0) return 0;
      int cont = 0;
      o_cross(va, p[1].v, v4.v);
      if (o_dot3(va, p[0].v) < 0) { p[2] = v4; cont = 1; }
      if (!cont) {
        o_cross(va, v4.v, p[2].v);
        if (o_dot3(va, p[0].v) < 0) { p[1] = v4; cont = 1; }
      }
      if (!cont) { p[3] = v4; ok = 1; break; }
      o_sub3(va, p[1].v, p[0].v); o_sub3(vb, p[2].v, p[0].v);
      o_cross(dir, va, vb); o_normalize3(dir);
    }
    if (!ok) return 0;
  }
  if (found == 2) {
    depth = o_norm3(p[1].v);
    o_copy3(nrm, p[1].v); o_normalize3(nrm);
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (p[1].v1[k] + p[1].v2[k]);
  } else {
    /* refinement: does the portal enclose the origin? */
    int hit = 0;
    for (int it = 0; it < MPR_ITERATIONS; it++) {
      mpr_portal_dir(p, dir);
      if (o_dot3(dir, p[1].v) >= 0) { hit = 1; break; }
      mpr_support(&A, &B, dir, &v4);
      if (o_dot3(v4.v, dir) < 0 || mpr_reach_tolerance(p, &v4, dir)) return 0;
      mpr_expand_portal(p, &v4);
    }
    if (!hit) return 0;
    /* penetration: push the portal to the surface */
    for (int it = 0; ; it++) {
      mpr_portal_dir(p, dir);
      mpr_support(&A, &B, dir, &v4);
      if (mpr_reach_tolerance(p, &v4, dir) || it >= MPR_ITERATIONS) {
        /* the portal point closest to the origin: penetration vector; its barycentric weights applied to the support points of
         * the two geoms give the two witness points, the contact sits half way between them */
        double w[3], bw[3];
        mpr_closest_on_triangle(p[1].v, p[2].v, p[3].v, w, bw);
        depth = o_norm3(w);
        if (depth < O_MINVAL) o_copy3(nrm, dir); else o_scl3(nrm, w, 1.0 / depth);
        for (int k = 0; k < 3; k++)
          pos[k] = 0.5 * (bw[0] * (p[1].v1[k] + p[1].v2[k]) + bw[1] * (p[2].v1[k] + p[2].v2[k]) + bw[2] * (p[3].v1[k] + p[3].v2[k]));
        break;
      }
      mpr_expand_portal(p, &v4);
    }
  }
  double dist = margin - depth;        /* both geoms were inflated by margin / 2 */
  if (dist > margin) return 0;
  o_zero(con->frame, 9);
  o_copy3(con->frame, nrm);
  con->dist = dist;
  o_copy3(con->pos, pos);
  return 1;
}

/* height field (geom 1) against a convex geom (MuJoCo: mjc_ConvexHField): the cells under the geom's bounding sphere are cut
 * into two triangular prisms each (top = the terrain triangle, bottom at -base) and every prism meets the geom through the
 * portal-refinement collider.  The 4 deepest contacts of a pair are kept (MuJoCo keeps up to 50: a deliberate cap, the contacts
 * dropped are the shallow duplicates neighbouring prisms report around the same touching point); more than HF_MAXCELL cells
 * under the geom raise the unsupported flag (the candidate fails loudly). */
#define HF_MAXCELL 100
static int hfield_convex(OContact *con, double margin, const double *hp, const double *hm, const double *hsize, int nrow, int ncol,
                         const double *data, MShape B, double rbound, int *overflow) {
  double dif[3], c[3];
  o_sub3(dif, B.pos, hp);
  o_mulmattvec3(c, hm, dif);
  double r = rbound + margin, rx = hsize[0], ry = hsize[1], elev = hsize[2], base = hsize[3];
  if (c[0] + r < -rx || c[0] - r > rx || c[1] + r < -ry || c[1] - r > ry || c[2] - r > elev || c[2] + r < -base) return 0;
  double dx = 2 * rx / (ncol - 1), dy = 2 * ry / (nrow - 1);
  int cmin = (int)floor((c[0] - r + rx) / dx), cmax = (int)floor((c[0] + r + rx) / dx);
  int rmin = (int)floor((c[1] - r + ry) / dy), rmax = (int)floor((c[1] + r + ry) / dy);
  if (cmin < 0) cmin = 0;
  if (rmin < 0) rmin = 0;
  if (cmax > ncol - 2) cmax = ncol - 2;
  if (rmax > nrow - 2) rmax = nrow - 2;
  if ((cmax - cmin + 1) * (rmax - rmin + 1) > HF_MAXCELL) { *overflow = 1; return 0; }
  int cnt = 0;
  for (int row = rmin; row <= rmax; row++) for (int col = cmin; col <= cmax; col++) {
    double x0 = -rx + col * dx, x1 = x0 + dx, y0 = -ry + row * dy, y1 = y0 + dy;
    double h00 = data[row * ncol + col] * elev, h01 = data[row * ncol + col + 1] * elev;
    double h10 = data[(row + 1) * ncol + col] * elev, h11 = data[(row + 1) * ncol + col + 1] * elev;
    if (fmax(fmax(h00, h01), fmax(h10, h11)) < c[2] - r) continue;      /* the geom is entirely above this cell */
    for (int tri = 0; tri < 2; tri++) {
      /* triangle 0: (x0,y0) (x1,y0) (x1,y1); triangle 1: (x0,y0) (x1,y1) (x0,y1) */
      double tx[3] = {x0, x1, tri == 0 ? x1 : x0}, ty[3] = {y0, tri == 0 ? y0 : y1, y1}, th[3] = {h00, tri == 0 ? h01 : h11, tri == 0 ? h11 : h10};
      double v[18], cen[3] = {0, 0, 0};
      for (int k = 0; k < 3; k++) {
        v[3 * k] = tx[k]; v[3 * k + 1] = ty[k]; v[3 * k + 2] = th[k];
        v[9 + 3 * k] = tx[k]; v[9 + 3 * k + 1] = ty[k]; v[9 + 3 * k + 2] = -base;
      }
      for (int k = 0; k < 6; k++) { cen[0] += v[3 * k]; cen[1] += v[3 * k + 1]; cen[2] += v[3 * k + 2]; }
      for (int k = 0; k < 3; k++) cen[k] *= 1.0 / 6.0;
      for (int k = 0; k < 6; k++) { v[3 * k] -= cen[0]; v[3 * k + 1] -= cen[1]; v[3 * k + 2] -= cen[2]; }
      double pp[3];
      o_mulmatvec3(pp, hm, cen);
      o_add3(pp, pp, hp);
      MShape P = {MJPC_GEOM_MESH, pp, hm, hsize, 0.5 * margin, v, 6};
      OContact t;
      if (convex_mpr_shapes(&t, margin, P, B)) {
        if (cnt < 4) con[cnt++] = t;
        else {        /* full: the new contact replaces the shallowest kept one if it is deeper (first such slot) */
          int w = 0;
          for (int k = 1; k < 4; k++) if (con[k].dist > con[w].dist) w = k;
          if (t.dist < con[w].dist) con[w] = t;
        }
      }
    }
  }
  return cnt;
}

/* plane against an ellipsoid: the ellipsoid's support point against the plane normal (mjc_PlaneConvex's construction) */
static int plane_convex(OContact *con, double margin, const double *pp, const double *pm, MShape E) {
  double n[3] = {pm[2], pm[5], pm[8]}, nd[3] = {-pm[2], -pm[5], -pm[8]}, sp[3], dif[3];
  E.margin = 0;
  mpr_support1(&E, nd, sp);
  o_sub3(dif, sp, pp);
  double dist = o_dot3(dif, n);
  if (dist > margin) return 0;
  o_zero(con->frame, 9);
  o_copy3(con->frame, n);
  con->dist = dist;
  o_addscl3(con->pos, sp, n, -0.5 * dist);
  return 1;
}

int oracle_collide_pair(const OModel *om, OData *d, int g1, int g2, double margin, OContact *con, int *unsupported) {
  const MjpcHipModel *m = &om->m;
  int t1 = m->geom_type[g1], t2 = m->geom_type[g2];
  const double *p1 = d->geom_xpos + 3 * g1, *p2 = d->geom_xpos + 3 * g2;
  const double *m1 = d->geom_xmat + 9 * g1, *m2 = d->geom_xmat + 9 * g2;
  const double *s1 = m->geom_size + 3 * g1, *s2 = m->geom_size + 3 * g2;
  /* pairs are stored with type1 <= type2 */
  /* cheap conservative pre-tests for the expensive pair types ("certainly apart" is exact): the cylinder's bounding capsule,
   * the box's bounding sphere against the capsule's segment */
  if (t2 == MJPC_GEOM_CYLINDER && (t1 == MJPC_GEOM_SPHERE || t1 == MJPC_GEOM_CAPSULE)) {
    OContact tmp[4];
    int nb = t1 == MJPC_GEOM_SPHERE ? sphere_capsule(tmp, margin, p1, s1[0], p2, m2, s2) : capsule_capsule(tmp, margin, p1, m1, s1, p2, m2, s2);
    if (nb == 0) return 0;
  } else if (t2 == MJPC_GEOM_BOX && (t1 == MJPC_GEOM_CAPSULE || t1 == MJPC_GEOM_CYLINDER)) {
    double a1[3] = {m1[2], m1[5], m1[8]};
    double r = s1[0] + m->geom_rbound[g2] + margin;
    if (seg_point_dist2(p1, a1, s1[1], p2) > r * r) return 0;
    double dif[3], q[3], al[3];
    o_sub3(dif, p1, p2);
    o_mulmattvec3(q, m2, dif);
    o_mulmattvec3(al, m2, a1);
    double rr = s1[0] + margin;                  /* the box's three face normals as separating axes */
    if (fabs(q[0]) - s1[1] * fabs(al[0]) > s2[0] + rr || fabs(q[1]) - s1[1] * fabs(al[1]) > s2[1] + rr ||
        fabs(q[2]) - s1[1] * fabs(al[2]) > s2[2] + rr) return 0;
    /* ... and the three axes  segment direction x box axis */
    double l0 = sqrt(al[1] * al[1] + al[2] * al[2]), l1 = sqrt(al[0] * al[0] + al[2] * al[2]), l2 = sqrt(al[0] * al[0] + al[1] * al[1]);
    if (fabs(q[2] * al[1] - q[1] * al[2]) > s2[1] * fabs(al[2]) + s2[2] * fabs(al[1]) + rr * l0) return 0;
    if (fabs(q[0] * al[2] - q[2] * al[0]) > s2[0] * fabs(al[2]) + s2[2] * fabs(al[0]) + rr * l1) return 0;
    if (fabs(q[1] * al[0] - q[0] * al[1]) > s2[0] * fabs(al[1]) + s2[1] * fabs(al[0]) + rr * l2) return 0;
  }
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
      case MJPC_GEOM_CYLINDER: return sphere_cylinder(con, margin, p1, s1[0], p2, m2, s2);
      default: break;
    }
  } else if (t1 == MJPC_GEOM_CAPSULE && t2 == MJPC_GEOM_CAPSULE) {
    return capsule_capsule(con, margin, p1, m1, s1, p2, m2, s2);
  } else if (t1 == MJPC_GEOM_CAPSULE && t2 == MJPC_GEOM_BOX) {
    return capsule_box(con, margin, p1, m1, s1, p2, m2, s2);
  } else if (t1 == MJPC_GEOM_CAPSULE && t2 == MJPC_GEOM_CYLINDER) {
    return capsule_cylinder(con, margin, p1, m1, s1, p2, m2, s2);
  } else if (t1 == MJPC_GEOM_BOX && t2 == MJPC_GEOM_BOX) {
    return box_box(con, margin, p1, m1, s1, p2, m2, s2);
  }
  /* height field against anything convex */
  if (t1 == MJPC_GEOM_HFIELD && t2 >= MJPC_GEOM_SPHERE) {
    MShape B = {t2, p2, m2, s2, 0.5 * margin, 0, 0};
    if (t2 == MJPC_GEOM_MESH) { int k = m->geom_dataid[g2]; B.vert = m->mesh_vert + 3 * m->mesh_vertadr[k]; B.nvert = m->mesh_vertnum[k]; }
    int h = m->geom_dataid[g1], over = 0;
    int n = hfield_convex(con, margin, p1, m1, m->hfield_size + 4 * h, m->hfield_nrow[h], m->hfield_ncol[h], m->hfield_data + m->hfield_adr[h], B,
                          m->geom_rbound[g2], &over);
    if (over) (*unsupported)++;
    return n;
  }
  /* ellipsoids and convex meshes: support point against a plane, the portal-refinement collider against everything else */
  if ((t1 == MJPC_GEOM_ELLIPSOID || t2 == MJPC_GEOM_ELLIPSOID || t1 == MJPC_GEOM_MESH || t2 == MJPC_GEOM_MESH) && t1 != MJPC_GEOM_HFIELD && t2 != MJPC_GEOM_HFIELD) {
    MShape A = {t1, p1, m1, s1, 0.5 * margin, 0, 0}, B = {t2, p2, m2, s2, 0.5 * margin, 0, 0};
    if (t1 == MJPC_GEOM_MESH) { int k = m->geom_dataid[g1]; A.vert = m->mesh_vert + 3 * m->mesh_vertadr[k]; A.nvert = m->mesh_vertnum[k]; }
    if (t2 == MJPC_GEOM_MESH) { int k = m->geom_dataid[g2]; B.vert = m->mesh_vert + 3 * m->mesh_vertadr[k]; B.nvert = m->mesh_vertnum[k]; }
    if (t1 == MJPC_GEOM_PLANE) return plane_convex(con, margin, p1, m1, B);
    return convex_mpr_shapes(con, margin, A, B);
  }
  /* cylinder-cylinder and cylinder-box: the cylinder's bounding capsule (same radius and half length) contains it, so a capsule
   * farther than the margin means certainly no contact (exact, cheap); otherwise the portal-refinement collider decides */
  if (t1 == MJPC_GEOM_CYLINDER && (t2 == MJPC_GEOM_CYLINDER || t2 == MJPC_GEOM_BOX)) {
    OContact tmp[4];
    int n = t2 == MJPC_GEOM_CYLINDER ? capsule_capsule(tmp, margin, p1, m1, s1, p2, m2, s2) : capsule_box(tmp, margin, p1, m1, s1, p2, m2, s2);
    if (n == 0) return 0;
    return convex_mpr(con, margin, t1, p1, m1, s1, t2, p2, m2, s2);
  }
  (*unsupported)++;
  return 0;
}

/* test access to the primitive colliders (tests/test_oracle_physics.py): geoms given directly, type1 <= type2;
 * out[k] = dist, pos[3], normal[3] */
int oracle_debug_collide(int t1, const double *s1, const double *p1, const double *m1, int t2, const double *s2, const double *p2,
                         const double *m2, double margin, double *out) {
  OContact con[8];
  int n = -1;
  if (t1 == MJPC_GEOM_SPHERE && t2 == MJPC_GEOM_BOX) n = sphere_box(con, margin, p1, s1[0], p2, m2, s2);
  else if (t1 == MJPC_GEOM_CAPSULE && t2 == MJPC_GEOM_BOX) n = capsule_box(con, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == MJPC_GEOM_BOX && t2 == MJPC_GEOM_BOX) n = box_box(con, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == MJPC_GEOM_CAPSULE && t2 == MJPC_GEOM_CAPSULE) n = capsule_capsule(con, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == MJPC_GEOM_SPHERE && t2 == MJPC_GEOM_CYLINDER) n = sphere_cylinder(con, margin, p1, s1[0], p2, m2, s2);
  else if (t1 == MJPC_GEOM_CAPSULE && t2 == MJPC_GEOM_CYLINDER) n = capsule_cylinder(con, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == MJPC_GEOM_PLANE && t2 == MJPC_GEOM_ELLIPSOID) { MShape E = {t2, p2, m2, s2, 0, 0, 0}; n = plane_convex(con, margin, p1, m1, E); }
  else if (t1 == MJPC_GEOM_CYLINDER || t1 == MJPC_GEOM_ELLIPSOID || t2 == MJPC_GEOM_ELLIPSOID) n = convex_mpr(con, margin, t1, p1, m1, s1, t2, p2, m2, s2);
  else if (t1 >= 100) n = convex_mpr(con, margin, t1 - 100, p1, m1, s1, t2, p2, m2, s2);   /* any supported pair through the portal collider */
  for (int k = 0; k < n; k++) {
    out[7 * k] = con[k].dist;
    o_copy3(out + 7 * k + 1, con[k].pos);
    o_copy3(out + 7 * k + 4, con[k].frame);
  }
  return n;
}

/* test access with convex meshes: vertices given directly (nvert = 0: not a mesh); plane-X or the portal collider */
int oracle_debug_collide_mesh(int t1, const double *s1, const double *p1, const double *m1, const double *v1, int n1,
                              int t2, const double *s2, const double *p2, const double *m2, const double *v2, int n2, double margin, double *out) {
  OContact con[2];
  memset(con, 0, sizeof(con));
  MShape A = {t1, p1, m1, s1, 0.5 * margin, v1, n1}, B = {t2, p2, m2, s2, 0.5 * margin, v2, n2};
  int n = t1 == MJPC_GEOM_PLANE ? plane_convex(con, margin, p1, m1, B) : convex_mpr_shapes(con, margin, A, B);
  for (int k = 0; k < n; k++) { out[7 * k] = con[k].dist; o_copy3(out + 7 * k + 1, con[k].pos); o_copy3(out + 7 * k + 4, con[k].frame); }
  return n;
}

/* test access: a height field (size[4], nrow x ncol data) against a primitive */
int oracle_debug_collide_hfield(const double *hsize, int nrow, int ncol, const double *data, const double *hp, const double *hm,
                                int t2, const double *s2, const double *p2, const double *m2, double rbound, double margin, double *out) {
  OContact con[4];
  memset(con, 0, sizeof(con));
  MShape B = {t2, p2, m2, s2, 0.5 * margin, 0, 0};
  int over = 0;
  int n = hfield_convex(con, margin, hp, hm, hsize, nrow, ncol, data, B, rbound, &over);
  for (int k = 0; k < n; k++) { out[7 * k] = con[k].dist; o_copy3(out + 7 * k + 1, con[k].pos); o_copy3(out + 7 * k + 4, con[k].frame); }
  return over ? -1 : n;
}
