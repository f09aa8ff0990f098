// collide.h — collision detection: bounding-sphere filter, ordered compaction, analytic colliders, portal-refinement collider (part of core.h)
// Included by core.h only, in this order: the files share one translation unit and its macros.
#pragma once
// ======================================================================================
// collision: bounding-sphere filter -> ordered compaction -> analytic narrow phase
// ======================================================================================
struct NPCon { double dist, pos[3], frame[6]; };
// The per-lane contact list (at most 4) must stay in registers: a run-time index would send the whole array to scratch
// memory, so slots are written / read through compile-time indices and value selects (no loops: the indices must be
// constants before the first SROA run).
DEV void np_sel(NPCon &d, const NPCon &v, bool p) {
  d.dist = p ? v.dist : d.dist;
  d.pos[0] = p ? v.pos[0] : d.pos[0]; d.pos[1] = p ? v.pos[1] : d.pos[1]; d.pos[2] = p ? v.pos[2] : d.pos[2];
  d.frame[0] = p ? v.frame[0] : d.frame[0]; d.frame[1] = p ? v.frame[1] : d.frame[1]; d.frame[2] = p ? v.frame[2] : d.frame[2];
  d.frame[3] = p ? v.frame[3] : d.frame[3]; d.frame[4] = p ? v.frame[4] : d.frame[4]; d.frame[5] = p ? v.frame[5] : d.frame[5];
}
DEV void np_put(NPCon *con, int idx, const NPCon &v) {
  np_sel(con[0], v, idx == 0); np_sel(con[1], v, idx == 1); np_sel(con[2], v, idx == 2); np_sel(con[3], v, idx == 3);
}
DEV NPCon np_get(const NPCon *con, int idx) {
  NPCon v = con[0];
  np_sel(v, con[1], idx == 1); np_sel(v, con[2], idx == 2); np_sel(v, con[3], idx == 3);
  return v;
}

DEV int np_sphere_sphere(NPCon *con, double margin, const double *p1, double r1, const double *p2, double r2) {
  double dif[3];
  d_sub3(dif, p2, p1);
  double cdist = d_norm3(dif), dist = cdist - r1 - r2;
  if (dist > margin) return 0;
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  if (cdist < D_MINVAL) con->frame[0] = 1; else d_scl3(con->frame, dif, 1.0 / cdist);
  con->dist = dist;
  d_addscl3(con->pos, p1, con->frame, r1 + 0.5 * dist);
  return 1;
}
DEV int np_plane_sphere(NPCon *con, double margin, const double *pp, const double *n, const double *cen, double r) {
  double dif[3];
  d_sub3(dif, cen, pp);
  double dist = d_dot3(dif, n) - r;
  if (dist > margin) return 0;
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  d_copy3(con->frame, n);
  con->dist = dist;
  d_addscl3(con->pos, cen, n, -(r + 0.5 * dist));
  return 1;
}
DEV int np_plane_capsule(NPCon *con, double margin, const double *pp, const double *pm, const double *cp, const double *cm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, axis[3] = {cm[2], cm[5], cm[8]}, seg[3], e[3];
  int cnt = 0;
  d_scl3(seg, axis, size[1]);
  d_add3(e, cp, seg);
  NPCon t;
  if (np_plane_sphere(&t, margin, pp, n, e, size[0])) { d_copy3(t.frame + 3, axis); con[0] = t; cnt++; }
  d_sub3(e, cp, seg);
  if (np_plane_sphere(&t, margin, pp, n, e, size[0])) { d_copy3(t.frame + 3, axis); np_put(con, cnt, t); cnt++; }
  return cnt;
}
DEV int np_plane_box(NPCon *con, double margin, const double *pp, const double *pm, const double *bp, const double *bm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, dif[3];
  d_sub3(dif, bp, pp);
  double dist = d_dot3(dif, n);
  int cnt = 0;
  for (int i = 0; i < 8; i++) {
    double vec[3] = {(i & 1) ? size[0] : -size[0], (i & 2) ? size[1] : -size[1], (i & 4) ? size[2] : -size[2]};
    double corner[3];
    d_mulmatvec3(corner, bm, vec);
    double ldist = d_dot3(n, corner);
    if (dist + ldist > margin || ldist > 0) continue;
    if (cnt >= 4) break;
    NPCon t, *q = &t;
    q->dist = dist + ldist;
    for (int k = 0; k < 6; k++) q->frame[k] = 0;
    d_copy3(q->frame, n);
    d_add3(corner, corner, bp);
    d_addscl3(q->pos, corner, n, -0.5 * q->dist);
    np_put(con, cnt, t);
    cnt++;
    if (cnt >= 4) break;
  }
  return cnt;
}
DEV int np_plane_cylinder(NPCon *con, double margin, const double *pp, const double *pm, const double *cp, const double *cm, const double *size) {
  double n[3] = {pm[2], pm[5], pm[8]}, axis[3] = {cm[2], cm[5], cm[8]};
  double prjaxis = d_dot3(n, axis);
  if (prjaxis > 0) { d_scl3(axis, axis, -1); prjaxis = -prjaxis; }
  double vec[3];
  d_sub3(vec, cp, pp);
  double dist0 = d_dot3(vec, n);
  d_scl3(vec, axis, prjaxis); d_sub3(vec, vec, n);
  double len2 = d_dot3(vec, vec);
  if (len2 >= D_MINVAL) d_scl3(vec, vec, size[0] / sqrt(len2));
  else { vec[0] = cm[0] * size[0]; vec[1] = cm[3] * size[0]; vec[2] = cm[6] * size[0]; }
  double prjvec = d_dot3(vec, n);
  d_scl3(axis, axis, size[1]); prjaxis *= size[1];
  int cnt = 0;
  if (dist0 + prjaxis + prjvec <= margin) {
    NPCon *q = con; cnt = 1;
    q->dist = dist0 + prjaxis + prjvec;
    d_add3(q->pos, cp, vec); d_add3(q->pos, q->pos, axis); d_addtoscl3(q->pos, n, -0.5 * q->dist);
    for (int k = 0; k < 6; k++) q->frame[k] = 0;
    d_copy3(q->frame, n);
  } else return 0;
  if (dist0 - prjaxis + prjvec <= margin) {
    NPCon *q = con + 1; cnt = 2;
    q->dist = dist0 - prjaxis + prjvec;
    d_add3(q->pos, cp, vec); d_sub3(q->pos, q->pos, axis); d_addtoscl3(q->pos, n, -0.5 * q->dist);
    for (int k = 0; k < 6; k++) q->frame[k] = 0;
    d_copy3(q->frame, n);
  }
  double prjvec1 = -0.5 * prjvec;
  if (dist0 + prjaxis + prjvec1 <= margin) {
    double vec1[3];
    d_cross(vec1, vec, axis);
    d_normalize3(vec1);
    d_scl3(vec1, vec1, size[0] * sqrt(3.0) / 2);
    for (int s = -1; s <= 1; s += 2) {
      NPCon t, *q = &t;
      q->dist = dist0 + prjaxis + prjvec1;
      d_add3(q->pos, cp, axis); d_addtoscl3(q->pos, vec, -0.5); d_addtoscl3(q->pos, vec1, (double)s);
      d_addtoscl3(q->pos, n, -0.5 * q->dist);
      for (int k = 0; k < 6; k++) q->frame[k] = 0;
      d_copy3(q->frame, n);
      np_put(con, cnt, t);
      cnt++;
    }
  }
  return cnt;
}
DEV int np_sphere_capsule(NPCon *con, double margin, const double *sp, double sr, const double *cp, const double *cm, const double *csize) {
  double axis[3] = {cm[2], cm[5], cm[8]}, vec[3], pt[3];
  d_sub3(vec, sp, cp);
  double x = d_clip(d_dot3(axis, vec), -csize[1], csize[1]);
  d_addscl3(pt, cp, axis, x);
  return np_sphere_sphere(con, margin, sp, sr, pt, csize[0]);
}
DEV int np_capsule_capsule(NPCon *con, double margin, const double *p1, const double *m1, const double *s1,
                           const double *p2, const double *m2, const double *s2) {
  double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]}, dif[3];
  d_sub3(dif, p1, p2);
  double len1 = s1[1], len2 = s2[1];
  double ma = d_dot3(a1, a1), mb = -d_dot3(a1, a2), mc = d_dot3(a2, a2);
  double u = -d_dot3(a1, dif), v = d_dot3(a2, dif);
  double det = ma * mc - mb * mb;
  if (fabs(det) >= D_MINVAL) {
    double x1 = (mc * u - mb * v) / det, x2 = (ma * v - mb * u) / det;
    if (x1 > len1) { x1 = len1; x2 = (v - mb * len1) / mc; }
    else if (x1 < -len1) { x1 = -len1; x2 = (v + mb * len1) / mc; }
    if (x2 > len2) { x2 = len2; x1 = d_clip((u - mb * len2) / ma, -len1, len1); }
    else if (x2 < -len2) { x2 = -len2; x1 = d_clip((u + mb * len2) / ma, -len1, len1); }
    double v1[3], v2[3];
    d_addscl3(v1, p1, a1, x1); d_addscl3(v2, p2, a2, x2);
    return np_sphere_sphere(con, margin, v1, s1[0], v2, s2[0]);
  }
  int cnt = 0;
  for (int s = -1; s <= 1 && cnt < 2; s += 2) {
    double e[3], w[3], pt[3];
    d_addscl3(e, p1, a1, s * len1);
    d_sub3(w, e, p2);
    double x = d_clip(d_dot3(a2, w), -len2, len2);
    d_addscl3(pt, p2, a2, x);
    NPCon t;
    if (np_sphere_sphere(&t, margin, e, s1[0], pt, s2[0])) { np_put(con, cnt, t); cnt++; }
  }
  return cnt;
}
DEV int np_sphere_box(NPCon *con, double margin, const double *sp, double sr, const double *bp, const double *bm, const double *bs) {
  double dif[3], cc[3], clamped[3];
  d_sub3(dif, sp, bp);
  d_mulmattvec3(cc, bm, dif);
  int inside = 1;
  for (int i = 0; i < 3; i++) { clamped[i] = d_clip(cc[i], -bs[i], bs[i]); if (clamped[i] != cc[i]) inside = 0; }
  double nloc[3], dist;
  if (!inside) {
    double dd[3]; d_sub3(dd, cc, clamped);
    double len = d_norm3(dd);
    dist = len - sr;
    if (dist > margin) return 0;
    d_scl3(nloc, dd, 1.0 / len);
  } else {
    int k = 0; double best = 1e300;
    for (int i = 0; i < 3; i++) { double pen = bs[i] - fabs(cc[i]); if (pen < best) { best = pen; k = i; } }
    nloc[0] = nloc[1] = nloc[2] = 0;
    double sgn = cc[k] >= 0 ? 1.0 : -1.0;
    if (k == 0) nloc[0] = sgn; else if (k == 1) nloc[1] = sgn; else nloc[2] = sgn;
    d_copy3(clamped, cc);
    if (k == 0) clamped[0] = sgn * bs[0]; else if (k == 1) clamped[1] = sgn * bs[1]; else clamped[2] = sgn * bs[2];
    dist = -best - sr;
  }
  double nw[3], surf[3];
  d_mulmatvec3(nw, bm, nloc);
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  d_scl3(con->frame, nw, -1);
  con->dist = dist;
  d_mulmatvec3(surf, bm, clamped); d_add3(surf, surf, bp);
  d_addscl3(con->pos, surf, nw, 0.5 * dist);
  return 1;
}

// capsule (geom1) vs box (geom2): closest point of the segment to the box = exact root of the monotone, piecewise-linear derivative
// (the CPU checker restates the same construction with the same operations), then sphere-box there and at the far end cap
DEV double capsule_box_g(const double *p0, const double *a, double h, const double *b, double s) {
  double g = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double q = p0[i] + (s * h) * a[i];
    double e = fabs(q) - b[i];
    g += (e > 0) ? (q > 0 ? e : -e) * a[i] : 0.0;
  }
  return g;
}
DEV int np_capsule_box(NPCon *con, double margin, const double *cp, const double *cm, const double *cs,
                       const double *bp, const double *bm, const double *bs) {
  double axis[3] = {cm[2], cm[5], cm[8]}, dif[3], p0[3], a[3];
  d_sub3(dif, cp, bp);
  d_mulmattvec3(p0, bm, dif);
  d_mulmattvec3(a, bm, axis);
  double h = cs[1], sstar;
  if (capsule_box_g(p0, a, h, bs, -1.0) >= 0) sstar = -1.0;
  else if (capsule_box_g(p0, a, h, bs, 1.0) <= 0) sstar = 1.0;
  else {
    // g is piecewise linear with breakpoints where a coordinate crosses a face (|q_i| = b_i): the bracket is narrowed at the
    // (at most six) breakpoints inside it, then g is linear on what is left and its root is exact
    double lo = -1.0, hi = 1.0, glo = capsule_box_g(p0, a, h, bs, -1.0), ghi = capsule_box_g(p0, a, h, bs, 1.0);
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
      for (int sg = -1; sg <= 1; sg += 2) {
        double den = h * a[i];
        if (fabs(den) < D_MINVAL) continue;
        double t = (sg * bs[i] - p0[i]) / den;
        if (!(t > lo && t < hi)) continue;
        double gt = capsule_box_g(p0, a, h, bs, t);
        if (gt < 0) { lo = t; glo = gt; } else { hi = t; ghi = gt; }
      }
    }
    // (a numerically flat piece is the zero-distance stretch of a segment that passes through the box: its left end, like the
    // leftmost point with g >= 0 everywhere else)
    double dg = ghi - glo;
    sstar = dg > 1e-15 ? lo - glo * (hi - lo) / dg : lo;
  }
  int cnt = 0;
  double pt[3];
  NPCon t;
  d_addscl3(pt, cp, axis, sstar * h);
  if (np_sphere_box(&t, margin, pt, cs[0], bp, bm, bs)) { con[0] = t; cnt++; }
  double s2 = sstar <= 0 ? 1.0 : -1.0;
  d_addscl3(pt, cp, axis, s2 * h);
  if (np_sphere_box(&t, margin, pt, cs[0], bp, bm, bs)) { np_put(con, cnt, t); cnt++; }
  return cnt;
}

// box (geom1 = A) vs box (geom2 = B): separating-axis test, then reference-face clipping (<= 4 contacts) or one edge-edge
// contact (DESIGN.md section 6 describes the construction).  Axis-indexed accesses go through selects so that
// nothing needs a run-time-indexed private array.
#define BB_TOL 1e-9
// a later candidate replaces an earlier one only if it is better by more than this: near-ties (a cube lying flat: several corners
// at the same depth to rounding) resolve to the lowest candidate index instead of flipping with the last bit of the pose
#define BB_TIE_D 1e-10
#define BB_TIE_A 1e-12
DEV double sel3(const double *v, int i) { return i == 0 ? v[0] : (i == 1 ? v[1] : v[2]); }
DEV void col3(double *r, const double *m, int k) { r[0] = k == 0 ? m[0] : (k == 1 ? m[1] : m[2]); r[1] = k == 0 ? m[3] : (k == 1 ? m[4] : m[5]); r[2] = k == 0 ? m[6] : (k == 1 ? m[7] : m[8]); }
struct BBSel { double x, y, d; int ok; };
// candidate q of the face case: 0-3 incident vertices, 4-7 reference corners, 8-23 edge crossings
struct BBFace { double c0[3], e1[3], e2[3], hu, hv, det; };
DEV BBSel bb_candidate(const BBFace &f, int q) {
  BBSel o; o.ok = 0; o.x = 0; o.y = 0; o.d = 0;
  if (q < 4) {
    double s1 = (q == 0 || q == 3) ? -1.0 : 1.0, s2 = (q < 2) ? -1.0 : 1.0;
    double x = f.c0[0] + s1 * f.e1[0] + s2 * f.e2[0], y = f.c0[1] + s1 * f.e1[1] + s2 * f.e2[1], d = f.c0[2] + s1 * f.e1[2] + s2 * f.e2[2];
    if (fabs(x) <= f.hu + BB_TOL && fabs(y) <= f.hv + BB_TOL) { o.ok = 1; o.x = x; o.y = y; o.d = d; }
  } else if (q < 8) {
    int k = q - 4;
    if (fabs(f.det) > 1e-14) {
      double cx = (k == 0 || k == 3) ? -f.hu : f.hu, cy = (k < 2) ? -f.hv : f.hv;
      double dx = cx - f.c0[0], dy = cy - f.c0[1];
      double al = (dx * f.e2[1] - dy * f.e2[0]) / f.det, be = (f.e1[0] * dy - f.e1[1] * dx) / f.det;
      if (fabs(al) <= 1.0 + BB_TOL && fabs(be) <= 1.0 + BB_TOL) { o.ok = 1; o.x = cx; o.y = cy; o.d = f.c0[2] + al * f.e1[2] + be * f.e2[2]; }
    }
  } else {
    int k = (q - 8) >> 2, e = (q - 8) & 3, k2 = (k + 1) & 3;
    double a1 = (k == 0 || k == 3) ? -1.0 : 1.0, a2 = (k < 2) ? -1.0 : 1.0, b1 = (k2 == 0 || k2 == 3) ? -1.0 : 1.0, b2 = (k2 < 2) ? -1.0 : 1.0;
    double px = f.c0[0] + a1 * f.e1[0] + a2 * f.e2[0], py = f.c0[1] + a1 * f.e1[1] + a2 * f.e2[1], pd = f.c0[2] + a1 * f.e1[2] + a2 * f.e2[2];
    double qx = f.c0[0] + b1 * f.e1[0] + b2 * f.e2[0], qy = f.c0[1] + b1 * f.e1[1] + b2 * f.e2[1], qd = f.c0[2] + b1 * f.e1[2] + b2 * f.e2[2];
    double dx = qx - px, dy = qy - py, dd = qd - pd;
    int xline = e < 2;
    double lim = (e & 1) ? 1.0 : -1.0;
    double num = xline ? lim * f.hu - px : lim * f.hv - py, den = xline ? dx : dy;
    if (!(fabs(den) < 1e-14)) {
      double s = num / den;
      if (!(s <= 0.0 || s >= 1.0)) {
        double ox = xline ? lim * f.hu : px + s * dx, oy = xline ? py + s * dy : lim * f.hv;
        if (!((xline ? fabs(oy) - f.hv : fabs(ox) - f.hu) > BB_TOL)) { o.ok = 1; o.x = ox; o.y = oy; o.d = pd + s * dd; }
      }
    }
  }
  return o;
}
// the face case of a box pair, set up by the pair's lane and finished by the whole wave (bb_face_select): the 24 candidate points
// are independent, one lane each; only the (order-dependent) selection among them is serial
struct BBPend { BBFace f; double pr[3], ru[3], rv[3], n[3], hn; int refA; };
// returns the number of contacts, or -3: face case, `pend` filled (bb_face_select / bb_face_finish complete it)
DEV int np_box_box(NPCon *con, double margin, const double *pa, const double *ma, const double *sa,
                   const double *pb, const double *mb, const double *sb, BBPend &pend) {
  double R[9], AR[9], t[3], tb[3], dif[3];
  d_sub3(dif, pb, pa);
  d_mulmattvec3(t, ma, dif);
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      double r = ma[i] * mb[j] + ma[3 + i] * mb[3 + j] + ma[6 + i] * mb[6 + j];
      R[3 * i + j] = r; AR[3 * i + j] = fabs(r);
    }
#pragma unroll
  for (int j = 0; j < 3; j++) tb[j] = t[0] * R[j] + t[1] * R[3 + j] + t[2] * R[6 + j];
  double best = -1e300; int code = -1, sep = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    double s = fabs(t[i]) - (sa[i] + sb[0] * AR[3 * i] + sb[1] * AR[3 * i + 1] + sb[2] * AR[3 * i + 2]);
    sep |= s > margin;
    if (s > best) { best = s; code = i; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++) {
    double s = fabs(tb[j]) - (sb[j] + sa[0] * AR[j] + sa[1] * AR[3 + j] + sa[2] * AR[6 + j]);
    sep |= s > margin;
    if (s > best) { best = s; code = 3 + j; }
  }
  if (sep) return 0;                 // separated along a face normal: most non-touching pairs leave here
  double ebest = -1e300; int ecode = -1;
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      double l2 = 1.0 - R[3 * i + j] * R[3 * i + j];
      if (l2 < 1e-6) continue;
      double proj = t[i2] * R[3 * i1 + j] - t[i1] * R[3 * i2 + j];
      double ra = sa[i1] * AR[3 * i2 + j] + sa[i2] * AR[3 * i1 + j];
      double rb = sb[j1] * AR[3 * i + j2] + sb[j2] * AR[3 * i + j1];
      double s = (fabs(proj) - (ra + rb)) / sqrt(l2);
      sep |= s > margin;
      if (s > ebest) { ebest = s; ecode = 3 * i + j; }
    }
  if (sep) return 0;
  if (ecode >= 0 && ebest > best + 0.05 * fabs(best) + BB_TOL) {
    int i = ecode / 3, j = ecode - 3 * i;
    double ai[3], bj[3], n[3];
    col3(ai, ma, i); col3(bj, mb, j);
    d_cross(n, ai, bj);
    d_normalize3(n);
    if (d_dot3(n, dif) < 0) d_scl3(n, n, -1);
    double ea[3], eb[3];
    d_copy3(ea, pa); d_copy3(eb, pb);
#pragma unroll
    for (int k = 0; k < 3; k++) {
      double ak[3] = {ma[k], ma[3 + k], ma[6 + k]}, bk[3] = {mb[k], mb[3 + k], mb[6 + k]};
      if (k != i) d_addtoscl3(ea, ak, d_dot3(n, ak) > 0 ? sa[k] : -sa[k]);
      if (k != j) d_addtoscl3(eb, bk, d_dot3(n, bk) > 0 ? -sb[k] : sb[k]);
    }
    double w[3]; d_sub3(w, eb, ea);
    double cc = (i == 0 ? sel3(R, j) : (i == 1 ? sel3(R + 3, j) : sel3(R + 6, j)));
    double d1 = d_dot3(w, ai), d2 = d_dot3(w, bj), den = 1.0 - cc * cc;
    double sai = sel3(sa, i), sbj = sel3(sb, j);
    double u = d_clip((d1 - cc * d2) / den, -sai, sai);
    double v = d_clip((cc * d1 - d2) / den, -sbj, sbj);
    double qa[3], qb[3];
    d_addscl3(qa, ea, ai, u); d_addscl3(qb, eb, bj, v);
    d_sub3(w, qb, qa);
    double dist = d_dot3(w, n);
    if (dist > margin) return 0;
    for (int k = 0; k < 6; k++) con->frame[k] = 0;
    d_copy3(con->frame, n);
    con->dist = dist;
    con->pos[0] = 0.5 * (qa[0] + qb[0]); con->pos[1] = 0.5 * (qa[1] + qb[1]); con->pos[2] = 0.5 * (qa[2] + qb[2]);
    return 1;
  }
  // ---- face case
  int refA = code < 3, ax = refA ? code : code - 3;
  double pr[3], mr[9], sr[3], pq[3], mq[9], sq[3];
#pragma unroll
  for (int k = 0; k < 3; k++) { pr[k] = refA ? pa[k] : pb[k]; sr[k] = refA ? sa[k] : sb[k]; pq[k] = refA ? pb[k] : pa[k]; sq[k] = refA ? sb[k] : sa[k]; }
#pragma unroll
  for (int k = 0; k < 9; k++) { mr[k] = refA ? ma[k] : mb[k]; mq[k] = refA ? mb[k] : ma[k]; }
  double sgn = (refA ? sel3(t, ax) : -sel3(tb, ax)) >= 0 ? 1.0 : -1.0;
  int u1 = (ax + 1) % 3, u2 = (ax + 2) % 3;
  double n[3], ru[3], rv[3];
  col3(n, mr, ax); d_scl3(n, n, sgn);
  col3(ru, mr, u1); col3(rv, mr, u2);
  double hn = sel3(sr, ax);
  BBFace f;
  f.hu = sel3(sr, u1); f.hv = sel3(sr, u2);
  double nl[3];
  d_mulmattvec3(nl, mq, n);
  int k = 0; double amax = fabs(nl[0]);
  if (fabs(nl[1]) > amax) { amax = fabs(nl[1]); k = 1; }
  if (fabs(nl[2]) > amax) { amax = fabs(nl[2]); k = 2; }
  int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
  double qk[3], q1[3], q2[3], cen[3], rel[3];
  col3(qk, mq, k); col3(q1, mq, k1); col3(q2, mq, k2);
  double sqk = sel3(sq, k), sq1 = sel3(sq, k1), sq2 = sel3(sq, k2);
  d_addscl3(cen, pq, qk, sel3(nl, k) > 0 ? -sqk : sqk);
  d_sub3(rel, cen, pr);
  f.c0[0] = d_dot3(rel, ru); f.c0[1] = d_dot3(rel, rv); f.c0[2] = d_dot3(rel, n) - hn;
  f.e1[0] = sq1 * d_dot3(q1, ru); f.e1[1] = sq1 * d_dot3(q1, rv); f.e1[2] = sq1 * d_dot3(q1, n);
  f.e2[0] = sq2 * d_dot3(q2, ru); f.e2[1] = sq2 * d_dot3(q2, rv); f.e2[2] = sq2 * d_dot3(q2, n);
  f.det = f.e1[0] * f.e2[1] - f.e1[1] * f.e2[0];
  pend.f = f; pend.hn = hn; pend.refA = refA;
  d_copy3(pend.pr, pr); d_copy3(pend.ru, ru); d_copy3(pend.rv, rv); d_copy3(pend.n, n);
  return -3;
}
// the wave's part of a face case: every lane below 24 evaluates one candidate point of the (wave-uniform) face problem, then the
// selection - deepest, farthest from it, extreme on either side, each with its tie rule, so in candidate order - reads them lane
// by lane.  All lanes return the same four picks.  (One-lane emulation build: the candidates in a local array.)
DEV void bb_face_select(const BBFace &f, double margin, BBSel &s0, BBSel &s1, BBSel &s2, BBSel &s3) {
#ifdef MJPC_EMU
  BBSel cand[24];
  for (int q = 0; q < 24; q++) cand[q] = bb_candidate(f, q);
#define BB_CAND(q) cand[q]
#else
  const BBSel mine = bb_candidate(f, LANE < 24 ? LANE : 0);
  auto bb_pick = [&](int q) { BBSel o; o.ok = __builtin_amdgcn_readlane(mine.ok, q); o.x = readlane_d(mine.x, q); o.y = readlane_d(mine.y, q); o.d = readlane_d(mine.d, q); return o; };
#define BB_CAND(q) bb_pick(q)
#endif
  s0.ok = s1.ok = s2.ok = s3.ok = 0;
  s0.x = s0.y = s0.d = 0; s1 = s0; s2 = s0; s3 = s0;
  double bd = 1e300;
  for (int q = 0; q < 24; q++) { BBSel cd = BB_CAND(q); if (cd.ok && cd.d <= margin && cd.d < bd - BB_TIE_D) { bd = cd.d; s0 = cd; } }
  if (!s0.ok) return;
  double far = 1e-16;
  for (int q = 0; q < 24; q++) {
    BBSel cd = BB_CAND(q);
    if (cd.ok && cd.d <= margin) { double r2 = (cd.x - s0.x) * (cd.x - s0.x) + (cd.y - s0.y) * (cd.y - s0.y); if (r2 > far + BB_TIE_A) { far = r2; s1 = cd; } }
  }
  if (s1.ok) {
    double lx = s1.x - s0.x, ly = s1.y - s0.y, amx = 1e-12, amn = -1e-12;
    for (int q = 0; q < 24; q++) {
      BBSel cd = BB_CAND(q);
      if (cd.ok && cd.d <= margin) {
        double ar = lx * (cd.y - s0.y) - ly * (cd.x - s0.x);
        if (ar > amx + BB_TIE_A) { amx = ar; s2 = cd; }
        if (ar < amn - BB_TIE_A) { amn = ar; s3 = cd; }
      }
    }
  }
#undef BB_CAND
}
// the pair's lane again: contacts from the picks
DEV int bb_face_finish(NPCon *con, const BBPend &pd, const BBSel &s0, const BBSel &s1, const BBSel &s2, const BBSel &s3) {
  int cnt = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    BBSel cd = q == 0 ? s0 : (q == 1 ? s1 : (q == 2 ? s2 : s3));
    if (!cd.ok) continue;
    NPCon o;
    double hgt = pd.hn + cd.d - 0.5 * cd.d;
    o.pos[0] = pd.pr[0] + cd.x * pd.ru[0] + cd.y * pd.rv[0] + hgt * pd.n[0];
    o.pos[1] = pd.pr[1] + cd.x * pd.ru[1] + cd.y * pd.rv[1] + hgt * pd.n[1];
    o.pos[2] = pd.pr[2] + cd.x * pd.ru[2] + cd.y * pd.rv[2] + hgt * pd.n[2];
    o.dist = cd.d;
    for (int e = 0; e < 6; e++) o.frame[e] = 0;
    if (pd.refA) d_copy3(o.frame, pd.n); else d_scl3(o.frame, pd.n, -1);
    np_put(con, cnt, o);
    cnt++;
  }
  return cnt;
}

// sphere (geom1) vs cylinder (geom2): closed form in the cylinder's frame (distance = hypot of the radial and axial excess)
DEV int np_sphere_cylinder(NPCon *con, double margin, const double *sp, double sr, const double *cp, const double *cm, const double *cs) {
  double dif[3], p[3];
  d_sub3(dif, sp, cp);
  d_mulmattvec3(p, cm, dif);
  double R = cs[0], h = cs[1];
  double rho = sqrt(p[0] * p[0] + p[1] * p[1]);
  double ux = rho > D_MINVAL ? p[0] / rho : 1.0, uy = rho > D_MINVAL ? p[1] / rho : 0.0;
  double er = rho - R, ez = fabs(p[2]) - h, sz = p[2] >= 0 ? 1.0 : -1.0;
  double q[3], nl[3], dist;
  if (er <= 0 && ez <= 0) {
    if (-er < -ez) { q[0] = ux * R; q[1] = uy * R; q[2] = p[2]; nl[0] = ux; nl[1] = uy; nl[2] = 0; dist = er - sr; }
    else { q[0] = p[0]; q[1] = p[1]; q[2] = sz * h; nl[0] = 0; nl[1] = 0; nl[2] = sz; dist = ez - sr; }
  } else {
    double cr = er > 0 ? R : rho;
    q[0] = ux * cr; q[1] = uy * cr; q[2] = ez > 0 ? sz * h : p[2];
    double d[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    double len = d_norm3(d);
    dist = len - sr;
    if (dist > margin) return 0;
    d_scl3(nl, d, 1.0 / len);
  }
  if (dist > margin) return 0;
  double nw[3], surf[3];
  d_mulmatvec3(nw, cm, nl);
  for (int k = 0; k < 6; k++) con->frame[k] = 0;
  d_scl3(con->frame, nw, -1);
  con->dist = dist;
  d_mulmatvec3(surf, cm, q); d_add3(surf, surf, cp);
  d_addscl3(con->pos, surf, nw, 0.5 * dist);
  return 1;
}
DEV double capsule_cylinder_g(const double *p0, const double *a, double hc, double R, double h, double s) {
  double q[3] = {p0[0] + (s * hc) * a[0], p0[1] + (s * hc) * a[1], p0[2] + (s * hc) * a[2]};
  double rho = sqrt(q[0] * q[0] + q[1] * q[1]);
  double g = 0, er = rho - R, ez = fabs(q[2]) - h;
  if (er > 0) g += er * (q[0] * a[0] + q[1] * a[1]) / rho;
  if (ez > 0) g += (q[2] > 0 ? ez : -ez) * a[2];
  return g;
}
DEV int np_capsule_cylinder(NPCon *con, double margin, const double *kp, const double *km, const double *ks,
                            const double *cp, const double *cm, const double *cs) {
  double axis[3] = {km[2], km[5], km[8]}, dif[3], p0[3], a[3];
  d_sub3(dif, kp, cp);
  d_mulmattvec3(p0, cm, dif);
  d_mulmattvec3(a, cm, axis);
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
  NPCon t;
  d_addscl3(pt, kp, axis, sstar * hc);
  if (np_sphere_cylinder(&t, margin, pt, ks[0], cp, cm, cs)) { con[0] = t; cnt++; }
  double s2 = sstar <= 0 ? 1.0 : -1.0;
  d_addscl3(pt, kp, axis, s2 * hc);
  if (np_sphere_cylinder(&t, margin, pt, ks[0], cp, cm, cs)) { np_put(con, cnt, t); cnt++; }
  return cnt;
}

// ---- convex pairs without an analytic collider (cylinder-cylinder, cylinder-box): Minkowski portal refinement ("XenoCollide",
// the algorithm class MuJoCo reaches through libccd for these pairs; tolerance 1e-6, 50 iterations).  Both geoms are inflated
// by margin / 2, one contact per pair; the penetration is measured where the ray from the centres' difference through the
// origin leaves the Minkowski difference, the contact sits half way between the two witness points.
#define MPR_TOLERANCE 1e-6
#define MPR_ITERATIONS 50
struct MShape { int type; const double *pos, *mat, *size; double margin; const double *vert; int nvert; };
struct MSup { double v[3], v1[3], v2[3]; };
DEV void mpr_support1(const MShape &s, const double *dir, double *out) {
  double l[3], v[3];
  d_mulmattvec3(l, s.mat, dir);
  if (s.type == 6) {
    for (int k = 0; k < 3; k++) v[k] = l[k] >= 0 ? s.size[k] : -s.size[k];
  } else if (s.type == 5) {
    double n = sqrt(l[0] * l[0] + l[1] * l[1]);
    if (n > D_MINVAL) { v[0] = s.size[0] * l[0] / n; v[1] = s.size[0] * l[1] / n; } else { v[0] = 0; v[1] = 0; }
    v[2] = l[2] >= 0 ? s.size[1] : -s.size[1];
  } else if (s.type == 7) {       // convex mesh: the vertex farthest along l (first of equals); vertices are read from HBM / L2
    double best = -1e300; int bi = 0;
    for (int i = 0; i < s.nvert; i++) {
      double t = s.vert[3 * i] * l[0] + s.vert[3 * i + 1] * l[1] + s.vert[3 * i + 2] * l[2];
      if (t > best) { best = t; bi = i; }
    }
    v[0] = s.vert[3 * bi]; v[1] = s.vert[3 * bi + 1]; v[2] = s.vert[3 * bi + 2];
  } else if (s.type == 4) {
    double a = s.size[0] * s.size[0] * l[0], b = s.size[1] * s.size[1] * l[1], c = s.size[2] * s.size[2] * l[2];
    double n = sqrt(a * l[0] + b * l[1] + c * l[2]);
    if (n > D_MINVAL) { v[0] = a / n; v[1] = b / n; v[2] = c / n; } else { v[0] = 0; v[1] = 0; v[2] = 0; }
  } else if (s.type == 3) {
    v[0] = s.size[0] * l[0]; v[1] = s.size[0] * l[1]; v[2] = s.size[0] * l[2] + (l[2] >= 0 ? s.size[1] : -s.size[1]);
  } else {
    v[0] = s.size[0] * l[0]; v[1] = s.size[0] * l[1]; v[2] = s.size[0] * l[2];
  }
  d_mulmatvec3(out, s.mat, v);
  d_add3(out, out, s.pos);
  d_addtoscl3(out, dir, s.margin);
}
DEV void mpr_support(const MShape &a, const MShape &b, const double *dir, MSup &s) {
  double nd[3] = {-dir[0], -dir[1], -dir[2]};
  mpr_support1(a, dir, s.v1);
  mpr_support1(b, nd, s.v2);
  d_sub3(s.v, s.v1, s.v2);
}
DEV void mpr_portal_dir(const MSup &p1, const MSup &p2, const MSup &p3, double *dir) {
  double a[3], b[3];
  d_sub3(a, p2.v, p1.v);
  d_sub3(b, p3.v, p1.v);
  d_cross(dir, a, b);
  d_normalize3(dir);
}
DEV int mpr_reach_tolerance(const MSup &p1, const MSup &p2, const MSup &p3, const MSup &v4, const double *dir) {
  double dv4 = d_dot3(v4.v, dir);
  double d1 = dv4 - d_dot3(p1.v, dir), d2 = dv4 - d_dot3(p2.v, dir), d3 = dv4 - d_dot3(p3.v, dir);
  double dm = fmin(d1, fmin(d2, d3));
  return dm <= MPR_TOLERANCE;
}
DEV void mpr_expand_portal(const MSup &p0, MSup &p1, MSup &p2, MSup &p3, const MSup &v4) {
  double v4v0[3];
  d_cross(v4v0, v4.v, p0.v);
  if (d_dot3(p1.v, v4v0) > 0) {
    if (d_dot3(p2.v, v4v0) > 0) p1 = v4; else p3 = v4;
  } else {
    if (d_dot3(p3.v, v4v0) > 0) p2 = v4; else p1 = v4;
  }
}
// closest point of the triangle (a, b, c) to the origin and its barycentric weights
DEV void mpr_closest_on_triangle(const double *a, const double *b, const double *c, double *w, double *bw) {
  double ab[3], ac[3];
  d_sub3(ab, b, a); d_sub3(ac, c, a);
  double d1 = -d_dot3(ab, a), d2 = -d_dot3(ac, a);
  if (d1 <= 0 && d2 <= 0) { d_copy3(w, a); bw[0] = 1; bw[1] = 0; bw[2] = 0; return; }
  double d3 = -d_dot3(ab, b), d4 = -d_dot3(ac, b);
  if (d3 >= 0 && d4 <= d3) { d_copy3(w, b); bw[0] = 0; bw[1] = 1; bw[2] = 0; return; }
  double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); d_addscl3(w, a, ab, v); bw[0] = 1 - v; bw[1] = v; bw[2] = 0; return; }
  double d5 = -d_dot3(ab, c), d6 = -d_dot3(ac, c);
  if (d6 >= 0 && d5 <= d6) { d_copy3(w, c); bw[0] = 0; bw[1] = 0; bw[2] = 1; return; }
  double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { double v = d2 / (d2 - d6); d_addscl3(w, a, ac, v); bw[0] = 1 - v; bw[1] = 0; bw[2] = v; return; }
  double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    double bc[3]; d_sub3(bc, c, b);
    double v = (d4 - d3) / ((d4 - d3) + (d5 - d6));
    d_addscl3(w, b, bc, v); bw[0] = 0; bw[1] = 1 - v; bw[2] = v; return;
  }
  double den = 1.0 / (va + vb + vc);
  d_addscl3(w, a, ab, vb * den);
  d_addtoscl3(w, ac, vc * den);
  bw[1] = vb * den; bw[2] = vc * den; bw[0] = 1 - bw[1] - bw[2];
}
// 1 contact (frame[0..2] = normal from geom 1 to geom 2) or 0
DEV int np_convex(NPCon *con, double margin, const MShape &A, const MShape &B) {
  const double *p1 = A.pos, *p2 = B.pos;
  MSup q0, q1, q2, q3, v4;
  double dir[3], va[3], vb[3], depth, nrm[3], pos[3];
  d_sub3(q0.v, p1, p2); d_copy3(q0.v1, p1); d_copy3(q0.v2, p2);
  if (d_dot3(q0.v, q0.v) < D_MINVAL * D_MINVAL) q0.v[0] += 1e-9;
  d_scl3(dir, q0.v, -1); d_normalize3(dir);
  mpr_support(A, B, dir, q1);
  if (d_dot3(q1.v, dir) <= 0) return 0;
  d_cross(dir, q0.v, q1.v);
  int found = 0;
  if (d_dot3(dir, dir) < D_MINVAL * D_MINVAL) found = 2;
  else {
    d_normalize3(dir);
    mpr_support(A, B, dir, q2);
    if (d_dot3(q2.v, dir) <= 0) return 0;
    d_sub3(va, q1.v, q0.v); d_sub3(vb, q2.v, q0.v);
    d_cross(dir, va, vb); d_normalize3(dir);
    if (d_dot3(dir, q0.v) > 0) { MSup t = q1; q1 = q2; q2 = t; d_scl3(dir, dir, -1); }
    int ok = 0;
    for (int it = 0; it < MPR_ITERATIONS; it++) {
      mpr_support(A, B, dir, v4);
      if (d_dot3(v4.v, dir) <= 0) return 0;
      int cont = 0;
      d_cross(va, q1.v, v4.v);
      if (d_dot3(va, q0.v) < 0) { q2 = v4; cont = 1; }
      if (!cont) {
        d_cross(va, v4.v, q2.v);
        if (d_dot3(va, q0.v) < 0) { q1 = v4; cont = 1; }
      }
      if (!cont) { q3 = v4; ok = 1; break; }
      d_sub3(va, q1.v, q0.v); d_sub3(vb, q2.v, q0.v);
      d_cross(dir, va, vb); d_normalize3(dir);
    }
    if (!ok) return 0;
  }
  if (found == 2) {
    depth = d_norm3(q1.v);
    d_copy3(nrm, q1.v); d_normalize3(nrm);
    for (int k = 0; k < 3; k++) pos[k] = 0.5 * (q1.v1[k] + q1.v2[k]);
  } else {
    int hit = 0;
    for (int it = 0; it < MPR_ITERATIONS; it++) {
      mpr_portal_dir(q1, q2, q3, dir);
      if (d_dot3(dir, q1.v) >= 0) { hit = 1; break; }
      mpr_support(A, B, dir, v4);
      if (d_dot3(v4.v, dir) < 0 || mpr_reach_tolerance(q1, q2, q3, v4, dir)) return 0;
      mpr_expand_portal(q0, q1, q2, q3, v4);
    }
    if (!hit) return 0;
    for (int it = 0; ; it++) {
      mpr_portal_dir(q1, q2, q3, dir);
      mpr_support(A, B, dir, v4);
      if (mpr_reach_tolerance(q1, q2, q3, v4, dir) || it >= MPR_ITERATIONS) {
        double w[3], bw[3];
        mpr_closest_on_triangle(q1.v, q2.v, q3.v, w, bw);
        depth = d_norm3(w);
        if (depth < D_MINVAL) d_copy3(nrm, dir); else d_scl3(nrm, w, 1.0 / depth);
        for (int k = 0; k < 3; k++)
          pos[k] = 0.5 * (bw[0] * (q1.v1[k] + q1.v2[k]) + bw[1] * (q2.v1[k] + q2.v2[k]) + bw[2] * (q3.v1[k] + q3.v2[k]));
        break;
      }
      mpr_expand_portal(q0, q1, q2, q3, v4);
    }
  }
  double dist = margin - depth;
  if (dist > margin) return 0;
  NPCon t;
  t.dist = dist;
  d_copy3(t.pos, pos);
  d_copy3(t.frame, nrm); t.frame[3] = 0; t.frame[4] = 0; t.frame[5] = 0;
  np_put(con, 0, t);
  return 1;
}

// height field (geom 1) against a convex geom: the cells under the geom's bounding sphere, two triangular prisms each, every
// prism through the portal-refinement collider; the 4 deepest contacts are kept; *overflow beyond HF_MAXCELL cells
#define HF_MAXCELL 100
DEV int np_hfield(NPCon *con, double margin, const double *hp, const double *hm, const double *hsize, int nrow, int ncol,
                  const double *data, const MShape &B, double rbound, int *overflow) {
  double dif[3], c[3];
  d_sub3(dif, B.pos, hp);
  d_mulmattvec3(c, hm, dif);
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
    if (fmax(fmax(h00, h01), fmax(h10, h11)) < c[2] - r) continue;
    for (int tri = 0; tri < 2; tri++) {
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
      d_mulmatvec3(pp, hm, cen);
      d_add3(pp, pp, hp);
      MShape P = {7, pp, hm, hsize, 0.5 * margin, v, 6};
      NPCon t[4];
      if (np_convex(t, margin, P, B)) {
        if (cnt < 4) { np_put(con, cnt, t[0]); cnt++; }
        else {        // full: the new contact replaces the shallowest kept one if it is deeper (first such slot)
          int w = 0;
          double dw = con[0].dist;
          if (con[1].dist > dw) { w = 1; dw = con[1].dist; }
          if (con[2].dist > dw) { w = 2; dw = con[2].dist; }
          if (con[3].dist > dw) { w = 3; dw = con[3].dist; }
          if (t[0].dist < dw) np_put(con, w, t[0]);
        }
      }
    }
  }
  return cnt;
}

// plane against an ellipsoid / a convex mesh: its support point against the plane normal
DEV int np_plane_convex(NPCon *con, double margin, const double *pp, const double *pm, MShape E) {
  double n[3] = {pm[2], pm[5], pm[8]}, nd[3] = {-pm[2], -pm[5], -pm[8]}, sp[3], dif[3];
  E.margin = 0;
  mpr_support1(E, nd, sp);
  d_sub3(dif, sp, pp);
  double dist = d_dot3(dif, n);
  if (dist > margin) return 0;
  NPCon t;
  t.dist = dist;
  d_addscl3(t.pos, sp, n, -0.5 * dist);
  d_copy3(t.frame, n); t.frame[3] = 0; t.frame[4] = 0; t.frame[5] = 0;
  np_put(con, 0, t);
  return 1;
}

// The rarely-met pair types (capsule-box, box-box, the cylinder and ellipsoid pairs).  n = -1: no collider and possibly touching.
// Only the out-of-line flavour of the narrow-phase batch (narrow_batch<true>) contains this code: the batch loop of
// collision() itself stays free of it and of any call inside the loop body's live ranges.
struct NPOut { NPCon c[4]; int n; };
DEV NPOut narrow_heavy(Ctx &c, int g1, int g2, double margin, BBPend &pend) {
  const DevModel &M = *c.M;
  NPOut o;
  int t1 = MI(geom_type)[g1], t2 = MI(geom_type)[g2];
  double p1[3], p2[3], m1[9], m2[9], s1[3], s2[3];
  d_copy3(p1, c.geom_xpos + 3 * g1); d_copy3(p2, c.geom_xpos + 3 * g2);
  for (int k = 0; k < 9; k++) { m1[k] = c.geom_xmat[9 * g1 + k]; m2[k] = c.geom_xmat[9 * g2 + k]; }
  d_copy3(s1, MD(geom_size) + 3 * g1); d_copy3(s2, MD(geom_size) + 3 * g2);
  o.n = -1;
  if (t1 == 3 && t2 == 6) o.n = np_capsule_box(o.c, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == 6 && t2 == 6) o.n = np_box_box(o.c, margin, p1, m1, s1, p2, m2, s2, pend);
  else if (t1 == 2 && t2 == 5) o.n = np_sphere_cylinder(o.c, margin, p1, s1[0], p2, m2, s2);
  else if (t1 == 3 && t2 == 5) o.n = np_capsule_cylinder(o.c, margin, p1, m1, s1, p2, m2, s2);
  else if (t1 == 1 && t2 >= 2) {
    // height field against anything convex
    MShape B = {t2, p2, m2, s2, 0.5 * margin, nullptr, 0};
    if (t2 == 7) { int k = M.geom_dataid[g2]; B.vert = M.mesh_vert + 3 * M.mesh_vertadr[k]; B.nvert = M.mesh_vertnum[k]; }
    int h = M.geom_dataid[g1], over = 0;
    o.n = np_hfield(o.c, margin, p1, m1, M.hfield_size + 4 * h, M.hfield_nrow[h], M.hfield_ncol[h], M.hfield_data + M.hfield_adr[h], B,
                    MD(geom_rbound)[g2], &over);
    if (over) o.n = -1;
  }
  else if ((t1 == 4 || t2 == 4 || t1 == 7 || t2 == 7) && t1 != 1 && t2 != 1) {
    // ellipsoids and convex meshes: support point against a plane, the portal-refinement collider against everything else
    MShape A = {t1, p1, m1, s1, 0.5 * margin, nullptr, 0}, B = {t2, p2, m2, s2, 0.5 * margin, nullptr, 0};
    if (t1 == 7) { int k = M.geom_dataid[g1]; A.vert = M.mesh_vert + 3 * M.mesh_vertadr[k]; A.nvert = M.mesh_vertnum[k]; }
    if (t2 == 7) { int k = M.geom_dataid[g2]; B.vert = M.mesh_vert + 3 * M.mesh_vertadr[k]; B.nvert = M.mesh_vertnum[k]; }
    o.n = t1 == 0 ? np_plane_convex(o.c, margin, p1, m1, B) : np_convex(o.c, margin, A, B);
  }
  else if (t1 == 5 && (t2 == 5 || t2 == 6)) {
    // cylinder-cylinder / cylinder-box: the cylinder's bounding capsule decides "certainly apart" (exact, cheap); otherwise the
    // portal-refinement collider
    NPCon tmp[4];
    int n = t2 == 5 ? np_capsule_capsule(tmp, margin, p1, m1, s1, p2, m2, s2) : np_capsule_box(tmp, margin, p1, m1, s1, p2, m2, s2);
    MShape A = {t1, p1, m1, s1, 0.5 * margin, nullptr, 0}, B = {t2, p2, m2, s2, 0.5 * margin, nullptr, 0};
    o.n = n == 0 ? 0 : np_convex(o.c, margin, A, B);
  }
  return o;
}

// squared distance from point q to the segment p +- h a (|a| = 1)
DEV double seg_point_dist2(const double *p, const double *a, double h, const double *q) {
  double w[3];
  d_sub3(w, q, p);
  double x = d_clip(d_dot3(a, w), -h, h);
  d_addtoscl3(w, a, -x);
  return d_dot3(w, w);
}

// returns the number of contacts; -2: a pair type handled by narrow_heavy().
// A cylinder that is not against a plane is first replaced by its bounding capsule (same radius and half length) and runs
// through the SAME sphere-capsule / capsule-capsule code as the real capsules of the wave (no extra divergent code path):
// "certainly apart" is exact, and only a cylinder whose bounding capsule touches goes out of line.  A capsule / cylinder
// against a box first tests its segment against the box's bounding sphere.
DEV int narrow_phase(Ctx &c, int g1, int g2, double margin, NPCon *con) {
  const DevModel &M = *c.M;
  const int t1 = MI(geom_type)[g1], t2 = MI(geom_type)[g2];
  const int cyl = (t1 == 5 || (t2 == 5 && t1 != 0));
  const int e1 = t1 == 5 ? 3 : t1, e2 = (t2 == 5 && t1 != 0) ? 3 : t2;
  if (e1 == 6 || e1 == 4 || e2 == 4 || e1 == 1 || e2 == 1 || e1 == 7 || e2 == 7) return -2;      // box-box (and what create() refuses)
  double p1[3], p2[3], m1[9], m2[9], s1[3], s2[3];
  d_copy3(p1, c.geom_xpos + 3 * g1); d_copy3(p2, c.geom_xpos + 3 * g2);
  for (int k = 0; k < 9; k++) { m1[k] = c.geom_xmat[9 * g1 + k]; m2[k] = c.geom_xmat[9 * g2 + k]; }
  d_copy3(s1, MD(geom_size) + 3 * g1); d_copy3(s2, MD(geom_size) + 3 * g2);
  int n;
  if (e1 == 0) {
    double nrm[3] = {m1[2], m1[5], m1[8]};
    if (e2 == 2) n = np_plane_sphere(con, margin, p1, nrm, p2, s2[0]);
    else if (e2 == 3) n = np_plane_capsule(con, margin, p1, m1, p2, m2, s2);
    else if (e2 == 6) n = np_plane_box(con, margin, p1, m1, p2, m2, s2);
    else n = np_plane_cylinder(con, margin, p1, m1, p2, m2, s2);
    return n;
  }
  if (e2 == 6 && e1 == 3) {         // capsule / cylinder against a box: cheap conservative separations, else out of line
    double a1[3] = {m1[2], m1[5], m1[8]};
    double r = s1[0] + MD(geom_rbound)[g2] + margin;
    if (seg_point_dist2(p1, a1, s1[1], p2) > r * r) return 0;         // segment against the box's bounding sphere
    double dif[3], q[3], al[3];
    d_sub3(dif, p1, p2);
    d_mulmattvec3(q, m2, dif);
    d_mulmattvec3(al, m2, a1);
    double rr = s1[0] + margin;                                        // the box's three face normals as separating axes
    if (fabs(q[0]) - s1[1] * fabs(al[0]) > s2[0] + rr || fabs(q[1]) - s1[1] * fabs(al[1]) > s2[1] + rr ||
        fabs(q[2]) - s1[1] * fabs(al[2]) > s2[2] + rr) return 0;
    // ... and the three axes  segment direction x box axis  (the segment projects to a point on them)
    {
      double l0 = sqrt(al[1] * al[1] + al[2] * al[2]), l1 = sqrt(al[0] * al[0] + al[2] * al[2]), l2 = sqrt(al[0] * al[0] + al[1] * al[1]);
      if (fabs(q[2] * al[1] - q[1] * al[2]) > s2[1] * fabs(al[2]) + s2[2] * fabs(al[1]) + rr * l0) return 0;      // a x e0 = (0, a2, -a1)
      if (fabs(q[0] * al[2] - q[2] * al[0]) > s2[0] * fabs(al[2]) + s2[2] * fabs(al[0]) + rr * l1) return 0;      // a x e1 = (-a2, 0, a0)
      if (fabs(q[1] * al[0] - q[0] * al[1]) > s2[0] * fabs(al[1]) + s2[1] * fabs(al[0]) + rr * l2) return 0;      // a x e2 = (a1, -a0, 0)
    }
    return -2;
  }
  if (cyl) {
    // cheap conservative separations before the bounding-capsule test proper (no divide, no square root): each segment against
    // the other geom's bounding sphere
    double a1[3] = {m1[2], m1[5], m1[8]}, a2[3] = {m2[2], m2[5], m2[8]};
    double h1 = e1 == 2 ? 0.0 : s1[1], h2 = e2 == 2 ? 0.0 : s2[1];
    double ra = s1[0] + h1 + s2[0] + margin, rb = s1[0] + s2[0] + h2 + margin;
    if (seg_point_dist2(p2, a2, h2, p1) > ra * ra) return 0;
    if (seg_point_dist2(p1, a1, h1, p2) > rb * rb) return 0;
  }
  if (e1 == 2) {
    if (e2 == 2) n = np_sphere_sphere(con, margin, p1, s1[0], p2, s2[0]);
    else if (e2 == 3) n = np_sphere_capsule(con, margin, p1, s1[0], p2, m2, s2);
    else n = np_sphere_box(con, margin, p1, s1[0], p2, m2, s2);
  } else {
    n = np_capsule_capsule(con, margin, p1, m1, s1, p2, m2, s2);
  }
  return (cyl && n > 0) ? -2 : n;
}

// contact parameters of a geom pair (mj_contactParam): prm = [friction 5, solref 2, solimp 5]; once per pair, not per contact
DEV void contact_param(Ctx &c, int g1, int g2, double *prm, int *dim) {
  const DevModel &M = *c.M; (void)M;
  int p1 = MI(geom_priority)[g1], p2 = MI(geom_priority)[g2];
  double fri[3];
  if (p1 != p2) {
    int g = p1 > p2 ? g1 : g2;
    *dim = MI(geom_condim)[g];
    for (int i = 0; i < 2; i++) prm[5 + i] = MD(geom_solref)[2 * g + i];
    for (int i = 0; i < 5; i++) prm[7 + i] = MD(geom_solimp)[5 * g + i];
    d_copy3(fri, MD(geom_friction) + 3 * g);
  } else {
    int d1 = MI(geom_condim)[g1], d2 = MI(geom_condim)[g2];
    *dim = d1 > d2 ? d1 : d2;
    double s1 = MD(geom_solmix)[g1], s2 = MD(geom_solmix)[g2], mix;
    if (s1 >= D_MINVAL && s2 >= D_MINVAL) mix = s1 / (s1 + s2);
    else if (s1 < D_MINVAL && s2 < D_MINVAL) mix = 0.5;
    else if (s1 < D_MINVAL) mix = 0.0;
    else mix = 1.0;
    double r10 = MD(geom_solref)[2 * g1], r20 = MD(geom_solref)[2 * g2];
    for (int i = 0; i < 2; i++) {
      double a = MD(geom_solref)[2 * g1 + i], b = MD(geom_solref)[2 * g2 + i];
      prm[5 + i] = (r10 > 0 && r20 > 0) ? mix * a + (1 - mix) * b : fmin(a, b);
    }
    for (int i = 0; i < 5; i++) prm[7 + i] = mix * MD(geom_solimp)[5 * g1 + i] + (1 - mix) * MD(geom_solimp)[5 * g2 + i];
    for (int i = 0; i < 3; i++) fri[i] = fmax(MD(geom_friction)[3 * g1 + i], MD(geom_friction)[3 * g2 + i]);
  }
  prm[0] = fri[0]; prm[1] = fri[0]; prm[2] = fri[1]; prm[3] = fri[2]; prm[4] = fri[2];
}

// one batch of (at most) NLANE active pairs: narrow phase per lane, ordered compaction, contact records.
// returns 0: done; 1 (HEAVY == false only): some pair needs narrow_heavy(), nothing was written; 2: contact buffer full
#define HX_HEAVY 46      // the previous step's first batch needed an out-of-line collider: this step starts with that flavour
template <bool HEAVY>
DEV int narrow_batch(Ctx &c, int base, int nactive, int *used_heavy = nullptr) {
  const DevModel &M = *c.M;
  int a = base + LANE, n = 0, g1 = 0, g2 = 0;
  double margin = 0, gap = 0;
  NPCon con[4] = {};
  BBPend pend; pend.refA = 0;
  if (a < nactive) {
    const int gg = c.active[a];           // (the broad phase leaves the pair's two geoms in the list, not its index)
    g1 = gg & 0xffff; g2 = (int)((unsigned)gg >> 16);
    margin = fmax(MD(geom_margin)[g1], MD(geom_margin)[g2]);
    gap = fmax(MD(geom_gap)[g1], MD(geom_gap)[g2]);
    n = narrow_phase(c, g1, g2, margin, con);
#if defined(MJPC_PROFILE_COLLISION) && MJPC_PROFILE_COLLISION == 2
    if constexpr (HEAVY) PROF(c, 10);
#endif
    if constexpr (HEAVY) {
      if (n == -2 && used_heavy) *used_heavy = 1;
      if (n == -2) {
        NPOut h = narrow_heavy(c, g1, g2, margin, pend);
        n = h.n; con[0] = h.c[0]; con[1] = h.c[1]; con[2] = h.c[2]; con[3] = h.c[3];
      }
      if (n < 0 && n != -3) { c.warning |= WARN_UNSUPPORTED; n = 0; }      // lane-local here; made wave-uniform below
    }
  }
#if defined(MJPC_PROFILE_COLLISION) && MJPC_PROFILE_COLLISION == 2
  if constexpr (HEAVY) PROF(c, 16);
#endif
  if constexpr (HEAVY) {
    // box pairs in their face case, one after the other with the whole wave (bb_face_select): the pair's lane hands its face
    // problem round, takes the picks back
#ifdef MJPC_EMU
    unsigned long long todo = n == -3 ? 1ull : 0ull;
#else
    unsigned long long todo = __builtin_amdgcn_ballot_w64(n == -3);
#endif
    while (todo) {
      const int L = __builtin_ctzll(todo);
      todo &= todo - 1;
      BBFace f; double mg;
#ifdef MJPC_EMU
      f = pend.f; mg = margin;
#else
      for (int k = 0; k < 3; k++) { f.c0[k] = readlane_d(pend.f.c0[k], L); f.e1[k] = readlane_d(pend.f.e1[k], L); f.e2[k] = readlane_d(pend.f.e2[k], L); }
      f.hu = readlane_d(pend.f.hu, L); f.hv = readlane_d(pend.f.hv, L); f.det = readlane_d(pend.f.det, L); mg = readlane_d(margin, L);
#endif
      BBSel s0, s1, s2, s3;
      bb_face_select(f, mg, s0, s1, s2, s3);
      if (LANE == L) n = bb_face_finish(con, pend, s0, s1, s2, s3);
    }
  }
  if constexpr (HEAVY) c.warning = wave_or_i(c.warning);
  else if (wave_any(n == -2)) return 1;
  int tot, off = wave_excl_scan(n, &tot);
  // contact buffer full (mjWARN_CONTACTFULL): the contacts that still fit are kept in pair order, the rest of the step goes on
  // with them (the candidate fails at the end of the step, but its constraint stage sees what the CPU path sees)
  const int full = c.ncon + tot > M.nconmax;
  if (full) c.warning |= WARN_CONTACTFULL;
  double prm[12]; int dim = 0;
  if (n > 0) contact_param(c, g1, g2, prm, &dim);
  for (int k = 0; k < n; k++) {
    int ci = c.ncon + off + k;
    if (ci >= M.nconmax) break;
    double *cc = c.contact + ci * c.M->con_stride;
#pragma unroll
    for (int q = 0; q < 5; q++) cc[CON_FRICTION + q] = prm[q];
    cc[CON_SOLREF] = prm[5]; cc[CON_SOLREF + 1] = prm[6];
#pragma unroll
    for (int q = 0; q < 5; q++) cc[CON_SOLIMP + q] = prm[7 + q];
    const NPCon cur = np_get(con, k);
    double fr[9];
    for (int q = 0; q < 6; q++) fr[q] = cur.frame[q];
    d_makeframe(fr);
    cc[CON_DIST] = cur.dist;
    d_copy3(cc + CON_POS, cur.pos);
    for (int q = 0; q < 9; q++) cc[CON_FRAME + q] = fr[q];
    cc[CON_INCLUDEMARGIN] = margin - gap;
    cc[CON_MU] = 0;
    int *ci_ = c.con_i + ci * CONI_STRIDE;
    ci_[0] = dim; ci_[1] = g1; ci_[2] = g2; ci_[3] = 0;
  }
  c.ncon = full ? M.nconmax : c.ncon + tot;
  return full ? 2 : 0;
}
// the batches from `base` on with every collider available (out of line: own registers, called from outside collision()'s loop)
struct BatchOut { int ncon, warning, first_heavy; };
DEV_NOINLINE BatchOut narrow_rest_heavy(const KParams *Kg, int base, int nactive, int ncon, int warning) {
  Ctx c;
  ctx_init(c, Kg, lds_base());
  c.ncon = ncon; c.warning = warning;
  int first = 0, used = 0;
  for (int b = base; b < nactive; b += NLANE) {
    used = 0;
    int st = narrow_batch<true>(c, b, nactive, &used);
    if (b == base) first = wave_any(used);
    if (st == 2) break;
  }
  BatchOut o;
  o.ncon = c.ncon; o.warning = c.warning; o.first_heavy = first;
  return o;
}

DEV void collision(Ctx &c) {
  const DevModel &M = *c.M;
  c.ncon = 0;
  if (M.disableflags & (1 << 4)) return;
  // (1) broad phase: ordered compaction of the pairs whose bounding volumes overlap.  One packed record per pair (host.h) and
  // the next batch's records requested before this batch is tested: where the tables live in HBM / L2 (no room for the LDS copy)
  // a batch no longer waits two dependent round trips
  int nactive = 0;
  const int npair = M.npair;
  int gg_n = 0; double mg_n = 0, r1_n = 0, r2_n = 0;
  if (LANE < npair) { gg_n = MI(pair_gg)[LANE]; const double *pp = MD(pair_bp) + 3 * LANE; mg_n = pp[0]; r1_n = pp[1]; r2_n = pp[2]; }
  for (int base = 0; base < npair; base += NLANE) {
    const int p = base + LANE, pn = p + NLANE;
    int pass = 0;
    const int gg = gg_n;
    const double margin = mg_n, r1 = r1_n, r2 = r2_n;
    if (pn < npair) { gg_n = MI(pair_gg)[pn]; const double *pp = MD(pair_bp) + 3 * pn; mg_n = pp[0]; r1_n = pp[1]; r2_n = pp[2]; }
    if (p < npair) {
      const int g1 = gg & 0xffff, g2 = (int)((unsigned)gg >> 16);
      double dif[3];
      d_sub3(dif, c.geom_xpos + 3 * g2, c.geom_xpos + 3 * g1);
      pass = 1;
      if (r1 < 0) {
        const double *mat = c.geom_xmat + 9 * g1;
        double n[3] = {mat[2], mat[5], mat[8]};
        if (d_dot3(dif, n) > margin + r2) pass = 0;
      } else if (r1 > 0 && r2 > 0) {
        double bound = r1 + r2 + margin;
        if (d_dot3(dif, dif) > bound * bound) pass = 0;
      }
    }
    int tot, off = wave_excl_scan(pass, &tot);
    if (pass && nactive + off < MAX_ACTIVE_PAIRS) c.active[nactive + off] = gg;
    nactive += tot;
  }
  if (nactive > MAX_ACTIVE_PAIRS) { c.warning |= WARN_CONTACTFULL; nactive = MAX_ACTIVE_PAIRS; }
  SYNC();
#if defined(MJPC_PROFILE_COLLISION) && MJPC_PROFILE_COLLISION == 1      // (diagnostic build: broad phase / cheap batches / out-of-line batches in the slots a pyramidal model leaves empty; level 2: inside the out-of-line batch - cheap colliders / out-of-line colliders / the rest)
  PROF(c, 10);
#endif
  // (2) narrow phase, one lane per active pair, contacts appended in pair order.  The loop only knows the cheap colliders; at
  // the first batch in which some pair needs an expensive one it stops (nothing of that batch is kept) and the out-of-line
  // flavour finishes the list from there.  The call sits behind the loop, so the loop's registers are not shaped by it.
  // (a candidate whose first batch went out of line in the previous step - a cube lying in the hand - starts there: the cheap
  // pass would only be thrown away again.  Both flavours write the same contacts.)
  int heavy_from = (nactive > 0 && uniform_i(c.misc[HX_HEAVY])) ? 0 : -1;
  if (heavy_from < 0)
    for (int base = 0; base < nactive; base += NLANE) {
      int st = narrow_batch<false>(c, base, nactive);
      if (st == 1) heavy_from = base;
      if (st != 0) break;
    }
  int again = 0;
#if defined(MJPC_PROFILE_COLLISION) && MJPC_PROFILE_COLLISION == 1
  PROF(c, 16);
#else
  PROF(c, 4);
#endif
  if (heavy_from >= 0) {
    BatchOut o = narrow_rest_heavy(c.K, heavy_from, nactive, c.ncon, c.warning);
    c.ncon = o.ncon; c.warning = o.warning;
    again = heavy_from == 0 && o.first_heavy;
  }
  if (LANE == 0) c.misc[HX_HEAVY] = again;
#ifdef MJPC_PROFILE_COLLISION
  PROF(c, 18);
#endif
  SYNC();
}

