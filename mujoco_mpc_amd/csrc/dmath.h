// dmath.h — fp64 vector / quaternion / spatial-algebra helpers for the device code.
// Semantics follow MuJoCo's mju_* conventions (quaternion [w,x,y,z]; spatial motion
// [angular; linear] expressed about the subtree-root centre of mass; 10-number inertia).
#pragma once
#include "spmd.h"

#define D_MINVAL 1e-15
#define D_PI 3.14159265358979323846

// 1/sqrt(x) to full fp64 accuracy without the IEEE sqrt/divide sequences (0 -> 0 so that T = x*rsqrt(x) = 0)
DEV double fast_rsqrt(double x) {
  if (!(x > 0)) return 0.0;
#ifdef MJPC_EMU
  return 1.0 / sqrt(x);
#else
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  y = y * (1.5 - 0.5 * x * y * y);
  return y;
#endif
}
// 1/x without the IEEE divide sequence (v_rcp_f64 + two Newton steps: ~1 ulp)
DEV double fast_rcp(double x) {
#ifdef MJPC_EMU
  return 1.0 / x;
#else
  double y = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, e, y);
#endif
}
// a / b and sqrt(x) without the IEEE sequences (~1 ulp; the emulation build keeps the IEEE operations)
DEV double d_div(double a, double b) {
#ifdef MJPC_EMU
  return a / b;
#else
  double r = fast_rcp(b), q = a * r;
  return __builtin_fma(r, __builtin_fma(-b, q, a), q);
#endif
}
DEV double d_sqrt(double x) {
#ifdef MJPC_EMU
  return sqrt(x);
#else
  if (!(x > 0)) return x == 0 ? 0.0 : sqrt(x);
  double r = fast_rsqrt(x), s = x * r;
  return __builtin_fma(0.5 * r, __builtin_fma(-s, s, x), s);
#endif
}
// x^p for the impedance sigmoid: the default power 2 (and 1, 3) without the generic pow
DEV double d_pow_small(double x, double p) {
#ifndef MJPC_EMU
  if (p == 2.0) return x * x;
  if (p == 1.0) return x;
  if (p == 3.0) return x * x * x;
#endif
  return pow(x, p);
}
DEV double d_dot3(const double *a, const double *b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
DEV double d_norm3(const double *a) { return sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }
DEV void d_copy3(double *r, const double *a) { r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; }
DEV void d_copy4(double *r, const double *a) { r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; r[3]=a[3]; }
DEV void d_add3(double *r, const double *a, const double *b) { r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; }
DEV void d_sub3(double *r, const double *a, const double *b) { r[0]=a[0]-b[0]; r[1]=a[1]-b[1]; r[2]=a[2]-b[2]; }
DEV void d_scl3(double *r, const double *a, double s) { r[0]=a[0]*s; r[1]=a[1]*s; r[2]=a[2]*s; }
DEV void d_addscl3(double *r, const double *a, const double *b, double s) { r[0]=a[0]+b[0]*s; r[1]=a[1]+b[1]*s; r[2]=a[2]+b[2]*s; }
DEV void d_addtoscl3(double *r, const double *b, double s) { r[0]+=b[0]*s; r[1]+=b[1]*s; r[2]+=b[2]*s; }
DEV void d_cross(double *r, const double *a, const double *b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0]=x; r[1]=y; r[2]=z;
}
DEV double d_normalize3(double *a) {
  double n = d_norm3(a);
  if (n < D_MINVAL) { a[0]=1; a[1]=0; a[2]=0; }
  else { double s = 1.0 / n; a[0]*=s; a[1]*=s; a[2]*=s; }
  return n;
}
DEV double d_normalize2(double *a) {
  double n = sqrt(a[0]*a[0] + a[1]*a[1]);
  if (n < D_MINVAL) { a[0]=1; a[1]=0; }
  else { double s = 1.0 / n; a[0]*=s; a[1]*=s; }
  return n;
}
DEV void d_normalize4(double *q) {
  double n2 = q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3];
  double s = fast_rsqrt(n2), n = n2 * s;                 // |q| and 1/|q| without the IEEE sqrt / divide sequences
  if (n < D_MINVAL) { q[0]=1; q[1]=0; q[2]=0; q[3]=0; }
  else if (fabs(n - 1) > D_MINVAL) { q[0]*=s; q[1]*=s; q[2]*=s; q[3]*=s; }
}
// sin and cos of a joint-sized angle: Cody-Waite reduction by pi/2 (two-term constant) + the fdlibm kernel polynomials
// (max error 1 ulp at |x| <= 20, checked against libm); large arguments fall back to libm
DEV void d_sincos(double x, double *sn, double *cs) {
#ifdef MJPC_EMU
  *sn = sin(x); *cs = cos(x);
#else
  if (!(fabs(x) < 64.0)) { *sn = sin(x); *cs = cos(x); return; }
  double k = rint(x * 0.63661977236758134308);
  double r = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
  double z = r * r;
  double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
  double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
  double s = r + r * z * ps;
  double c = 1.0 - 0.5 * z + z * z * pc;
  int q = ((int)k) & 3;
  double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
  *sn = (q & 2) ? -s1 : s1;
  *cs = ((q + 1) & 2) ? -c1 : c1;
#endif
}
DEV double d_clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
DEV void d_mulmatvec3(double *r, const double *m, const double *v) {
  double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
DEV void d_mulmattvec3(double *r, const double *m, const double *v) {
  double x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  double y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  double z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
DEV void d_mulquat(double *r, const double *a, const double *b) {
  double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0]=w; r[1]=x; r[2]=y; r[3]=z;
}
DEV void d_quat2mat(double *m, const double *q) {
  double q00=q[0]*q[0], q01=q[0]*q[1], q02=q[0]*q[2], q03=q[0]*q[3];
  double q11=q[1]*q[1], q12=q[1]*q[2], q13=q[1]*q[3];
  double q22=q[2]*q[2], q23=q[2]*q[3], q33=q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33;  m[4] = q00 - q11 + q22 - q33;  m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03);  m[2] = 2*(q13 + q02);
  m[3] = 2*(q12 + q03);  m[5] = 2*(q23 - q01);
  m[6] = 2*(q13 - q02);  m[7] = 2*(q23 + q01);
}
DEV void d_rotvecquat(double *r, const double *v, const double *q) {
  double m[9]; d_quat2mat(m, q); d_mulmatvec3(r, m, v);
}
DEV void d_axisangle2quat(double *q, const double *axis, double angle) {
  if (angle == 0) { q[0]=1; q[1]=0; q[2]=0; q[3]=0; return; }
  double s, cq;
  d_sincos(angle * 0.5, &s, &cq);
  q[0] = cq; q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
DEV void d_quatintegrate(double *q, const double *vel, double scale) {
  double ax[3] = {vel[0], vel[1], vel[2]};
  double angle = scale * d_normalize3(ax);
  double qr[4], t[4];
  d_axisangle2quat(qr, ax, angle);
  d_normalize4(q);
  d_mulquat(t, q, qr);
  q[0]=t[0]; q[1]=t[1]; q[2]=t[2]; q[3]=t[3];
  d_normalize4(q);
}
DEV void d_subquat(double *r, const double *qa, const double *qb) {
  double qn[4] = {qb[0], -qb[1], -qb[2], -qb[3]}, qd[4];
  d_mulquat(qd, qn, qa);
  double ax[3] = {qd[1], qd[2], qd[3]};
  double s = d_normalize3(ax);
  double speed = 2 * atan2(s, qd[0]);
  if (speed > D_PI) speed -= 2 * D_PI;
  d_scl3(r, ax, speed);
}
DEV void d_inertcom(double *r, const double *inert, const double *mat, const double *dif, double mass) {
  double t[9], I[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3*i+j] = mat[3*i+j] * inert[j];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
    I[3*i+j] = t[3*i]*mat[3*j] + t[3*i+1]*mat[3*j+1] + t[3*i+2]*mat[3*j+2];
  double d2 = dif[0]*dif[0] + dif[1]*dif[1] + dif[2]*dif[2];
  r[0] = I[0] + mass * (d2 - dif[0]*dif[0]);
  r[1] = I[4] + mass * (d2 - dif[1]*dif[1]);
  r[2] = I[8] + mass * (d2 - dif[2]*dif[2]);
  r[3] = I[1] - mass * dif[0]*dif[1];
  r[4] = I[2] - mass * dif[0]*dif[2];
  r[5] = I[5] - mass * dif[1]*dif[2];
  r[6] = mass * dif[0]; r[7] = mass * dif[1]; r[8] = mass * dif[2];
  r[9] = mass;
}
DEV void d_mulinertvec(double *r, const double *i, const double *v) {
  r[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  r[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  r[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  r[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  r[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  r[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
DEV void d_crossmotion(double *r, const double *vel, const double *v) {
  double a[3], b[3], c[3];
  d_cross(a, vel, v); d_cross(b, vel, v + 3); d_cross(c, vel + 3, v);
  r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; r[3]=b[0]+c[0]; r[4]=b[1]+c[1]; r[5]=b[2]+c[2];
}
DEV void d_crossforce(double *r, const double *vel, const double *f) {
  double a[3], b[3], c[3];
  d_cross(a, vel, f); d_cross(b, vel + 3, f + 3); d_cross(c, vel, f + 3);
  r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; r[3]=c[0]; r[4]=c[1]; r[5]=c[2];
}
DEV void d_makeframe(double *f) {
  d_normalize3(f);
  if (d_norm3(f + 3) < 0.5) {
    f[3]=0; f[4]=0; f[5]=0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double d = d_dot3(f, f + 3);
  f[3] -= f[0]*d; f[4] -= f[1]*d; f[5] -= f[2]*d;
  d_normalize3(f + 3);
  d_cross(f + 6, f, f + 3);
}
