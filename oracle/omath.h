/*
 * oracle/omath.h — TEST INFRASTRUCTURE ONLY (CPU oracle), never linked into the product.
 *
 * Small fp64 vector / quaternion / spatial-algebra helpers.  These restate the published
 * semantics of MuJoCo 3.1.4's mju_* utilities (third-party dependency of the reference,
 * fetched by /root/reference/CMakeLists.txt:58-61 and therefore NOT in /root/reference).
 * Reference call sites that rely on them: mjpc/tasks/quadruped/quadruped.cc:70,712-713,
 * mjpc/tasks/shadow_reorient/hand.cc:61, mjpc/trajectory.cc:158,198 (mj_step/mj_forward).
 * Parity at this boundary is UNPINNED (no MuJoCo here); see oracle/README.md.
 */
#ifndef ORACLE_OMATH_H_
#define ORACLE_OMATH_H_
#include <math.h>
#include <string.h>

#define O_MINVAL 1e-15
#define O_PI 3.14159265358979323846

static inline void o_zero(double *r, int n) { memset(r, 0, sizeof(double) * (size_t)n); }
static inline void o_copy(double *r, const double *a, int n) { memcpy(r, a, sizeof(double) * (size_t)n); }
static inline double o_dot(const double *a, const double *b, int n) {
  double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s;
}
static inline double o_dot3(const double *a, const double *b) { return a[0]*b[0] + a[1]*b[1] + a[2]*b[2]; }
static inline double o_norm3(const double *a) { return sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }
static inline double o_norm(const double *a, int n) { return sqrt(o_dot(a, a, n)); }
static inline void o_copy3(double *r, const double *a) { r[0]=a[0]; r[1]=a[1]; r[2]=a[2]; }
static inline void o_add3(double *r, const double *a, const double *b) { r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2]; }
static inline void o_sub3(double *r, const double *a, const double *b) { r[0]=a[0]-b[0]; r[1]=a[1]-b[1]; r[2]=a[2]-b[2]; }
static inline void o_scl3(double *r, const double *a, double s) { r[0]=a[0]*s; r[1]=a[1]*s; r[2]=a[2]*s; }
static inline void o_addscl3(double *r, const double *a, const double *b, double s) { r[0]=a[0]+b[0]*s; r[1]=a[1]+b[1]*s; r[2]=a[2]+b[2]*s; }
static inline void o_addtoscl3(double *r, const double *b, double s) { r[0]+=b[0]*s; r[1]+=b[1]*s; r[2]+=b[2]*s; }
static inline void o_cross(double *r, const double *a, const double *b) {
  double x = a[1]*b[2] - a[2]*b[1], y = a[2]*b[0] - a[0]*b[2], z = a[0]*b[1] - a[1]*b[0];
  r[0]=x; r[1]=y; r[2]=z;
}
/* mju_normalize3: returns the norm; degenerate -> (1,0,0) */
static inline double o_normalize3(double *a) {
  double n = o_norm3(a);
  if (n < O_MINVAL) { a[0]=1; a[1]=0; a[2]=0; }
  else { double s = 1.0 / n; a[0]*=s; a[1]*=s; a[2]*=s; }
  return n;
}
static inline double o_normalize(double *a, int n) {
  double nn = o_norm(a, n);
  if (nn < O_MINVAL) { a[0] = 1; for (int i = 1; i < n; i++) a[i] = 0; }
  else { double s = 1.0 / nn; for (int i = 0; i < n; i++) a[i] *= s; }
  return nn;
}
static inline double o_normalize4(double *q) {
  double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
  if (n < O_MINVAL) { q[0]=1; q[1]=0; q[2]=0; q[3]=0; }
  else if (fabs(n - 1) > O_MINVAL) { double s = 1.0 / n; q[0]*=s; q[1]*=s; q[2]*=s; q[3]*=s; }
  return n;
}
static inline double o_clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* r = m(3x3 row-major) * v */
static inline void o_mulmatvec3(double *r, const double *m, const double *v) {
  double x = m[0]*v[0] + m[1]*v[1] + m[2]*v[2];
  double y = m[3]*v[0] + m[4]*v[1] + m[5]*v[2];
  double z = m[6]*v[0] + m[7]*v[1] + m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
/* r = m^T * v */
static inline void o_mulmattvec3(double *r, const double *m, const double *v) {
  double x = m[0]*v[0] + m[3]*v[1] + m[6]*v[2];
  double y = m[1]*v[0] + m[4]*v[1] + m[7]*v[2];
  double z = m[2]*v[0] + m[5]*v[1] + m[8]*v[2];
  r[0]=x; r[1]=y; r[2]=z;
}
static inline void o_mulquat(double *r, const double *a, const double *b) {
  double w = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
  double x = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
  double y = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1];
  double z = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0];
  r[0]=w; r[1]=x; r[2]=y; r[3]=z;
}
static inline void o_negquat(double *r, const double *q) { r[0]=q[0]; r[1]=-q[1]; r[2]=-q[2]; r[3]=-q[3]; }
static inline void o_quat2mat(double *m, const double *q) {
  double q00=q[0]*q[0], q01=q[0]*q[1], q02=q[0]*q[2], q03=q[0]*q[3];
  double q11=q[1]*q[1], q12=q[1]*q[2], q13=q[1]*q[3];
  double q22=q[2]*q[2], q23=q[2]*q[3], q33=q[3]*q[3];
  m[0] = q00 + q11 - q22 - q33;  m[4] = q00 - q11 + q22 - q33;  m[8] = q00 - q11 - q22 + q33;
  m[1] = 2*(q12 - q03);  m[2] = 2*(q13 + q02);
  m[3] = 2*(q12 + q03);  m[5] = 2*(q23 - q01);
  m[6] = 2*(q13 - q02);  m[7] = 2*(q23 + q01);
}
/* rotate vector by quaternion */
static inline void o_rotvecquat(double *r, const double *v, const double *q) {
  double m[9]; o_quat2mat(m, q); o_mulmatvec3(r, m, v);
}
static inline void o_axisangle2quat(double *q, const double *axis, double angle) {
  if (angle == 0) { q[0]=1; q[1]=0; q[2]=0; q[3]=0; return; }
  double s = sin(angle * 0.5);
  q[0] = cos(angle * 0.5); q[1] = axis[0]*s; q[2] = axis[1]*s; q[3] = axis[2]*s;
}
/* mju_quatIntegrate: q <- q * exp(vel*scale/2), renormalised */
static inline void o_quatintegrate(double *q, const double *vel, double scale) {
  double ax[3] = {vel[0], vel[1], vel[2]};
  double angle = scale * o_normalize3(ax);
  double qr[4], t[4];
  o_axisangle2quat(qr, ax, angle);
  o_normalize4(q);
  o_mulquat(t, q, qr);
  q[0]=t[0]; q[1]=t[1]; q[2]=t[2]; q[3]=t[3];
  o_normalize4(q);
}
/* mju_quat2Vel */
static inline void o_quat2vel(double *r, const double *q, double dt) {
  double ax[3] = {q[1], q[2], q[3]};
  double s = o_normalize3(ax);
  double speed = 2 * atan2(s, q[0]);
  if (speed > O_PI) speed -= 2 * O_PI;
  speed /= dt;
  o_scl3(r, ax, speed);
}
/* mju_subQuat: 3D velocity that rotates qb into qa */
static inline void o_subquat(double *r, const double *qa, const double *qb) {
  double qn[4], qd[4]; o_negquat(qn, qb); o_mulquat(qd, qn, qa); o_quat2vel(r, qd, 1);
}

/* ---- spatial algebra in MuJoCo's "com-based" convention: motion = [ang(3); lin(3)],
 *      inertia = [Ixx Iyy Izz Ixy Ixz Iyz  m*dx m*dy m*dz  m] -------------------------- */
static inline void o_inertcom(double *r, const double *inert, const double *mat, const double *dif, double mass) {
  /* rotated diagonal inertia: mat * diag * mat^T */
  double t[9];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3*i+j] = mat[3*i+j] * inert[j];
  double I[9];
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
static inline void o_mulinertvec(double *r, const double *i, const double *v) {
  r[0] = i[0]*v[0] + i[3]*v[1] + i[4]*v[2] - i[8]*v[4] + i[7]*v[5];
  r[1] = i[3]*v[0] + i[1]*v[1] + i[5]*v[2] + i[8]*v[3] - i[6]*v[5];
  r[2] = i[4]*v[0] + i[5]*v[1] + i[2]*v[2] - i[7]*v[3] + i[6]*v[4];
  r[3] = i[8]*v[1] - i[7]*v[2] + i[9]*v[3];
  r[4] = i[6]*v[2] - i[8]*v[0] + i[9]*v[4];
  r[5] = i[7]*v[0] - i[6]*v[1] + i[9]*v[5];
}
/* motion cross motion */
static inline void o_crossmotion(double *r, const double *vel, const double *v) {
  double a[3], b[3], c[3];
  o_cross(a, vel, v);          /* w x v_ang */
  o_cross(b, vel, v + 3);      /* w x v_lin */
  o_cross(c, vel + 3, v);      /* vlin x v_ang */
  r[0]=a[0]; r[1]=a[1]; r[2]=a[2];
  r[3]=b[0]+c[0]; r[4]=b[1]+c[1]; r[5]=b[2]+c[2];
}
/* motion cross force */
static inline void o_crossforce(double *r, const double *vel, const double *f) {
  double a[3], b[3], c[3];
  o_cross(a, vel, f);          /* w x torque */
  o_cross(b, vel + 3, f + 3);  /* v x force */
  o_cross(c, vel, f + 3);      /* w x force */
  r[0]=a[0]+b[0]; r[1]=a[1]+b[1]; r[2]=a[2]+b[2];
  r[3]=c[0]; r[4]=c[1]; r[5]=c[2];
}
/* mju_makeFrame: frame[0:3] = normal given, frame[3:6] optional hint */
static inline void o_makeframe(double *f) {
  o_normalize3(f);
  if (o_norm3(f + 3) < 0.5) {
    f[3]=0; f[4]=0; f[5]=0;
    if (f[1] < 0.5 && f[1] > -0.5) f[4] = 1; else f[5] = 1;
  }
  double d = o_dot3(f, f + 3);
  f[3] -= f[0]*d; f[4] -= f[1]*d; f[5] -= f[2]*d;
  o_normalize3(f + 3);
  o_cross(f + 6, f, f + 3);
}
#endif
