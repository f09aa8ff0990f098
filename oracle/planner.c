/*
 * oracle/planner.c — TEST INFRASTRUCTURE ONLY (CPU oracle).
 *
 * MJPC side of the path, restated from:
 *   mjpc/spline/spline.cc:103-156,240-277   TimeSpline::Sample / CubicCoefficients / Slope
 *   mjpc/planners/sampling/policy.cc:52-59  SamplingPolicy::Action (sample + Clamp)
 *   mjpc/planners/sampling/planner.cc:313-339  AddNoiseToPolicy
 *   mjpc/planners/sampling/planner.cc:342-380  Rollouts (ThreadPool fan-out)
 *   mjpc/trajectory.cc:100-210,312-326      NoisyRollout, UpdateReturn
 *   mjpc/planners/sampling/planner.cc:168-181  partial_sort -> winner (lowest index on ties)
 *   mjpc/threadpool.cc:30-85                FIFO pool, one mjData per worker
 * The reference's absl::BitGen (planner.cc:318) is unseedable; noise here is an explicit
 * tensor or Philox4x32-10 + Box-Muller, the same definition the device uses.
 */
#include <stdlib.h>
#include <stdio.h>
#include <pthread.h>
#include <time.h>
#include "oracle.h"
#include "omath.h"

/* ---- TimeSpline::Sample ---------------------------------------------------------------- */
static double slope(const double *times, const double *values, int P, int dim, int node, int k) {   /* spline.cc:259-277 */
  if (node == 0)
    return (values[dim + k] - values[k]) / (times[1] - times[0]);
  if (node == P - 1)
    return (values[node * dim + k] - values[(node - 1) * dim + k]) / (times[node] - times[node - 1]);
  return 0.5 * (values[(node + 1) * dim + k] - values[node * dim + k]) / (times[node + 1] - times[node]) +
         0.5 * (values[node * dim + k] - values[(node - 1) * dim + k]) / (times[node] - times[node - 1]);
}

void oracle_spline_sample(const double *times, const double *values, int P, int dim, int interp, double time, double *out) {
  if (P == 0) { for (int i = 0; i < dim; i++) out[i] = 0.0; return; }
  int upper = 0;                                  /* std::upper_bound */
  while (upper < P && !(time < times[upper])) upper++;
  if (upper == P) { for (int i = 0; i < dim; i++) out[i] = values[(P - 1) * dim + i]; return; }
  if (upper == 0) { for (int i = 0; i < dim; i++) out[i] = values[i]; return; }
  int lower = upper - 1;
  double t = (time - times[lower]) / (times[upper] - times[lower]);
  switch (interp) {
    case MJPC_SPLINE_ZERO:
      for (int i = 0; i < dim; i++) out[i] = values[lower * dim + i];
      return;
    case MJPC_SPLINE_LINEAR:
      for (int i = 0; i < dim; i++) out[i] = values[lower * dim + i] * (1 - t) + values[upper * dim + i] * t;
      return;
    default: {
      double lo = times[lower], up = times[upper];
      double c0 = 2.0 * t*t*t - 3.0 * t*t + 1.0;
      double c1 = (t*t*t - 2.0 * t*t + t) * (up - lo);
      double c2 = -2.0 * t*t*t + 3 * t*t;
      double c3 = (t*t*t - t*t) * (up - lo);
      for (int i = 0; i < dim; i++) {
        double p0 = values[lower * dim + i];
        double m0 = slope(times, values, P, dim, lower, i);
        double m1 = slope(times, values, P, dim, upper, i);
        double p1 = values[upper * dim + i];
        out[i] = c0 * p0 + c1 * m0 + c2 * p1 + c3 * m1;
      }
    }
  }
}

/* ---- Philox4x32-10 + Box-Muller -------------------------------------------------------- */
void oracle_philox(uint64_t seed, uint64_t stream, uint32_t c0, uint32_t c1, uint32_t out[4]) {
  uint32_t c[4] = {c0, c1, (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}
static double philox_normal(uint64_t seed, uint64_t stream, uint32_t i, uint32_t e) {
  uint32_t o[4];
  oracle_philox(seed, stream, i, e, o);
  uint64_t x1 = ((uint64_t)o[0] << 32) | o[1], x2 = ((uint64_t)o[2] << 32) | o[3];
  double u1 = (double)((x1 >> 11) + 1) * (1.0 / 9007199254740992.0);
  double u2 = (double)(x2 >> 11) * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cos(2.0 * O_PI * u2);
}
/* eps[n*P*nu] (row r = candidate i0+r), sel[n] (1 => second std, prob 0.2 when sigma2 > 0) */
void oracle_noise(uint64_t seed, uint64_t stream, int i0, int n, int P, int nu, double sigma2, double *eps, int *sel) {
  for (int r = 0; r < n; r++) {
    uint32_t i = (uint32_t)(i0 + r);
    for (int e = 0; e < P * nu; e++) eps[(size_t)r * P * nu + e] = philox_normal(seed, stream, i, (uint32_t)e);
    if (sel) {
      uint32_t o[4];
      oracle_philox(seed, stream, i, 0xFFFFFFFFu, o);
      uint64_t x = ((uint64_t)o[0] << 32) | o[1];
      double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
      sel[r] = (sigma2 > 0 && u < 0.2) ? 1 : 0;
    }
  }
}

/* ---- one rollout ----------------------------------------------------------------------- */
static void get_trace(const OModel *om, const OData *d, double *trace) {   /* utilities.cc:250-267 */
  for (int i = 0; i < om->t.num_trace; i++) {
    int id = om->t.trace_objid[i];
    const double *src;
    switch (om->t.trace_objtype[i]) {
      case MJPC_OBJ_SITE: src = d->site_xpos + 3 * id; break;
      case MJPC_OBJ_GEOM: src = d->geom_xpos + 3 * id; break;
      case MJPC_OBJ_BODY: src = d->xipos + 3 * id; break;
      default: src = d->xpos + 3 * id;
    }
    o_copy3(trace + 3 * i, src);
  }
}

#define XFRC_STREAM 0x5846524300000000ull    /* "XFRC": separates the force noise from the knot noise of the same plan */
void oracle_rollout(const OModel *om, OData *d, const MjpcHipPlanInput *in, const double *knots, int row, OPlanOutput *out) {
  const MjpcHipModel *m = &om->m;
  const MjpcHipTask *t = &om->t;
  int nq = m->nq, nv = m->nv, na = m->na, nu = m->nu, H = in->horizon, P = in->num_spline_points;
  int ds = nq + nv + na, nr = t->num_residual, ntr = 3 * t->num_trace;
  double *states = out->states + (size_t)row * H * ds, *actions = out->actions + (size_t)row * H * nu;
  double *times = out->times + (size_t)row * H, *residual = out->residual + (size_t)row * H * nr;
  double *costs = out->costs + (size_t)row * H, *trace = out->trace + (size_t)row * H * ntr;
  int failure = 0;
  for (int i = 0; i < m->nmocap; i++) {
    o_copy3(d->mocap_pos + 3 * i, in->mocap + 7 * i);
    o_copy(d->mocap_quat + 4 * i, in->mocap + 7 * i + 3, 4);
  }
  if (m->nuserdata) o_copy(d->userdata, in->userdata, m->nuserdata);
  o_copy(states, in->state, ds);
  o_copy(d->qpos, in->state, nq);
  o_copy(d->qvel, in->state + nq, nv);
  times[0] = in->time;
  d->time = in->time;
  d->warning = 0;
  o_zero(d->qacc_warmstart, nv);     /* deterministic warm start */
  d->xfrc_on = in->xfrc_std > 0;
  o_zero(d->xfrc_applied, 6 * m->nbody);
  for (int s = 0; s < H - 1; s++) {
    oracle_spline_sample(in->knot_times, knots, P, nu, in->interpolation, d->time, actions + s * nu);
    for (int k = 0; k < nu; k++)
      actions[s * nu + k] = o_clip(actions[s * nu + k], m->actuator_ctrlrange[2 * k], m->actuator_ctrlrange[2 * k + 1]);
    o_copy(d->ctrl, actions + s * nu, nu);
    if (in->xfrc_std > 0) {            /* trajectory.cc:147-155: Ornstein-Uhlenbeck in discrete time */
      double rate = exp(-m->timestep / in->xfrc_rate), scale = in->xfrc_std * sqrt(1 - rate * rate);
      uint32_t gi = (uint32_t)(in->candidate_offset + row);
      for (int i = 0; i < 6 * m->nbody; i++)
        d->xfrc_applied[i] = rate * d->xfrc_applied[i] +
                             scale * philox_normal(in->seed, in->stream ^ XFRC_STREAM, gi, (uint32_t)(s * 6 * m->nbody + i));
    }
    oracle_step(om, d);
    o_copy(residual + s * nr, d->sensordata, nr);
    get_trace(om, d, trace + s * ntr);
    if (d->warning) { failure = 1; break; }
    o_copy(states + (s + 1) * ds, d->qpos, nq);
    o_copy(states + (s + 1) * ds + nq, d->qvel, nv);
    times[s + 1] = d->time;
  }
  out->unsupported += d->unsupported; d->unsupported = 0;
  if (failure) { out->failure[row] = 1; out->returns[row] = MJPC_MAX_RETURN; return; }
  if (H > 1) o_copy(actions + (H - 1) * nu, actions + (H - 2) * nu, nu);
  else o_zero(actions + (H - 1) * nu, nu);
  oracle_forward(om, d);
  o_copy(residual + (H - 1) * nr, d->sensordata, nr);
  get_trace(om, d, trace + (H - 1) * ntr);
  /* UpdateReturn, trajectory.cc:312-326 */
  double total = 0;
  for (int s = 0; s < H; s++) {
    costs[s] = oracle_cost_value(t, residual + s * nr, NULL);
    total += costs[s];
  }
  total /= (H > 1 ? H : 1);
  out->returns[row] = total;
  out->failure[row] = 0;
}

/* ---- plan step with a FIFO pool -------------------------------------------------------- */
typedef struct {
  const OModel *om; const MjpcHipPlanInput *in; OPlanOutput *out;
  const double *eps; const int *sel;
  int next; int unsupported;
  pthread_mutex_t mtx;
} PlanJob;

static void make_candidate_knots(const OModel *om, const MjpcHipPlanInput *in, int i, const double *eps, const int *sel, double *knots) {
  const MjpcHipModel *m = &om->m;
  int P = in->num_spline_points, nu = m->nu;
  o_copy(knots, in->knot_values, P * nu);
  if (in->candidate_knots) { o_copy(knots, in->candidate_knots + (size_t)i * P * nu, P * nu); return; }   /* robust planner: explicit policies */
  if (i == in->nominal_index) return;              /* planner.cc:361 (index 0); cross_entropy/planner.cc:412 (extra rollout) */
  if (in->noise_std) {                             /* cross_entropy/planner.cc:340-375: absolute per-parameter std */
    for (int p = 0; p < P; p++) {
      for (int k = 0; k < nu; k++) knots[p * nu + k] += in->noise_std[p * nu + k] * eps[((size_t)i * P + p) * nu + k];
      for (int k = 0; k < nu; k++)
        knots[p * nu + k] = o_clip(knots[p * nu + k], m->actuator_ctrlrange[2 * k], m->actuator_ctrlrange[2 * k + 1]);
    }
    return;
  }
  double std = in->noise_exploration[0];
  if (in->noise_exploration[1] > 0 && sel && sel[i]) std = in->noise_exploration[1];
  for (int p = 0; p < P; p++) {
    for (int k = 0; k < nu; k++) {
      double scale = 0.5 * (m->actuator_ctrlrange[2 * k + 1] - m->actuator_ctrlrange[2 * k]);
      double noise = (scale * std) * eps[((size_t)i * P + p) * nu + k];
      knots[p * nu + k] += noise;
    }
    for (int k = 0; k < nu; k++)
      knots[p * nu + k] = o_clip(knots[p * nu + k], m->actuator_ctrlrange[2 * k], m->actuator_ctrlrange[2 * k + 1]);
  }
}

static void *worker(void *arg) {
  PlanJob *job = (PlanJob *)arg;
  struct timespec w0, w1, c0, c1; int dbg = getenv("ORACLE_DEBUG_THREADS") != NULL; int cnt = 0;
  if (dbg) { clock_gettime(CLOCK_MONOTONIC, &w0); clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c0); }
  OData *d = oracle_make_data(job->om);
  int P = job->in->num_spline_points, nu = job->om->m.nu;
  OPlanOutput local = *job->out;
  local.unsupported = 0;
  for (;;) {
    pthread_mutex_lock(&job->mtx);
    int r = job->next++;
    pthread_mutex_unlock(&job->mtx);
    if (r >= job->in->num_local) break;
    int i = job->in->candidate_offset + r;
    double *knots = job->out->knots + (size_t)r * P * nu;
    make_candidate_knots(job->om, job->in, i, job->eps, job->sel, knots);
    oracle_rollout(job->om, d, job->in, knots, r, &local); cnt++;
  }
  if (dbg) { clock_gettime(CLOCK_MONOTONIC, &w1); clock_gettime(CLOCK_THREAD_CPUTIME_ID, &c1);
    fprintf(stderr, "worker: %d rollouts wall %.3f cpu %.3f\n", cnt, (w1.tv_sec-w0.tv_sec)+(w1.tv_nsec-w0.tv_nsec)*1e-9, (c1.tv_sec-c0.tv_sec)+(c1.tv_nsec-c0.tv_nsec)*1e-9); }
  pthread_mutex_lock(&job->mtx);
  job->unsupported += local.unsupported;
  pthread_mutex_unlock(&job->mtx);
  oracle_free_data(d);
  return NULL;
}

int oracle_plan(const OModel *om, const MjpcHipPlanInput *in, OPlanOutput *out, int nthreads) {
  int N = in->num_trajectory, P = in->num_spline_points, nu = om->m.nu;
  double *eps_own = NULL; int *sel_own = NULL;
  const double *eps = in->noise_eps; const int *sel = in->noise_sel;
  if (!eps) {
    eps_own = (double *)malloc(sizeof(double) * (size_t)N * P * nu + 8);
    sel_own = (int *)malloc(sizeof(int) * (size_t)N + 8);
    oracle_noise(in->seed, in->stream, 0, N, P, nu, in->noise_exploration[1], eps_own, sel_own);
    eps = eps_own; sel = sel_own;
  }
  PlanJob job;
  job.om = om; job.in = in; job.out = out; job.eps = eps; job.sel = sel; job.next = 0; job.unsupported = 0;
  pthread_mutex_init(&job.mtx, NULL);
  if (nthreads < 1) nthreads = 1;
  if (nthreads == 1) worker(&job);
  else {
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
    for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &job);
    for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
    free(th);
  }
  pthread_mutex_destroy(&job.mtx);
  out->unsupported = job.unsupported;
  /* winner: first minimum (partial_sort with '<' keeps the lowest index on ties) */
  int w = 0;
  for (int r = 1; r < in->num_local; r++) if (out->returns[r] < out->returns[w]) w = r;
  out->winner = in->candidate_offset + w;
  free(eps_own); free(sel_own);
  return 0;
}
