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
  if (na > 0) o_copy(d->act, in->state + nq + nv, na);
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
    if (na > 0) o_copy(states + (s + 1) * ds + nq + nv, d->act, na);
    times[s + 1] = d->time;
  }
  if (failure) { out->unsupported += d->unsupported; d->unsupported = 0; out->failure[row] = d->warning; out->returns[row] = MJPC_MAX_RETURN; return; }
  if (H > 1) o_copy(actions + (H - 1) * nu, actions + (H - 2) * nu, nu);
  else o_zero(actions + (H - 1) * nu, nu);
  oracle_forward(om, d);
  o_copy(residual + (H - 1) * nr, d->sensordata, nr);
  get_trace(om, d, trace + (H - 1) * ntr);
  out->unsupported += d->unsupported; d->unsupported = 0;
  /* warnings of the terminal forward pass fail the candidate too (the reference leaves them in mjData for the worker's next
   * rollout to trip over, trajectory.cc:183-206: scheduling-dependent; here the definition is per candidate, like the device) */
  if (d->warning) { out->failure[row] = d->warning; out->returns[row] = MJPC_MAX_RETURN; return; }
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

/* ---- plan step on a persistent FIFO pool (mjpc/threadpool.cc:30-85) ------------------------
 * The reference creates its workers once (ThreadPool ctor, threadpool.cc:30-49), every worker owns one mjData for
 * its whole life (Planner::ResizeMjData, planners/planner.cc:23-33; indexed by ThreadPool::WorkerId()),
 * Schedule() pushes one closure per candidate on a mutex-protected queue (threadpool.cc:51-60), the workers pop in
 * FIFO order (threadpool.cc:62-85) and the planner blocks in WaitCount() until all N have run
 * (sampling/planner.cc:379).  OPool is that: the threads and their OData arenas survive across plan steps. */
struct OPool {
  const OModel *om;
  int nthreads;
  pthread_t *th;
  OData **data;               /* one arena per worker, reused across rollouts and plan steps */
  double **eps_row;           /* per-worker scratch for the candidate's noise row */
  pthread_mutex_t mtx;
  pthread_cond_t cv_job, cv_done;
  const MjpcHipPlanInput *in; OPlanOutput *out;
  int next, total, done, shutdown, started, unsupported;
};
typedef struct { OPool *pool; int id; } PoolArg;

static void candidate_knots(const OModel *om, const MjpcHipPlanInput *in, int i, const double *eps_row, int sel_i, double *knots) {
  const MjpcHipModel *m = &om->m;
  int P = in->num_spline_points, nu = m->nu;
  o_copy(knots, in->knot_values, P * nu);
  if (in->candidate_knots) { o_copy(knots, in->candidate_knots + (size_t)i * P * nu, P * nu); return; }   /* robust planner: explicit policies */
  if (i == in->nominal_index) return;              /* planner.cc:361 (index 0); cross_entropy/planner.cc:412 (extra rollout) */
  if (in->noise_std) {                             /* cross_entropy/planner.cc:340-375: absolute per-parameter std */
    for (int p = 0; p < P; p++) {
      for (int k = 0; k < nu; k++) knots[p * nu + k] += in->noise_std[p * nu + k] * eps_row[p * nu + k];
      for (int k = 0; k < nu; k++)
        knots[p * nu + k] = o_clip(knots[p * nu + k], m->actuator_ctrlrange[2 * k], m->actuator_ctrlrange[2 * k + 1]);
    }
    return;
  }
  double std = in->noise_exploration[0];
  if (in->noise_exploration[1] > 0 && sel_i) std = in->noise_exploration[1];
  for (int p = 0; p < P; p++) {
    for (int k = 0; k < nu; k++) {
      double scale = 0.5 * (m->actuator_ctrlrange[2 * k + 1] - m->actuator_ctrlrange[2 * k]);
      double noise = (scale * std) * eps_row[p * nu + k];
      knots[p * nu + k] += noise;
    }
    for (int k = 0; k < nu; k++)
      knots[p * nu + k] = o_clip(knots[p * nu + k], m->actuator_ctrlrange[2 * k], m->actuator_ctrlrange[2 * k + 1]);
  }
}

/* one queued closure of SamplingPlanner::Rollouts (planner.cc:352-376): copy the nominal policy, add noise, roll out */
static void run_candidate(OPool *pl, int id, int r) {
  const MjpcHipPlanInput *in = pl->in;
  int P = in->num_spline_points, nu = pl->om->m.nu;
  int i = in->candidate_offset + r;
  const double *row; int sel_i = 0;
  if (in->noise_eps) { row = in->noise_eps + (size_t)i * P * nu; sel_i = in->noise_sel ? in->noise_sel[i] : 0; }
  else { oracle_noise(in->seed, in->stream, i, 1, P, nu, in->noise_exploration[1], pl->eps_row[id], &sel_i); row = pl->eps_row[id]; }
  double *knots = pl->out->knots + (size_t)r * P * nu;
  candidate_knots(pl->om, in, i, row, sel_i, knots);
  OPlanOutput local = *pl->out;
  local.unsupported = 0;
  oracle_rollout(pl->om, pl->data[id], in, knots, r, &local);
  if (local.unsupported) { pthread_mutex_lock(&pl->mtx); pl->unsupported += local.unsupported; pthread_mutex_unlock(&pl->mtx); }
}

static void *pool_worker(void *arg) {
  PoolArg *pa = (PoolArg *)arg;
  OPool *pl = pa->pool; int id = pa->id;
  free(pa);
  pthread_mutex_lock(&pl->mtx);
  pl->started++;
  pthread_cond_broadcast(&pl->cv_done);
  for (;;) {
    while (!pl->shutdown && pl->next >= pl->total) pthread_cond_wait(&pl->cv_job, &pl->mtx);
    if (pl->shutdown) break;
    int r = pl->next++;                              /* pop the head of the queue */
    pthread_mutex_unlock(&pl->mtx);
    run_candidate(pl, id, r);
    pthread_mutex_lock(&pl->mtx);
    if (++pl->done == pl->total) pthread_cond_broadcast(&pl->cv_done);
  }
  pthread_mutex_unlock(&pl->mtx);
  return NULL;
}

OPool *oracle_pool_create(const OModel *om, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  OPool *pl = (OPool *)calloc(1, sizeof(OPool));
  pl->om = om; pl->nthreads = nthreads;
  pl->data = (OData **)calloc((size_t)nthreads, sizeof(OData *));
  pl->eps_row = (double **)calloc((size_t)nthreads, sizeof(double *));
  for (int i = 0; i < nthreads; i++) {
    pl->data[i] = oracle_make_data(om);
    pl->eps_row[i] = (double *)malloc(sizeof(double) * (size_t)(MJPC_MAX_HORIZON * (om->m.nu + 1)));
  }
  pthread_mutex_init(&pl->mtx, NULL);
  pthread_cond_init(&pl->cv_job, NULL); pthread_cond_init(&pl->cv_done, NULL);
  if (nthreads > 1) {                                 /* one thread: the caller runs the queue itself */
    pl->th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    for (int i = 0; i < nthreads; i++) {
      PoolArg *pa = (PoolArg *)malloc(sizeof(PoolArg)); pa->pool = pl; pa->id = i;
      pthread_create(&pl->th[i], NULL, pool_worker, pa);
    }
    pthread_mutex_lock(&pl->mtx);
    while (pl->started < nthreads) pthread_cond_wait(&pl->cv_done, &pl->mtx);
    pthread_mutex_unlock(&pl->mtx);
  }
  return pl;
}

void oracle_pool_destroy(OPool *pl) {
  if (!pl) return;
  if (pl->th) {
    pthread_mutex_lock(&pl->mtx); pl->shutdown = 1; pthread_cond_broadcast(&pl->cv_job); pthread_mutex_unlock(&pl->mtx);
    for (int i = 0; i < pl->nthreads; i++) pthread_join(pl->th[i], NULL);
    free(pl->th);
  }
  for (int i = 0; i < pl->nthreads; i++) { oracle_free_data(pl->data[i]); free(pl->eps_row[i]); }
  free(pl->data); free(pl->eps_row);
  pthread_mutex_destroy(&pl->mtx); pthread_cond_destroy(&pl->cv_job); pthread_cond_destroy(&pl->cv_done);
  free(pl);
}

int oracle_pool_threads(const OPool *pl) { return pl ? pl->nthreads : 0; }

/* one plan step: Schedule N closures, WaitCount(N), pick the winner */
int oracle_pool_plan(OPool *pl, const MjpcHipPlanInput *in, OPlanOutput *out) {
  if (in->num_spline_points > MJPC_MAX_HORIZON) return -1;
  pthread_mutex_lock(&pl->mtx);
  pl->in = in; pl->out = out; pl->unsupported = 0; pl->done = 0; pl->next = 0;
  if (pl->th) {
    pl->total = in->num_local;
    pthread_cond_broadcast(&pl->cv_job);
    while (pl->done < pl->total) pthread_cond_wait(&pl->cv_done, &pl->mtx);
    pl->total = 0; pl->next = 0;
    pthread_mutex_unlock(&pl->mtx);
  } else {
    pthread_mutex_unlock(&pl->mtx);
    for (int r = 0; r < in->num_local; r++) run_candidate(pl, 0, r);
  }
  out->unsupported = pl->unsupported;
  /* winner: first minimum (partial_sort with '<' keeps the lowest index on ties) */
  int w = 0;
  for (int r = 1; r < in->num_local; r++) if (out->returns[r] < out->returns[w]) w = r;
  out->winner = in->candidate_offset + w;
  return 0;
}

/* convenience for the tests: a pool that lives for one plan step */
int oracle_plan(const OModel *om, const MjpcHipPlanInput *in, OPlanOutput *out, int nthreads) {
  OPool *pl = oracle_pool_create(om, nthreads);
  int rc = oracle_pool_plan(pl, in, out);
  oracle_pool_destroy(pl);
  return rc;
}
