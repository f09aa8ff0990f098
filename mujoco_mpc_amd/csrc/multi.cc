// multi.cc — one planner process driving one rollout engine per GPU (include/mjpc_hip.h, "one planner, several GPUs").
//
// The reference fans its candidates out over the threads of ONE process (mjpc/planners/sampling/planner.cc:342-380); the
// drop-in keeps that shape: the planner object lives in one host process and owns G engines.  Candidates are independent given
// (x0, nominal spline, cost) so the batch is block-partitioned with no data-path exchange; the only cross-device step is the
// elite pick, a lexicographic min over G (return, global index) pairs.
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/mjpc_hip.h"

struct MjpcHipMulti {
  std::vector<MjpcHipEngine *> eng;
  std::vector<int> off, cnt;            // shard of engine k in the last plan: [off[k], off[k] + cnt[k])
  std::vector<std::vector<double>> knots;    // per-shard local-elite knots of the last plan
  int max_samples = 0, nu = 0, nr = 0, ntr = 0, ds = 0, last_N = 0, last_H = 0, last_P = 0;
};

static thread_local std::string g_multi_error;
extern "C" {

MjpcHipMulti *mjpc_hip_multi_create(const MjpcHipModel *model, const MjpcHipTask *task, int max_samples, int max_horizon,
                                    int n_devices, const int *devices) {
  if (!model || !task || n_devices < 1 || max_samples < 1) return nullptr;
  MjpcHipMulti *m = new MjpcHipMulti();
  m->max_samples = max_samples; m->nu = model->nu; m->nr = task->num_residual; m->ntr = 3 * task->num_trace;
  m->ds = model->nq + model->nv + model->na;
  int per = (max_samples + n_devices - 1) / n_devices;
  for (int k = 0; k < n_devices; k++) {
    MjpcHipEngine *e = mjpc_hip_create(model, task, per, max_horizon, devices ? devices[k] : k);
    if (!e) { mjpc_hip_multi_destroy(m); return nullptr; }       // mjpc_hip_last_error() holds the reason
    if (n_devices > 1) mjpc_hip_set_fetch_mode(e, MJPC_FETCH_SUMMARY);
    m->eng.push_back(e);
  }
  m->off.assign(n_devices, 0); m->cnt.assign(n_devices, 0); m->knots.resize(n_devices);
  return m;
}

void mjpc_hip_multi_destroy(MjpcHipMulti *m) {
  if (!m) return;
  for (MjpcHipEngine *e : m->eng) mjpc_hip_destroy(e);
  delete m;
}

int mjpc_hip_multi_num_devices(const MjpcHipMulti *m) { return m ? (int)m->eng.size() : 0; }
MjpcHipEngine *mjpc_hip_multi_engine(MjpcHipMulti *m, int k) { return (m && k >= 0 && k < (int)m->eng.size()) ? m->eng[k] : nullptr; }

int mjpc_hip_multi_set_task(MjpcHipMulti *m, const MjpcHipTask *task) {
  if (!m) return -1;
  for (MjpcHipEngine *e : m->eng) { int rc = mjpc_hip_set_task(e, task); if (rc != 0) return rc; }
  return 0;
}

int mjpc_hip_multi_plan(MjpcHipMulti *m, const MjpcHipPlanInput *in, MjpcHipPlanOutput *out) {
  if (!m || !in || !out) return -1;
  const int G = (int)m->eng.size(), N = in->num_trajectory;
  if (G == 1) {                                           // one device: the packed single-copy path of the plain engine
    MjpcHipPlanInput one = *in;
    one.candidate_offset = 0; one.num_local = N;
    m->off[0] = 0; m->cnt[0] = N; m->last_N = N; m->last_H = in->horizon; m->last_P = in->num_spline_points;
    return mjpc_hip_plan(m->eng[0], &one, out);
  }
  if (N < 1 || N > m->max_samples) return -1;
  // block partition: the first N % G shards get one candidate more; engines without candidates sit the plan out
  const int base = N / G, extra = N % G;
  int o = 0;
  for (int k = 0; k < G; k++) { m->off[k] = o; m->cnt[k] = base + (k < extra ? 1 : 0); o += m->cnt[k]; }
  for (int k = 0; k < G; k++) {                           // enqueue every shard (asynchronous: the devices run concurrently)
    if (!m->cnt[k]) continue;
    MjpcHipPlanInput sub = *in;
    sub.candidate_offset = m->off[k]; sub.num_local = m->cnt[k];
    int rc = mjpc_hip_plan_async(m->eng[k], &sub);
    if (rc != 0) { for (int j = 0; j < k; j++) if (m->cnt[j]) { MjpcHipPlanOutput dump; memset(&dump, 0, sizeof(dump)); mjpc_hip_plan_fetch(m->eng[j], &dump); } return rc; }
  }
  const size_t PN = (size_t)in->num_spline_points * m->nu;
  int owner = -1, best_index = 0; double best_value = 0, noise_us = 0, rollouts_us = 0;
  int rc_all = 0;
  for (int k = 0; k < G; k++) {                           // summaries: returns + failure flags + local elite (value, index, knots)
    if (!m->cnt[k]) continue;
    m->knots[k].resize(PN);
    MjpcHipPlanOutput o_k;
    memset(&o_k, 0, sizeof(o_k));
    o_k.returns = out->returns ? out->returns + m->off[k] : nullptr;
    o_k.failure = out->failure ? out->failure + m->off[k] : nullptr;
    o_k.winner_knots = m->knots[k].data();
    int rc = mjpc_hip_plan_fetch(m->eng[k], &o_k);
    if (rc == -3) continue;                               // this shard has no finite return: no local winner, the others still count
    if (rc != 0) { rc_all = rc; continue; }
    noise_us = std::max(noise_us, o_k.noise_compute_time_us); rollouts_us = std::max(rollouts_us, o_k.rollouts_compute_time_us);
    // lexicographic (return, global index): shards are visited in ascending index order, so a strict < keeps the lowest index
    if (owner < 0 || o_k.winner_return < best_value) { owner = k; best_value = o_k.winner_return; best_index = o_k.winner; }
  }
  m->last_N = N; m->last_H = in->horizon; m->last_P = in->num_spline_points;
  if (rc_all != 0 || owner < 0) return rc_all != 0 ? rc_all : -3;     // -3: no shard has a finite winner (mjpc_hip_last_error says so)
  out->winner = best_index; out->winner_return = best_value;
  out->noise_compute_time_us = noise_us; out->rollouts_compute_time_us = rollouts_us;
  m->last_N = N; m->last_H = in->horizon; m->last_P = in->num_spline_points;
  // only the owner of the global winner copies a trajectory to the host
  MjpcHipPlanOutput rows = *out;
  rows.returns = nullptr; rows.failure = nullptr;
  int rc = mjpc_hip_get_candidate(m->eng[owner], best_index - m->off[owner], &rows);
  if (rc != 0) return rc;
  return 0;
}

static int owner_of(const MjpcHipMulti *m, int index) {
  for (size_t k = 0; k < m->eng.size(); k++) if (index >= m->off[k] && index < m->off[k] + m->cnt[k]) return (int)k;
  return -1;
}

int mjpc_hip_multi_get_candidate(MjpcHipMulti *m, int index, MjpcHipPlanOutput *out) {
  if (!m || !out) return -1;
  int k = owner_of(m, index);
  if (k < 0) return -1;
  int rc = mjpc_hip_get_candidate(m->eng[k], index - m->off[k], out);
  out->winner = index;
  return rc;
}

int mjpc_hip_multi_get_knots(MjpcHipMulti *m, double *knots) {
  if (!m || !knots) return -1;
  const size_t PN = (size_t)m->last_P * m->nu;
  for (size_t k = 0; k < m->eng.size(); k++) if (m->cnt[k]) { int rc = mjpc_hip_get_knots(m->eng[k], knots + (size_t)m->off[k] * PN); if (rc != 0) return rc; }
  return 0;
}

int mjpc_hip_multi_get_traces(MjpcHipMulti *m, double *traces) {
  if (!m || !traces) return -1;
  const size_t row = (size_t)m->last_H * m->ntr;
  for (size_t k = 0; k < m->eng.size(); k++) if (m->cnt[k]) { int rc = mjpc_hip_get_traces(m->eng[k], traces + (size_t)m->off[k] * row); if (rc != 0) return rc; }
  return 0;
}

}  // extern "C"
