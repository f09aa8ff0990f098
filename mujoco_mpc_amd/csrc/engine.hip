// engine.hip — HIP kernels + C ABI (include/mjpc_hip.h) of the Predictive-Sampling rollout engine.
//
// Kernels (gfx950, wave64):
//   noise_kernel    Philox4x32-10 + Box-Muller standard normals  eps[nlocal, P, nu]
//   rollout_kernel  (rollout_cached.hip / rollout_direct.hip)  ONE WORKGROUP PER CANDIDATE (grid = nlocal blocks x 64*MJPC_WAVES threads: an owner wave on the
//                   critical path + helper / side waves on the CU's other SIMDs, see spmd.h); the candidate's
//                   whole mjData-equivalent lives in dynamic LDS for all H steps; HBM traffic is only
//                   the Trajectory record (coalesced row writes by the owning wave) + model reads
//                   that hit L2 / the scalar cache.  Replaces planner.cc:342-380 + trajectory.cc:100-210.
//   argmin_kernel   wavefront (value, index) min-reduction, lowest index wins ties
//                   (planner.cc:168-181 partial_sort -> trajectory_order[0]).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "model.h"
#include "spmd.h"
#include "philox.h"
#include "host.h"
#include "../../include/mjpc_hip_debug.h"

// ------------------------------------------------------------------------------ kernels
// the rollout kernels live in their own translation units (rollout_cached.hip, rollout_direct.hip; see rollout_tu.inc)
typedef void (*RolloutFn)(const KParams);
extern "C" RolloutFn mjpc_pick_rollout_cached(int nv, int *exact);
extern "C" RolloutFn mjpc_pick_rollout_direct(int nv, int *exact);
extern "C" RolloutFn mjpc_pick_rollout_dense2(int nv, int *exact);
extern "C" RolloutFn mjpc_pick_rollout_dense2h(int nv, int *exact);
extern "C" int mjpc_rollout_threads_cached(void);

// Capacity tiers.  One candidate per CU leaves every SIMD with a single, mostly stalled wave; two resident workgroups per CU
// raise throughput ~1.6x once a shard has more candidates than CUs, but need <= 80 KiB of LDS each.  The dense tier gets
// there with a smaller contact / constraint-row capacity than the model asks for (and, to afford 100 rows / 28 contacts, reads the spline knots and the Hessian entry table from L2); a candidate that overflows it is flagged
// (MJPC_WARN_CONTACTFULL / CNSTRFULL) and the full-capacity kernel re-runs exactly those candidates right behind it on
// the stream (all other workgroups of that launch exit at once).  Rollouts are deterministic and independent, so the result
// is the same as running everything at full capacity: no candidate is lost to the smaller buffers.
#define TIERB_NEFCMAX 128    // first capacity tried for the dense tier (rows); contacts = rows / 4 + 2
#define TIERB_NEFCMIN 40
#define TIERB_LDS_LIMIT (80 * 1024)
#define TIERB_HOT_KEEP 75       // per cent of the plain dense tier's rows the hot-cached variant must still hold to be preferred

// eps[r, e] for global candidate (offset + r), element e = p*nu + k; sel[r] = second-std choice
extern "C" __global__ void noise_kernel(double *eps, int *sel, unsigned long long seed, unsigned long long stream,
                                        int offset, int nlocal, int PN, double sigma1) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)nlocal * PN;
  if (idx < total) {
    int r = (int)(idx / PN), e = (int)(idx - (size_t)r * PN);
    eps[idx] = philox_normal(seed, stream, (unsigned)(offset + r), (unsigned)e);
  }
  if (idx < (size_t)nlocal) {
    unsigned o[4];
    philox4x32_10(seed, stream, (unsigned)(offset + (int)idx), 0xFFFFFFFFu, o);
    unsigned long long x = ((unsigned long long)o[0] << 32) | o[1];
    double u = (double)(x >> 11) * (1.0 / 9007199254740992.0);
    sel[idx] = (sigma1 > 0 && u < 0.2) ? 1 : 0;
  }
}

// winner[0] = local index of the first minimum of returns[0..n), winner_val[0] = its value
extern "C" __global__ void __launch_bounds__(64) argmin_kernel(const double *returns, int n, int *winner, double *winner_val) {
  int lane = threadIdx.x;
  double best = 1.0e300; int bi = 0x7fffffff;
  for (int i = lane; i < n; i += 64) { double v = returns[i]; if (v < best) { best = v; bi = i; } }   // strict <: keeps the lowest index per lane
  for (int o = 32; o > 0; o >>= 1) {
    double ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
    if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { winner[0] = bi; winner_val[0] = best; }
}

// gathers everything the host wants after a plan step into ONE contiguous buffer (one D2H copy instead of eleven):
//   [winner index, winner return | returns[nl] | failure[nl] (as doubles) | winner knots | winner rows: states, actions, times,
//    residual, costs, trace]      (the rows only when a.rows: a shard of a multi-device plan sends the summary alone and the
//    owner of the global winner fetches its rows afterwards)
struct PackArgs {
  const int *winner; const double *winner_val, *returns; const int *failure;
  const double *states, *actions, *times, *residual, *costs, *trace, *knots;
  int nl, H, P, ds, nu, nr, ntr, rows;
  double *dst;
};
extern "C" __global__ void __launch_bounds__(256) pack_kernel(const PackArgs a) {
  int w = a.winner[0];
  int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
  double *d = a.dst;
  if (tid == 0) { d[0] = (double)w; d[1] = a.winner_val[0]; }
  d += 2;
  for (int i = tid; i < a.nl; i += nth) { d[i] = a.returns[i]; d[a.nl + i] = (double)a.failure[i]; }
  d += 2 * a.nl;
  if (w < 0 || w >= a.nl) return;
  size_t H = (size_t)a.H, r = (size_t)w;
  const double *src[7] = {a.knots + r * (size_t)a.P * a.nu, a.states + r * H * a.ds, a.actions + r * H * a.nu, a.times + r * H,
                          a.residual + r * H * a.nr, a.costs + r * H, a.trace + r * H * a.ntr};
  size_t cnt[7] = {(size_t)a.P * a.nu, H * a.ds, H * a.nu, H, H * a.nr, H, H * a.ntr};
  for (int k = 0; k < (a.rows ? 7 : 1); k++) {
    for (size_t i = tid; i < cnt[k]; i += nth) d[i] = src[k][i];
    d += cnt[k];
  }
}

// ------------------------------------------------------------------------------ host side
static thread_local std::string g_error;
static void set_error(const std::string &s) { g_error = s; }

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return -2; } } while (0)
// inside mjpc_hip_create only: `e` (when already allocated) and everything it owns are released on the error path
#define HIPCHKP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_error(std::string(#x) + ": " + hipGetErrorString(e_)); mjpc_hip_destroy(e); return nullptr; } } while (0)

struct MjpcHipEngine {
  int device = 0;
  PackedModel pm;
  int *d_ib = nullptr; double *d_db = nullptr;
  KParams K;
  int max_local = 0, max_horizon = 0, P_max = 0;
  int nq = 0, nv = 0, nu = 0, nmocap = 0, nr = 0, ntr = 0, ds = 0, nuserdata = 0;
  double *d_userdata = nullptr;      // mjData.userdata of the plan's state (State::CopyTo, states/state.cc:128-135): carried for residuals that read it
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  // device buffers
  double *d_state = nullptr, *d_mocap = nullptr, *d_kt = nullptr, *d_kv = nullptr, *d_eps = nullptr, *d_std = nullptr, *d_cand = nullptr; size_t cand_cap = 0;
  int *d_sel = nullptr;
  size_t eps_cap = 0;
  double *d_states = nullptr, *d_actions = nullptr, *d_times = nullptr, *d_residual = nullptr, *d_costs = nullptr,
         *d_trace = nullptr, *d_knots = nullptr, *d_returns = nullptr, *d_winner_val = nullptr;
  size_t knots_cap = 0;
  int *d_failure = nullptr, *d_diag = nullptr, *d_winner = nullptr;
  long long *d_prof = nullptr;
  double *d_frame = nullptr; int nbody = 0, nsite = 0;
  double *d_ckpt = nullptr; int ckpt_stride = 0;          // dense-tier checkpoints for the retry launch
  // pinned host staging
  double *d_pack = nullptr, *h_pack = nullptr; size_t pack_cap = 0;   // packed plan result (device / pinned host)
  double *h_small = nullptr;   // state | mocap | knot_times | knot_values | noise_std
  void *h_task[2] = {nullptr, nullptr}; hipEvent_t ev_task[2] = {nullptr, nullptr}; int task_slot = 0;   // set_task staging
  // last plan
  int last_H = 0, last_P = 0, last_nlocal = 0, last_offset = 0, pending = 0;
  // kernel timing accumulation
  double acc_rollout_us = 0, acc_total_us = 0; int acc_n = 0;
  size_t lds_bytes = 0;
  RolloutFn kernel = nullptr; bool cached = true;
  int fault = 0;               // diagnostics knob fault_inject (mjpc_hip_debug.h; test-suite only)
  int summary_only = 0, last_summary = 0;      // mjpc_hip_set_fetch_mode
  int last_dense = 0;
  // dense tier (two workgroups per CU), see "Capacity tiers" above
  RolloutFn kernelB = nullptr; Lay layB; int nefcB = 0, nconB = 0, cacheB_i = 0, cacheB_d = 0; size_t ldsB = 0; int num_cu = 256, force_tier = 0;
};

static int upload_model(MjpcHipEngine *e) {
  HIPCHK(hipMemcpy(e->d_ib, e->pm.ib.data(), e->pm.ib.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(e->d_db, e->pm.db.data(), e->pm.db.size() * sizeof(double), hipMemcpyHostToDevice));
  e->K.M = mjpc_host::relocate(e->pm, e->d_ib, e->d_db);
  e->K.L = e->pm.L;
  e->K.ibase = e->d_ib; e->K.dbase = e->d_db; e->K.cache_i = (int)e->pm.cache_i; e->K.cache_d = (int)e->pm.cache_d;
  return 0;
}

extern "C" {
void mjpc_hip_destroy(MjpcHipEngine *e);

const char *mjpc_hip_last_error(void) { return g_error.c_str(); }
void mjpc_hip_debug_set(const char *name, const char *value) { if (name) mjpc_host::debug_set(name, value); }
int mjpc_hip_version(void) { return MJPC_HIP_ABI_VERSION; }
int mjpc_hip_sizeof_model(void) { return (int)sizeof(MjpcHipModel); }
int mjpc_hip_sizeof_task(void) { return (int)sizeof(MjpcHipTask); }
int mjpc_hip_sizeof_plan_input(void) { return (int)sizeof(MjpcHipPlanInput); }
int mjpc_hip_sizeof_plan_output(void) { return (int)sizeof(MjpcHipPlanOutput); }

MjpcHipEngine *mjpc_hip_create(const MjpcHipModel *model, const MjpcHipTask *task, int max_local, int max_horizon, int device) {
  if (!model || !task || max_local < 1 || max_horizon < 1 || max_horizon > MJPC_MAX_HORIZON) { set_error("mjpc_hip_create: invalid argument"); return nullptr; }
  if (model->struct_size != (int)sizeof(MjpcHipModel) || task->struct_size != (int)sizeof(MjpcHipTask)) {
    set_error("mjpc_hip_create: MjpcHipModel / MjpcHipTask struct_size does not match this library (ABI revision " + std::to_string(MJPC_HIP_ABI_VERSION) +
              ": header and library out of step, or struct_size not set)");
    return nullptr;
  }
  int ndev = 0;
  MjpcHipEngine *e = nullptr;
  HIPCHKP(hipGetDeviceCount(&ndev));
  if (ndev < 1 || device < 0 || device >= ndev) { set_error("mjpc_hip_create: no such HIP device"); return nullptr; }
  HIPCHKP(hipSetDevice(device));
  e = new MjpcHipEngine();
  e->device = device;
  e->P_max = 36;     // MaxSamplingSplinePoints (mjpc/planners/sampling/planner.h:35-36)
  // kernel flavour: tables cached in LDS when that fits next to the candidate's state and a compile-time-nv kernel exists
  // there; otherwise the flavour that reads them from HBM / L2
  {
    int exact_c = 0, exact_d = 0;
    RolloutFn kc = mjpc_pick_rollout_cached(model->nv, &exact_c), kd = mjpc_pick_rollout_direct(model->nv, &exact_d);
    bool use_cache = !(exact_d && !exact_c);
    if (mjpc_host::debug_knob("no_model_cache")) use_cache = false;     // diagnostics knob (mjpc_hip_debug.h)
    // (the flavour without the whole copy still keeps the hot prefix - kinematic / tree tables - in LDS: hot_only)
    // (a compile-time-nv kernel solves with the Hessian in registers: its layout has no scaled-row table, host.h)
    if (!mjpc_host::build(e->pm, model, task, e->P_max, use_cache, false, !use_cache, use_cache ? exact_c != 0 : exact_d != 0)) { set_error("mjpc_hip_create: " + e->pm.error); delete e; return nullptr; }
    if (use_cache && (size_t)e->pm.L.total_doubles * sizeof(double) > 160 * 1024) {
      use_cache = false;
      if (!mjpc_host::build(e->pm, model, task, e->P_max, false, false, true, exact_d != 0)) { set_error("mjpc_hip_create: " + e->pm.error); delete e; return nullptr; }
    }
    e->kernel = use_cache ? kc : kd;
    // dense tier: needs a compile-time-nv kernel of that flavour, a model that asks for more capacity than the tier's and a
    // layout of <= 80 KiB
    int exact_b = 0;
    RolloutFn kb = mjpc_pick_rollout_dense2(model->nv, &exact_b);
    // (a model with a noslip pass runs at full capacity only: the pass recomputes qacc from efc_force, and the last commit's
    // force bits - unlike the iterate - are not pinned across kernel flavours, so the tiers would agree to rounding, not bit for bit)
    if (exact_b && model->noslip_iterations <= 0) {
      // capacity of the dense tier: the largest (rows, contacts = rows / 4 + 2) not above the model's own whose lean layout fits
      // 80 KiB; below 40 rows the retry pass would be the rule, not the exception
      int cap_e = 0, cap_c = 0;
      std::string cap;
      if (mjpc_host::debug_knob("dense_tier_cap", &cap)) {       // diagnostics knob "nefcmax,nconmax": a tiny dense tier forces the retry pass
        int a = 0, b = 0;
        if (sscanf(cap.c_str(), "%d,%d", &a, &b) == 2 && a > 0 && b > 0 && a <= e->pm.M.nefcmax && b <= e->pm.M.nconmax) { cap_e = a; cap_c = b; }
      }
      int first = e->pm.M.nefcmax < TIERB_NEFCMAX ? e->pm.M.nefcmax : TIERB_NEFCMAX;
      // two variants of the flavour: with the hot prefix of the tables in LDS (rollout_dense2h.hip; costs capacity) and without.
      // The hot-cached one is taken when it exists for this dof count and still holds TIERB_HOT_KEEP of the plain one's rows
      int exact_h = 0;
      RolloutFn kh = mjpc_pick_rollout_dense2h(model->nv, &exact_h);
      struct Cand { RolloutFn k = nullptr; Lay lay; int nefc = 0, ncon = 0, ci = 0, cd = 0; size_t lds = 0; } plain, hot;
      for (int variant = 0; variant < 2; variant++) {
        Cand &cd = variant ? hot : plain;
        if (variant && (!exact_h || cap_e)) break;          // (a forced tiny tier is the plain flavour's test case)
        for (int ne = cap_e ? cap_e : first; ne >= (cap_e ? cap_e : TIERB_NEFCMIN) && !cd.k; ne -= 4) {
          MjpcHipModel mb = *model;
          mb.nefcmax = ne; mb.nconmax = cap_e ? cap_c : ne / 4 + 2;
          if (mb.nconmax > e->pm.M.nconmax) mb.nconmax = e->pm.M.nconmax;
          PackedModel pmB;
          if (mjpc_host::build(pmB, &mb, task, e->P_max, false, true, variant == 1, true) && (size_t)pmB.L.total_doubles * sizeof(double) <= TIERB_LDS_LIMIT) {
            cd.k = variant ? kh : kb; cd.lay = pmB.L; cd.nefc = pmB.M.nefcmax; cd.ncon = pmB.M.nconmax;
            cd.ci = (int)pmB.cache_i; cd.cd = (int)pmB.cache_d;
            cd.lds = (size_t)pmB.L.total_doubles * sizeof(double);
          }
        }
      }
      const Cand &use = (hot.k && plain.k && hot.nefc * 100 >= plain.nefc * TIERB_HOT_KEEP) ? hot : plain;
      if (use.k) { e->kernelB = use.k; e->layB = use.lay; e->nefcB = use.nefc; e->nconB = use.ncon; e->cacheB_i = use.ci; e->cacheB_d = use.cd; e->ldsB = use.lds; }
    }
    std::string tier, fi;                                 // diagnostics knobs: "A" = never the dense tier, "B" = always (when it exists)
    e->force_tier = mjpc_host::debug_knob("tier", &tier) ? (tier[0] == 'B' ? 2 : 1) : 0;
    e->fault = (mjpc_host::debug_knob("fault_inject", &fi) && fi == "sync") ? 1 : 0;
    e->cached = use_cache;
  }
  memset(&e->K, 0, sizeof(e->K));
  e->max_local = max_local; e->max_horizon = max_horizon;
  e->nq = model->nq; e->nv = model->nv; e->nu = model->nu; e->nmocap = model->nmocap;
  e->nr = task->num_residual; e->ntr = 3 * task->num_trace; e->ds = model->nq + model->nv + model->na;
  e->lds_bytes = (size_t)e->pm.L.total_doubles * sizeof(double);
  if (e->lds_bytes > 160 * 1024) { set_error("mjpc_hip_create: per-candidate state exceeds 160 KiB of LDS; lower nconmax/nefcmax"); delete e; return nullptr; }
  HIPCHKP(hipMalloc(&e->d_ib, e->pm.ib.size() * sizeof(int)));
  HIPCHKP(hipMalloc(&e->d_db, e->pm.db.size() * sizeof(double)));
  if (upload_model(e) != 0) { mjpc_hip_destroy(e); return nullptr; }
  HIPCHKP(hipStreamCreate(&e->stream));
  for (int i = 0; i < 4; i++) HIPCHKP(hipEventCreate(&e->ev[i]));
  size_t NL = (size_t)max_local, H = (size_t)max_horizon;
  HIPCHKP(hipMalloc(&e->d_state, sizeof(double) * (e->ds + 1)));
  HIPCHKP(hipMalloc(&e->d_mocap, sizeof(double) * (7 * e->nmocap + 7)));
  e->nuserdata = model->nuserdata;
  HIPCHKP(hipMalloc(&e->d_userdata, sizeof(double) * (e->nuserdata + 1)));
  HIPCHKP(hipMalloc(&e->d_kt, sizeof(double) * e->P_max));
  HIPCHKP(hipMalloc(&e->d_kv, sizeof(double) * (e->P_max * e->nu + 1)));
  HIPCHKP(hipMalloc(&e->d_std, sizeof(double) * (e->P_max * e->nu + 1)));
  HIPCHKP(hipMalloc(&e->d_sel, sizeof(int) * NL));
  HIPCHKP(hipMalloc(&e->d_states, sizeof(double) * NL * H * e->ds));
  HIPCHKP(hipMalloc(&e->d_actions, sizeof(double) * NL * H * (e->nu + 1)));
  HIPCHKP(hipMalloc(&e->d_times, sizeof(double) * NL * H));
  HIPCHKP(hipMalloc(&e->d_residual, sizeof(double) * NL * H * (e->nr + 1)));
  HIPCHKP(hipMalloc(&e->d_costs, sizeof(double) * NL * H));
  HIPCHKP(hipMalloc(&e->d_trace, sizeof(double) * NL * H * (e->ntr + 1)));
  HIPCHKP(hipMalloc(&e->d_returns, sizeof(double) * NL));
  HIPCHKP(hipMalloc(&e->d_failure, sizeof(int) * NL));
  HIPCHKP(hipMalloc(&e->d_diag, sizeof(int) * NL * 4));
  HIPCHKP(hipMalloc(&e->d_winner, sizeof(int) * 2));
  e->nbody = model->nbody; e->nsite = model->nsite;
  HIPCHKP(hipMalloc(&e->d_frame, sizeof(double) * (18 * (size_t)e->nbody + 3 * (size_t)e->nsite + 1)));
  HIPCHKP(hipMalloc(&e->d_prof, sizeof(long long) * NL * 24));
  HIPCHKP(hipMemset(e->d_prof, 0, sizeof(long long) * NL * 24));
  HIPCHKP(hipMalloc(&e->d_winner_val, sizeof(double) * 2));
  HIPCHKP(hipHostMalloc(&e->h_small, sizeof(double) * (e->ds + 7 * e->nmocap + e->P_max * (2 * e->nu + 1) + model->nuserdata + 16)));
  HIPCHKP(hipFuncSetAttribute((const void *)e->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->lds_bytes));
  if (e->kernelB) {
    HIPCHKP(hipFuncSetAttribute((const void *)e->kernelB, hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->ldsB));
    e->ckpt_stride = 7 + e->nq + 2 * e->nv + model->na + 1;
    HIPCHKP(hipMalloc(&e->d_ckpt, sizeof(double) * NL * e->ckpt_stride));
    HIPCHKP(hipMemset(e->d_ckpt, 0, sizeof(double) * NL * e->ckpt_stride));
  }
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cu = prop.multiProcessorCount; }
  return e;
}

void mjpc_hip_destroy(MjpcHipEngine *e) {
  if (!e) return;
  hipSetDevice(e->device);
  if (e->stream) hipStreamSynchronize(e->stream);
  void *bufs[] = {e->d_userdata, e->d_cand, e->d_std, e->d_ib, e->d_db, e->d_state, e->d_mocap, e->d_kt, e->d_kv, e->d_eps, e->d_sel, e->d_states, e->d_actions,
                  e->d_times, e->d_residual, e->d_costs, e->d_trace, e->d_knots, e->d_returns, e->d_failure, e->d_diag,
                  e->d_winner, e->d_winner_val, e->d_prof, e->d_frame, e->d_ckpt};
  for (void *b : bufs) if (b) hipFree(b);
  if (e->h_small) hipHostFree(e->h_small);
  for (int i = 0; i < 2; i++) { if (e->h_task[i]) hipHostFree(e->h_task[i]); if (e->ev_task[i]) hipEventDestroy(e->ev_task[i]); }
  if (e->h_pack) hipHostFree(e->h_pack);
  if (e->d_pack) hipFree(e->d_pack);
  for (int i = 0; i < 4; i++) if (e->ev[i]) hipEventDestroy(e->ev[i]);
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
}

int mjpc_hip_set_task(MjpcHipEngine *e, const MjpcHipTask *task) {
  if (!e || !task) { set_error("mjpc_hip_set_task: invalid argument"); return -1; }
  if (task->struct_size != (int)sizeof(MjpcHipTask)) { set_error("mjpc_hip_set_task: MjpcHipTask.struct_size does not match this library"); return -1; }
  HIPCHK(hipSetDevice(e->device));
  if (task->num_residual != e->nr || 3 * task->num_trace != e->ntr) { set_error("mjpc_hip_set_task: residual/trace dimensions changed"); return -1; }
  if (e->pending) { set_error("mjpc_hip_set_task: a plan step is in flight (call mjpc_hip_plan_fetch first)"); return -1; }
  if (!mjpc_host::repack_task(e->pm, task)) { set_error("mjpc_hip_set_task: " + e->pm.error); return -1; }
  // only the task block changes (cost table, residual parameters, frozen ResidualFn state): one small stream-ordered copy per
  // buffer out of the engine's own packed image, ahead of the next plan's kernels; the model and key-frame tables stay put
  const PackedModel &pm = e->pm;
  size_t bi = pm.task_i_cap * sizeof(int), bd = pm.task_d_cap * sizeof(double);
  int slot = e->task_slot ^= 1;                       // two pinned staging slots: the copy of the previous call may still be in flight
  if (!e->h_task[slot]) {
    HIPCHK(hipHostMalloc(&e->h_task[slot], bi + bd + 16));
    HIPCHK(hipEventCreateWithFlags(&e->ev_task[slot], hipEventDisableTiming));
  } else HIPCHK(hipEventSynchronize(e->ev_task[slot]));
  char *stage = (char *)e->h_task[slot];
  memcpy(stage, pm.db.data() + pm.task_d0, bd);
  memcpy(stage + bd, pm.ib.data() + pm.task_i0, bi);
  HIPCHK(hipMemcpyAsync(e->d_db + pm.task_d0, stage, bd, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_ib + pm.task_i0, stage + bd, bi, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipEventRecord(e->ev_task[slot], e->stream));
  e->K.M = mjpc_host::relocate(e->pm, e->d_ib, e->d_db);
  return 0;
}

int mjpc_hip_plan_async(MjpcHipEngine *e, const MjpcHipPlanInput *in) {
  if (!e || !in) { set_error("mjpc_hip_plan: invalid argument"); return -1; }
  if (e->pending) { set_error("mjpc_hip_plan_async: the previous plan step has not been fetched (one plan in flight per engine)"); return -1; }
  int P = in->num_spline_points, H = in->horizon, nl = in->num_local, nu = e->nu;
  if (P < 1 || P > e->P_max) { set_error("mjpc_hip_plan: num_spline_points out of range (1..36)"); return -1; }
  if (H < 1 || H > e->max_horizon) { set_error("mjpc_hip_plan: horizon out of range"); return -1; }
  if (nl < 1 || nl > e->max_local || in->candidate_offset < 0 || in->candidate_offset + nl > in->num_trajectory) { set_error("mjpc_hip_plan: candidate range out of bounds"); return -1; }
  if (in->interpolation < 0 || in->interpolation > 2) { set_error("mjpc_hip_plan: unknown interpolation"); return -1; }
  if (!in->state || !in->knot_times || !in->knot_values || (e->nmocap > 0 && !in->mocap)) { set_error("mjpc_hip_plan: null input"); return -1; }
  HIPCHK(hipSetDevice(e->device));
  size_t need = (size_t)nl * P * nu;
  if (need > e->eps_cap) {
    if (e->d_eps) HIPCHK(hipFree(e->d_eps));
    HIPCHK(hipMalloc(&e->d_eps, sizeof(double) * (need + 1)));
    e->eps_cap = need;
  }
  if (need > e->knots_cap) {
    if (e->d_knots) HIPCHK(hipFree(e->d_knots));
    HIPCHK(hipMalloc(&e->d_knots, sizeof(double) * ((size_t)e->max_local * P * nu + 1)));
    e->knots_cap = (size_t)e->max_local * P * nu;
  }
  // small inputs through pinned staging
  double *hs = e->h_small;
  memcpy(hs, in->state, sizeof(double) * e->ds);
  if (e->nmocap) memcpy(hs + e->ds, in->mocap, sizeof(double) * 7 * e->nmocap);
  double *hkt = hs + e->ds + 7 * e->nmocap, *hkv = hkt + e->P_max;
  memcpy(hkt, in->knot_times, sizeof(double) * P);
  memcpy(hkv, in->knot_values, sizeof(double) * P * nu);
  HIPCHK(hipMemcpyAsync(e->d_state, hs, sizeof(double) * e->ds, hipMemcpyHostToDevice, e->stream));
  if (e->nmocap) HIPCHK(hipMemcpyAsync(e->d_mocap, hs + e->ds, sizeof(double) * 7 * e->nmocap, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_kt, hkt, sizeof(double) * P, hipMemcpyHostToDevice, e->stream));
  HIPCHK(hipMemcpyAsync(e->d_kv, hkv, sizeof(double) * P * nu, hipMemcpyHostToDevice, e->stream));
  if (e->nuserdata) {
    double *hu = hkv + 2 * e->P_max * nu;
    if (in->userdata) memcpy(hu, in->userdata, sizeof(double) * e->nuserdata); else memset(hu, 0, sizeof(double) * e->nuserdata);
    HIPCHK(hipMemcpyAsync(e->d_userdata, hu, sizeof(double) * e->nuserdata, hipMemcpyHostToDevice, e->stream));
  }
  if (in->noise_std) {
    memcpy(hkv + e->P_max * nu, in->noise_std, sizeof(double) * P * nu);
    HIPCHK(hipMemcpyAsync(e->d_std, hkv + e->P_max * nu, sizeof(double) * P * nu, hipMemcpyHostToDevice, e->stream));
  }
  HIPCHK(hipEventRecord(e->ev[0], e->stream));
  if (in->noise_eps) {
    HIPCHK(hipMemcpyAsync(e->d_eps, in->noise_eps + (size_t)in->candidate_offset * P * nu, sizeof(double) * need, hipMemcpyHostToDevice, e->stream));
    if (in->noise_sel) HIPCHK(hipMemcpyAsync(e->d_sel, in->noise_sel + in->candidate_offset, sizeof(int) * nl, hipMemcpyHostToDevice, e->stream));
    else HIPCHK(hipMemsetAsync(e->d_sel, 0, sizeof(int) * nl, e->stream));
  } else {
    size_t total = need > (size_t)nl ? need : (size_t)nl;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(noise_kernel, dim3(blocks), dim3(256), 0, e->stream, e->d_eps, e->d_sel, (unsigned long long)in->seed,
                       (unsigned long long)in->stream, in->candidate_offset, nl, P * nu, in->noise_exploration[1]);
  }
  KParams &K = e->K;
  K.state = e->d_state; K.mocap = e->d_mocap; K.userdata = e->d_userdata; K.nuserdata = e->nuserdata; K.knot_times = e->d_kt; K.knot_values = e->d_kv; K.noise_eps = e->d_eps; K.noise_sel = e->d_sel;
  K.time = in->time; K.sigma0 = in->noise_exploration[0]; K.sigma1 = in->noise_exploration[1];
  K.seed = in->seed; K.stream = in->stream;
  K.P = P; K.interp = in->interpolation; K.H = H; K.N = in->num_trajectory; K.offset = in->candidate_offset; K.nlocal = nl;
  K.use_device_noise = in->noise_eps ? 0 : 1;
  K.fault = e->fault;
  K.noise_std = in->noise_std ? e->d_std : nullptr; K.nominal_index = in->nominal_index;
  K.cand_knots = nullptr; K.xfrc_std = in->xfrc_std; K.xfrc_rate = in->xfrc_rate;
  if (in->xfrc_std > 0 && !(in->xfrc_rate > 0)) { set_error("mjpc_hip_plan: xfrc_rate must be positive when xfrc_std > 0"); return -1; }
  if (in->candidate_knots) {
    if (need > e->cand_cap) {
      if (e->d_cand) HIPCHK(hipFree(e->d_cand));
      HIPCHK(hipMalloc(&e->d_cand, sizeof(double) * (need + 1)));
      e->cand_cap = need;
    }
    HIPCHK(hipMemcpyAsync(e->d_cand, in->candidate_knots + (size_t)in->candidate_offset * P * nu, sizeof(double) * need, hipMemcpyHostToDevice, e->stream));
    K.cand_knots = e->d_cand;
  }
  K.states = e->d_states; K.actions = e->d_actions; K.times = e->d_times; K.residual = e->d_residual; K.costs = e->d_costs;
  K.trace = e->d_trace; K.knots = e->d_knots; K.returns = e->d_returns; K.failure = e->d_failure; K.diag = e->d_diag; K.prof = e->d_prof; K.frame = e->d_frame;
  HIPCHK(hipEventRecord(e->ev[1], e->stream));
  K.retry = 0; K.tier = 0; K.ckpt = e->d_ckpt; K.ckpt_stride = e->ckpt_stride;
  const bool dense = e->kernelB && e->force_tier != 1 && (nl > e->num_cu || e->force_tier == 2) && !(in->xfrc_std > 0);   // (the lean layout has no body-force block)
  if (dense) {
    KParams KB = K;
    KB.M.nefcmax = e->nefcB; KB.M.nconmax = e->nconB; KB.L = e->layB; KB.cache_i = e->cacheB_i; KB.cache_d = e->cacheB_d; KB.tier = 1;
    hipLaunchKernelGGL(e->kernelB, dim3(nl), dim3(mjpc_rollout_threads_cached()), e->ldsB, e->stream, KB);
    K.retry = 1;                       // full capacity for whoever overflowed the dense tier (usually nobody: the launch drains at once)
  }
  hipLaunchKernelGGL(e->kernel, dim3(nl), dim3(mjpc_rollout_threads_cached()), e->lds_bytes, e->stream, K);
  K.retry = 0;
  e->last_dense = dense;
  HIPCHK(hipEventRecord(e->ev[2], e->stream));
  hipLaunchKernelGGL(argmin_kernel, dim3(1), dim3(64), 0, e->stream, e->d_returns, nl, e->d_winner, e->d_winner_val);
  HIPCHK(hipEventRecord(e->ev[3], e->stream));
  {
    size_t rows = (e->summary_only ? 0 : (size_t)H * (e->ds + nu + 2 + e->nr + e->ntr)) + (size_t)P * nu;
    size_t need_pack = 2 + 2 * (size_t)nl + rows;
    if (need_pack > e->pack_cap) {
      if (e->d_pack) HIPCHK(hipFree(e->d_pack));
      if (e->h_pack) HIPCHK(hipHostFree(e->h_pack));
      HIPCHK(hipMalloc(&e->d_pack, sizeof(double) * need_pack));
      HIPCHK(hipHostMalloc(&e->h_pack, sizeof(double) * need_pack));
      e->pack_cap = need_pack;
    }
    PackArgs pa{e->d_winner, e->d_winner_val, e->d_returns, e->d_failure, e->d_states, e->d_actions, e->d_times, e->d_residual,
                e->d_costs, e->d_trace, e->d_knots, nl, H, P, e->ds, nu, e->nr, e->ntr, e->summary_only ? 0 : 1, e->d_pack};
    e->last_summary = e->summary_only;
    hipLaunchKernelGGL(pack_kernel, dim3(8), dim3(256), 0, e->stream, pa);
    HIPCHK(hipMemcpyAsync(e->h_pack, e->d_pack, sizeof(double) * need_pack, hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(hipGetLastError());
  e->last_H = H; e->last_P = P; e->last_nlocal = nl; e->last_offset = in->candidate_offset; e->pending = 1;
  return 0;
}

static int fetch_rows(MjpcHipEngine *e, int local, MjpcHipPlanOutput *out) {
  size_t H = (size_t)e->last_H, P = (size_t)e->last_P, r = (size_t)local;
  if (out->states) HIPCHK(hipMemcpyAsync(out->states, e->d_states + r * H * e->ds, sizeof(double) * H * e->ds, hipMemcpyDeviceToHost, e->stream));
  if (out->actions) HIPCHK(hipMemcpyAsync(out->actions, e->d_actions + r * H * e->nu, sizeof(double) * H * e->nu, hipMemcpyDeviceToHost, e->stream));
  if (out->times) HIPCHK(hipMemcpyAsync(out->times, e->d_times + r * H, sizeof(double) * H, hipMemcpyDeviceToHost, e->stream));
  if (out->residual) HIPCHK(hipMemcpyAsync(out->residual, e->d_residual + r * H * e->nr, sizeof(double) * H * e->nr, hipMemcpyDeviceToHost, e->stream));
  if (out->costs) HIPCHK(hipMemcpyAsync(out->costs, e->d_costs + r * H, sizeof(double) * H, hipMemcpyDeviceToHost, e->stream));
  if (out->trace && e->ntr) HIPCHK(hipMemcpyAsync(out->trace, e->d_trace + r * H * e->ntr, sizeof(double) * H * e->ntr, hipMemcpyDeviceToHost, e->stream));
  if (out->winner_knots) HIPCHK(hipMemcpyAsync(out->winner_knots, e->d_knots + r * P * e->nu, sizeof(double) * P * e->nu, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(hipStreamSynchronize(e->stream));
  return 0;
}

int mjpc_hip_plan_fetch(MjpcHipEngine *e, MjpcHipPlanOutput *out) {
  if (!e || !out || !e->last_nlocal) { set_error("mjpc_hip_plan_fetch: nothing planned"); return -1; }
  HIPCHK(hipSetDevice(e->device));
  hipError_t sync_rc = hipStreamSynchronize(e->stream);       // the packed result of plan_async is in pinned host memory now
  // whatever this fetch returns, the plan step is over: the engine accepts the next plan_async / set_task (a failed fetch used
  // to leave `pending` set for good)
  const int was_pending = e->pending;
  e->pending = 0;
  if (sync_rc != hipSuccess) { set_error(std::string("mjpc_hip_plan_fetch: hipStreamSynchronize: ") + hipGetErrorString(sync_rc)); return -2; }
  if (was_pending) {
    float t01 = 0, t12 = 0, t03 = 0;
    hipEventElapsedTime(&t01, e->ev[0], e->ev[1]); hipEventElapsedTime(&t12, e->ev[1], e->ev[2]); hipEventElapsedTime(&t03, e->ev[0], e->ev[3]);
    out->noise_compute_time_us = 1e3 * t01; out->rollouts_compute_time_us = 1e3 * t12;
    e->acc_rollout_us += 1e3 * t12; e->acc_total_us += 1e3 * t03; e->acc_n++;
  }
  const double *p = e->h_pack;
  int nl = e->last_nlocal;
  int wl = (int)p[0]; double wv = p[1];
  p += 2;
  if (out->returns) memcpy(out->returns, p, sizeof(double) * nl);
  if (out->failure) for (int i = 0; i < nl; i++) out->failure[i] = (int)p[nl + i];
  p += 2 * (size_t)nl;
  // no candidate with a comparable return (every return NaN, e.g. a NaN cost weight): returns[] / failure[] above are valid,
  // there is no winner
  if (wl < 0 || wl >= nl) { set_error("mjpc_hip_plan_fetch: no finite return among the candidates (argmin out of range)"); return -3; }
  out->winner = e->last_offset + wl; out->winner_return = wv;
  size_t H = (size_t)e->last_H, P = (size_t)e->last_P;
  double *dst[7] = {out->winner_knots, out->states, out->actions, out->times, out->residual, out->costs, out->trace};
  size_t cnt[7] = {P * e->nu, H * e->ds, H * e->nu, H, H * e->nr, H, H * e->ntr};
  for (int k = 0; k < (e->last_summary ? 1 : 7); k++) {        // summary mode: the rows stay on the device (mjpc_hip_get_candidate)
    if (dst[k] && cnt[k]) memcpy(dst[k], p, sizeof(double) * cnt[k]);
    p += cnt[k];
  }
  return 0;
}

int mjpc_hip_set_fetch_mode(MjpcHipEngine *e, int mode) {
  if (!e || (mode != MJPC_FETCH_WINNER_ROWS && mode != MJPC_FETCH_SUMMARY)) { set_error("mjpc_hip_set_fetch_mode: invalid argument"); return -1; }
  e->summary_only = mode == MJPC_FETCH_SUMMARY;
  return 0;
}

int mjpc_hip_plan(MjpcHipEngine *e, const MjpcHipPlanInput *in, MjpcHipPlanOutput *out) {
  int rc = mjpc_hip_plan_async(e, in);
  if (rc != 0) return rc;
  return mjpc_hip_plan_fetch(e, out);
}

int mjpc_hip_get_frame(MjpcHipEngine *e, double *xpos, double *xmat, double *site_xpos, double *subtree_com, double *subtree_linvel) {
  if (!e || e->last_nlocal < 1) { set_error("mjpc_hip_get_frame: no finished plan"); return -1; }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  size_t nb = (size_t)e->nbody, ns = (size_t)e->nsite;
  if (xpos) HIPCHK(hipMemcpy(xpos, e->d_frame, sizeof(double) * 3 * nb, hipMemcpyDeviceToHost));
  if (xmat) HIPCHK(hipMemcpy(xmat, e->d_frame + 3 * nb, sizeof(double) * 9 * nb, hipMemcpyDeviceToHost));
  if (site_xpos && ns) HIPCHK(hipMemcpy(site_xpos, e->d_frame + 12 * nb, sizeof(double) * 3 * ns, hipMemcpyDeviceToHost));
  if (subtree_com) HIPCHK(hipMemcpy(subtree_com, e->d_frame + 12 * nb + 3 * ns, sizeof(double) * 3 * nb, hipMemcpyDeviceToHost));
  if (subtree_linvel) HIPCHK(hipMemcpy(subtree_linvel, e->d_frame + 15 * nb + 3 * ns, sizeof(double) * 3 * nb, hipMemcpyDeviceToHost));
  return 0;
}

int mjpc_hip_get_knots(MjpcHipEngine *e, double *knots) {
  if (!e || !knots || e->last_nlocal < 1) { set_error("mjpc_hip_get_knots: no finished plan"); return -1; }
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(knots, e->d_knots, sizeof(double) * (size_t)e->last_nlocal * e->last_P * e->nu, hipMemcpyDeviceToHost));
  return 0;
}

int mjpc_hip_get_candidate(MjpcHipEngine *e, int local_index, MjpcHipPlanOutput *out) {
  if (!e || !out || local_index < 0 || local_index >= e->last_nlocal) { set_error("mjpc_hip_get_candidate: index out of range"); return -1; }
  HIPCHK(hipSetDevice(e->device));
  if (out->returns) HIPCHK(hipMemcpyAsync(out->returns, e->d_returns + local_index, sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (out->failure) HIPCHK(hipMemcpyAsync(out->failure, e->d_failure + local_index, sizeof(int), hipMemcpyDeviceToHost, e->stream));
  out->winner = e->last_offset + local_index;
  return fetch_rows(e, local_index, out);
}

int mjpc_hip_kernel_time(MjpcHipEngine *e, double *avg_rollout_us, double *avg_total_us) {
  if (!e) return 0;
  int n = e->acc_n;
  if (avg_rollout_us) *avg_rollout_us = n ? e->acc_rollout_us / n : 0;
  if (avg_total_us) *avg_total_us = n ? e->acc_total_us / n : 0;
  e->acc_rollout_us = 0; e->acc_total_us = 0; e->acc_n = 0;
  return n;
}

int mjpc_hip_device_ptrs(MjpcHipEngine *e, void **returns, void **states, void **residual) {
  if (!e) return -1;
  if (returns) *returns = e->d_returns;
  if (states) *states = e->d_states;
  if (residual) *residual = e->d_residual;
  return 0;
}

// every local candidate's trace rows [num_local][H][3*num_trace] of the last plan in one copy (GUI: SamplingPlanner::Traces)
int mjpc_hip_get_traces(MjpcHipEngine *e, double *traces) {
  if (!e || !traces || !e->last_nlocal) { set_error("mjpc_hip_get_traces: no finished plan"); return -1; }
  if (!e->ntr) return 0;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(traces, e->d_trace, sizeof(double) * (size_t)e->last_nlocal * e->last_H * e->ntr, hipMemcpyDeviceToHost));
  return 0;
}

// every local candidate's Trajectory arrays of the last plan (any pointer may be NULL); diag = per candidate
// [Newton iterations summed over the steps, max contacts, max constraint rows, warning bits]
int mjpc_hip_get_all_candidates(MjpcHipEngine *e, double *states, double *actions, double *times, double *residual,
                             double *costs, double *trace, double *knots, int *diag) {
  if (!e || !e->last_nlocal) { set_error("mjpc_hip_get_all_candidates: no finished plan"); return -1; }
  HIPCHK(hipSetDevice(e->device));
  size_t n = (size_t)e->last_nlocal, H = (size_t)e->last_H, P = (size_t)e->last_P;
  HIPCHK(hipStreamSynchronize(e->stream));
  if (states) HIPCHK(hipMemcpy(states, e->d_states, sizeof(double) * n * H * e->ds, hipMemcpyDeviceToHost));
  if (actions) HIPCHK(hipMemcpy(actions, e->d_actions, sizeof(double) * n * H * e->nu, hipMemcpyDeviceToHost));
  if (times) HIPCHK(hipMemcpy(times, e->d_times, sizeof(double) * n * H, hipMemcpyDeviceToHost));
  if (residual) HIPCHK(hipMemcpy(residual, e->d_residual, sizeof(double) * n * H * e->nr, hipMemcpyDeviceToHost));
  if (costs) HIPCHK(hipMemcpy(costs, e->d_costs, sizeof(double) * n * H, hipMemcpyDeviceToHost));
  if (trace && e->ntr) HIPCHK(hipMemcpy(trace, e->d_trace, sizeof(double) * n * H * e->ntr, hipMemcpyDeviceToHost));
  if (knots) HIPCHK(hipMemcpy(knots, e->d_knots, sizeof(double) * n * P * e->nu, hipMemcpyDeviceToHost));
  if (diag) HIPCHK(hipMemcpy(diag, e->d_diag, sizeof(int) * n * 4, hipMemcpyDeviceToHost));
  return 0;
}

int mjpc_hip_lds_bytes(MjpcHipEngine *e) { return e ? (int)e->lds_bytes : 0; }
void mjpc_hip_debug_dense_capacity(MjpcHipEngine *e, int *nefc, int *ncon, int *hot) {
  if (nefc) *nefc = (e && e->kernelB) ? e->nefcB : 0;
  if (ncon) *ncon = (e && e->kernelB) ? e->nconB : 0;
  if (hot) *hot = (e && e->kernelB && e->cacheB_d > 0) ? 1 : 0;
}
// LDS bytes of the dense (two workgroups per CU) tier, 0 when the model has none; *used_last = 1 when the last plan ran on it
int mjpc_hip_dense_tier(MjpcHipEngine *e, int *used_last) {
  if (!e) return 0;
  if (used_last) *used_last = e->last_dense;
  return e->kernelB ? (int)e->ldsB : 0;
}

// host-only (no HIP call): bytes of LDS one candidate would occupy; use_cache bit 0: with / without the LDS copy of the model
// tables, bit 1: the dense tier's lean layout (knots and entry tables stay in global memory).
// < 0: the model is refused (mjpc_hip_last_error tells why)
int mjpc_hip_layout_bytes(const MjpcHipModel *model, const MjpcHipTask *task, int use_cache) {
  if (!model || !task) { set_error("mjpc_hip_layout_bytes: invalid argument"); return -1; }
  PackedModel pm;
  int exact = 0;
  if (use_cache & 2) mjpc_pick_rollout_dense2(model->nv, &exact); else if (use_cache & 1) mjpc_pick_rollout_cached(model->nv, &exact); else mjpc_pick_rollout_direct(model->nv, &exact);
  if (!mjpc_host::build(pm, model, task, 36, (use_cache & 1) != 0, (use_cache & 2) != 0, (use_cache & 3) == 0, exact != 0)) { set_error("mjpc_hip_layout_bytes: " + pm.error); return -1; }
  return (int)((size_t)pm.L.total_doubles * sizeof(double));
}

#ifdef MJPC_PROFILE
// MJPC_PROFILE builds only (tools/profile_phases.py): per-candidate phase cycle counters of the last plan ([nlocal][24] int64)
int mjpc_hip_debug_fetch_prof(MjpcHipEngine *e, long long *prof) {
  if (!e || !e->last_nlocal) return -1;
  HIPCHK(hipSetDevice(e->device));
  HIPCHK(hipStreamSynchronize(e->stream));
  HIPCHK(hipMemcpy(prof, e->d_prof, sizeof(long long) * (size_t)e->last_nlocal * 24, hipMemcpyDeviceToHost));
  return 0;
}
#endif

}  // extern "C"
