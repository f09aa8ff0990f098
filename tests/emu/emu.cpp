// tests/emu/emu.cpp — TEST INFRASTRUCTURE ONLY.
// Compiles the product kernel source (mujoco_mpc_amd/csrc/core.h) in its 1-lane emulation mode so
// the CPU test tier can check the kernel's logic against the oracle before a GPU run.
// Never loaded by the product: mujoco_mpc_amd/capi.py only ever opens libmjpc_hip.so.
#define MJPC_EMU 1
#include <stdlib.h>
#include <vector>
#include "../../mujoco_mpc_amd/csrc/core.h"
#include "../../mujoco_mpc_amd/csrc/host.h"

struct EmuOut {
  double *returns; int *failure; double *states, *actions, *times, *residual, *costs, *trace, *knots; int *diag;
};

extern "C" int emu_plan(const MjpcHipModel *m, const MjpcHipTask *t, const MjpcHipPlanInput *in, EmuOut *out) {
  PackedModel pm;
  if (!mjpc_host::build(pm, m, t, in->num_spline_points > 0 ? in->num_spline_points : 1)) return -1;
  KParams K;
  memset(&K, 0, sizeof(K));
  K.M = mjpc_host::relocate(pm, pm.ib.data(), pm.db.data());
  K.L = pm.L;
  K.frame = nullptr;
  K.ibase = pm.ib.data(); K.dbase = pm.db.data(); K.cache_i = (int)pm.cache_i; K.cache_d = (int)pm.cache_d;
  int P = in->num_spline_points, nu = m->nu, nl = in->num_local;
  std::vector<double> eps((size_t)nl * P * nu + 1, 0.0);
  std::vector<int> sel(nl + 1, 0);
  if (in->noise_eps) for (size_t i = 0; i < (size_t)nl * P * nu; i++) eps[i] = in->noise_eps[(size_t)in->candidate_offset * P * nu + i];
  if (in->noise_sel) for (int i = 0; i < nl; i++) sel[i] = in->noise_sel[in->candidate_offset + i];
  K.state = in->state; K.mocap = in->mocap; K.knot_times = in->knot_times; K.knot_values = in->knot_values;
  K.noise_eps = eps.data(); K.noise_sel = sel.data(); K.noise_std = in->noise_std; K.nominal_index = in->nominal_index;
  K.cand_knots = in->candidate_knots ? in->candidate_knots + (size_t)in->candidate_offset * in->num_spline_points * m->nu : nullptr;
  K.xfrc_std = in->xfrc_std; K.xfrc_rate = in->xfrc_rate;
  K.time = in->time; K.sigma0 = in->noise_exploration[0]; K.sigma1 = in->noise_exploration[1];
  K.seed = in->seed; K.stream = in->stream;
  K.P = P; K.interp = in->interpolation; K.H = in->horizon; K.N = in->num_trajectory; K.offset = in->candidate_offset; K.nlocal = nl;
  K.states = out->states; K.actions = out->actions; K.times = out->times; K.residual = out->residual; K.costs = out->costs;
  K.trace = out->trace; K.knots = out->knots; K.returns = out->returns; K.failure = out->failure; K.diag = out->diag;
  std::vector<double> lds((size_t)pm.L.total_doubles + 16);
  for (int r = 0; r < nl; r++) {
    for (auto &v : lds) v = 0.0 / 0.0;      // poison: catches reads of uninitialised LDS
    g_emu_lds = lds.data(); g_emu_r = r;
    rollout<0>(&K);
  }
  return pm.L.total_doubles;
}
