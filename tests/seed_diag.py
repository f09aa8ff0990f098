"""Diagnostic (GPU box): one fuzz seed of tests/random_models.py, HIP engine vs oracle, candidate by candidate (failure codes, returns,
first step that deviates).  Usage: python tests/seed_diag.py SEED.  Test infrastructure only (uses oracle/)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from random_models import random_model
from test_random_models import _plan_inputs
from mujoco_mpc_amd.planner import HipBackend
seed = int(sys.argv[1])
m, task, d = random_model(seed)
print("nv", m["nv"], "na", m["na"], "neq", m["neq"], m["eq_type"], "integrator", m["integrator"], "cone", m["cone"], "ntendon", m["ntendon"], m["tendon_frictionloss"])
P, H, N, kt, kv, eps, sel = _plan_inputs(m, seed)
a = ol.Oracle(m, task).plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=8)
be = HipBackend(m, task, max_samples=N, max_horizon=H)
out = be.plan(state=d["state"], mocap=None, time=0.0, knot_times=kt, knot_values=kv, interpolation=2, num_trajectory=N, horizon=H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
b = be.fetch_all(N, H, P)
print("failure", a["failure"], out["failure"]); print("returns", a["returns"], out["returns"]); print("winner", a["winner"], out["winner"])
for i in range(N):
    dev = np.abs(b["states"][i] - a["states"][i]).max(axis=1) / (np.abs(a["states"][i]).max() + 1e-300)
    first = np.argmax(dev > 1e-6) if (dev > 1e-6).any() else -1
    print(i, "max rel dev", dev.max(), "first step > 1e-6:", first, "diag", b["diag"][i])
