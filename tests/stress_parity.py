"""Diagnostic (GPU box): randomised plans (random nominal splines, larger noise) on the A1 / humanoid tracking / humanoid walk against the CPU
oracle: same winners and failure flags, returns within 1e-5.  Test infrastructure only (uses oracle/)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, oracle_lib as ol
from mujoco_mpc_amd.modelgen import quadruped, humanoid_track, humanoid_walk, shadow_hand, walker
from mujoco_mpc_amd.planner import HipBackend
def rel(a, b): return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)
worst = 0
for name, fn, P, H, N, sig in (("quadruped", quadruped, 3, 60, 48, 0.08), ("humanoid", humanoid_track, 16, 60, 32, 0.2), ("walk", humanoid_walk, 3, 40, 32, 0.1),
                              ("hand", shadow_hand, 5, 40, 32, 0.2), ("walker", walker, 3, 60, 48, 0.5)):
    m, task, d = fn()
    o = ol.Oracle(m, task)
    be = HipBackend(m, task, max_samples=N, max_horizon=H)
    mocap = d["mocap"] if len(d["mocap"]) else None
    for seed in range(12):
        kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(seed).uniform(-0.2, 0.2, (P, m["nu"])) + (np.asarray(d["ctrl0"]) if "ctrl0" in d else 0.0)
        ref = o.plan(d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=(sig, 0.0), seed=seed, stream=seed, nthreads=32)
        for rep in range(2):
            out = be.plan(state=d["state"], mocap=mocap, time=0.0, knot_times=kt, knot_values=kv, interpolation=2, num_trajectory=N, horizon=H,
                          sigma=(sig, 0.0), seed=seed, stream=seed)
            r = rel(out["returns"], ref["returns"])
            worst = max(worst, r)
            assert np.array_equal(out["failure"] != 0, ref["failure"] != 0), (name, seed)
            assert r < 1e-5, (name, seed, r)
            assert out["winner"] == ref["winner"], (name, seed)
    print(name, "ok; worst rel dev of returns so far", worst, "failures in ref:", int((ref["failure"] != 0).sum()))
    be.close()
