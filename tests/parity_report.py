"""Diagnostic (GPU box): deviation of the HIP engine from the CPU oracle on the C2 workload shape (quadruped, H=100, the
bench's knots/sigma/seed), for a handful of candidates.  Prints the max relative deviation of states / costs / returns so
that drift introduced by a kernel change is visible beyond the short-horizon parity tests.  Test infrastructure only."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
from mujoco_mpc_amd.modelgen import quadruped, humanoid_track
from mujoco_mpc_amd.planner import HipBackend

def rel(a, b): return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)

def report(name, fn, N, H, P, sigma, dt_span):
    m, task, d = fn()
    o = ol.Oracle(m, task)
    kt = np.linspace(0, dt_span, P); kv = np.zeros((P, m["nu"]))
    mocap = d["mocap"] if len(d["mocap"]) else None
    ref = o.plan(d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=(sigma, 0.0), seed=0x5EED, stream=0, nthreads=8)
    be = HipBackend(m, task, max_samples=N, max_horizon=H)
    out = be.plan(state=d["state"], mocap=mocap, time=0.0, knot_times=kt, knot_values=kv, interpolation=2, num_trajectory=N,
                  horizon=H, sigma=(sigma, 0.0), seed=0x5EED, stream=0)
    allc = be.fetch_all(N, H, P)
    print(f"{name}: N={N} H={H}  rel dev states {rel(allc['states'], ref['states']):.3e}  costs {rel(allc['costs'], ref['costs']):.3e}  "
          f"returns {rel(out['returns'], ref['returns']):.3e}  winner {out['winner']}=={ref['winner']}  "
          f"newton iters/step {allc['diag'][:, 0].mean() / H:.2f}")
    be.close()

if __name__ == "__main__":
    report("quadruped C2-shape", quadruped, 32, 100, 3, 0.04, 0.99)
    report("humanoid C3-shape", humanoid_track, 16, 128, 16, 0.15, 0.635)
