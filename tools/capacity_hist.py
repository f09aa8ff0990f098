"""Diagnostic: how many constraint rows / contacts the candidates of each bench workload really need (per-candidate maxima over
the horizon), to size the capacity tiers."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mujoco_mpc_amd import modelgen
from mujoco_mpc_amd.planner import HipBackend
import bench
from mujoco_mpc_amd import capi
capi.debug_set("tier", "A")          # full capacity only: the histogram must not be clipped by the dense tier
for wl in sys.argv[1:] or ["quadruped", "humanoid", "hand"]:
    gen, n, H, P, interp, sigma, _, _ = bench.WORKLOADS[wl]
    n = min(n, 1024)
    m, task, d = getattr(modelgen, gen)()
    shift = (H - 1) * m["timestep"] / (P if interp == 0 else max(P - 1, 1))
    kt = np.arange(P) * shift if interp == 0 else np.linspace(0.0, (H - 1) * m["timestep"], P)
    kv = np.tile(np.asarray(d["ctrl0"], float), (P, 1)) if "ctrl0" in d else np.zeros((P, m["nu"]))
    be = HipBackend(m, task, max_samples=n, max_horizon=H)
    for i in range(4):
        out = be.plan(state=d["state"], mocap=d["mocap"], time=0.0, knot_times=kt, knot_values=kv, interpolation=interp,
                      num_trajectory=n, horizon=H, sigma=(sigma, 0.0), seed=0x5EED, stream=i)
        kv = out["winner_knots"]
        a = be.fetch_all(n, H, P)
        ncon, nefc = a["diag"][:, 1], a["diag"][:, 2]
        q = [50, 90, 99, 100]
        print(f"{wl} plan {i}: N={n} failures {int((out['failure'] != 0).sum())} ncon pct{q} {np.percentile(ncon, q)} nefc pct{q} {np.percentile(nefc, q)}"
              f"  newton iters/step {a['diag'][:, 0].mean() / H:.2f}  kernel {out['rollouts_compute_time_us']:.0f} us", flush=True)
        for cap in ((48, 12), (64, 16), (80, 20), (100, 28)):
            print(f"    fits {cap}: {np.mean((nefc <= cap[0]) & (ncon <= cap[1])):.3f}")
    be.close()
