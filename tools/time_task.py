"""Diagnostic: kernel time of one plan step of a registry model (tools/time_task.py name N [key=value ...] for the generator)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mujoco_mpc_amd import capi
if os.environ.get("ENGINE_LIB"):      # a variant built by tools/mkvariant.sh
    capi.ENGINE_PATH = os.path.abspath(os.environ["ENGINE_LIB"])
from mujoco_mpc_amd.modelgen import REGISTRY
from mujoco_mpc_amd.planner import HipBackend
name, N = sys.argv[1], int(sys.argv[2])
kw = {k: (int(v) if v.lstrip("-").isdigit() else v == "True" if v in ("True", "False") else float(v)) for k, v in (a.split("=") for a in sys.argv[3:])}
import mujoco_mpc_amd.modelgen.tasks as T
gen = (lambda: getattr(T, name)(**kw)) if (kw or name not in REGISTRY) else REGISTRY[name]
m, task, d = gen()
H, P = d["horizon"], d["P"]
kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.tile(d["ctrl0"], (P, 1)) if "ctrl0" in d else np.zeros((P, m["nu"]))
mocap = d["mocap"] if len(d["mocap"]) else None
be = HipBackend(m, task, max_samples=N, max_horizon=H)
ts = []
for i in range(6):
    out = be.plan(state=d["state"], mocap=mocap, time=0.0, knot_times=kt, knot_values=kv, interpolation=d["interp"], num_trajectory=N, horizon=H,
                  sigma=d["sigma"], seed=0x5EED, stream=i)
    ts.append(out["rollouts_compute_time_us"])
allc = be.fetch_all(N, H, P)
print(name, kw, "N", N, "H", H, "kernel us (median of 5)", float(np.median(ts[1:])), "us per step", float(np.median(ts[1:])) / H, "failures", int((out["failure"] != 0).sum()),
      "newton iters/step", allc["diag"][:, 0].mean() / H, "max ncon", int(allc["diag"][:, 1].max()), "max nefc", int(allc["diag"][:, 2].max()))
