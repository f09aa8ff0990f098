import sys, numpy as np, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import oracle_lib as ol
from mujoco_mpc_amd.modelgen import quadruped, quadruped_hill
from mujoco_mpc_amd.planner import HipBackend
from mujoco_mpc_amd import cplanner
m, task, d = quadruped()
N, H, P = 4096, 100, 3
kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.zeros((P, m["nu"]))
kw = dict(state=d["state"], mocap=d["mocap"], time=0.0, knot_times=kt, knot_values=kv, interpolation=2, num_trajectory=N, horizon=H, sigma=(0.04, 0.0), seed=0x5EED, stream=0)
be = HipBackend(m, task, max_samples=N, max_horizon=H)
out = be.plan(**kw)
o = ol.Oracle(m, task)
t0 = time.time()
ref = o.plan(d["state"], d["mocap"], 0.0, kt, kv, 2, N, H, sigma=(0.04, 0.0), seed=0x5EED, stream=0, nthreads=16)
st2 = d["state"].copy(); st2[:m["nq"]] = np.nextafter(st2[:m["nq"]], np.inf)
ref2 = o.plan(st2, d["mocap"], 0.0, kt, kv, 2, N, H, sigma=(0.04, 0.0), seed=0x5EED, stream=0, nthreads=16)
print("oracle s", time.time() - t0)
r = np.abs(out["returns"] - ref["returns"]) / np.abs(ref["returns"])
rs = np.abs(ref2["returns"] - ref["returns"]) / np.abs(ref["returns"])
for name, x in (("kernel-vs-oracle", r), ("oracle-vs-oracle(1ulp)", rs)):
    print(name, "max %.2e p99 %.2e p50 %.2e  >1e-5: %d  >1e-6: %d  >1e-7: %d" % (x.max(), np.percentile(x, 99), np.percentile(x, 50), (x > 1e-5).sum(), (x > 1e-6).sum(), (x > 1e-7).sum()), "argmax", int(x.argmax()))
be.close()
# hill closed loop: progress for several horizons / durations
m, task, d = quadruped_hill()
for horizon, total, nsamp in ((26, 1.5, 128), (26, 4.0, 128), (51, 4.0, 128), (51, 4.0, 256)):
    num = dict(sampling_spline_points=5, sampling_exploration=0.3, sampling_trajectories=nsamp, sampling_representation=2)
    p = cplanner.SamplingPlanner(); p.Initialize(m, task, num, max_samples=nsamp, max_horizon=horizon); p.Reset(horizon)
    goal0 = d["mocap"][:3].copy(); dist0 = np.linalg.norm(d["state"][:2] - goal0[:2])
    t0 = time.time()
    res = cplanner.testspeed(p, d["state"], d["mocap"], horizon=horizon, steps_per_planning_iteration=1, total_time=total)
    print("hill H", horizon, "T", total, "N", nsamp, "fail", res["failure"], "z %.3f" % res["state"][2], "dist0 %.3f -> %.3f" % (dist0, np.linalg.norm(res["state"][:2] - goal0[:2])), "moved_on", not np.allclose(res["mocap"][:3], goal0), "wall %.1fs" % (time.time() - t0))
    p.close()
