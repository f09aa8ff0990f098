"""Diagnostic (GPU box): Humanoid Interact closed loop from the scene's home key in the four task modes (interact.h:31-45)."""
import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from mujoco_mpc_amd import cplanner
from mujoco_mpc_amd.modelgen import humanoid_interact
m, task, d = humanoid_interact()
for mode in (0, 1, 2, 3):
    num = dict(sampling_spline_points=3, sampling_exploration=0.05, sampling_trajectories=128, sampling_representation=0)
    p = cplanner.SamplingPlanner(); p.Initialize(m, task, num, max_samples=128, max_horizon=24); p.Reset(24)
    res = cplanner.testspeed(p, d["state"], None, horizon=24, steps_per_planning_iteration=1, total_time=2.0, mode=mode, mode_time=0.0)
    q = res["state"]
    print("mode", mode, "fail", res["failure"], "torso z %.3f -> %.3f" % (d["state"][2], q[2]), "avg cost %.3f" % res["average_cost"], "cost first/last %.3f %.3f" % (res["cost_per_step"][0], res["cost_per_step"][-1]), "wall %.1f" % res["wall_seconds"])
    p.close()
