"""Generates tests/golden/*.npz: plan-step regression fixtures in the drop-in dump format of SURVEY.md section 8c-4.

Every fixture is ONE plan step of the hot path: inputs (x0, mocap, time, the nominal spline, the INJECTED standard-normal
noise tensor, sigma, horizon) -> outputs named after the members of mjpc::Trajectory / SamplingPlanner (trajectory.h:74-86,
sampling/planner.h:115-162): `total_return[N]`, `failure[N]`, `winner`, and the winner's `states / actions / times / residual /
costs / trace`, plus every candidate's knot values (`candidate_policy[i].plan`).

The values here come from this build's CPU oracle (oracle/), so they are REGRESSION data, not reference parity: MuJoCo is not
in this image.  Anyone with a built reference can overwrite the output arrays with a real
`Trajectory` dump for the same inputs (the noise is injected, nothing is hidden in an RNG) and both test tiers will then
check against MuJoCo itself:  python tests/golden/make_golden.py  regenerates the files from the oracle.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

# name: (generator, N, H, P, interpolation, sigma, hold the defaults' ctrl0 as nominal plan)
CONFIGS = {
    "c1_cartpole_16x50": ("cartpole", 16, 50, 10, 2, 0.5, False),           # BASELINE configs[0], the reference's CPU case
    "particle_10x11": ("particle", 10, 11, 11, 2, 0.01, False),             # test fixture of rollout_test / sampling_planner_test
    "quadruped_8x30": ("quadruped", 8, 30, 3, 2, 0.04, False),              # configs[1]/[3] model, reduced size
    "humanoid_track_4x24": ("humanoid_track", 4, 24, 16, 2, 0.15, False),   # configs[2] model, reduced size
    "humanoid_interact_4x24": ("humanoid_interact", 4, 24, 3, 0, 0.05, False),   # registry Humanoid Interact (task.xml:31-37: horizon 0.35 s, 3 zero-order points, exploration 0.05), from the scene's home key
    "shadow_hand_6x24": ("shadow_hand", 6, 24, 5, 0, 0.1, True),            # configs[4] model (synthetic hand), reduced size
    "walker_10x80": ("walker", 10, 80, 3, 2, 0.5, False),                   # mjpc/tasks/walker/task.xml:10-15 (horizon 0.8 s, 3 points, exploration 0.5)
    "acrobot_10x100": ("acrobot", 10, 100, 10, 2, 0.05, False),             # mjpc/tasks/acrobot/task.xml:9-17, half the horizon
    "ball_chain_6x60": ("ball_chain", 6, 60, 4, 2, 0.4, False),             # limited ball joints, tendon spring / damper / cross-branch limit
    "ball_chain_friction_6x80": ("ball_chain_friction", 6, 80, 4, 2, 0.4, False),   # tendon friction loss rows (cross-branch and in-pattern)
    "swimmer_6x101": ("swimmer", 6, 101, 10, 2, 0.3, False),                # mjpc/tasks/swimmer/task.xml:9-16 (10 spline points), half the horizon; agent_integrator 2 (the full implicit integrator), as the XML asks
    "quadrotor_8x51": ("quadrotor", 8, 51, 5, 2, 0.3, True),                # mjpc/tasks/quadrotor/task.xml:13-19 (horizon 0.5 s, 5 points, exploration 0.3), hover nominal
    "fingers_8x60": ("fingers_grasp", 8, 60, 5, 2, 0.04, False),             # registry Fingers (task.xml:9-19: 5 spline points, exploration 0.04, implicit 5 ms steps, noslip 5), from a pinch grasp
    "welded_6x80": ("welded", 6, 80, 4, 2, 0.5, False),                     # weld equalities (arm-to-free-body, explicit relpose, puck welded to a mocap body)
    "linkage_6x80": ("linkage", 6, 80, 4, 2, 0.5, False),                   # equality constraints (joint coupling, four-bar connect, pinned free body) next to contacts
    "servo_arm_6x80": ("servo_arm", 6, 80, 4, 2, 0.5, False),               # implicitfast integrator (velocity servos, saturating force range, damped tendon)
    "filter_arm_6x80": ("filter_arm", 6, 80, 4, 2, 0.4, False),             # activation states (filter / filterexact / clamped integrator)
    "quadruped_hill_8x26": ("quadruped_hill", 8, 26, 5, 2, 0.3, False),     # mjpc/tasks/quadruped/task_hill.xml:9-14 (horizon 0.25 s, 5 points, exploration 0.3)
    "terrain_balls_6x60": ("terrain_balls", 6, 60, 3, 2, 0.5, False),       # height-field terrain
    "cylinder_pile_6x50": ("cylinder_pile", 6, 50, 3, 2, 0.5, False),       # cylinder / ellipsoid pairs through the portal-refinement collider
}


def inputs(name):
    from mujoco_mpc_amd.modelgen import REGISTRY
    gen, N, H, P, interp, sigma, hold = CONFIGS[name]
    m, task, d = REGISTRY[gen]()
    dt = m["timestep"]
    kt = np.arange(P) * ((H - 1) * dt / P) if interp == 0 else np.linspace(0, (H - 1) * dt, P)
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    kv = np.tile(d["ctrl0"], (P, 1)) if hold else rng.uniform(-0.2, 0.2, (P, m["nu"]))
    lo, hi = np.asarray(m["actuator_ctrlrange"]).reshape(-1, 2).T
    kv = np.clip(kv, lo, hi)
    eps = rng.standard_normal((N, P, m["nu"]))
    sel = np.zeros(N, np.int32)
    mocap = np.asarray(d["mocap"], float)
    return m, task, dict(state=np.asarray(d["state"], float), mocap=mocap, time=np.float64(0.0), knot_times=kt, knot_values=kv,
                         interpolation=np.int32(interp), horizon=np.int32(H), num_trajectory=np.int32(N),
                         noise_exploration=np.array([sigma, 0.0]), noise_eps=eps, noise_sel=sel)


def main():
    import oracle_lib as ol
    for name in (sys.argv[1:] or CONFIGS):
        m, task, inp = inputs(name)
        o = ol.Oracle(m, task)
        N, H = int(inp["num_trajectory"]), int(inp["horizon"])
        r = o.plan(inp["state"], inp["mocap"] if inp["mocap"].size else None, float(inp["time"]), inp["knot_times"], inp["knot_values"],
                   int(inp["interpolation"]), N, H, sigma=tuple(inp["noise_exploration"]), noise_eps=inp["noise_eps"], noise_sel=inp["noise_sel"])
        w = r["winner"]
        out = dict(total_return=r["returns"], failure=r["failure"], winner=np.int32(w), candidate_knots=r["knots"],
                   states=r["states"][w], actions=r["actions"][w], times=r["times"][w], residual=r["residual"][w],
                   costs=r["costs"][w], trace=r["trace"][w], all_costs=r["costs"])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **{"in_" + k: v for k, v in inp.items()}, **{"out_" + k: v for k, v in out.items()},
                            source=np.array("oracle (this build's CPU restatement) - regression data, not reference parity"))
        print(name, "winner", w, "return", r["returns"][w], "failures", int((r["failure"] != 0).sum()))


if __name__ == "__main__":
    main()
