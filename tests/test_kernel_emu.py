"""CPU tier: the product kernel source (mujoco_mpc_amd/csrc/core.h) compiled in 1-lane emulation mode vs the
oracle, plus ABI checks that need no GPU.  The emulation library is test infrastructure only."""
import ctypes
import os
import re

import numpy as np
import pytest

import emu_lib
import oracle_lib as ol
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.modelgen import REGISTRY, cartpole, humanoid_track, particle, quadruped

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.mark.parametrize("name,P,H,N,sigma,tol", [("particle", 11, 26, 6, (0.3, 0.0), 1e-12), ("cartpole", 10, 50, 8, (0.5, 0.0), 1e-12),
                                                   ("quadruped", 3, 30, 6, (0.04, 0.0), 1e-5),
                                                   ("walker", 3, 80, 6, (0.5, 0.0), 1e-5), ("acrobot", 10, 100, 6, (0.05, 0.0), 1e-9),   # registry tasks beyond the BASELINE configs
                                                   ("quadruped_hill", 5, 26, 6, (0.3, 0.0), 1e-5),  # the A1 on the fractal height field (task_hill.xml)
                                                   ("terrain_balls", 3, 60, 6, (0.5, 0.0), 1e-5),   # height field: prisms through the portal-refinement collider
                                                   ("cylinder_pile", 3, 50, 6, (0.5, 0.0), 1e-5),   # cylinder-box / cylinder-cylinder through the portal-refinement collider
                                                   ("particle_timevarying", 5, 51, 6, (0.3, 0.0), 1e-12), ("particle_fixed", 5, 51, 6, (0.3, 0.0), 1e-12),   # registry Particle / ParticleFixed
                                                   ("swimmer", 10, 101, 6, (0.3, 0.0), 1e-9),       # registry Swimmer: inertia-box fluid forces, filter actuators, planar root, the XML's full implicit integrator
                                                   ("quadrotor", 5, 51, 6, (0.3, 0.0), 1e-9),       # registry Quadrotor: site transmissions (thrust + reaction torque), 15 declared / 13 written residuals
                                                   ("linkage", 4, 80, 6, (0.5, 0.0), 1e-9),         # equality constraints: joint coupling across branches, four-bar connect, pinned free body
                                                   ("welded", 4, 80, 6, (0.5, 0.0), 1e-9),          # weld equalities: arm-to-free-body, explicit relpose, free body welded to a mocap body
                                                   ("fingers", 5, 60, 6, (0.3, 0.0), 1e-6),         # registry Fingers: noslip pass, site transmissions against a reference site with integrated-velocity servos, condim 6
                                                   ("fingers_grasp", 5, 40, 6, (0.05, 0.0), 1e-5),  # the same with the object pinched between the fingers and lifted: the noslip pass at work on condim-6 contacts
                                                   ("site_servo", 4, 80, 6, (0.3, 0.0), 1e-9),      # site transmissions against reference sites (tilted, on another tree / on the same branch), affine bias, activation, force range
                                                   ("noslip_elliptic3", 4, 60, 6, (0.5, 0.0), 1e-6), ("noslip_elliptic4", 4, 60, 6, (0.5, 0.0), 1e-6), ("noslip_elliptic6", 4, 60, 6, (0.5, 0.0), 1e-6),
                                                   ("noslip_pyramidal3", 4, 60, 6, (0.5, 0.0), 1e-6), ("noslip_pyramidal6", 4, 60, 6, (0.5, 0.0), 1e-6),      # noslip pass: joint / tendon friction loss, contact friction of either cone
                                                   ("servo_arm", 4, 80, 6, (0.5, 0.0), 1e-9),       # mjINT_IMPLICITFAST: velocity servos, saturating force range, damped tendon
                                                   ("filter_arm", 4, 80, 6, (0.4, 0.0), 1e-9),      # activation states: filter / filterexact / clamped integrator actuators
                                                   ("ball_chain", 4, 60, 6, (0.4, 0.0), 1e-5),      # limited ball joints, tendon spring / damper / cross-branch limit
                                                   ("humanoid_track", 16, 30, 4, (0.15, 0.0), 1e-5),
                                                   ("humanoid_stand", 3, 24, 4, (0.05, 0.0), 1e-5), ("humanoid_walk", 3, 24, 4, (0.05, 0.0), 1e-5),
                                                   ("humanoid_interact", 3, 24, 4, (0.05, 0.0), 1e-5)])    # registry Humanoid Interact: humanoid + armchair (capsule-box contacts), 68 residuals
def test_kernel_source_matches_oracle(name, P, H, N, sigma, tol):
    m, task, d = REGISTRY[name]()
    o = ol.Oracle(m, task)
    kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(0).uniform(-0.3, 0.3, (P, m["nu"]))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    mocap = d["mocap"] if len(d["mocap"]) else None
    a = o.plan(d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=sigma, noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=sigma, noise_eps=eps, noise_sel=sel)
    assert np.array_equal(a["knots"], b["knots"]) and np.array_equal(a["times"], b["times"]) and np.array_equal(a["actions"], b["actions"])
    for k in ("states", "residual", "costs", "trace", "returns"):
        assert _rel(b[k], a[k]) < tol, k
    assert int(np.argmin(b["returns"])) == a["winner"]
    assert b["lds_doubles"] * 8 <= 160 * 1024          # per-candidate state must fit one CU's LDS


@pytest.mark.parametrize("motion, time0", [(3, 0.0), (8, 1.1), (9, 16.9)])
def test_tracking_motions_other_than_jump_kernel_source_matches_oracle(motion, time0):
    """tracking.cc:43-66: the task's other modes (Cartwheel, Run, Walk): the residual interpolates the key frames of THAT motion
    (first key = sum of the lengths before it) and clamps at its last key (Run: 39 keys = 1.27 s, the rollout from t = 1.1 s runs past
    it; Walk from t = 16.9 s likewise); the host Transition of the closed-loop harness sets the mocap bodies from the same table."""
    from mujoco_mpc_amd.modelgen import humanoid_track
    from mujoco_mpc_amd.modelgen.tasks import HUMANOID_MOTIONS
    m, task, d = humanoid_track(motion=motion)
    lengths = [121, 154, 115, 78, 145, 188, 260, 279, 39, 510]
    assert len(HUMANOID_MOTIONS) == 10 and list(task["int_data"][:3]) == [motion, sum(lengths[:motion]), lengths[motion]]
    assert m["nkey"] == sum(lengths)
    o = ol.Oracle(m, task)
    P, H, N = 4, 30, 4
    kt = time0 + np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(0).uniform(-0.3, 0.3, (P, m["nu"]))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    a = o.plan(d["state"], d["mocap"], time0, kt, kv, 2, N, H, sigma=(0.15, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], d["mocap"], time0, kt, kv, 2, N, H, sigma=(0.15, 0.0), noise_eps=eps, noise_sel=sel)
    for k in ("states", "residual", "costs", "returns"):
        assert _rel(b[k], a[k]) < 1e-5, k
    # the tracking residuals really read this motion's keys: against motion 0's table the same rollout prices differently
    m0, task0, _ = humanoid_track(motion=0)
    a0 = ol.Oracle(m0, task0).plan(d["state"], d["mocap"], time0, kt, kv, 2, N, H, sigma=(0.15, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    assert _rel(a0["states"], a["states"]) < 1e-12 and _rel(a0["returns"], a["returns"]) > 1e-3


@pytest.mark.parametrize("name, integrator, tol", [("swimmer", 3, 1e-9), ("swimmer", 0, 1e-9), ("servo_arm", 2, 1e-9), ("ball_chain", 2, 1e-5), ("humanoid_walk", 2, 1e-5),
                                                   ("quadrotor", 2, 1e-9)])
def test_implicit_integrators_kernel_source_matches_oracle(name, integrator, tol):
    """mjINT_IMPLICIT (2) and implicitfast with fluid forces (swimmer, 3): the dense M - h dF/dv of the integration step - fluid
    derivative, bias-force derivative by exact central differences over hinge chains (servo arm), ball joints (ball chain), a free
    joint with ball / hinge limbs under contact (humanoid), site-driven free body (quadrotor) - and its LU solve, kernel source vs oracle"""
    m, task, d = REGISTRY[name]()
    m = dict(m); m["integrator"] = integrator
    o = ol.Oracle(m, task)
    P, H, N = 4, 40, 4
    kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(0).uniform(-0.3, 0.3, (P, m["nu"]))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    mocap = d["mocap"] if len(d["mocap"]) else None
    a = o.plan(d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
    assert not a["failure"].any() and not b["failure"].any()
    for k in ("states", "residual", "costs", "returns"):
        assert _rel(b[k], a[k]) < tol, k
    # and the integrator matters: the same plan stepped with Euler ends elsewhere
    if integrator == 2:
        m0 = dict(m); m0["integrator"] = 0
        a0 = ol.Oracle(m0, task).plan(d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
        assert _rel(a0["states"], a["states"]) > 1e-7


def test_tendon_friction_loss_kernel_source_matches_oracle():
    """mjCNSTR_FRICTION_TENDON rows: a cross-branch tendon (dense Hessian builds) and a one-joint tendon (inside the pattern) with
    friction loss; rollouts visit the saturated zones (gradient-only rows) and the quadratic zone."""
    from mujoco_mpc_amd.modelgen import ball_chain
    m, task, d = ball_chain(tendon_frictionloss=0.3)
    o = ol.Oracle(m, task)
    P, H, N = 4, 80, 6
    kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(0).uniform(-0.3, 0.3, (P, m["nu"]))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    a = o.plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.4, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.4, 0.0), noise_eps=eps, noise_sel=sel)
    m0, task0, _ = ball_chain()
    a0 = ol.Oracle(m0, task0).plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.4, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    assert _rel(a["states"], a0["states"]) > 1e-2              # the friction changes the motion
    assert not a["failure"].any() and not b["failure"].any()
    for k in ("states", "residual", "costs", "trace", "returns"):
        assert _rel(b[k], a[k]) < 1e-5, k
    assert int(np.argmin(b["returns"])) == a["winner"]


@pytest.mark.parametrize("cone", [0, 1])
def test_shadow_hand_kernel_source_matches_oracle(cone):
    """a8.4 on the CPU tier: the synthetic Shadow hand (position servos, tendon-coupled actuators, capsule-box / box-box /
    sphere-box contacts, pyramidal and elliptic cones) through the kernel source vs the oracle."""
    m, task, d = REGISTRY["shadow_hand"](cone=cone)
    o = ol.Oracle(m, task)
    P, H, N = 5, 30, 5
    kt = np.arange(P) * ((H - 1) * m["timestep"] / P); kv = np.tile(d["ctrl0"], (P, 1))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    a = o.plan(d["state"], None, 0.0, kt, kv, 0, N, H, sigma=(0.1, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], None, 0.0, kt, kv, 0, N, H, sigma=(0.1, 0.0), noise_eps=eps, noise_sel=sel)
    assert a["unsupported"] == 0 and not a["failure"].any() and not b["failure"].any()
    assert np.array_equal(a["knots"], b["knots"]) and np.array_equal(a["actions"], b["actions"])
    for k in ("states", "residual", "costs", "trace", "returns"):
        assert _rel(b[k], a[k]) < 1e-5, k
    assert b["diag"][:, 1].max() >= 5                    # contacts between hand and cube are in play
    assert int(np.argmin(b["returns"])) == a["winner"]


def test_lds_budget_of_every_baseline_model():
    """The whole per-candidate state must fit one CU's 160 KiB of LDS in the flavour the engine would pick (host-only query)."""
    lib = ctypes.CDLL(capi.ENGINE_PATH)
    lib.mjpc_hip_layout_bytes.argtypes = [ctypes.POINTER(capi.MjpcHipModel), ctypes.POINTER(capi.MjpcHipTask), ctypes.c_int]
    for name, cached in (("cartpole", 1), ("quadruped", 1), ("humanoid_track", 1), ("shadow_hand", 0)):
        m, task, _ = REGISTRY[name]()
        cm = capi.CModel(m, task)
        n = lib.mjpc_hip_layout_bytes(ctypes.byref(cm.c_model), ctypes.byref(cm.c_task), cached)
        assert 0 < n <= 160 * 1024, (name, n)


def test_models_the_engine_cannot_roll_out_are_refused_at_create():
    """no silent approximation.  Host-side validation (mjpc_host::build) refuses meshes / height fields that can collide,
    user data, oversized buffers; a cylinder next to a capsule is accepted (conservative run-time test)."""
    from mujoco_mpc_amd.modelgen.builder import BALL, BOX, CYLINDER, ELLIPSOID, FREE, HFIELD, HINGE, MESH, PLANE, SPHERE, ModelBuilder
    from mujoco_mpc_amd.modelgen.tasks import make_task
    lib = ctypes.CDLL(capi.ENGINE_PATH)
    lib.mjpc_hip_layout_bytes.argtypes = [ctypes.POINTER(capi.MjpcHipModel), ctypes.POINTER(capi.MjpcHipTask), ctypes.c_int]
    lib.mjpc_hip_last_error.restype = ctypes.c_char_p
    task = make_task(3, [(1, 0, 1.0)])

    def check(edit, expect):
        b = ModelBuilder()
        b.geom(0, "floor", PLANE, size=(1, 1, 0.1))
        body = b.body("a", 0, pos=(0, 0, 1))
        b.joint(body, "f", FREE)
        b.geom(body, "g", SPHERE, size=(0.1,))
        edit(b, body)
        m = b.compile()
        task["num_residual"] = 1
        cm = capi.CModel(m, task)
        n = lib.mjpc_hip_layout_bytes(ctypes.byref(cm.c_model), ctypes.byref(cm.c_task), 1)
        if expect is None:
            assert n > 0, lib.mjpc_hip_last_error()
        else:
            assert n < 0 and expect in lib.mjpc_hip_last_error().decode(), lib.mjpc_hip_last_error()

    check(lambda b, body: None, None)
    check(lambda b, body: b.geom(body, "e", ELLIPSOID, size=(0.1, 0.2, 0.3)), None)      # ellipsoids go through the portal-refinement collider
    check(lambda b, body: b.geom(0, "t", HFIELD, size=(1, 1, 0.1)), "no usable data")                             # a height field without samples
    check(lambda b, body: b.geom(0, "t", HFIELD, hfield=dict(size=(1, 1, 0.2, 0.1), data=np.zeros((4, 4))), contype=2, conaffinity=0), None)   # fine: only the sphere meets it
    check(lambda b, body: b.geom(body, "e", MESH, size=(0.1, 0.2, 0.3)), "no usable vertex data")           # a mesh geom without vertices
    check(lambda b, body: b.geom(body, "e", MESH, mesh=[[0.1, 0, 0], [-0.1, 0, 0], [0, 0.1, 0], [0, 0, 0.1], [0, -0.05, -0.05]]), None)

    def ball(b, body):
        c = b.body("c", body)
        b.joint(c, "ball", BALL, limited=True, range=(0, 1))
        b.geom(c, "cg", SPHERE, size=(0.05,))
    check(ball, None)                                 # limited ball joints have their limit row now

    def tfric(b, body):
        c = b.body("c", body)
        b.joint(c, "h", HINGE, axis=(0, 1, 0))
        b.geom(c, "cg", SPHERE, size=(0.05,))
        b.tendon("t", ["h"], [1.0], frictionloss=0.1)
    check(tfric, None)                                # tendon friction loss has its friction row now

    # mjOption settings the engine does not implement are refused, not ignored
    def option(field, value, expect):
        b = ModelBuilder()
        b.geom(0, "floor", PLANE, size=(1, 1, 0.1))
        body = b.body("a", 0, pos=(0, 0, 1))
        b.joint(body, "f", FREE)
        b.geom(body, "g", SPHERE, size=(0.1,))
        m = dict(b.compile(), **{field: value})
        task["num_residual"] = 1
        cm = capi.CModel(m, task)
        n = lib.mjpc_hip_layout_bytes(ctypes.byref(cm.c_model), ctypes.byref(cm.c_task), 1)
        if expect is None:
            assert n > 0, lib.mjpc_hip_last_error()
        else:
            assert n < 0 and expect in lib.mjpc_hip_last_error().decode(), lib.mjpc_hip_last_error()
    option("solver", 0, "Newton"); option("solver", 1, "Newton"); option("solver", 2, None)
    option("integrator", 1, "RK4"); option("integrator", 2, None); option("integrator", 3, None)       # RK4 refused; Euler, implicit, implicitfast accepted
    option("density", 1000.0, None); option("viscosity", 0.1, None)                    # inertia-box fluid forces: with every accepted integrator
    for integ in (2, 3):                                                                # (the implicit ones carry the fluid forces' velocity derivative since round 3)
        bf = ModelBuilder(integrator=integ, density=10.0)
        bodyf = bf.body("a", 0, pos=(0, 0, 1)); bf.joint(bodyf, "f", FREE); bf.geom(bodyf, "g", SPHERE, size=(0.1,))
        cmf = capi.CModel(bf.compile(), task)
        assert lib.mjpc_hip_layout_bytes(ctypes.byref(cmf.c_model), ctypes.byref(cmf.c_task), 1) > 0, lib.mjpc_hip_last_error()
        assert lib.mjpc_hip_layout_bytes(ctypes.byref(cmf.c_model), ctypes.byref(cmf.c_task), 2) < 0 and b"dense-tier" in lib.mjpc_hip_last_error()
    option("noslip_iterations", 3, None); option("noslip_iterations", -1, "noslip")           # the noslip pass is built (csrc/noslip.h)
    option("disableflags", 1 << 6, "disableflags"); option("disableflags", 1 << 14, "disableflags")          # gravity, eulerdamp
    option("disableflags", (1 << 0) | (1 << 2) | (1 << 3) | (1 << 4) | (1 << 12), None)
    option("enableflags", 1 << 0, "override"); option("enableflags", 1 << 1, None)
    option("unsupported", 1, "outside the engine's model view")
    option("na", 2, "stateful actuators")                   # activation states must belong to integrator / filter actuators

    check(lambda b, body: b.connect(body, 0, (0, 0, 0.1)), None)          # connect / weld / joint / tendon equalities have rows; a flex does not
    check(lambda b, body: b.weld(body, 0), None)
    b = ModelBuilder()
    body = b.body("a", 0, pos=(0, 0, 1)); b.joint(body, "f", FREE); b.geom(body, "g", SPHERE, size=(0.1,))
    b.connect(body, 0, (0, 0, 0.1))
    m = b.compile(); m["eq_type"][0] = 4
    cm = capi.CModel(m, task)
    assert lib.mjpc_hip_layout_bytes(ctypes.byref(cm.c_model), ctypes.byref(cm.c_task), 1) < 0 and b"only connect, weld, joint and tendon equalities" in lib.mjpc_hip_last_error()

    def bigcon(b, body):
        b.nconmax = 100
    check(bigcon, "nconmax")

    def cyl(b, body):
        c = b.body("c", 0, pos=(1, 0, 1))
        b.joint(c, "h", HINGE)
        b.geom(c, "cyl", CYLINDER, size=(0.1, 0.2))
        b.geom(c, "box", BOX, size=(0.1, 0.1, 0.1), pos=(0, 0, 0.5))
    check(cyl, None)


def test_cross_entropy_noise_mode_in_oracle_and_kernel_source():
    """ABI extension for the Cross-Entropy planner (cross_entropy/planner.cc:340-415): absolute per-parameter std, every
    candidate perturbed except `nominal_index`."""
    m, task, d = cartpole()
    P, H, N = 4, 20, 7
    kt = np.linspace(0, 0.19, P); kv = np.random.default_rng(3).uniform(-0.2, 0.2, (P, 1))
    eps, _ = ol.noise(5, 0, 0, N, P, 1)
    std = np.array([0.05, 0.4, 3.0, 0.0])
    o = ol.Oracle(m, task)
    a = o.plan(d["state"], None, 0.0, kt, kv, 2, N, H, noise_eps=eps, noise_std=std, nominal_index=N - 1)
    expect = np.clip(kv[None] + std[None, :, None] * eps.reshape(N, P, 1), -1.0, 1.0)
    expect[N - 1] = kv
    assert np.array_equal(a["knots"], expect)
    assert np.any(a["knots"][0] != kv) and np.any(np.abs(a["knots"]) == 1.0)         # candidate 0 is perturbed; clamping is hit
    b = emu_lib.plan(m, task, d["state"], None, 0.0, kt, kv, 2, N, H, noise_eps=eps, noise_std=std, nominal_index=N - 1)
    assert np.array_equal(a["knots"], b["knots"])
    assert _rel(b["returns"], a["returns"]) < 1e-12


def test_noisy_rollout_force_noise_obeys_newtons_law_and_matches_kernel_source():
    """ABI extension for the robust planner (Trajectory::NoisyRollout, trajectory.cc:100-210): explicit candidate policies and
    Ornstein-Uhlenbeck xfrc_applied noise.  On the particle (mass 0.3 on two damped slide joints, motors of gear 1) every step
    must satisfy the implicit-damping Euler update with the force  ctrl + xfrc  where xfrc follows the OU recursion on the
    Philox normals of stream ^ "XFRC"."""
    m, task, d = particle(timestep=0.01)
    P, H, N = 3, 12, 3
    kt = np.array([0.0, 0.05, 0.2]); rng = np.random.default_rng(11)
    cand = rng.uniform(-0.5, 0.5, (N, P, 2))
    std, rate_t, seed, stream = 0.7, 0.05, 21, 5
    o = ol.Oracle(m, task)
    a = o.plan(d["state"], d["mocap"], 0.0, kt, np.zeros((P, 2)), 1, N, H, candidate_knots=cand, xfrc_std=std, xfrc_rate=rate_t,
               seed=seed, stream=stream)
    assert np.array_equal(a["knots"], cand)                               # explicit policies: used verbatim
    nb = m["nbody"]; pm = nb - 1                                          # the point mass is the last body
    eps, _ = ol.noise(seed, stream ^ 0x5846524300000000, 0, N, H, 6 * nb)
    rate = np.exp(-m["timestep"] / rate_t); scale = std * np.sqrt(1 - rate * rate)
    h, mass, damp = m["timestep"], 0.3, 1.0
    for i in range(N):
        x = np.zeros(6 * nb)
        for t in range(H - 1):
            x = rate * x + scale * eps[i, t]
            f = x[6 * pm:6 * pm + 2]                                      # force on the point mass, x / y components
            v0, v1 = a["states"][i, t, 2:4], a["states"][i, t + 1, 2:4]
            u = a["actions"][i, t]
            expect = v0 + h * (u + f - damp * v0) / (mass + h * damp)     # Euler, implicit in the joint damping
            assert np.abs(v1 - expect).max() < 1e-12
    b0 = o.plan(d["state"], d["mocap"], 0.0, kt, np.zeros((P, 2)), 1, N, H, candidate_knots=cand)
    assert np.abs(b0["states"] - a["states"]).max() > 1e-4               # the noise matters
    e = emu_lib.plan(m, task, d["state"], d["mocap"], 0.0, kt, np.zeros((P, 2)), 1, N, H, candidate_knots=cand, xfrc_std=std,
                     xfrc_rate=rate_t, seed=seed, stream=stream)
    assert np.array_equal(e["knots"], cand) and _rel(e["states"], a["states"]) < 1e-12 and _rel(e["returns"], a["returns"]) < 1e-12
    # a contact-rich model with forces and torques on every body
    m, task, d = quadruped()
    kt = np.linspace(0, 0.19, 3); cand = np.random.default_rng(2).uniform(-0.2, 0.2, (2, 3, 12))
    a = o = None
    o = ol.Oracle(m, task)
    a = o.plan(d["state"], d["mocap"], 0.0, kt, np.zeros((3, 12)), 2, 2, 20, candidate_knots=cand, xfrc_std=2.0, xfrc_rate=0.1, seed=3, stream=1)
    e = emu_lib.plan(m, task, d["state"], d["mocap"], 0.0, kt, np.zeros((3, 12)), 2, 2, 20, candidate_knots=cand, xfrc_std=2.0, xfrc_rate=0.1,
                     seed=3, stream=1)
    assert _rel(e["states"], a["states"]) < 1e-5 and _rel(e["returns"], a["returns"]) < 1e-5


def test_engine_library_exports_every_declared_symbol():
    """libmjpc_hip.so loads without a GPU and exports exactly what include/mjpc_hip.h declares."""
    import __graft_entry__ as g
    so = g.build_engine()
    lib = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "mjpc_hip.h")).read() + open(os.path.join(ROOT, "include", "mjpc_hip_debug.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mjpc_hip_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(capi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.mjpc_hip_version.restype = ctypes.c_int
    assert lib.mjpc_hip_version() == capi.ABI_VERSION == int(re.search(r"#define MJPC_HIP_ABI_VERSION (\d+)", hdr).group(1))
    # the ctypes layouts and the compiled structs agree (what capi.load_engine() and mjpc_hip_create check at run time)
    for what, ctype in (("model", capi.MjpcHipModel), ("task", capi.MjpcHipTask), ("plan_input", capi.MjpcHipPlanInput), ("plan_output", capi.MjpcHipPlanOutput)):
        assert getattr(lib, "mjpc_hip_sizeof_" + what)() == ctypes.sizeof(ctype), what
    # no environment variable steers the product library any more (diagnostics go through mjpc_hip_debug_set)
    csrc = os.path.join(ROOT, "mujoco_mpc_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".h", ".hip", ".cc", ".inc")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_ctypes_structs_match_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "mjpc_hip.h")).read()
    body = hdr[hdr.index("typedef struct MjpcHipModel {"):hdr.index("} MjpcHipModel;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.replace("typedef struct MjpcHipModel {", "").strip()
        if not stmt:
            continue
        stmt = re.sub(r"^(const\s+)?(int|double)\s+", "", stmt)
        for part in stmt.split(","):
            n = part.strip().lstrip("*").strip()
            n = re.sub(r"\[\d+\]", "", n)
            if n:
                names.append(n)
    assert names == [f[0] for f in capi.MjpcHipModel._fields_]


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mujoco_mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("the CPU oracle under oracle/", ""), f


def test_disabled_and_inactive_equalities_are_dropped_the_same_way():
    """mjDSBL_EQUALITY and eq_active0 = 0: the host leaves the rows out of its table, the oracle skips them; the linkage model (joint,
    connect and tendon equalities) gives the same rollouts either way, with fewer rows, and the fingers are no longer coupled."""
    from mujoco_mpc_amd.modelgen import linkage
    m, task, d = linkage()
    P, H, N = 4, 60, 4
    kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(3).uniform(-0.5, 0.5, (P, m["nu"]))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    base = emu_lib.plan(m, task, d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
    act = np.array(m["eq_active0"]).copy(); act[0] = 0                       # the finger coupling switched off
    for mm in (dict(m, disableflags=m["disableflags"] | (1 << 1)), dict(m, eq_active0=act)):
        a = ol.Oracle(mm, task).plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
        b = emu_lib.plan(mm, task, d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
        assert np.array_equal(a["failure"], b["failure"]) and not a["failure"].any()
        for k in ("states", "costs", "returns"):
            assert _rel(b[k], a[k]) < 1e-7, k
        assert b["diag"][:, 2].min() < base["diag"][:, 2].min()
        s = b["states"]
        assert np.abs(s[:, :, 1] - (-s[:, :, 2] + 0.1 * s[:, :, 2] ** 2)).max() > 2e-2


@pytest.mark.parametrize("flags", [1 << 2, 1 << 3, (1 << 2) | (1 << 3), 1 << 0])
def test_disabled_constraint_kinds_are_dropped_the_same_way(flags):
    """mjDSBL_FRICTIONLOSS / LIMIT / CONSTRAINT: the host drops the rows from its lists, the oracle skips them in make_constraint;
    a model with friction loss, hinge / ball / tendon limits and contacts gives the same rollouts either way."""
    from random_models import random_model
    m, task, d = random_model(5)
    m = dict(m, disableflags=flags)
    o = ol.Oracle(m, task)
    P, H, N = 4, 40, 4
    kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(3).uniform(-0.5, 0.5, (P, m["nu"]))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    a = o.plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
    assert np.array_equal(a["failure"], b["failure"])
    for k in ("states", "costs", "returns"):
        assert _rel(b[k], a[k]) < 1e-5, k
    base = emu_lib.plan(dict(m, disableflags=0), task, d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
    assert b["diag"][:, 2].max() < base["diag"][:, 2].max()           # fewer constraint rows than with everything enabled
    if flags & 1:
        assert b["diag"][:, 2].max() == 0 and b["diag"][:, 1].max() == 0
