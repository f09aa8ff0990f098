"""Analytic / invariant checks of the oracle's physics restatement (CPU only).

MuJoCo is not available here (parity unpinned at the mj_step boundary); these tests
pin the restatement against closed forms and conservation laws instead.
"""
import math

import numpy as np
import pytest

import oracle_lib as ol
from mujoco_mpc_amd.modelgen import ModelBuilder, cartpole, mass_matrix, particle, quadruped
from mujoco_mpc_amd.modelgen.builder import BOX, CAPSULE, FREE, HINGE, PLANE, SLIDE, SPHERE
from mujoco_mpc_amd.modelgen.tasks import TASK_COPYSTATE, make_task


def _copy_task(m):
    n = m["nq"] + m["nv"]
    return make_task(TASK_COPYSTATE, [(n, 0, 1.0)])


def test_particle_implicit_damped_integrator_closed_form():
    m, task, _ = particle(timestep=0.01)
    o = ol.Oracle(m, task)
    h, mass, b = 0.01, 0.3, 1.0
    x = np.array([0.05, -0.1]); v = np.array([0.2, 0.1]); u = np.array([0.7, -0.4])
    q, qd, t, _, w = o.step(x, v, ctrl=u, nstep=25)
    for _ in range(25):
        v = v + h * (u - b * v) / (mass + h * b)           # Euler, implicit in damping
        x = x + h * v
    assert w == 0 and abs(t - 0.25) < 1e-12
    assert np.allclose(q, x, rtol=0, atol=1e-13) and np.allclose(qd, v, rtol=0, atol=1e-13)


def test_mass_matrix_matches_independent_numpy_for_all_models():
    rng = np.random.default_rng(1)
    for f in (particle, cartpole, quadruped):
        m, task, d = f()
        o = ol.Oracle(m, task)
        q = d["state"][:m["nq"]].copy()
        q[-min(m["nq"], 12):] += rng.uniform(-0.3, 0.3, min(m["nq"], 12))
        if m["nq"] == 19:
            q[3:7] = rng.normal(size=4); q[3:7] /= np.linalg.norm(q[3:7])
        r = o.forward(q, mocap=d["mocap"] if len(d["mocap"]) else None)
        Mq, _, _ = mass_matrix(m, q)
        assert np.abs(Mq - r["qM"]).max() < 1e-13
        assert np.all(np.linalg.eigvalsh(r["qM"]) > 0)


def test_cartpole_equations_of_motion_textbook():
    m, task, _ = cartpole()
    o = ol.Oracle(m, task)
    mc = 1.0; mp = 0.1
    # pole capsule: COM at l=0.5 from the hinge; inertia about COM from the compiled model (axis y)
    l = 0.5
    pole = m["names"]["body"]["pole_1"]
    Mq, _, _ = mass_matrix(m, np.zeros(2))
    I_hinge = Mq[1, 1]                                    # Ic + mp l^2 at theta=0
    g = 9.81
    for th, x_d, th_d, u in [(0.3, 0.1, -0.5, 0.4), (2.5, -0.3, 1.2, -1.0), (-1.0, 0.0, 0.0, 0.0)]:
        r = o.forward([0.2, th], [x_d, th_d], [u])
        # generalized coords (x, theta), theta=0 upright, rotation about +y: pole tip at (sin th, cos th)
        Mt = np.array([[mc + mp, mp * l * math.cos(th)], [mp * l * math.cos(th), I_hinge]])
        bias = np.array([-mp * l * math.sin(th) * th_d ** 2, -mp * g * l * math.sin(th)])
        tau = np.array([10.0 * u - 1e-4 * x_d, -1e-4 * th_d])
        acc = np.linalg.solve(Mt, tau - bias)
        assert np.allclose(r["qM"], Mt, atol=1e-12)
        assert np.allclose(r["qacc"], acc, rtol=1e-10, atol=1e-10)


def test_free_flight_momentum_quadruped():
    """Internal forces (actuators, damping, friction loss, Coriolis) cannot accelerate the COM: as h -> 0 the
    COM acceleration of the free-flying robot is exactly gravity (Euler's O(h) error shrinks with h)."""
    errs = []
    for h in (1e-4, 1e-5):
        m, task, d = quadruped(timestep=h)
        o = ol.Oracle(m, task)
        rng = np.random.default_rng(3)
        q = d["state"][:19].copy(); q[2] = 2.0             # far above the floor: no contacts
        q[7:] += rng.uniform(-0.2, 0.2, 12)
        v = rng.normal(size=18) * 0.5
        trunk = m["names"]["body"]["trunk"]
        r0 = o.forward(q, v, mocap=d["mocap"])
        assert r0["ncon"] == 0
        q1, v1, _, _, w = o.step(q, v, ctrl=rng.uniform(-1, 1, 12), mocap=d["mocap"], nstep=1)
        r1 = o.forward(q1, v1, mocap=d["mocap"])
        acc = (r1["subtree_linvel"][trunk] - r0["subtree_linvel"][trunk]) / h
        assert w == 0
        errs.append(np.abs(acc - [0, 0, -9.81]).max())
    assert errs[1] < 5e-4 and errs[1] < 0.2 * errs[0]      # first-order convergence to exactly g


def _ball_on_plane(tilt=0.0, friction=1.0, condim=3, cone=1, r=0.1, mass=1.0, margin=0.0):
    b = ModelBuilder(timestep=0.002, cone=cone, impratio=1.0)
    b.geom(0, "floor", PLANE, size=(0, 0, 1), friction=(friction, 0.005, 0.0001), condim=condim, margin=margin,
           quat=(math.cos(tilt / 2), 0, math.sin(tilt / 2), 0))
    ball = b.body("ball", 0, pos=(0, 0, r))
    b.joint(ball, "root", FREE)
    b.geom(ball, "ball", SPHERE, size=(r,), mass=mass, friction=(friction, 0.005, 0.0001), condim=condim)
    m = b.compile()
    return m, _copy_task(m)


def test_sphere_rests_on_plane_with_weight_as_contact_force():
    m, task = _ball_on_plane(condim=1)
    o = ol.Oracle(m, task)
    q = np.array([0, 0, 0.1, 1, 0, 0, 0.0]); v = np.zeros(6)
    q, v, _, _, w = o.step(q, v, nstep=1500)
    r = o.forward(q, v)
    assert w == 0 and r["ncon"] == 1 and r["nefc"] == 1
    assert abs(v[2]) < 1e-6 and 0.09 < q[2] < 0.1          # small static penetration
    assert r["efc_force"][0] == pytest.approx(9.81, rel=1e-4)
    assert np.abs(r["qacc"]).max() < 1e-4


@pytest.mark.parametrize("condim", [3, 6])
def test_elliptic_friction_slide_vs_stick_on_incline(condim):
    mu = 0.5
    # gentle slope: sticks (friction angle atan(0.5)=26.6deg)
    m, task = _ball_on_plane(tilt=0.2, friction=mu, condim=condim)
    o = ol.Oracle(m, task)
    nrm = np.array([math.sin(0.2), 0, math.cos(0.2)])
    q = np.concatenate([nrm * 0.0995, [1, 0, 0, 0.0]]); v = np.zeros(6)
    q1, v1, _, _, w = o.step(q, v, nstep=400)
    r = o.forward(q1, v1)
    assert w == 0 and r["nefc"] == condim
    # a sphere on an incline rolls; the contact point must not slip: v_contact ~ 0
    R = 0.1
    # world angular velocity = xmat * local; near-identity orientation change handled by using speeds
    speed = np.linalg.norm(v1[:3]); omega = np.linalg.norm(v1[3:])
    assert speed > 0.05                                     # it does move (rolls downhill)
    assert abs(speed - omega * R) / speed < 0.05           # rolling without slipping
    # steep slope, small mu: slips.  Evaluate the forward dynamics at a sliding, penetrating state.
    m, task = _ball_on_plane(tilt=1.2, friction=0.1, condim=condim)
    o = ol.Oracle(m, task)
    nrm = np.array([math.sin(1.2), 0, math.cos(1.2)])
    tangent = np.array([math.cos(1.2), 0, -math.sin(1.2)])
    q = np.concatenate([nrm * 0.0995, [1, 0, 0, 0.0]])
    v = np.concatenate([0.5 * tangent, np.zeros(3)])
    r = o.forward(q, v)
    f = r["efc_force"]
    assert r["nefc"] == condim and f[0] > 0
    assert math.hypot(f[1], f[2]) == pytest.approx(0.1 * f[0], rel=1e-6)   # on the cone boundary: |ft| = mu*fn
    # Newton's law on the ball: m*a_t = m*g*sin(theta) - mu*fn ; m*a_n = fn - m*g*cos(theta)
    assert r["qacc"][:3] @ tangent == pytest.approx(9.81 * math.sin(1.2) - 0.1 * f[0], rel=1e-6)
    assert r["qacc"][:3] @ nrm == pytest.approx(f[0] - 9.81 * math.cos(1.2), rel=1e-6)


def test_joint_limit_balances_applied_force():
    m, task, _ = particle(timestep=0.01)
    o = ol.Oracle(m, task)
    q, v, _, _, w = o.step([0.28, 0.0], [0.0, 0.0], ctrl=[1.0, 0.0], nstep=600)
    r = o.forward(q, v, ctrl=[1.0, 0.0])
    assert w == 0 and r["nefc"] == 1
    assert 0.29 < q[0] < 0.30 and abs(v[0]) < 1e-6
    assert r["efc_force"][0] == pytest.approx(1.0, rel=1e-4)    # limit force = motor force (gear 1)


def test_frictionloss_stick_and_slip():
    def build(torque):
        b = ModelBuilder(timestep=0.002, gravity=(0, 0, 0))
        arm = b.body("arm", 0)
        b.joint(arm, "h", HINGE, axis=(0, 0, 1), frictionloss=0.5)
        b.geom(arm, "g", BOX, size=(0.1, 0.1, 0.1), mass=1.0)
        b.actuator("a", "h", gear=1.0, ctrlrange=(-2, 2))
        m = b.compile()
        return m, _copy_task(m)
    m, task = build(0)
    o = ol.Oracle(m, task)
    I = m["body_inertia"][1][0]
    r = o.forward([0.0], [0.0], [0.3])
    assert abs(r["qacc"][0]) < 0.3 / I * 0.2                # below the loss: (almost) stuck
    q, v, _, _, _ = o.step([0.0], [0.0], ctrl=[0.3], nstep=500)
    # soft friction: steady creep where -D*(0 - aref) = tau, aref = -B*v, R = (1-d0)/d0 * invweight0
    R = (1 - 0.9) / 0.9 / I; B = 2 / (0.95 * 0.02)
    assert v[0] == pytest.approx(0.3 * R / B, rel=1e-6)
    r = o.forward([0.0], [1.0], [1.5])                      # sliding: tau - f
    assert r["qacc"][0] == pytest.approx((1.5 - 0.5) / I, rel=1e-6)


def test_energy_drift_small_for_undamped_pendulum_chain():
    b = ModelBuilder(timestep=0.0005)
    l1 = b.body("l1", 0, pos=(0, 0, 2))
    b.joint(l1, "j1", HINGE, axis=(0, 1, 0))
    b.geom(l1, "g1", CAPSULE, size=(0.03, 0), fromto=(0, 0, 0, 0, 0, -0.5))
    l2 = b.body("l2", l1, pos=(0, 0, -0.5))
    b.joint(l2, "j2", HINGE, axis=(1, 0, 0))
    b.geom(l2, "g2", CAPSULE, size=(0.03, 0), fromto=(0, 0, 0, 0, 0.1, -0.4))
    l3 = b.body("l3", l2, pos=(0, 0.1, -0.4))
    b.joint(l3, "j3", SLIDE, axis=(0, 0, 1), stiffness=50.0)
    b.geom(l3, "g3", SPHERE, size=(0.05,))
    m = b.compile()
    o = ol.Oracle(m, _copy_task(m))

    def energy(q, v):
        r = o.forward(q, v)
        ke = 0.5 * v @ r["qM"] @ v
        xp, _, xm, _, _ = __import__("mujoco_mpc_amd.modelgen.builder", fromlist=["kinematics"]).kinematics(m, np.asarray(q, float))
        pe = sum(m["body_mass"][i] * 9.81 * (xp[i] + xm[i] @ m["body_ipos"][i])[2] for i in range(1, m["nbody"]))
        pe += 0.5 * 50.0 * q[2] ** 2
        return ke + pe
    q = np.array([1.0, 0.5, 0.02]); v = np.array([0.3, -0.8, 0.1])
    e0 = energy(q, v)
    q1, v1, _, _, w = o.step(q, v, nstep=2000)
    e1 = energy(q1, v1)
    assert w == 0 and abs(e1 - e0) / abs(e0) < 2e-3          # semi-implicit Euler: O(h) drift only


def test_quadruped_stands_on_four_feet_and_ground_raycast():
    m, task, d = quadruped()
    o = ol.Oracle(m, task)
    q = d["state"][:19].copy(); v = np.zeros(18)
    q1, v1, _, _, w = o.step(q, v, mocap=d["mocap"], nstep=100)   # 1 s, zero torque, settles on the legs
    r = o.forward(q1, v1, mocap=d["mocap"])
    assert w == 0
    assert r["ncon"] >= 4
    total_mass = m["body_subtreemass"][m["names"]["body"]["trunk"]]
    # Gait residual (indices 7..10) uses Ground(): foot z - (ground + 0.02 + step); ground = floor at -0.01
    feet = [m["names"]["geom"][n] for n in ("FL", "HL", "FR", "HR")]
    from mujoco_mpc_amd.modelgen.tasks import select_value
    gz = r["geom_xpos"][feet, 2]
    step_h = 0.06 * 1.0                                      # amplitude * StepHeight(phase 0) with duty 0
    assert np.allclose(r["sensordata"][7:11], gz - (-0.01 + 0.02 + step_h), atol=1e-12)
    # vertical force balance when (nearly) at rest: sum of normal forces ~ weight
    if np.abs(v1).max() < 1e-2:
        fn = sum(r["efc_force"][12 + 6 * k] for k in range(4)) if r["ncon"] == 4 else None
        if fn is not None:
            assert fn == pytest.approx(total_mass * 9.81, rel=0.05)


def test_pyramidal_cone_slides_with_mu_times_normal_force():
    """cone=pyramidal, condim 3: edges Jn +- mu*Jt.  Sliding along the first tangent axis (y for a z-normal, by the
    contact-frame rule) saturates one edge pair: |f_t| = mu * f_n."""
    m, task = _ball_on_plane(tilt=0.0, friction=0.4, condim=3, cone=0)
    o = ol.Oracle(m, task)
    q = np.array([0, 0, 0.0995, 1, 0, 0, 0.0]); v = np.array([0, 0.8, 0, 0, 0, 0.0])
    r = o.forward(q, v)
    assert r["ncon"] == 1 and r["nefc"] == 4                  # 2*(dim-1) pyramid edges
    f = r["efc_force"]
    assert np.all(f >= 0)
    # edge forces -> normal and tangential components: fn = sum(f), f_t1 = mu*(f0 - f1), f_t2 = mu*(f2 - f3)
    fn = f.sum(); ft1 = 0.4 * (f[0] - f[1]); ft2 = 0.4 * (f[2] - f[3])
    assert fn > 0 and abs(ft2) < 1e-9 * fn
    assert r["qacc"][1] == pytest.approx(ft1 / 1.0, rel=1e-9)      # only friction acts along y (mass 1)
    assert r["qacc"][2] == pytest.approx(fn / 1.0 - 9.81, rel=1e-9)
    assert abs(ft1) <= 0.4 * fn * (1 + 1e-12) and ft1 < 0


def test_fixed_tendon_limit_couples_two_joints():
    b = ModelBuilder(timestep=0.002, gravity=(0, 0, 0))
    l1 = b.body("l1", 0)
    b.joint(l1, "j1", HINGE, axis=(0, 0, 1))
    b.geom(l1, "g1", BOX, size=(0.1, 0.05, 0.05), pos=(0.1, 0, 0), mass=1.0)
    l2 = b.body("l2", l1, pos=(0.2, 0, 0))
    b.joint(l2, "j2", HINGE, axis=(0, 0, 1))
    b.geom(l2, "g2", BOX, size=(0.1, 0.05, 0.05), pos=(0.1, 0, 0), mass=1.0)
    b.actuator("a1", "j1", gear=1.0, ctrlrange=(-5, 5))
    b.tendon("t", ["j1", "j2"], [0.5, -0.5], limited=True, range=(-0.3, 0.2))
    m = b.compile()
    o = ol.Oracle(m, _copy_task(m))
    r = o.forward([0.1, 0.0], [0, 0], [1.0])
    assert r["nefc"] == 0                                      # length 0.05 inside the range
    r = o.forward([0.6, 0.1], [0, 0], [1.0])                   # length 0.25 > 0.2: upper limit active
    assert r["nefc"] == 1 and r["efc_force"][0] > 0
    # constraint force acts through J = -(0.5, -0.5): opposite torques on the two joints
    assert r["qfrc_constraint"][0] == pytest.approx(-0.5 * r["efc_force"][0])
    assert r["qfrc_constraint"][1] == pytest.approx(0.5 * r["efc_force"][0])
    # held against the limit: simulate and check it settles near the limit with force balancing the motor
    q, v, _, _, w = o.step([0.3, 0.0], [0, 0], ctrl=[0.5], nstep=4000)
    assert w == 0 and 0.5 * q[0] - 0.5 * q[1] < 0.2 + 0.02


def test_humanoid_model_matches_known_mass_and_mocap_pose():
    from mujoco_mpc_amd.modelgen import humanoid_track
    m, task, d = humanoid_track()
    assert m["nq"] == 28 and m["nv"] == 27 and m["nu"] == 21 and task["num_residual"] == 141 and task["num_term"] == 21
    torso = m["names"]["body"]["torso"]
    assert m["body_subtreemass"][torso] == pytest.approx(40.84, abs=0.05)     # dm_control humanoid total mass
    o = ol.Oracle(m, task)
    r = o.forward(d["state"][:28], d["state"][28:], mocap=d["mocap"])
    assert r["warning"] == 0 and r["ncon"] == 4 and r["nefc"] == 16           # two feet, 2 capsule ends each, 4 edges
    pos_res = r["sensordata"][42:93]                                         # marker position residuals at key 0
    assert np.abs(pos_res).max() < 0.06                                       # model pose agrees with its mocap markers


def test_humanoid_stand_and_walk_residuals_at_the_upright_pose():
    """stand.cc:41-94 / walk.cc:44-166 on the oracle at qpos0 (upright, at rest): closed-form values from the numpy kinematics
    of the model generator (site / inertial-frame positions), independent of the oracle's own kinematics."""
    from mujoco_mpc_amd.modelgen import humanoid_stand, humanoid_walk, kinematics

    def frames(m):
        xpos, xquat, xmat, _, _ = kinematics(m, m["qpos0"])
        xmat = np.asarray(xmat).reshape(-1, 3, 3); xpos = np.asarray(xpos).reshape(-1, 3)
        xipos = xpos + np.einsum("bij,bj->bi", xmat, np.asarray(m["body_ipos"]).reshape(-1, 3))
        sb = np.asarray(m["site_bodyid"])
        site = xpos[sb] + np.einsum("bij,bj->bi", xmat[sb], np.asarray(m["site_pos"]).reshape(-1, 3))
        return xipos, site
    m, task, d = humanoid_stand()
    o = ol.Oracle(m, task)
    f = o.forward(m["qpos0"], ctrl=np.full(m["nu"], 0.25))
    r = f["sensordata"]
    names = m["names"]
    xi, sx = frames(m)
    feet = np.array([sx[names["site"][k]] for k in ("sp0", "sp1", "sp2", "sp3")])
    assert abs(r[0] - (xi[names["body"]["head"]][2] - feet[:, 2].mean() - 1.4)) < 1e-12
    com = f["subtree_com"][names["body"]["torso"]]
    assert abs(r[1] - np.linalg.norm(feet[:, :2].mean(axis=0) - com[:2])) < 1e-12      # at rest the capture point is the CoM
    assert np.all(r[2:4] == 0) and np.all(r[4:25] == 0) and np.all(r[25:46] == 0.25)
    m, task, d = humanoid_walk()
    o = ol.Oracle(m, task)
    f = o.forward(m["qpos0"], ctrl=np.full(m["nu"], -0.5))
    r = f["sensordata"]
    xi, sx = frames(m); b = m["names"]["body"]
    th = xi[b["torso"]][2]
    assert abs(r[0] - (th - 1.35)) < 1e-12
    assert abs(r[1] - (0.5 * (xi[b["foot_left"]][2] + xi[b["foot_right"]][2]) - xi[b["pelvis"]][2] - 0.2)) < 1e-12
    assert np.all(np.abs(r[4:12]) < 1e-12)                                # upright: all z axes are (0,0,1)
    assert np.all(r[12:33] == m["qpos0"][7:])                             # posture
    standing = th / np.sqrt(th * th + 0.45 * 0.45) - 0.4
    assert abs(r[33] - standing * (0.0 - 0.5)) < 1e-12                    # at rest: speed error = -speed goal
    assert np.all(np.abs(r[34:36]) < 1e-12) and np.all(r[36:57] == -0.5)


# ---------------------------------------------------------------------------------------------------------------------
# colliders of round 2 (capsule-box, box-box, sphere-cylinder, capsule-cylinder): closed-form placements and properties
# ---------------------------------------------------------------------------------------------------------------------
def _collide(t1, s1, p1, m1, t2, s2, p2, m2, margin=0.0):
    import ctypes as C
    L = ol.lib()
    dp = C.POINTER(C.c_double)
    L.oracle_debug_collide.argtypes = [C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, C.c_double, dp]
    a = [np.ascontiguousarray(x, float).ravel() for x in (list(s1) + [0.0] * (3 - len(s1)), p1, m1, list(s2) + [0.0] * (3 - len(s2)), p2, m2)]
    out = np.zeros(56)
    n = L.oracle_debug_collide(t1, *[x.ctypes.data_as(dp) for x in a[:3]], t2, *[x.ctypes.data_as(dp) for x in a[3:]], margin, out.ctypes.data_as(dp))
    return out[:7 * max(n, 0)].reshape(max(n, 0), 7), n


def _rot(axis, ang):
    c, s = np.cos(ang), np.sin(ang)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]), "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[axis]


def test_box_box_face_edge_and_corner_cases():
    """box-box (own construction, oracle/collide.c): a cube resting on a large face gives its four bottom corners at the
    penetration depth with the face normal; rotating it in the plane keeps four contacts; a cube overhanging a smaller box is
    clipped to that box's face; crossed edges give one contact at the crossing; a corner-down cube one contact."""
    I = np.eye(3)
    big, cube, pen = [0.5, 0.5, 0.1], [0.022] * 3, 1e-3
    c, n = _collide(6, big, [0, 0, 0], I, 6, cube, [0.1, 0.05, 0.1 + 0.022 - pen], I)
    assert n == 4 and np.allclose(c[:, 0], -pen) and np.allclose(c[:, 4:], [0, 0, 1]) and np.allclose(c[:, 3], 0.1 - pen / 2)
    assert sorted(map(tuple, np.round(c[:, 1:3], 6))) == sorted([(0.078, 0.028), (0.078, 0.072), (0.122, 0.028), (0.122, 0.072)])
    c, n = _collide(6, big, [0, 0, 0], I, 6, cube, [0.1, 0.05, 0.1 + 0.022 - pen], _rot("z", np.pi / 4))
    assert n == 4 and np.allclose(c[:, 0], -pen)
    assert np.allclose(sorted(np.hypot(c[:, 1] - 0.1, c[:, 2] - 0.05)), [0.022 * np.sqrt(2)] * 4)
    c, n = _collide(6, [0.01, 0.01, 0.01], [0, 0, 0], I, 6, [0.05] * 3, [0, 0, 0.01 + 0.05 - pen], _rot("z", 0.3))
    assert n == 4 and np.allclose(np.abs(c[:, 1:3]), 0.01) and np.allclose(c[:, 0], -pen)        # clipped to the small box's face
    c, n = _collide(6, [0.05] * 3, [0, 0, 0], _rot("x", np.pi / 4), 6, [0.05] * 3, [0, 0, 2 * 0.05 * np.sqrt(2) - 2e-3],
                    _rot("y", np.pi / 4) @ _rot("z", np.pi / 2))
    assert n == 1 and abs(c[0, 0] + 2e-3) < 1e-12 and np.allclose(c[0, 1:3], 0, atol=1e-12) and np.allclose(np.abs(c[0, 4:]), [0, 0, 1])
    c, n = _collide(6, big, [0, 0, 0], I, 6, cube, [0, 0, 0.1 + 0.022 * np.sqrt(3) - pen], _rot("x", np.arctan(np.sqrt(2))) @ _rot("z", np.pi / 4))
    assert n == 1 and abs(c[0, 0] + pen) < 1e-9 and np.allclose(c[0, 1:3], 0, atol=1e-9)
    _, n = _collide(6, big, [0, 0, 0], I, 6, cube, [0.1, 0.05, 0.1 + 0.022 + 1e-4], I)
    assert n == 0                                                                                    # separated: no contact
    c, n = _collide(6, big, [0, 0, 0], I, 6, cube, [0.1, 0.05, 0.1 + 0.022 + 1e-4], I, margin=1e-3)
    assert n == 4 and np.allclose(c[:, 0], 1e-4)                                                     # inside the margin: positive distance


def test_capsule_box_and_cylinder_colliders():
    I = np.eye(3)
    pen = 1e-3
    c, n = _collide(3, [0.01, 0.03], [0, 0, 0.1 + 0.01 - pen], _rot("y", np.pi / 2), 6, [0.5, 0.5, 0.1], [0, 0, 0], I)
    assert n == 2 and np.allclose(c[:, 0], -pen) and np.allclose(sorted(c[:, 1]), [-0.03, 0.03]) and np.allclose(c[:, 4:], [0, 0, -1])
    c, n = _collide(3, [0.01, 0.03], [0, 0, 0.1 + 0.04 - pen], I, 6, [0.5, 0.5, 0.1], [0, 0, 0], I)
    assert n == 1 and abs(c[0, 0] + pen) < 1e-12                                                     # poking: one contact
    c, n = _collide(2, [0.05], [0.01, 0, 0.04 + 0.05 - pen], I, 5, [0.04, 0.04], [0, 0, 0], I)
    assert n == 1 and abs(c[0, 0] + pen) < 1e-12 and np.allclose(c[0, 4:], [0, 0, -1])              # sphere on the cap
    c, n = _collide(2, [0.05], [0.04 + 0.05 - 2e-3, 0, 0.01], I, 5, [0.04, 0.04], [0, 0, 0], I)
    assert n == 1 and abs(c[0, 0] + 2e-3) < 1e-12 and np.allclose(c[0, 4:], [-1, 0, 0])             # on the lateral surface
    c, n = _collide(2, [0.05], [0.07, 0, 0.07], I, 5, [0.04, 0.04], [0, 0, 0], I)
    assert n == 1 and abs(c[0, 0] - (np.hypot(0.03, 0.03) - 0.05)) < 1e-12                           # on the rim
    c, n = _collide(3, [0.01, 0.03], [0, 0, 0.04 + 0.01 - pen], _rot("y", np.pi / 2), 5, [0.04, 0.04], [0, 0, 0], I)
    assert n == 2 and np.allclose(c[:, 0], -pen)                                                     # capsule lying on the cap


def test_conservative_pretests_never_drop_a_true_contact():
    """The cheap separations in front of the expensive colliders (bounding capsule / sphere, box face and cross axes) must be
    exact in the 'apart' direction: a two-body model goes through the full pair path (with pre-tests) and must report the same
    number of contacts as the bare collider on random configurations."""
    from mujoco_mpc_amd.modelgen.builder import BOX, CAPSULE, CYLINDER, FREE, SPHERE, ModelBuilder
    from mujoco_mpc_amd.modelgen.tasks import make_task
    rng = np.random.default_rng(0)

    def q2m(q):
        w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    for t1, s1, t2, s2 in ((CAPSULE, (0.02, 0.08), BOX, (0.1, 0.06, 0.04)), (CAPSULE, (0.02, 0.08), CYLINDER, (0.05, 0.04)),
                           (SPHERE, (0.03,), CYLINDER, (0.05, 0.04)), (BOX, (0.05, 0.03, 0.02), BOX, (0.1, 0.06, 0.04))):
        b = ModelBuilder()
        a = b.body("a", 0); b.joint(a, "fa", FREE); b.geom(a, "g1", t1, size=s1)
        c = b.body("c", 0); b.joint(c, "fc", FREE); b.geom(c, "g2", t2, size=s2)
        m = b.compile()
        o = ol.Oracle(m, make_task(3, [(26, 0, 1.0)]))
        hits = 0
        for _ in range(1500):
            qa, qb = rng.normal(size=4), rng.normal(size=4)
            qa /= np.linalg.norm(qa); qb /= np.linalg.norm(qb)
            p1 = rng.uniform(-0.18, 0.18, 3)
            _, n = _collide(t1, s1, p1, q2m(qa), t2, s2, np.zeros(3), q2m(qb))
            f = o.forward(np.concatenate([p1, qa, np.zeros(3), qb]))
            assert f["ncon"] == n and f["unsupported"] == 0
            hits += n > 0
        assert hits > 50                                     # the sample really contains contacts


def test_cube_comes_to_rest_on_a_box_and_in_the_synthetic_hand():
    """box-box in the dynamics: a free cube dropped on a welded box settles on its top face (height = box top + half size, no
    drift); the synthetic Shadow hand holding its grasp targets keeps the cube above the palm instead of dropping it to the floor."""
    from mujoco_mpc_amd.modelgen.builder import BOX, FREE, PLANE, ModelBuilder
    from mujoco_mpc_amd.modelgen.tasks import make_task
    b = ModelBuilder(timestep=0.005)
    b.geom(0, "floor", PLANE, size=(1, 1, 0.1))
    b.geom(0, "table", BOX, size=(0.2, 0.2, 0.05), pos=(0, 0, 0.05))
    c = b.body("cube", 0, pos=(0.03, -0.02, 0.13))
    b.joint(c, "f", FREE)
    b.geom(c, "cube", BOX, size=(0.022, 0.022, 0.022), mass=0.126)
    m = b.compile()
    o = ol.Oracle(m, make_task(3, [(13, 0, 1.0)]))
    q = np.array([0.03, -0.02, 0.13, np.cos(0.2), 0, 0, np.sin(0.2)]); v = np.zeros(6)
    q, v, _, _, w = o.step(q, v, nstep=400)
    assert w == 0 and abs(q[2] - (0.1 + 0.022)) < 1e-3 and np.abs(v).max() < 1e-3 and abs(q[0] - 0.03) < 1e-3
    from mujoco_mpc_amd.modelgen import shadow_hand
    m, task, d = shadow_hand()
    o = ol.Oracle(m, task)
    q, v, _, _, w = o.step(d["state"][:35], np.zeros(33), ctrl=d["ctrl0"], nstep=150)
    assert w == 0 and q[6] > -0.02 and np.abs(v).max() < 0.2        # the cube (qpos[4:7]) stays in the hand; the floor is at -0.2


@pytest.mark.parametrize("kw,bit", [(dict(nconmax=6, nefcmax=128), 8), (dict(nconmax=32, nefcmax=44), 16),
                                    (dict(nconmax=32, nefcmax=47, cone=1), 16)])
def test_full_buffers_fail_the_candidate_and_leave_no_half_built_cone(kw, bit):
    """mjWARN_CONTACTFULL / mjWARN_CNSTRFULL (utilities.cc:787-799): a buffer that fills in the middle of a contact's rows drops
    that contact whole (every later loop walks rows by contact dimension); the candidate fails with that bit and kMaxReturnValue.
    Run under `make -C oracle asan` to check the memory side."""
    from oracle_backend import OracleBackend
    from mujoco_mpc_amd.modelgen.tasks import shadow_hand
    m, task, d = shadow_hand(**kw)
    P, H = 5, 30
    kt = np.arange(P) * ((H - 1) * m["timestep"] / P)
    r = OracleBackend(m, task).plan(state=d["state"], mocap=None, time=0.0, knot_times=kt, knot_values=np.tile(d["ctrl0"], (P, 1)),
                                    interpolation=0, num_trajectory=16, horizon=H, sigma=(0.3, 0.0), seed=0, stream=0)
    assert (r["failure"] & bit).any() and (r["failure"] == 0).any(), r["failure"]
    assert np.all(r["returns"][r["failure"] != 0] == 1.0e6)


def test_ball_joint_limit_row_and_static_balance():
    """mj_instantiateLimit for mjJNT_BALL: value = rotation angle of the joint quaternion, dist = max(range) - angle, one row with
    J = -axis at the joint's three dofs.  A pendulum on a limited ball joint pushed by a constant torque comes to rest with the
    limit force balancing the torque, a little beyond the range (soft constraint)."""
    from mujoco_mpc_amd.modelgen.builder import BALL
    b = ModelBuilder(timestep=0.002, gravity=(0, 0, 0))
    l1 = b.body("l1", 0)
    b.joint(l1, "ball", BALL, limited=True, range=(0, 0.4), damping=0.5)
    b.geom(l1, "g1", SPHERE, size=(0.1,), pos=(0, 0, -0.3), mass=1.0)
    m = b.compile()
    o = ol.Oracle(m, _copy_task(m))
    ang = 0.2
    r = o.forward([math.cos(ang / 2), 0, math.sin(ang / 2), 0], [0, 0, 0])
    assert r["nefc"] == 0                                              # 0.2 rad: inside the 0.4 rad cone
    ang = 0.5
    ax = np.array([0.6, 0.8, 0.0])
    r = o.forward([math.cos(ang / 2), *(math.sin(ang / 2) * ax)], [0, 0, 0])
    assert r["nefc"] == 1 and r["efc_force"][0] > 0
    assert np.allclose(r["qfrc_constraint"], -ax * r["efc_force"][0], rtol=1e-12, atol=1e-14)   # pushes back along -axis
    # rotations beyond pi come back as the short way round with the axis flipped (mju_quat2Vel)
    ang = 2 * math.pi - 0.5
    r = o.forward([math.cos(ang / 2), *(math.sin(ang / 2) * ax)], [0, 0, 0])
    assert r["nefc"] == 1 and np.allclose(r["qfrc_constraint"], ax * r["efc_force"][0], rtol=1e-12, atol=1e-14)
    # constant torque 0.3 about y (xfrc-free: apply through a motor on a hinge is not available on a ball joint; use qvel kick and damping)
    q = np.array([1.0, 0, 0, 0]); v = np.array([0.0, 3.0, 0.0])
    q, v, _, _, w = o.step(q, v, nstep=3000)
    assert w == 0
    angle = 2 * math.atan2(np.linalg.norm(q[1:]), q[0])
    assert angle < 0.4 + 0.05 and np.linalg.norm(v) < 1e-3               # stopped by the limit (and the damper), not spinning on


def test_tendon_spring_and_damper_closed_form():
    """mj_passive, tendon part: force = k * (dead band edge - length) - b * velocity along J^T, length = sum coef * qpos."""
    b = ModelBuilder(timestep=0.002, gravity=(0, 0, 0))
    l1 = b.body("l1", 0)
    b.joint(l1, "s1", SLIDE, axis=(1, 0, 0))
    b.geom(l1, "g1", SPHERE, size=(0.1,), mass=2.0)
    l2 = b.body("l2", 0, pos=(0, 1, 0))
    b.joint(l2, "s2", SLIDE, axis=(1, 0, 0))
    b.geom(l2, "g2", SPHERE, size=(0.1,), mass=4.0)
    b.tendon("t", ["s1", "s2"], [2.0, -1.0], stiffness=30.0, damping=1.5, springlength=(-0.1, 0.2))
    m = b.compile()
    o = ol.Oracle(m, _copy_task(m))
    for q, v in (([0.3, 0.1], [0.5, -0.25]), ([-0.2, 0.1], [0.0, 0.4]), ([0.05, 0.0], [1.0, 1.0])):
        length = 2 * q[0] - q[1]; vel = 2 * v[0] - v[1]
        frc = 30.0 * (0.2 - length) if length > 0.2 else (30.0 * (-0.1 - length) if length < -0.1 else 0.0)
        frc -= 1.5 * vel
        r = o.forward(q, v)
        assert r["qacc"][0] == pytest.approx(2.0 * frc / 2.0, rel=1e-12, abs=1e-14)
        assert r["qacc"][1] == pytest.approx(-1.0 * frc / 4.0, rel=1e-12, abs=1e-14)
    # springlength=None: resting length = the length at qpos0 (0 here)
    b.tendons[0]["springlength"] = None
    m2 = b.compile()
    assert np.allclose(m2["tendon_lengthspring"], 0.0)


def test_joint_actuator_force_range_closed_form():
    """jnt_actfrclimited: the SUM of the actuator forces on a scalar joint is clamped to jnt_actfrcrange (end of mj_fwdActuation)."""
    b = ModelBuilder(timestep=0.002, gravity=(0, 0, 0), contact=False)
    l = b.body("l", 0); b.joint(l, "h", HINGE, axis=(0, 1, 0), actuatorfrcrange=(-0.5, 0.8)); b.geom(l, "g", SPHERE, size=(0.1,), pos=(0, 0, -0.5), mass=2.0)
    b.actuator("m1", "h", gear=1.0, ctrlrange=(-2, 2)); b.actuator("m2", "h", gear=2.0, ctrlrange=(-2, 2))
    m = b.compile()
    o = ol.Oracle(m, _copy_task(m))
    inertia = 2.0 * (0.5 ** 2 + 0.4 * 0.1 ** 2)
    for u, tau in (((0.2, 0.1), 0.4), ((1.0, 0.5), 0.8), ((-1.0, -0.4), -0.5), ((1.5, -0.5), 0.5)):
        assert o.forward([0.1], [0.0], ctrl=list(u))["qacc"][0] == pytest.approx(tau / inertia, rel=1e-12)


def test_fluid_forces_inertia_box_closed_form():
    """mj_inertiaBoxFluidModel on a free box (half extents a, b, c => equivalent box 2a x 2b x 2c): quadratic drag -1/2 rho A |v| v per
    axis, viscous drag -3 pi d mu v with d the mean box size, their angular counterparts, wind subtracted from the linear velocity;
    in a rotated body the forces follow the body axes."""
    a, bb, cc, mass, rho, mu = 0.1, 0.06, 0.03, 0.5, 1000.0, 0.2
    def box(density, viscosity, wind=(0, 0, 0)):
        b = ModelBuilder(timestep=0.002, gravity=(0, 0, 0), contact=False, density=density, viscosity=viscosity, wind=wind)
        l = b.body("l", 0); b.joint(l, "f", FREE); b.geom(l, "g", BOX, size=(a, bb, cc), mass=mass)
        m = b.compile()
        return ol.Oracle(m, _copy_task(m))
    q0 = np.array([0, 0, 0, 1, 0, 0, 0.0]); v = np.array([1.0, -2.0, 0.5])
    area = np.array([4 * bb * cc, 4 * a * cc, 4 * a * bb]); ext = 2 * np.array([a, bb, cc]); diam = ext.mean()
    r = box(rho, 0).forward(q0, np.concatenate([v, np.zeros(3)]))
    assert np.allclose(r["qacc"][:3], -0.5 * rho * area * np.abs(v) * v / mass, rtol=1e-12)
    r = box(0, mu).forward(q0, np.concatenate([v, np.zeros(3)]))
    assert np.allclose(r["qacc"][:3], -3 * np.pi * diam * mu * v / mass, rtol=1e-12)
    r = box(rho, mu, wind=(0.3, 0, 0)).forward(q0, np.concatenate([v, np.zeros(3)]))
    vr = v - np.array([0.3, 0, 0])
    assert np.allclose(r["qacc"][:3], (-0.5 * rho * area * np.abs(vr) * vr - 3 * np.pi * diam * mu * vr) / mass, rtol=1e-12)
    w = 4.0; Ixx = mass / 3 * (bb ** 2 + cc ** 2)                                     # spin about the body's x axis: no gyroscopic term
    r = box(rho, mu).forward(q0, np.array([0, 0, 0, w, 0, 0]))
    tq = -rho * ext[0] * (ext[1] ** 4 + ext[2] ** 4) * abs(w) * w / 64 - np.pi * diam ** 3 * mu * w
    assert np.allclose(r["qacc"][3:], [tq / Ixx, 0, 0], rtol=1e-10, atol=1e-12) and np.abs(r["qacc"][:3]).max() < 1e-12
    q90 = np.array([0, 0, 0, np.cos(np.pi / 4), 0, 0, np.sin(np.pi / 4)])              # rotated 90 deg about z: body x along world y
    r = box(rho, 0).forward(q90, np.array([0, 1.5, 0, 0, 0, 0]))
    assert np.allclose(r["qacc"][:3], [0, -0.5 * rho * area[0] * 1.5 * 1.5 / mass, 0], rtol=1e-10, atol=1e-12)


def test_gravity_compensation_closed_form():
    """mj_passive gravcomp: -gravity * mass * gravcomp at the body's com.  A pendulum with gravcomp 1 feels no gravity torque, with 0.5
    half of it; compensation of the outer link also unloads the inner joint through the point Jacobian."""
    def pend(gc):
        b = ModelBuilder(timestep=0.002, contact=False)
        l1 = b.body("l1", 0, gravcomp=gc[0]); b.joint(l1, "h1", HINGE, axis=(0, 1, 0)); b.geom(l1, "g1", SPHERE, size=(0.05,), pos=(0, 0, -0.4), mass=1.0)
        l2 = b.body("l2", l1, pos=(0, 0, -0.4), gravcomp=gc[1]); b.joint(l2, "h2", HINGE, axis=(0, 1, 0)); b.geom(l2, "g2", SPHERE, size=(0.05,), pos=(0, 0, -0.3), mass=0.5)
        m = b.compile()
        return ol.Oracle(m, _copy_task(m))
    q = [0.7, -0.4]
    full = pend((1.0, 1.0)).forward(q, [0, 0])
    assert np.abs(full["qacc"]).max() < 1e-10                                       # weightless: no acceleration from rest
    none, half, outer = (pend(gc).forward(q, [0, 0]) for gc in ((0, 0), (0.5, 0.5), (0.0, 1.0)))
    M = none["qM"]
    tau_none = M @ none["qacc"]                                                     # = -gravity torque (no other force at rest)
    assert np.allclose(M @ half["qacc"], 0.5 * tau_none, rtol=1e-10)
    g2 = np.array([-0.5 * 9.81 * (0.4 * np.sin(q[0]) + 0.3 * np.sin(q[0] + q[1])), -0.5 * 9.81 * 0.3 * np.sin(q[0] + q[1])])   # link 2's share
    assert np.allclose(M @ outer["qacc"], tau_none - g2, rtol=1e-9, atol=1e-12)


def test_site_transmission_closed_form():
    """mjTRN_SITE: the gear wrench acts in the site frame, qfrc = J_site^T wrench.  The quadrotor hovers at m g / 4 per rotor (reaction
    torques cancel), climbs at (sum f - m g) / m in the body's z direction also when tilted, one rotor more than the others rolls /
    pitches it by r x f / I, and a yaw torque imbalance turns it about z."""
    from mujoco_mpc_amd.modelgen import quadrotor
    m, task, d = quadrotor()
    o = ol.Oracle(m, task)
    mass, g = 1.325, 9.81
    q = np.array(d["state"][:7]); hover = mass * g / 4
    r = o.forward(q, np.zeros(6), ctrl=np.full(4, hover))
    assert np.abs(r["qacc"]).max() < 1e-9
    r = o.forward(q, np.zeros(6), ctrl=np.full(4, hover + 0.5))
    assert np.allclose(r["qacc"], [0, 0, 2.0 / mass, 0, 0, 0], atol=1e-9)
    tilt = np.array([0, 0, 1.0, np.cos(0.3), np.sin(0.3), 0, 0])                  # rolled by 0.6 rad about x, free of the floor
    r = o.forward(tilt, np.zeros(6), ctrl=np.full(4, hover))
    zb = np.array([0, -np.sin(0.6), np.cos(0.6)])
    assert np.allclose(r["qacc"][:3], 4 * hover * zb / mass - np.array([0, 0, g]), atol=1e-9) and np.abs(r["qacc"][3:]).max() < 1e-9
    ctrl = np.full(4, hover); ctrl[1] += 0.4                                       # rotor 2 at (-0.14, 0.18), reaction torque +0.0201
    r = o.forward(np.array([0, 0, 1.0, 1, 0, 0, 0]), np.zeros(6), ctrl=ctrl)
    Ixx, Iyy, Izz = [mass / 3 * v for v in (0.16 ** 2 + 0.03 ** 2, 0.12 ** 2 + 0.03 ** 2, 0.12 ** 2 + 0.16 ** 2)]
    assert np.allclose(r["qacc"][3:], [0.18 * 0.4 / Ixx, 0.14 * 0.4 / Iyy, 0.0201 * 0.4 / Izz], rtol=1e-9)


def test_equality_constraints_against_equivalent_joints():
    """mj_instantiateEquality.  (i) connect: a free ball pinned to the world 0.3 m above its centre swings like the same ball on a
    hinge at that point (soft constraint: agreement to 2 mm over half a second, pin error below 1 mm); (ii) joint equality q1 = q2
    between two identical pendulums, only one of them driven: both follow the motion of one pendulum with half the torque."""
    h, L = 0.002, 0.3
    b = ModelBuilder(timestep=h, contact=False)
    ball = b.body("ball", 0, pos=(0, 0, -L)); b.joint(ball, "f", FREE); b.geom(ball, "g", SPHERE, size=(0.04,), mass=0.2)
    b.connect(ball, 0, (0, 0, L), solref=(0.004, 1.0))
    m = b.compile()
    o = ol.Oracle(m, _copy_task(m))
    b2 = ModelBuilder(timestep=h, contact=False)
    l = b2.body("l", 0); b2.joint(l, "h", HINGE, axis=(0, 1, 0)); b2.geom(l, "g", SPHERE, size=(0.04,), pos=(0, 0, -L), mass=0.2)
    m2 = b2.compile()
    o2 = ol.Oracle(m2, _copy_task(m2))
    vx = 0.8
    q, v = np.array([0, 0, -L, 1, 0, 0, 0.0]), np.array([vx, 0, 0, 0, -vx / L, 0])
    q2, v2 = np.array([0.0]), np.array([-vx / L])
    for _ in range(5):
        q, v, *_ = o.step(q, v, nstep=50); q2, v2, *_ = o2.step(q2, v2, nstep=50)
        centre = np.array([-L * np.sin(q2[0]), 0, -L * np.cos(q2[0])])
        assert np.abs(q[:3] - centre).max() < 2e-3
        w, x, y, z = q[3:7]
        top = q[:3] + L * np.array([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)])       # the pinned point
        assert np.abs(top).max() < 1e-3
    # (ii) joint equality
    def pend(n, coupled, gear):
        bb = ModelBuilder(timestep=h, contact=False)
        for k in range(n):
            l = bb.body(f"l{k}", 0, pos=(k, 0, 0)); bb.joint(l, f"h{k}", HINGE, axis=(0, 1, 0), damping=0.01)
            bb.geom(l, f"g{k}", SPHERE, size=(0.05,), pos=(0, 0, -0.4), mass=0.5)
        if coupled:
            bb.joint_equality("h0", "h1", solref=(0.004, 1.0))
        bb.actuator("m", "h0", gear=gear, ctrlrange=(-1, 1))
        mm = bb.compile()
        return ol.Oracle(mm, _copy_task(mm))
    oc, o1 = pend(2, True, 1.0), pend(1, False, 0.5)
    qc, vc, *_ = oc.step([0.3, 0.3], [0, 0], ctrl=[0.8], nstep=400)
    q1, v1, *_ = o1.step([0.3], [0], ctrl=[0.8], nstep=400)
    assert abs(qc[0] - qc[1]) < 2e-3 and abs(qc[0] - q1[0]) < 5e-3 and abs(vc[0] - v1[0]) < 2e-2


def _nudge(m, qpos, dof, eps):
    """qpos moved by eps along one dof (mj_integratePos: free / ball rotations are body-frame)"""
    from mujoco_mpc_amd.modelgen.builder import BALL, quat_mul
    q = np.array(qpos, float)
    for j in range(m["njnt"]):
        da, qa, ty = m["jnt_dofadr"][j], m["jnt_qposadr"][j], m["jnt_type"][j]
        n = {FREE: 6, BALL: 3}.get(ty, 1)
        if not da <= dof < da + n:
            continue
        k = dof - da
        if ty in (HINGE, SLIDE) or (ty == FREE and k < 3):
            q[qa + k] += eps
        else:
            o = qa + 3 if ty == FREE else qa
            ax = np.zeros(3); ax[k - 3 if ty == FREE else k] = 1.0
            dq = np.concatenate([[math.cos(eps / 2)], math.sin(eps / 2) * ax])
            q[o:o + 4] = quat_mul(q[o:o + 4], dq)
    return q


def test_weld_equality_rows():
    """mjEQ_WELD in mj_instantiateEquality: six rows (anchor offset, torquescale * vec of the relative rotation) that vanish at the
    reference pose; the Jacobian rows are the derivatives of the residuals along every dof (central differences); one impedance from
    the norm of all six; diagApprox translational / rotational; a welded free body sags under gravity by the soft-constraint closed
    form  m g R / (K imp)  and holds its orientation against its off-centre weight."""
    from mujoco_mpc_amd.modelgen.builder import BALL
    h = 0.002
    b = ModelBuilder(timestep=h, contact=False)
    l1 = b.body("l1", 0, pos=(0, 0, 1)); b.joint(l1, "h1", HINGE, axis=(0, 1, 0)); b.geom(l1, "g1", CAPSULE, size=(0.03, 0.2), pos=(0.2, 0, 0), euler=(0, 90, 0), mass=0.5)
    l2 = b.body("l2", l1, pos=(0.4, 0, 0)); b.joint(l2, "b2", BALL); b.geom(l2, "g2", CAPSULE, size=(0.03, 0.15), pos=(0.15, 0, 0), euler=(0, 90, 0), mass=0.3)
    f = b.body("f", 0, pos=(0.75, 0.05, 1.02), quat=(0.9, 0.1, -0.3, 0.2)); b.joint(f, "ff", FREE); b.geom(f, "gf", BOX, size=(0.05, 0.04, 0.03), mass=0.4)
    b.weld(l2, f, anchor=(0.02, -0.01, 0.03), torquescale=0.7, solref=(0.01, 1.0))
    w = b.body("w", 0, pos=(-1, 0, 1)); b.joint(w, "fw", FREE); b.geom(w, "gw", SPHERE, size=(0.05,), pos=(0.1, 0, 0), mass=0.25)
    b.weld(w, 0, relpose=(0.3, 0.2, -1.1, 0.8, 0.0, 0.6, 0.0), solref=(0.006, 1.0))          # explicit relpose, not the qpos0 one
    m = b.compile()
    assert list(m["eq_type"]) == [1, 1] and m["eq_data"][0, 10] == 0.7 and m["eq_data"][1, 10] == 1.0
    assert np.allclose(m["eq_data"][1, 3:10], [0.3, 0.2, -1.1, 0.8, 0, 0.6, 0])
    o = ol.Oracle(m, _copy_task(m))
    c0 = o.constraints(m["qpos0"])
    assert len(c0["pos"]) == 12 and np.abs(c0["pos"][:6]).max() < 1e-15 and np.abs(c0["pos"][6:]).max() > 0.1      # the first weld holds at qpos0
    rng = np.random.default_rng(5)
    q = np.array(m["qpos0"], float)
    for d in range(m["nv"]):
        q = _nudge(m, q, d, rng.uniform(-0.4, 0.4))
    c = o.constraints(q, rng.normal(size=m["nv"]))
    eps = 1e-6
    for d in range(m["nv"]):
        fd = (o.constraints(_nudge(m, q, d, eps))["pos"] - o.constraints(_nudge(m, q, d, -eps))["pos"]) / (2 * eps)
        assert np.abs(fd - c["J"][:, d]).max() < 2e-8, (d, fd, c["J"][:, d])
    iw = m["body_invweight0"].reshape(-1, 2)
    assert np.allclose(c["diag"][:3], iw[l2, 0] + iw[f, 0]) and np.allclose(c["diag"][3:6], iw[l2, 1] + iw[f, 1])
    assert np.allclose(c["diag"][6:9], iw[w, 0]) and np.allclose(c["diag"][9:], iw[w, 1])
    for k in (0, 6):                                                                  # one impedance for the six rows
        ratio = c["R"][k:k + 6] / c["diag"][k:k + 6]
        assert np.allclose(ratio, ratio[0], rtol=1e-12)
    # a free body welded to the world where it is, its mass off the body origin: static sag of the soft constraint
    bb = ModelBuilder(timestep=h, contact=False)
    s = bb.body("s", 0, pos=(0, 0, 1)); bb.joint(s, "fs", FREE); bb.geom(s, "gs", SPHERE, size=(0.05,), pos=(0.1, 0, 0), mass=0.25)
    bb.weld(s, 0, solref=(0.006, 1.0))
    ms = bb.compile()
    os_ = ol.Oracle(ms, _copy_task(ms))
    qs, vs = np.array(ms["qpos0"], float), np.zeros(6)
    qs, vs, *_ = os_.step(qs, vs, nstep=8000)          # the tilt mode is slow
    assert np.abs(vs).max() < 1e-6
    cs = os_.constraints(qs)
    tc, dmax = 0.006, 0.95
    K = 1 / (dmax * dmax * tc * tc)
    imp = 1 / (1 + cs["R"][2] / cs["diag"][2])
    assert cs["pos"][2] == pytest.approx(-0.25 * 9.81 * cs["R"][2] / (K * imp), rel=1e-5)          # D K imp pos = - m g
    assert abs(cs["pos"][2]) < 1e-3 and 1e-5 < np.abs(cs["pos"][3:]).max() < 5e-3                    # small tilt from the off-centre weight


@pytest.mark.parametrize("cone, condim", [(0, 3), (0, 4), (0, 6), (1, 3), (1, 4), (1, 6)])
def test_noslip_pass_removes_the_creep_of_soft_contacts(cone, condim):
    """mj_solNoSlip (opt.noslip_iterations): a box resting under gravity with a tangential component below the friction limit.  The
    soft friction rows let it creep (velocity ~ R * force); re-solving the friction dimensions without the regulariser stops it."""
    def drift(ns):
        b = ModelBuilder(timestep=0.005, gravity=(1.5, 0.4, -9.81), cone=cone, contact=True)
        b.noslip_iterations = ns
        b.geom(0, "floor", PLANE, size=(0, 0, 0.05))
        o = b.body("box", 0, pos=(0, 0, 0.05)); b.joint(o, "f", FREE)
        b.geom(o, "g", BOX, size=(0.05, 0.04, 0.05), condim=condim, friction=(0.5, 0.005, 0.0001), mass=0.2)
        m = b.compile()
        q, v, *_ = ol.Oracle(m, _copy_task(m)).step(np.array(m["qpos0"], float), np.zeros(6), nstep=200)
        return np.hypot(*v[:2])
    creep, held = drift(0), drift(5)
    assert creep > 3e-4 and held < creep / 100


def test_noslip_leaves_normal_forces_and_limits_alone():
    """the pass only touches friction-loss rows and friction dimensions: a frictionless (condim 1) contact and a joint at its limit
    give the same step with and without it; a joint with friction loss held by it against a load below the loss does not move at all"""
    def build(ns, condim=1, floss=0.0):
        b = ModelBuilder(timestep=0.004, contact=True)
        b.noslip_iterations = ns
        b.geom(0, "floor", PLANE, size=(0, 0, 0.05), condim=condim)
        ball = b.body("ball", 0, pos=(0, 0, 0.049)); b.joint(ball, "f", FREE); b.geom(ball, "g", SPHERE, size=(0.05,), condim=condim, mass=0.3)
        arm = b.body("arm", 0, pos=(1, 0, 0.5)); b.joint(arm, "h", HINGE, axis=(0, 1, 0), limited=True, range=(-0.2, 0.2), frictionloss=floss)
        b.geom(arm, "ga", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0.3, 0, 0), mass=0.2)
        m = b.compile()
        return m, ol.Oracle(m, _copy_task(m))
    m, o5 = build(5); _, o0 = build(0)
    q = np.array(m["qpos0"], float); q[7] = 0.21; v = np.zeros(7); v[0] = 0.3
    a, b_ = o5.step(q, v, nstep=20), o0.step(q, v, nstep=20)
    assert np.abs(a[0] - b_[0]).max() < 1e-13 and np.abs(a[1] - b_[1]).max() < 1e-12
    # gravity torque on the arm 0.2 * 9.81 * 0.15 = 0.29 < frictionloss 0.5: soft row -> slow creep; noslip -> none
    m, o5 = build(5, floss=0.5); _, o0 = build(0, floss=0.5)
    q = np.array(m["qpos0"], float); v = np.zeros(7)
    a, b_ = o5.step(q, v, nstep=50), o0.step(q, v, nstep=50)
    assert abs(b_[1][6]) > 1e-4 and abs(a[1][6]) < 1e-9


def test_site_transmission_with_a_reference_site():
    """mj_transmission, mjTRN_SITE with refsite: (i) the Fingers arrangement - a body on three slide joints, servos on its site
    relative to a world site - is the same system as position servos on the joints themselves; (ii) in general the length is the
    site's position in the reference site's frame dotted with the gear, and the moment is its derivative along the dofs that do
    not turn the reference frame (central differences), cleared on the dofs both sites hang from."""
    def finger(site_servo):
        b = ModelBuilder(timestep=0.005, contact=False)
        world = b.site(0, "world")
        f = b.body("f", 0, pos=(0.1, -0.2, 0.3))
        for ax, v in (("x", (1, 0, 0)), ("y", (0, 1, 0)), ("z", (0, 0, 1))):
            b.joint(f, ax, SLIDE, axis=v, damping=0.3)
        b.geom(f, "g", SPHERE, size=(0.02,), mass=0.05)
        s = b.site(f, "s")
        for k, ax in enumerate("xyz"):
            g6 = [0.0] * 6; g6[k] = 1.0
            kw = dict(gainprm=(200, 0, 0), biastype=1, biasprm=(0, -200, 0), ctrlrange=(-0.99, 0.99), dyntype=1, actlimited=True, actrange=(-1, 1))
            if site_servo:
                b.actuator(f"a{ax}", site=s, refsite=world, gear6=g6, **kw)
            else:
                b.actuator(f"a{ax}", ax, **kw)
        m = b.compile()
        return m, ol.Oracle(m, _copy_task(m))
    ms, os_ = finger(True); mj, oj = finger(False)
    assert list(ms["actuator_refsite"]) == [0, 0, 0] and list(mj["actuator_refsite"]) == [-1, -1, -1]
    # joint transmissions measure qpos, the site one the world position: start the activations accordingly
    qs, vs = np.zeros(3), np.array([0.2, -0.1, 0.3])
    act_site, act_joint = np.array([0.15, -0.25, 0.4]), np.array([0.05, -0.05, 0.1])
    def run(o, m, act):
        P, H, N = 3, 40, 2
        kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.array([[0.5, -0.3, 0.2]] * P)
        eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
        return o.plan(np.concatenate([qs, vs, act]), None, 0.0, kt, kv, 0, N, H, sigma=(0.0, 0.0), noise_eps=eps, noise_sel=sel, nthreads=2)["states"][0]
    a, b_ = run(os_, ms, act_site), run(oj, mj, act_joint)
    assert np.abs(a[:, :6] - b_[:, :6]).max() < 1e-12 and np.abs(a[-1, :3]).max() > 0.01
    # (ii) a site on a two-link arm against a tilted reference site on a sliding cart (another tree) and against one on the first link
    b = ModelBuilder(timestep=0.005, contact=False)
    cart = b.body("cart", 0, pos=(0.5, 0.2, 0)); b.joint(cart, "cx", SLIDE, axis=(1, 0, 0)); b.geom(cart, "gc", BOX, size=(0.05, 0.05, 0.05), mass=1.0)
    ref = b.site(cart, "ref", pos=(0.02, 0.0, 0.05), quat=(0.9, 0.1, -0.3, 0.2))
    l1 = b.body("l1", 0, pos=(0, 0, 0.4)); b.joint(l1, "h1", HINGE, axis=(0, 1, 0)); b.geom(l1, "g1", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0.3, 0, 0), mass=0.3)
    ref1 = b.site(l1, "ref1", pos=(0.1, 0, 0.02))
    l2 = b.body("l2", l1, pos=(0.3, 0, 0)); b.joint(l2, "h2", HINGE, axis=(0, 0, 1)); b.geom(l2, "g2", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0.2, 0, 0), mass=0.2)
    tip = b.site(l2, "tip", pos=(0.2, 0.01, 0))
    gear = (0.3, -1.0, 0.5)
    b.actuator("cartref", site=tip, refsite=ref, gear6=gear + (0, 0, 0), gainprm=(0, 0, 0), biastype=1, biasprm=(0, -1, 0))     # force = -length
    b.actuator("linkref", site=tip, refsite=ref1, gear6=gear + (0, 0, 0), gainprm=(0, 0, 0), biastype=1, biasprm=(0, -1, 0))
    m = b.compile()
    o = ol.Oracle(m, _copy_task(m))
    q = np.array([0.13, 0.4, -0.7])
    f0, qf = o.actuation(q)
    eps = 1e-6
    grad = np.array([(o.actuation(q + eps * np.eye(3)[d])[0] - o.actuation(q - eps * np.eye(3)[d])[0]) / (2 * eps) for d in range(3)])   # d(-length)/dq
    # both actuators at once: qfrc = sum_a moment_a * force_a; separate them by switching one bias off
    def moment(a):
        mm = dict(m); bp = np.array(m["actuator_biasprm"], float).copy(); bp[1 - a] = 0; mm["actuator_biasprm"] = bp
        fa, qa = ol.Oracle(mm, _copy_task(mm)).actuation(q)
        return qa / fa[a]
    m0, m1 = moment(0), moment(1)
    assert np.allclose(m0, -grad[:, 0], atol=1e-8)                        # nothing turns the cart's frame: the moment is the full derivative
    assert m1[0] == 0 and m1[1] == 0                                      # h1 carries both sites: cleared (and the cart does not move either)
    assert m1[2] == pytest.approx(-grad[2, 1], abs=1e-8)                  # h2 moves the tip only
    # the length itself: R_ref^T (x_tip - x_ref) . gear
    from mujoco_mpc_amd.modelgen.builder import kinematics, quat2mat, quat_mul
    xpos, xquat, xmat, *_ = kinematics(m, q)
    x_tip = xpos[l2] + xmat[l2] @ np.array([0.2, 0.01, 0]); x_ref = xpos[cart] + xmat[cart] @ np.array([0.02, 0.0, 0.05])
    R_ref = xmat[cart] @ quat2mat(np.array([0.9, 0.1, -0.3, 0.2]) / np.linalg.norm([0.9, 0.1, -0.3, 0.2]))
    assert -f0[0] == pytest.approx(np.array(gear) @ (R_ref.T @ (x_tip - x_ref)), rel=1e-12)


def test_implicitfast_integrator_closed_form():
    """mjINT_IMPLICITFAST on one hinge with a position servo (kp, kv) and joint damping b:  v' = v + h tau / (I + h (b + g^2 kv)),
    tau = g (kp (u - g q) - kv g v) - b v;  Euler keeps only b in the denominator; with the servo force on its range the velocity
    term drops out of the denominator again."""
    h, kp, kv, bdamp, g = 0.01, 30.0, 2.0, 0.1, 1.5
    inertia = 2.0 * (0.5 ** 2 + 0.4 * 0.1 ** 2)

    def build(integrator, forcerange=None):
        b = ModelBuilder(timestep=h, gravity=(0, 0, 0), contact=False, integrator=integrator)
        l1 = b.body("l1", 0)
        b.joint(l1, "h", HINGE, axis=(0, 1, 0), damping=bdamp)
        b.geom(l1, "g", SPHERE, size=(0.1,), pos=(0, 0, -0.5), mass=2.0)
        b.actuator("s", "h", gainprm=(kp, 0, 0), biastype=1, biasprm=(0, -kp, -kv), gear=g, ctrlrange=(-1, 1),
                   forcelimited=forcerange is not None, forcerange=forcerange or (0, 0))
        m = b.compile()
        return ol.Oracle(m, make_task(3, [(1, 0, 1.0), (1, 0, 1.0)]))
    q, v, u = 0.2, 1.5, 0.6
    frc = kp * (u - g * q) - kv * g * v
    tau = g * frc - bdamp * v
    for integ, extra in ((0, 0.0), (3, g * g * kv)):
        q1, v1, *_ = build(integ).step([q], [v], ctrl=[u], nstep=1)
        assert v1[0] == pytest.approx(v + h * tau / (inertia + h * (bdamp + extra)), rel=1e-12)
        assert q1[0] == pytest.approx(q + h * v1[0], rel=1e-12)
    lim = 0.5 * abs(frc)                                                     # the servo saturates: force = -lim, no velocity derivative
    q1, v1, *_ = build(3, (-lim, lim)).step([q], [v], ctrl=[u], nstep=1)
    assert v1[0] == pytest.approx(v + h * (g * np.sign(frc) * lim - bdamp * v) / (inertia + h * bdamp), rel=1e-12)


def test_activation_states_closed_forms():
    """mj_fwdActuation / mj_advance with stateful actuators (na > 0): filter act' = (ctrl - act) / tau under Euler, filterexact with
    the exact decay, a clamped integrator; the force of a stateful actuator is gain * act (velocity of a free hinge follows)."""
    h, tau, gain = 0.004, 0.05, 2.0
    b = ModelBuilder(timestep=h, gravity=(0, 0, 0), contact=False)
    l1 = b.body("l1", 0)
    b.joint(l1, "h", HINGE, axis=(0, 1, 0))
    b.geom(l1, "g", SPHERE, size=(0.1,), pos=(0, 0, -0.5), mass=2.0)
    l2 = b.body("l2", 0, pos=(1, 0, 0))
    b.joint(l2, "h2", HINGE, axis=(0, 1, 0))
    b.geom(l2, "g2", SPHERE, size=(0.1,), pos=(0, 0, -0.5), mass=2.0)
    b.actuator("f", "h", gainprm=(gain, 0, 0), ctrlrange=(-1, 1), dyntype=2, dynprm=tau)
    b.actuator("e", "h2", gainprm=(0.0, 0, 0), ctrlrange=(-1, 1), dyntype=3, dynprm=tau)
    b.actuator("i", "h2", gainprm=(0.0, 0, 0), ctrlrange=(-1, 1), dyntype=1, actlimited=True, actrange=(-0.1, 0.1))
    m = b.compile()
    assert m["na"] == 3 and list(m["actuator_actadr"]) == [0, 1, 2]
    task = make_task(3, [(2, 0, 1.0), (2, 0, 1.0), (3, 0, 1.0)])
    o = ol.Oracle(m, task)
    H, u = 60, np.array([0.8, -0.5, 0.7])
    a = o.plan(np.array([0, 0, 0, 0, 0.1, 0.2, -0.05]), None, 0.0, np.array([0.0]), u[None, :], 0, 1, H, sigma=(0.0, 0.0), nthreads=1)
    k = np.arange(H)
    act = a["states"][0, :, 4:]
    assert np.allclose(act[:, 0], u[0] + (0.1 - u[0]) * (1 - h / tau) ** k, rtol=1e-12, atol=1e-14)          # Euler filter
    assert np.allclose(act[:, 1], u[1] + (0.2 - u[1]) * np.exp(-k * h / tau), rtol=1e-12, atol=1e-14)        # exact filter
    assert np.allclose(act[:, 2], np.minimum(-0.05 + k * h * u[2], 0.1), rtol=1e-12, atol=1e-14)            # clamped integrator
    inertia = 2.0 * (0.5 ** 2 + 0.4 * 0.1 ** 2)
    vel = np.concatenate([[0.0], np.cumsum(h * gain * act[:-1, 0] / inertia)])                               # force = gain * act
    assert np.allclose(a["states"][0, :, 2], vel, rtol=1e-10, atol=1e-13)
    assert np.array_equal(a["residual"][0, :, 4:], act)                                                      # residual row t sees act_t


def test_tendon_friction_loss_row():
    """mjCNSTR_FRICTION_TENDON: a friction row along the tendon.  (i) far from rest the row saturates: joint torque = coef *
    frictionloss against the motion; (ii) a one-joint tendon with coefficient c and friction loss f is the same constraint as a
    joint friction loss c * f (R scales with c^2 through tendon_invweight0, the zone edges with c)."""
    def pend(tendon_f, dof_f):
        b = ModelBuilder(timestep=0.002, gravity=(0, 0, 0))
        l1 = b.body("l1", 0)
        b.joint(l1, "h", HINGE, axis=(0, 1, 0), frictionloss=dof_f)
        b.geom(l1, "g", SPHERE, size=(0.1,), pos=(0, 0, -0.5), mass=2.0)
        b.tendon("t", ["h"], [1.5], frictionloss=tendon_f)
        return b.compile()
    mt, md = pend(0.4, 0.0), pend(0.0, 0.6)
    ot, od = ol.Oracle(mt, _copy_task(mt)), ol.Oracle(md, _copy_task(md))
    inertia = 2.0 * (0.5 ** 2 + 0.4 * 0.1 ** 2)
    for v in (3.0, -2.0):
        r = ot.forward([0.2], [v])
        assert r["qacc"][0] == pytest.approx(-np.sign(v) * 1.5 * 0.4 / inertia, rel=1e-9)
    for v in (3.0, 0.5, 1e-3, 0.0, -2e-4, -1.0):      # saturated and quadratic zones
        assert ot.forward([0.2], [v])["qacc"][0] == pytest.approx(od.forward([0.2], [v])["qacc"][0], rel=1e-9, abs=1e-12)
    # a rollout: the pendulum under gravity comes to rest where the tendon friction holds it
    mt["gravity"] = np.array([0, 0, -9.81]); md["gravity"] = np.array([0, 0, -9.81])
    ot, od = ol.Oracle(mt, _copy_task(mt)), ol.Oracle(md, _copy_task(md))
    (qt, vt, *_), (qd, vd, *_) = ot.step([0.8], [0.0], nstep=3000), od.step([0.8], [0.0], nstep=3000)
    assert qt[0] == pytest.approx(qd[0], rel=1e-7) and vt[0] == pytest.approx(vd[0], abs=1e-9)
    assert abs(vt[0]) < 5e-3 and 0 < abs(np.sin(qt[0])) <= 0.6 / (2.0 * 9.81 * 0.5) * 1.05      # held (soft constraint: slow creep) inside the static band


def test_portal_refinement_collider_against_closed_forms():
    """Convex pairs without an analytic collider (cylinder-cylinder, cylinder-box; MuJoCo: libccd MPR) go through the portal
    refinement collider of oracle/collide.c.  Type 100 + t sends any supported primitive through it: spheres must reproduce the
    closed form to round-off; a cylinder on a box face / lying on it / hovering inside the margin / crossing another cylinder
    must give the known depth, normal (geom 1 -> geom 2) and a contact half way between the surfaces, to the 1e-6 tolerance."""
    I = np.eye(3)
    rng = np.random.default_rng(0)

    def randrot():
        q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    for _ in range(500):
        r1, r2 = rng.uniform(0.02, 0.2, 2); d = rng.normal(size=3); d /= np.linalg.norm(d); gap = rng.uniform(-0.5, 0.3) * min(r1, r2)
        c, n = _collide(102, [r1], [0, 0, 0], randrot(), 2, [r2], d * (r1 + r2 + gap), randrot(), margin=0.01)
        if gap <= 0.01:
            assert n == 1 and abs(c[0, 0] - gap) < 1e-10 and np.abs(c[0, 4:] - d).max() < 1e-9 and np.abs(c[0, 1:4] - d * (r1 + gap / 2)).max() < 1e-9
        else:
            assert n == 0
    pen = 1e-3
    c, n = _collide(5, [0.03, 0.05], [0.01, 0.02, 0.1 + 0.05 - pen], I, 6, [0.5, 0.5, 0.1], [0, 0, 0], I)          # standing on the face
    assert n == 1 and abs(c[0, 0] + pen) < 1e-6 and np.allclose(c[0, 4:], [0, 0, -1], atol=1e-6) and abs(c[0, 3] - (0.1 - pen / 2)) < 1e-6
    assert np.hypot(c[0, 1] - 0.01, c[0, 2] - 0.02) <= 0.03 + 1e-9                                                   # inside the cap's footprint
    c, n = _collide(5, [0.03, 0.05], [0.01, 0.02, 0.1 + 0.03 - pen], _rot("x", np.pi / 2), 6, [0.5, 0.5, 0.1], [0, 0, 0], I)   # lying on it
    assert n == 1 and abs(c[0, 0] + pen) < 1e-6 and np.allclose(c[0, 4:], [0, 0, -1], atol=1e-5) and abs(c[0, 1] - 0.01) < 1e-3
    c, n = _collide(5, [0.03, 0.05], [0, 0, 0.1 + 0.05 + 5e-4], I, 6, [0.5, 0.5, 0.1], [0, 0, 0], I, margin=1e-3)       # inside the margin
    assert n == 1 and abs(c[0, 0] - 5e-4) < 1e-6 and abs(c[0, 3] - 0.10025) < 1e-6
    _, n = _collide(5, [0.03, 0.05], [0, 0, 0.1 + 0.05 + 2e-3], I, 6, [0.5, 0.5, 0.1], [0, 0, 0], I, margin=1e-3)
    assert n == 0
    c, n = _collide(5, [0.03, 0.1], [0, 0, 0], _rot("x", np.pi / 2), 5, [0.02, 0.1], [0, 0, 0.05 - pen], _rot("y", np.pi / 2))   # crossed
    assert n == 1 and abs(c[0, 0] + pen) < 1e-6 and np.allclose(c[0, 4:], [0, 0, 1], atol=1e-5) and np.allclose(c[0, 1:4], [0, 0, 0.03 - pen / 2], atol=1e-4)
    # ellipsoids: plane-ellipsoid is a closed form (height of the support point: sqrt(sum (a_i (R^T n)_i)^2)); an ellipsoid with
    # three equal semi-axes is a sphere for the portal collider
    for _ in range(200):
        R = randrot(); a = rng.uniform(0.02, 0.1, 3); z = rng.uniform(0.0, 0.12)
        h = np.sqrt(np.sum((a * (R.T @ np.array([0, 0, 1.0]))) ** 2))
        c, n = _collide(0, [1, 1, 0.1], [0, 0, 0], I, 4, a, [0.3, -0.2, z], R, margin=0.002)
        if z - h <= 0.002:
            assert n == 1 and abs(c[0, 0] - (z - h)) < 1e-12 and np.allclose(c[0, 4:], [0, 0, 1]) and abs(c[0, 3] - (z - h) / 2) < 1e-12
        else:
            assert n == 0
    for _ in range(200):
        r1, r2 = rng.uniform(0.02, 0.2, 2); d = rng.normal(size=3); d /= np.linalg.norm(d); gap = rng.uniform(-0.3, 0.02) * min(r1, r2)
        c, n = _collide(2, [r1], [0, 0, 0], I, 4, [r2, r2, r2], d * (r1 + r2 + gap), randrot(), margin=0.01)
        assert n == 1 and abs(c[0, 0] - gap) < 1e-9 and np.abs(c[0, 4:] - d).max() < 1e-8
    c, n = _collide(4, [0.06, 0.04, 0.03], [0.1, 0.05, 0.1 + 0.03 - pen], I, 6, [0.5, 0.5, 0.1], [0, 0, 0], I)       # ellipsoid on a box face
    assert n == 1 and abs(c[0, 0] + pen) < 1e-6 and np.allclose(c[0, 4:], [0, 0, -1], atol=1e-4) and np.allclose(c[0, 1:3], [0.1, 0.05], atol=2e-3)
    # a cylinder resting on the box of the cylinder_pile model stays there (one contact under the cap, friction holds it)
    from mujoco_mpc_amd.modelgen import cylinder_pile
    m, task, d = cylinder_pile()
    o = ol.Oracle(m, task)
    q, v, _, _, w = o.step(d["state"][:m["nq"]], d["state"][m["nq"]:], ctrl=[0.0], nstep=150)
    assert w == 0 and abs(q[2] - 0.26) < 3e-3 and np.hypot(q[0], q[1]) < 5e-3 and abs(q[9] - 0.35) < 1e-2


def _collide_mesh(t1, s1, p1, m1, v1, t2, s2, p2, m2, v2, margin=0.0):
    import ctypes as C
    L = ol.lib()
    dp = C.POINTER(C.c_double)
    L.oracle_debug_collide_mesh.argtypes = [C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, C.c_int, C.c_double, dp]
    arr = lambda x: np.ascontiguousarray(x, float).ravel()
    a = [arr(list(s1) + [0.0] * (3 - len(s1))), arr(p1), arr(m1), arr(v1 if v1 is not None else [0.0]),
         arr(list(s2) + [0.0] * (3 - len(s2))), arr(p2), arr(m2), arr(v2 if v2 is not None else [0.0])]
    out = np.zeros(14)
    n = L.oracle_debug_collide_mesh(t1, *[x.ctypes.data_as(dp) for x in a[:4]], 0 if v1 is None else len(v1),
                                    t2, *[x.ctypes.data_as(dp) for x in a[4:]], 0 if v2 is None else len(v2), margin, out.ctypes.data_as(dp))
    return out[:7 * max(n, 0)].reshape(max(n, 0), 7), n


def test_convex_mesh_collider_against_closed_forms():
    """Convex meshes (collision = hull of the vertices, support = vertex farthest along the direction): a plane meets the lowest
    vertex; a cube given as an 8-vertex mesh behaves like the box primitive against a sphere and on a box face."""
    I = np.eye(3)
    rng = np.random.default_rng(1)
    tet = np.array([[0.1, 0, -0.03], [-0.05, 0.08, -0.03], [-0.05, -0.08, -0.03], [0, 0, 0.09]])
    for _ in range(100):
        q = rng.normal(size=4); q /= np.linalg.norm(q); w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        pz = rng.uniform(0.0, 0.12)
        low = (tet @ R.T)[:, 2].min() + pz
        c, n = _collide_mesh(0, [1, 1, 0.1], [0, 0, 0], I, None, 7, [0.1, 0.1, 0.1], [0.2, -0.1, pz], R, tet, margin=0.003)
        if low <= 0.003:
            assert n == 1 and abs(c[0, 0] - low) < 1e-12 and np.allclose(c[0, 4:], [0, 0, 1]) and abs(c[0, 3] - low / 2) < 1e-12
        else:
            assert n == 0
    h = 0.05
    cube = np.array([[sx * h, sy * h, sz * h] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], float)
    for _ in range(100):                                   # sphere over a face of the cube: same contact as the box primitive
        r = rng.uniform(0.02, 0.06); xy = rng.uniform(-0.03, 0.03, 2); gap = rng.uniform(-1e-3, 5e-4)
        p = [xy[0], xy[1], h + r + gap]
        a, na = _collide(2, [r], p, I, 6, [h, h, h], [0, 0, 0], I, margin=1e-3)
        b, nb = _collide_mesh(2, [r], p, I, None, 7, [h, h, h], [0, 0, 0], I, cube, margin=1e-3)
        # (depth to the 1e-6 tolerance; on a curved feature the witness point / normal are only good to ~sqrt(tolerance * radius))
        assert na == nb == 1 and abs(a[0, 0] - b[0, 0]) < 1e-6 and np.allclose(a[0, 4:], b[0, 4:], atol=1e-3) and np.allclose(a[0, 1:4], b[0, 1:4], atol=3e-4)
    c, n = _collide_mesh(7, [h, h, h], [0.1, 0.05, 0.1 + h - 1e-3], _rot("z", 0.4), cube, 6, [0.5, 0.5, 0.1], [0, 0, 0], I, None)      # cube mesh on a box face
    assert n == 1 and abs(c[0, 0] + 1e-3) < 1e-6 and np.allclose(c[0, 4:], [0, 0, -1], atol=1e-6) and abs(c[0, 3] - (0.1 - 5e-4)) < 1e-6


def _collide_hfield(hsize, data, t2, s2, p2, m2, rbound, margin=0.0):
    import ctypes as C
    L = ol.lib()
    dp = C.POINTER(C.c_double)
    L.oracle_debug_collide_hfield.argtypes = [dp, C.c_int, C.c_int, dp, dp, dp, C.c_int, dp, dp, dp, C.c_double, C.c_double, dp]
    arr = lambda x: np.ascontiguousarray(x, float).ravel()
    data = np.asarray(data, float)
    a = [arr(hsize), arr(data), arr([0, 0, 0]), arr(np.eye(3)), arr(list(s2) + [0.0] * (3 - len(s2))), arr(p2), arr(m2)]
    out = np.zeros(28)
    n = L.oracle_debug_collide_hfield(a[0].ctypes.data_as(dp), data.shape[0], data.shape[1], a[1].ctypes.data_as(dp), a[2].ctypes.data_as(dp),
                                      a[3].ctypes.data_as(dp), t2, a[4].ctypes.data_as(dp), a[5].ctypes.data_as(dp), a[6].ctypes.data_as(dp), rbound, margin,
                                      out.ctypes.data_as(dp))
    return out[:7 * max(n, 0)].reshape(max(n, 0), 7), n


def test_height_field_collider_against_closed_forms():
    """Height field vs convex geom = the terrain prisms under the geom through the portal-refinement collider: on a flat field a
    sphere meets the plane z = height (deepest contact), on a ramp the plane of the ramp; nothing above the geom's reach."""
    I = np.eye(3)
    flat = np.full((9, 9), 0.5)                          # height 0.5 * elevation 0.4 = 0.2
    hs = [1.0, 1.0, 0.4, 0.1]
    for x, y in ((0.03, 0.04), (-0.4, 0.31), (0.13, -0.02)):
        c, n = _collide_hfield(hs, flat, 2, [0.05], [x, y, 0.2 + 0.05 - 1e-3], I, 0.05)
        assert n >= 1
        k = np.argmin(c[:, 0])
        assert abs(c[k, 0] + 1e-3) < 1e-5 and np.allclose(c[k, 4:], [0, 0, 1], atol=1e-3) and abs(c[k, 3] - (0.2 - 5e-4)) < 1e-4
    _, n = _collide_hfield(hs, flat, 2, [0.05], [0.1, 0.1, 0.2 + 0.05 + 1e-3], I, 0.05)
    assert n == 0
    c, n = _collide_hfield(hs, flat, 2, [0.05], [0.1, 0.1, 0.2 + 0.05 + 5e-4], I, 0.05, margin=1e-3)
    assert n >= 1 and abs(c[:, 0].min() - 5e-4) < 1e-5
    ramp = np.tile(np.linspace(0.0, 1.0, 9), (9, 1))     # height = 0.2 * (x + 1): a plane of slope 0.2
    nrm = np.array([-0.2, 0, 1.0]); nrm /= np.linalg.norm(nrm)
    x0 = 0.17; z0 = 0.2 * (x0 + 1)
    p = np.array([x0, 0.1, z0]) + nrm * (0.05 - 1e-3)
    c, n = _collide_hfield(hs, ramp, 2, [0.05], p, I, 0.05)
    assert n >= 1
    k = np.argmin(c[:, 0])
    assert abs(c[k, 0] + 1e-3) < 1e-5 and np.allclose(c[k, 4:], nrm, atol=1e-3)
    # a box lying on the flat field touches many prisms (one contact each): the four deepest are kept
    c, n = _collide_hfield(hs, flat, 6, [0.2, 0.15, 0.05], [0.05, 0.02, 0.2 + 0.05 - 1e-3], I, 0.26)
    assert n == 4 and np.allclose(c[:, 0], -1e-3, atol=1e-5)


def test_implicit_integrator_velocity_derivatives_against_finite_differences():
    """mjINT_IMPLICIT (mj_implicit, mjd_smooth_vel): (a) d qfrc_bias / d qvel - the oracle takes it as the step-1 central difference
    of its own RNE, exact because the bias force is a quadratic polynomial of qvel: checked here against small-step differences
    AND by the polynomial identity itself (the same matrix from steps 1 and 1e-3); (b) the inertia-box fluid derivative
    (mjd_inertiaBoxFluid), analytic in the oracle, against finite differences of qfrc_passive; (c) structure: the bias derivative of
    a body at rest vanishes, the fluid one is symmetric positive semi-definite."""
    from mujoco_mpc_amd.modelgen import humanoid_walk, swimmer
    rng = np.random.default_rng(3)
    for gen in (swimmer, humanoid_walk):
        m, task, d = gen()
        o = ol.Oracle(m, task)
        nq, nv = m["nq"], m["nv"]
        qpos = d["state"][:nq].copy()
        qvel = rng.normal(0, 1.5, nv)
        base = o.vel_derivatives(qpos, qvel)
        assert base["warning"] == 0
        for eps, tol in ((1e-3, 1e-9), (1e-6, 2e-6)):
            fd_b = np.zeros((nv, nv)); fd_f = np.zeros((nv, nv))
            for j in range(nv):
                e = np.zeros(nv); e[j] = eps
                p, q = o.vel_derivatives(qpos, qvel + e), o.vel_derivatives(qpos, qvel - e)
                fd_b[:, j] = (p["qfrc_bias"] - q["qfrc_bias"]) / (2 * eps)
                fd_f[:, j] = -(p["qfrc_passive"] - q["qfrc_passive"]) / (2 * eps)
            scale = np.abs(base["dbias"]).max() + 1e-300
            assert np.abs(fd_b - base["dbias"]).max() / scale < tol, (gen.__name__, eps)          # quadratic: step 1 == step 1e-3 to round-off
            if gen is swimmer and eps == 1e-6:
                damp = np.diag(m["dof_damping"])                                              # qfrc_passive also holds -damping * qvel
                assert np.abs(fd_f - damp - base["dfluid"]).max() / (np.abs(base["dfluid"]).max() + 1e-300) < 1e-5
        rest = o.vel_derivatives(qpos, np.zeros(nv))
        assert np.abs(rest["dbias"]).max() < 1e-12
        if gen is swimmer:
            assert np.abs(base["dfluid"] - base["dfluid"].T).max() < 1e-12 * np.abs(base["dfluid"]).max()
            assert np.linalg.eigvalsh(0.5 * (base["dfluid"] + base["dfluid"].T)).min() > -1e-9 * np.abs(base["dfluid"]).max()
        else:
            assert np.abs(base["dfluid"]).max() == 0.0
        assert np.abs(base["dbias"] - base["dbias"].T).max() > 1e-3 * np.abs(base["dbias"]).max()     # really non-symmetric: LU, not Cholesky


def test_implicit_integrator_step_solves_its_linear_system():
    """one step of each integrator from the same state: v+ - v = h (M - h dF/dv)^-1 (qfrc_smooth + qfrc_constraint) with dF/dv =
    -(damping + fluid derivative) for implicitfast and additionally -d bias / d qvel for implicit (mj_implicit); at rest all three
    integrators coincide with Euler's implicit-damping step"""
    from mujoco_mpc_amd.modelgen import swimmer
    rng = np.random.default_rng(5)
    steps = {}
    for integ in (0, 3, 2):
        m, task, d = swimmer(integrator=integ)
        o = ol.Oracle(m, task)
        nq, nv, h = m["nq"], m["nv"], m["timestep"]
        qpos = d["state"][:nq].copy(); qvel = rng.normal(0, 2.0, nv) if integ == 0 else steps["qvel0"]
        steps.setdefault("qvel0", qvel)
        f = o.forward(qpos, qvel)
        der = o.vel_derivatives(qpos, qvel)
        rhs = f["qM"] @ f["qacc"]                      # = qfrc_smooth + qfrc_constraint (no constraints in the swimmer)
        A = f["qM"] + h * np.diag(m["dof_damping"])
        if integ in (2, 3):
            A = A + h * der["dfluid"]
        if integ == 2:
            A = A + h * der["dbias"]
        _, v1, _, _, w = o.step(qpos, qvel)
        assert w == 0
        assert np.abs((v1 - qvel) / h - np.linalg.solve(A, rhs)).max() < 1e-9 * np.abs(rhs).max()
        steps[integ] = v1
    assert np.abs(steps[2] - steps[3]).max() > 1e-6 and np.abs(steps[3] - steps[0]).max() > 1e-6          # the three really differ


def test_humanoid_interact_residual_against_closed_forms():
    """interact.cc:31-186 on the oracle, in a tilted pose at rest and in motion: every term from the numpy kinematics of the model
    generator (xmat columns, inertial-frame positions), independent of the oracle's own kinematics; contact pairs and the facing target
    (empty in the reference's default key frame) are exercised through the frozen state."""
    from mujoco_mpc_amd.modelgen import humanoid_interact, kinematics
    pairs = [("hand_right", (0.0, 0.0, -0.05), "chair", (0.1, 0.3, 0.2)), ("pelvis", (0.0, 0.0, -0.1), 0, (-0.35, 0.0, 0.4))]
    m, task, d = humanoid_interact(contact_pairs=pairs, facing_target=(1.0, 2.0))
    assert task["num_residual"] == 68 and task["num_term"] == 13
    o = ol.Oracle(m, task)
    nq, nv, nu = m["nq"], m["nv"], m["nu"]
    qpos = d["state"][:nq].copy()
    rng = np.random.default_rng(0)
    qvel = rng.normal(0, 0.3, nv); ctrl = rng.uniform(-0.5, 0.5, nu)
    r = o.forward(qpos, qvel, ctrl)["sensordata"]
    xpos, xquat, xmat, _, _ = kinematics(m, qpos)
    xmat = np.asarray(xmat).reshape(-1, 3, 3); xpos = np.asarray(xpos).reshape(-1, 3)
    xipos = xpos + np.einsum("bij,bj->bi", xmat, np.asarray(m["body_ipos"]).reshape(-1, 3))
    b = m["names"]["body"]
    for k, name in enumerate(("torso", "pelvis", "foot_right", "foot_left")):
        assert abs(r[k] - abs(xmat[b[name]][2, 2] - 1.0)) < 1e-12
    assert abs(r[4] - abs(xipos[b["head"]][2] - 1.4)) < 1e-12 and abs(r[5] - abs(xipos[b["torso"]][2] - 1.3)) < 1e-12
    knee = 0.5 * (xipos[b["shin_left"]][:2] + xipos[b["shin_right"]][:2]); foot = 0.5 * (xipos[b["foot_left"]][:2] + xipos[b["foot_right"]][:2])
    assert abs(r[6] - np.linalg.norm(knee - foot)) < 1e-12
    f = o.forward(qpos, qvel, ctrl)
    assert abs(r[7] - np.linalg.norm(f["subtree_com"][b["torso"]][:2] - foot)) < 1e-12
    ximat_t = xmat[b["torso"]] @ _quat2mat(np.asarray(m["body_iquat"]).reshape(-1, 4)[b["torso"]])
    tgt = np.array([1.0, 2.0]) - xipos[b["torso"]][:2]; tgt /= np.linalg.norm(tgt)
    assert abs(r[8] - np.linalg.norm(tgt - ximat_t[:2, 0])) < 1e-12
    assert np.allclose(r[11:32], qvel[6:], atol=0) and np.allclose(r[32:53], ctrl, atol=0)
    g1 = xpos[b["hand_right"]] + xmat[b["hand_right"]] @ np.array(pairs[0][1]); g2 = xpos[b["chair"]] + xmat[b["chair"]] @ np.array(pairs[0][3])
    assert np.allclose(r[53:56], np.abs(g1 - g2), atol=1e-12)
    g1 = xpos[b["pelvis"]] + xmat[b["pelvis"]] @ np.array(pairs[1][1])
    assert np.allclose(r[56:59], np.abs(g1 - np.array(pairs[1][3])), atol=1e-12) and np.all(r[59:68] == 0)
    # the reference's default: no pairs, no facing target -> those terms vanish; at rest the velocity terms do too
    m0, task0, d0 = humanoid_interact()
    r0 = ol.Oracle(m0, task0).forward(qpos)["sensordata"]
    assert r0[8] == 0 and np.all(r0[53:] == 0) and np.all(r0[9:53] == 0) and np.allclose(r0[:8], r[:8], atol=1e-12)


def _quat2mat(q):
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
