"""Task models for the BASELINE configs, authored with ModelBuilder.

Numbers are transcribed from the reference's MJCF task files (facts, not code):
  particle   mjpc/test/testdata/particle_task.xml, particle.xml
  cartpole   mjpc/tasks/cartpole/task.xml, cartpole.xml.patch  (+ dm_control cartpole.xml `pole`
             default class and motor, recalled: SURVEY.md Appendix D)
  quadruped  mjpc/tasks/quadruped/task_flat.xml, a1.xml.patch (+ menagerie a1.xml collision
             default classes, recalled: SURVEY.md Appendix D)
Each function returns (model, task, defaults).
"""
from __future__ import annotations

import math
import struct

import numpy as np

from .builder import (BALL, BOX, CAPSULE, CYLINDER, ELLIPSOID, FREE, HFIELD, HINGE, PLANE, SLIDE, SPHERE, ModelBuilder)

TASK_PARTICLE, TASK_CARTPOLE, TASK_QUADRUPED, TASK_COPYSTATE, TASK_HUMANOID_TRACK, TASK_HUMANOID_STAND, TASK_HUMANOID_WALK = 0, 1, 2, 3, 4, 5, 6
TASK_SHADOW_REORIENT = 7
TASK_WALKER, TASK_ACROBOT = 8, 9
OBJ_BODY, OBJ_XBODY, OBJ_GEOM, OBJ_SITE = 1, 2, 5, 6
NORM_NPARAM = {-1: 0, 0: 0, 1: 2, 2: 1, 3: 1, 5: 1, 6: 1, 7: 2, 8: 1}   # mjpc/norm.cc:25-47


def select_value(i: int) -> float:
    """residual_select_* numerics are int64 bit-cast into a double (mjpc/utilities.cc:207-211)."""
    return struct.unpack("<d", struct.pack("<q", int(i)))[0]


def make_task(task_id, terms, parameters=(), risk=0.0, traces=(), int_data=(), dbl_data=()):
    """terms: list of (dim, norm, weight, [norm params]) == the <user> sensors (mjpc/task.cc:203-238)."""
    return dict(
        task_id=task_id,
        num_residual=sum(t[0] for t in terms), num_term=len(terms), num_trace=len(traces),
        dim_norm_residual=np.array([t[0] for t in terms], np.int32),
        norm=np.array([t[1] for t in terms], np.int32),
        num_norm_parameter=np.array([NORM_NPARAM[t[1]] for t in terms], np.int32),
        weight=np.array([t[2] for t in terms], float),
        norm_parameter=np.array([p for t in terms for p in (t[3] if len(t) > 3 else [])][:], float),
        risk=risk, num_parameter=len(parameters), parameters=np.array(parameters, float),
        trace_objtype=np.array([t[0] for t in traces], np.int32), trace_objid=np.array([t[1] for t in traces], np.int32),
        num_int=len(int_data), int_data=np.array(int_data, np.int32),
        num_dbl=len(dbl_data), dbl_data=np.array(dbl_data, float))


# ----------------------------------------------------------------------------------- particle
def particle(timestep=0.1, copystate=False):
    b = ModelBuilder(timestep=timestep, contact=False)
    goal = b.body("goal", 0, pos=(0.25, 0, 0.01), mocap=True)
    b.geom(goal, "goal", SPHERE, size=(0.01,), contype=0, conaffinity=0)
    b.geom(0, "ground", PLANE, size=(0.3, 0.3, 0.1))
    pm = b.body("pointmass", 0, pos=(0, 0, 0.01))
    b.joint(pm, "root_x", SLIDE, axis=(1, 0, 0), limited=True, range=(-0.29, 0.29), damping=1.0)
    b.joint(pm, "root_y", SLIDE, axis=(0, 1, 0), limited=True, range=(-0.29, 0.29), damping=1.0)
    b.geom(pm, "pointmass", SPHERE, size=(0.01,), mass=0.3)
    tip = b.site(pm, "tip")
    b.actuator("x_motor", "root_x", gear=1.0, ctrlrange=(-1, 1))
    b.actuator("y_motor", "root_y", gear=1.0, ctrlrange=(-1, 1))
    b.key("home", [1.0, 2.0])
    m = b.compile()
    if copystate:
        # rollout_test.cc: residual copies the state; 4 residuals, quadratic
        task = make_task(TASK_COPYSTATE, [(2, 0, 5.0), (2, 0, 0.1)], parameters=[0.05, -0.1], risk=1.0,
                         traces=[(OBJ_SITE, tip)])
    else:
        task = make_task(TASK_PARTICLE, [(2, 0, 5.0), (2, 0, 0.1)], parameters=[0.05, -0.1], risk=1.0,
                         traces=[(OBJ_SITE, tip)])
    defaults = dict(N=10, P=11, sigma=(0.01, 0.0), interp=2, horizon=11, state=np.zeros(4),
                    mocap=np.array([0.25, 0, 0.01, 1, 0, 0, 0.0]))
    return m, task, defaults


TASK_PARTICLE_TIMEVARYING, TASK_PARTICLE_FIXED = 11, 12


def particle_task(fixed=False, timestep=0.01):
    """The registry's Particle / ParticleFixed tasks (mjpc/tasks/particle/particle.cc, task_timevarying.xml): the point mass of
    particle() with the 6-residual cost Position (w 5) / Velocity (w 0.1) / Control (w 0.1), risk 1; the goal is a Lissajous curve of
    the time (Particle) or the mocap body (ParticleFixed).  Agent settings of the XML: horizon 0.5 s, 5 spline points, exploration 0.01."""
    m, _, d = particle(timestep=timestep)
    tip = 0
    task = make_task(TASK_PARTICLE_FIXED if fixed else TASK_PARTICLE_TIMEVARYING, [(2, 0, 5.0), (2, 0, 0.1), (2, 0, 0.1)], risk=1.0,
                     traces=[(OBJ_SITE, tip)], int_data=[tip])
    defaults = dict(d, N=10, P=5, sigma=(0.01, 0.0), interp=2, horizon=51, state=np.array([0.05, -0.1, 0.0, 0.0]))
    return m, task, defaults


TASK_SWIMMER = 14


def swimmer(timestep=0.01, integrator=2):
    """mjpc/tasks/swimmer (swimmer.cc:33-61, task.xml, swimmer.xml.patch on dm_control's swimmer): six 10 g links in a medium of
    density 1000 (inertia-box fluid forces), planar root (two slides and a hinge), five limited hinge joints with a weak spring, driven
    through first-order filters (dyntype filter, 0.6 s); contacts disabled; target = a mocap body.  Cost / agent settings are the
    reference's (horizon 2 s, 10 spline points, exploration 0.05), also `agent_integrator` 2 = the full implicit integrator
    (task.xml:11; mjINT_IMPLICIT: velocity derivatives of the fluid and bias forces, LU), the default here since round 3."""
    b = ModelBuilder(timestep=timestep, contact=False, density=1000.0, integrator=integrator)
    b.geom(0, "ground", PLANE, size=(2, 2, 0.01))
    head = b.body("head", 0, pos=(0, 0, 0.05))
    b.joint(head, "rootx", SLIDE, axis=(1, 0, 0), pos=(0, -0.05, 0))
    b.joint(head, "rooty", SLIDE, axis=(0, 1, 0), pos=(0, -0.05, 0))
    b.joint(head, "rootz", HINGE, axis=(0, 0, 1), pos=(0, -0.05, 0))
    b.geom(head, "inertial", BOX, size=(0.001, 0.05, 0.01), mass=0.01)
    nose = b.geom(head, "nose", SPHERE, size=(0.004,), pos=(0, -0.06, 0), mass=0.0)
    parent = head
    for k in range(5):
        seg = b.body(f"segment_{k}", parent, pos=(0, 0.1, 0))
        b.joint(seg, f"joint_{k}", HINGE, axis=(0, 0, 1), pos=(0, -0.05, 0), limited=True, range=(-math.pi / 2, math.pi / 2), stiffness=0.001, armature=1e-6,
                solreflimit=(0.05, 0.3), solimplimit=(0, 0.8, 0.1, 0.5, 2))
        b.geom(seg, f"inertial_{k}", BOX, size=(0.001, 0.05, 0.01), mass=0.01)
        b.actuator(str(k), f"joint_{k}", gainprm=(2e-3, 0, 0), ctrlrange=(-1, 1), dyntype=2, dynprm=0.6)
        parent = seg
    target = b.body("target", 0, pos=(1, 1, 0.05), mocap=True)
    b.geom(target, "target_g", SPHERE, size=(0.05,), contype=0, conaffinity=0)
    m = b.compile()
    task = make_task(TASK_SWIMMER, [(5, 0, 0.1), (2, 2, 10.0, [0.04])], traces=[(OBJ_GEOM, nose)], int_data=[nose, 0])
    state = np.concatenate([m["qpos0"], np.zeros(m["nv"]), np.zeros(m["na"])])
    defaults = dict(N=10, P=10, sigma=(0.05, 0.0), interp=2, horizon=201, state=state, mocap=np.array([0.3, -0.2, 0.05, 1, 0, 0, 0.0]))
    return m, task, defaults


TASK_QUADROTOR = 13
QUADROTOR_STAGES = [(1.2, 0.0, 0.75), (2.3, 0.6, 1.5), (2.7, 0.95, 1.5), (4.6, 0.4, 0.75), (5.0, -1.8, 0.75), (3.4, -2.5, 0.75), (2.5, -2.25, 1.45),
                    (2.5, -2.25, 2.25), (1.5, -1.75, 1.85), (1.05, -1.75, 1.3), (0.1, -1.4, 0.75), (0.0, 0.0, 0.75)]      # task.xml:80-93 (key mpos)


def quadrotor(timestep=0.01, stage=0):
    """mjpc/tasks/quadrotor (quadrotor.cc:37-95, task.xml): a 1.325 kg free body with four thrust motors through site transmissions
    (gear = unit force along the site's z plus the rotor's reaction torque, quadrotor.xml.patch:57-58 for the menagerie Skydio X2),
    floor and two gate posts to bump into; goal = a mocap body that Transition moves through the 12 keyframe positions.  The
    body / rotor geometry is a synthetic box (the menagerie mesh is not in the reference tree); cost terms, agent settings
    (horizon 0.5 s, 5 spline points, exploration 0.3) and stage goals are the reference's."""
    b = ModelBuilder(timestep=timestep, contact=True)
    b.geom(0, "floor", PLANE, size=(100, 100, 0.2))
    x2 = b.body("x2", 0, pos=(0, 0, 0.04), quat=(0, 0, 0, 1))       # resting on the floor, like the reference's start
    b.joint(x2, "x2_free", FREE)
    b.geom(x2, "x2_g", BOX, size=(0.12, 0.16, 0.03), mass=1.325)
    for k, (x, y, tq) in enumerate(((-0.14, -0.18, -0.0201), (-0.14, 0.18, 0.0201), (0.14, 0.18, -0.0201), (0.14, -0.18, 0.0201))):
        s = b.site(x2, f"thrust{k + 1}", pos=(x, y, 0.05))
        b.actuator(f"thrust{k + 1}", site=s, gear6=(0, 0, 1, 0, 0, tq), ctrlrange=(0, 13))
    goal = b.body("goal", 0, pos=QUADROTOR_STAGES[stage], mocap=True)
    b.geom(goal, "goal_g", SPHERE, size=(0.1,), contype=0, conaffinity=0)
    for k, (x, y) in enumerate(((1.2, 0.45), (1.2, -0.45))):                 # gate posts either side of the first waypoint
        b.geom(0, f"post{k}", BOX, pos=(x, y, 0.75), size=(0.03, 0.03, 0.75))
    m = b.compile()
    stages = np.array([list(p) + [1, 0, 0, 0] for p in QUADROTOR_STAGES], float)
    task = make_task(TASK_QUADROTOR, [(3, 0, 25.0), (3, 0, 1.25), (3, 0, 1.25), (4, 0, 1.0e-3), (2, 0, 0.0)], traces=[(OBJ_BODY, x2)],
                     int_data=[x2, int(stage)], dbl_data=list(stages.ravel()))
    q = np.array(m["qpos0"], float)
    hover = 1.325 * 9.81 / 4
    defaults = dict(N=10, P=5, sigma=(0.3, 0.0), interp=2, horizon=51, state=np.concatenate([q, np.zeros(m["nv"])]), mocap=stages[stage].copy(),
                    ctrl0=np.full(4, hover))
    return m, task, defaults


TASK_FINGERS = 16


def fingers(timestep=0.005, noslip_iterations=5, integrator=2, grasp=False):
    """mjpc/tasks/fingers (fingers.cc:31-62, task.xml): two free-floating spherical fingers on three slide joints each, driven by
    integrated-velocity servos (intvelocity: integrator activation, kp 1000, site transmission relative to a world reference site),
    a thin box (condim 6) to bring to a gravity-compensated target pose; elliptic cones, `noslip_iterations` 5, the implicit
    integrator at the agent's 5 ms step.  Residual: finger - object (2 x 3), distance of three object sites to the target's, control."""
    b = ModelBuilder(timestep=timestep, cone=1, contact=True, integrator=integrator)
    b.noslip_iterations = int(noslip_iterations)
    world = b.site(0, "world")
    b.geom(0, "floor", PLANE, size=(0, 0, 0.05))
    obj = b.body("object", 0); b.joint(obj, "red_rgb_object/", FREE)
    b.geom(obj, "object", BOX, size=(0.05, 0.01, 0.1), condim=6, friction=(0.2, 0.005, 0.0001), mass=0.2, solref=(0.004, 1.0))
    s_obj = [b.site(obj, str(k), pos=p) for k, p in enumerate(((0.12, 0, 0), (0, 0.08, 0), (0, 0, 0.08)))]
    tgt = b.body("target", 0, gravcomp=1.0); b.joint(tgt, "joint1", FREE)          # (unnamed in the XML)
    b.geom(tgt, "target", BOX, size=(0.039, 0.008, 0.09), contype=0, conaffinity=0)
    s_tgt = [b.site(tgt, f"{k}t", pos=p) for k, p in enumerate(((0.12, 0, 0), (0, 0.08, 0), (0, 0, 0.08)))]
    fb, fs = [], []
    for name in ("a", "b"):
        f = b.body(f"finger_{name}", 0, gravcomp=1.0)
        for ax, v in (("x", (1, 0, 0)), ("y", (0, 1, 0)), ("z", (0, 0, 1))):
            b.joint(f, f"{name.upper()}_{ax}", SLIDE, axis=v)
        b.geom(f, f"finger_{name}", SPHERE, size=(0.02,), condim=6)
        fb.append(f); fs.append(b.site(f, f"finger_{name}"))
    b.exclude(obj, tgt)
    for name, s in zip(("A", "B"), fs):
        for k, ax in enumerate("xyz"):
            g6 = [0.0] * 6; g6[k] = 1.0
            b.actuator(f"{name}_{ax}", site=s, refsite=world, gear6=g6, gainprm=(1000, 0, 0), biastype=1, biasprm=(0, -1000, 0), ctrlrange=(-0.99, 0.99),
                       dyntype=1, actlimited=True, actrange=(0, 1.4) if ax == "z" else (-1, 1))
    home = [0, 0, 0.3, 1, 0, 0, 0, 0, 0, 0.12, 1, 0, 1, 0, 0, 0.1, 0.3, 0, -0.1, 0.3]
    home = np.array(home, float); home[10:14] /= np.linalg.norm(home[10:14])
    b.key("home", home)
    m = b.compile()
    task = make_task(TASK_FINGERS, [(6, 2, 0.35, [0.02]), (3, 6, 1.0, [0.05]), (6, 6, 0.05, [0.01])], traces=[(OBJ_SITE, fs[0]), (OBJ_SITE, fs[1])],
                     int_data=[fb[0], fb[1], obj] + s_obj + s_tgt)
    # the "home" key puts the OBJECT up at 0.3 and the target on the floor in the XML's body order (object first): [object 7, target 7, A 3, B 3]
    q = home.copy()
    act = np.array([0, 0.1, 0.3, 0, -0.1, 0.3], float)
    if grasp:       # test state: the object stands on the floor, pinched on its thin sides by the fingers, whose servo targets squeeze and lift
        q[0:7] = [0, 0, 0.1, 1, 0, 0, 0]; q[7:14] = [0.3, 0, 0.3, 1, 0, 0, 0]
        q[14:17] = [0, 0.0295, 0.1]; q[17:20] = [0, -0.0295, 0.1]
        act = np.array([0, 0.022, 0.16, 0, -0.022, 0.16], float)
    defaults = dict(N=60, P=5, sigma=(0.04, 0.0), interp=2, horizon=101, state=np.concatenate([q, np.zeros(m["nv"]), act]), mocap=np.zeros(0))
    return m, task, defaults


def noslip_mix(cone=1, condim=3, noslip_iterations=5, timestep=0.005):
    """Test model for the noslip pass: gravity with a tangential component; a box and a capsule on the floor (contacts of the given
    cone / condim), a hinged arm with joint friction loss under its own weight, two sliders tied by a tendon with friction loss and
    a motor pushing one of them against it, a limited joint at its stop.  Residual = state (TASK_COPYSTATE)."""
    b = ModelBuilder(timestep=timestep, gravity=(1.2, 0.5, -9.81), cone=cone, contact=True)
    b.noslip_iterations = int(noslip_iterations)
    b.geom(0, "floor", PLANE, size=(0, 0, 0.05), condim=condim)
    box = b.body("box", 0, pos=(0, 0, 0.05)); b.joint(box, "box_f", FREE)
    b.geom(box, "box_g", BOX, size=(0.05, 0.04, 0.05), condim=condim, friction=(0.5, 0.005, 0.0001), mass=0.2)
    cap = b.body("cap", 0, pos=(0.4, 0, 0.03), quat=(0.92, 0, 0, 0.39)); b.joint(cap, "cap_f", FREE)
    b.geom(cap, "cap_g", CAPSULE, size=(0.03, 0.08), euler=(0, 90, 0), condim=condim, friction=(0.7, 0.01, 0.001), mass=0.15)
    arm = b.body("arm", 0, pos=(-0.5, 0, 0.5)); b.joint(arm, "arm_h", HINGE, axis=(0, 1, 0), frictionloss=0.5, limited=True, range=(-0.4, 0.4))
    b.geom(arm, "arm_g", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0.3, 0, 0), mass=0.2, contype=0, conaffinity=0)
    for k in range(2):
        sl = b.body(f"sl{k}", 0, pos=(-0.5, 0.3 + 0.2 * k, 0.3)); b.joint(sl, f"sl{k}_j", SLIDE, axis=(0, 0, 1), damping=0.2)
        b.geom(sl, f"sl{k}_g", SPHERE, size=(0.03,), mass=0.1, contype=0, conaffinity=0)
    b.tendon("t", ["sl0_j", "sl1_j"], [1.0, -0.5], frictionloss=0.8)
    b.actuator("push", "sl0_j", gear=1.0, ctrlrange=(-1, 1))
    tip = b.site(arm, "tip", pos=(0.3, 0, 0))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(m["nq"], 0, 1.0), (m["nv"], 0, 0.1)], traces=[(OBJ_SITE, tip)])
    q = m["qpos0"].copy(); v = np.zeros(m["nv"])
    q[14] = 0.39; v[0] = 0.2; v[6] = -0.1; v[13] = 0.5
    defaults = dict(N=6, P=4, sigma=(0.5, 0.0), interp=2, horizon=60, state=np.concatenate([q, v]), mocap=np.zeros(0))
    return m, task, defaults


def site_servo(timestep=0.005, integrator=0):
    """Test model for site transmissions with a reference site: the tip of a two-link arm servoed (position gains, one with an
    integrator activation) along three axes of a tilted reference site that rides a sliding cart, and along one axis of a reference
    site on the arm's own first link (moment cleared on the shared hinge).  Residual = state (TASK_COPYSTATE)."""
    b = ModelBuilder(timestep=timestep, contact=False, integrator=integrator)
    cart = b.body("cart", 0, pos=(0.5, 0.2, 0)); b.joint(cart, "cx", SLIDE, axis=(1, 0, 0), damping=0.5); b.geom(cart, "gc", BOX, size=(0.05, 0.05, 0.05), mass=1.0)
    ref = b.site(cart, "ref", pos=(0.02, 0.0, 0.05), quat=(0.9, 0.1, -0.3, 0.2))
    l1 = b.body("l1", 0, pos=(0, 0, 0.4)); b.joint(l1, "h1", HINGE, axis=(0, 1, 0), damping=0.05); b.geom(l1, "g1", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0.3, 0, 0), mass=0.3)
    ref1 = b.site(l1, "ref1", pos=(0.1, 0, 0.02))
    l2 = b.body("l2", l1, pos=(0.3, 0, 0)); b.joint(l2, "h2", HINGE, axis=(0, 0, 1), damping=0.05); b.geom(l2, "g2", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0.2, 0, 0), mass=0.2)
    tip = b.site(l2, "tip", pos=(0.2, 0.01, 0))
    for k, ax in enumerate("xyz"):
        g6 = [0.0] * 6; g6[k] = 1.0
        b.actuator(f"ref_{ax}", site=tip, refsite=ref, gear6=g6, gainprm=(8, 0, 0), biastype=1, biasprm=(0.1, -8, -0.5 if integrator == 0 else 0.0), ctrlrange=(-0.5, 0.5),
                   dyntype=1 if k == 2 else 0, actlimited=k == 2, actrange=(-0.3, 0.3))
    b.actuator("link_y", site=tip, refsite=ref1, gear6=(0.2, 1.0, 0, 0, 0, 0), gainprm=(5, 0, 0), biastype=1, biasprm=(0, -5, 0), ctrlrange=(-0.5, 0.5), forcelimited=True, forcerange=(-0.6, 0.6))
    b.actuator("cart_m", "cx", gear=2.0, ctrlrange=(-1, 1))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(m["nq"], 0, 1.0), (m["nv"], 0, 0.1), (m["na"], 0, 0.1)], traces=[(OBJ_SITE, tip)])
    q = np.array([0.13, 0.4, -0.7]); v = np.array([0.2, -0.5, 1.0])
    defaults = dict(N=6, P=4, sigma=(0.3, 0.0), interp=2, horizon=80, state=np.concatenate([q, v, [0.05]]), mocap=np.zeros(0))
    return m, task, defaults


def linkage(timestep=0.004):
    """Test model for equality constraints: a gripper-like pair of fingers on one palm whose hinge angles are coupled by a joint
    equality (different branches: dense Hessian builds), a four-bar loop closed by a connect constraint between two chain ends, a
    tendon equality tying the wrist to the bars' mean angle, and a free ball hung from the world by a connect; a box on the floor for contacts next to them.  Residual = state (TASK_COPYSTATE)."""
    b = ModelBuilder(timestep=timestep, gravity=(0, 0, -9.81), contact=True)
    b.geom(0, "floor", PLANE, pos=(0, 0, -0.6), size=(2, 2, 0.1))
    palm = b.body("palm", 0, pos=(0, 0, 0))
    b.joint(palm, "wrist", HINGE, axis=(0, 1, 0), damping=0.05, armature=0.01)
    b.geom(palm, "palm_g", BOX, size=(0.06, 0.03, 0.02), mass=0.4)
    for name, x in (("fl", -0.05), ("fr", 0.05)):
        f = b.body(name, palm, pos=(x, 0, -0.02))
        b.joint(f, name + "_j", HINGE, axis=(0, 1, 0), damping=0.02, armature=0.002, limited=True, range=(-1.0, 1.0))
        b.geom(f, name + "_g", CAPSULE, size=(0.01, 0), fromto=(0, 0, 0, 0, 0, -0.12), mass=0.05)
    b.joint_equality("fl_j", "fr_j", polycoef=(0.0, -1.0, 0.1, 0, 0))            # mirrored closing with a slight quadratic term
    # four-bar: two chains from the world, ends tied together
    a1 = b.body("a1", 0, pos=(0.4, 0, 0)); b.joint(a1, "a1_j", HINGE, axis=(0, 1, 0), damping=0.01); b.geom(a1, "a1_g", CAPSULE, size=(0.012, 0), fromto=(0, 0, 0, 0, 0, -0.2), mass=0.1)
    a2 = b.body("a2", a1, pos=(0, 0, -0.2)); b.joint(a2, "a2_j", HINGE, axis=(0, 1, 0), damping=0.01); b.geom(a2, "a2_g", CAPSULE, size=(0.012, 0), fromto=(0, 0, 0, 0.15, 0, 0), mass=0.1)
    c1 = b.body("c1", 0, pos=(0.55, 0, 0)); b.joint(c1, "c1_j", HINGE, axis=(0, 1, 0), damping=0.01); b.geom(c1, "c1_g", CAPSULE, size=(0.012, 0), fromto=(0, 0, 0, 0, 0, -0.2), mass=0.1)
    b.connect(a2, c1, (0.15, 0, 0))
    b.tendon("t_w", ["wrist"], [1.0]); b.tendon("t_c", ["c1_j", "a1_j"], [0.5, 0.5])
    b.tendon_equality("t_w", "t_c", polycoef=(0.0, 0.5, 0, 0, 0), solref=(0.05, 1.0))        # the wrist follows the bars' mean angle (cross-branch)
    ball = b.body("ball", 0, pos=(-0.4, 0, -0.3)); b.joint(ball, "ball_f", FREE); b.geom(ball, "ball_g", SPHERE, size=(0.04,), mass=0.2)
    b.connect(ball, 0, (0, 0, 0.3), solref=(0.01, 1.0))
    box = b.body("box", 0, pos=(0.1, 0, -0.55)); b.joint(box, "box_f", FREE); b.geom(box, "box_g", BOX, size=(0.05, 0.05, 0.05), mass=0.3)
    tip = b.site(a2, "tip", pos=(0.15, 0, 0))
    b.actuator("wrist_m", "wrist", gear=0.5, ctrlrange=(-1, 1))
    b.actuator("fl_m", "fl_j", gear=0.2, ctrlrange=(-1, 1))
    b.actuator("bar_m", "a1_j", gear=0.5, ctrlrange=(-1, 1))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(m["nq"], 0, 1.0), (m["nv"], 0, 0.1)], traces=[(OBJ_SITE, tip)])
    q = m["qpos0"].copy(); v = np.zeros(m["nv"])
    v[0] = 1.0; v[1] = 2.0; v[2] = -2.0; v[3] = 1.5; v[6:9] = [0.8, 0.3, 0.0]
    defaults = dict(N=6, P=4, sigma=(0.5, 0.0), interp=2, horizon=80, state=np.concatenate([q, v]), mocap=np.zeros(0))
    return m, task, defaults


def welded(timestep=0.004):
    """Test model for weld equalities: a two-link arm (hinge + ball) whose hand is welded to a free tool with an off-centre anchor and
    torquescale 0.7 (both sides move; cross-branch rows), a free lamp welded to the world with an explicit relpose it starts away
    from (it is pulled in), and a free puck welded to a mocap body (the MJPC way of dragging an object); a box on the floor for
    contacts next to them.  Residual = state (TASK_COPYSTATE)."""
    b = ModelBuilder(timestep=timestep, gravity=(0, 0, -9.81), contact=True)
    b.geom(0, "floor", PLANE, pos=(0, 0, -0.6), size=(2, 2, 0.1))
    l1 = b.body("l1", 0, pos=(0, 0, 0.2)); b.joint(l1, "sh", HINGE, axis=(0, 1, 0), damping=0.05, armature=0.01)
    b.geom(l1, "l1_g", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0.25, 0, 0), mass=0.4)
    l2 = b.body("l2", l1, pos=(0.25, 0, 0)); b.joint(l2, "el", BALL, damping=0.03)
    b.geom(l2, "l2_g", CAPSULE, size=(0.018, 0), fromto=(0, 0, 0, 0.2, 0, 0), mass=0.25)
    tool = b.body("tool", 0, pos=(0.5, 0.01, 0.21), quat=(0.98, 0.05, -0.1, 0.15)); b.joint(tool, "tool_f", FREE)
    b.geom(tool, "tool_g", BOX, size=(0.04, 0.02, 0.015), mass=0.15)
    b.weld(l2, tool, anchor=(-0.03, 0.0, 0.01), torquescale=0.7, solref=(0.01, 1.0))
    lamp = b.body("lamp", 0, pos=(-0.4, 0.3, 0.1)); b.joint(lamp, "lamp_f", FREE); b.geom(lamp, "lamp_g", SPHERE, size=(0.04,), pos=(0.05, 0, 0), mass=0.2)
    b.weld(lamp, 0, relpose=(0.38, -0.3, -0.15, 0.96, 0.0, 0.28, 0.0), solref=(0.02, 1.0))
    hold = b.body("hold", 0, pos=(-0.3, -0.4, -0.2), mocap=True)
    puck = b.body("puck", 0, pos=(-0.3, -0.4, -0.2)); b.joint(puck, "puck_f", FREE); b.geom(puck, "puck_g", CYLINDER, size=(0.05, 0.02), mass=0.3)
    b.weld(puck, hold, solref=(0.02, 1.0), solimp=(0.9, 0.95, 0.01, 0.5, 2))
    box = b.body("box", 0, pos=(0.1, 0, -0.55)); b.joint(box, "box_f", FREE); b.geom(box, "box_g", BOX, size=(0.05, 0.05, 0.05), mass=0.3)
    tip = b.site(tool, "tip", pos=(0.04, 0, 0))
    b.actuator("sh_m", "sh", gear=1.5, ctrlrange=(-1, 1))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(m["nq"], 0, 1.0), (m["nv"], 0, 0.1)], traces=[(OBJ_SITE, tip)])
    q = m["qpos0"].copy(); v = np.zeros(m["nv"])
    v[0] = 1.5; v[1:4] = [0.5, -1.0, 0.8]; v[4:7] = [0.2, 0.1, -0.3]; v[13:16] = [0.0, 0.0, 2.0]
    mocap = np.array([-0.25, -0.35, -0.1, 0.92, 0.0, 0.0, 0.39])          # the hold has moved and turned: the puck follows
    defaults = dict(N=6, P=4, sigma=(0.5, 0.0), interp=2, horizon=80, state=np.concatenate([q, v]), mocap=mocap)
    return m, task, defaults


def servo_arm(timestep=0.005, integrator=3):
    """Test model for mjINT_IMPLICITFAST: a three-link arm on position servos with velocity gains (kv: the bias' velocity term), one
    of them with a force range it saturates, a velocity servo through a fixed tendon along the chain, a damped tendon and gravity
    compensation (full / half / none per link); stiff
    enough that Euler (integrator=0) and implicitfast (3) give visibly different trajectories.  Residual = state (TASK_COPYSTATE)."""
    b = ModelBuilder(timestep=timestep, gravity=(0, 0, -9.81), contact=False, integrator=integrator)
    parent = 0
    for k, (axis, ln) in enumerate((((0, 1, 0), 0.3), ((0, 1, 0), 0.25), ((1, 0, 0), 0.2))):
        body = b.body(f"l{k}", parent, pos=(0, 0, 0) if k == 0 else (0, 0, -(0.3, 0.25)[k - 1]), gravcomp=(1.0, 0.5, 0.0)[k])      # menagerie-arm style gravity compensation
        b.joint(body, f"j{k}", HINGE, axis=axis, damping=0.02, armature=0.002, actuatorfrcrange=(-1.2, 1.2) if k == 0 else None)      # joint-level clamp
        b.geom(body, f"g{k}", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0, 0, -ln), mass=0.3 - 0.08 * k)
        parent = body
    tip = b.site(parent, "tip", pos=(0, 0, -0.2))
    b.tendon("chain", ["j0", "j1"], [1.0, 0.5], damping=0.3)
    b.tendon("drive", ["j1", "j2"], [1.0, -1.0])
    b.actuator("s0", "j0", gainprm=(20.0, 0, 0), biastype=1, biasprm=(0, -20.0, -1.5), ctrlrange=(-1.5, 1.5))
    b.actuator("s1", "j1", gainprm=(15.0, 0, 0), biastype=1, biasprm=(0, -15.0, -1.0), ctrlrange=(-1.5, 1.5), forcelimited=True, forcerange=(-1.0, 1.0), gear=1.5)
    b.actuator("v2", tendon="drive", gainprm=(0.8, 0, 0), biastype=1, biasprm=(0, 0, -0.8), ctrlrange=(-2, 2))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(3, 0, 1.0), (3, 0, 0.1)], traces=[(OBJ_SITE, tip)])
    q = np.array([0.5, -0.4, 0.3]); v = np.array([2.0, -3.0, 1.0])
    defaults = dict(N=6, P=4, sigma=(0.5, 0.0), interp=2, horizon=80, state=np.concatenate([q, v]), mocap=np.zeros(0))
    return m, task, defaults


def filter_arm(timestep=0.005):
    """Test model for activation states (na > 0): a three-link arm whose joints are driven through a first-order filter (the
    swimmer's dyntype="filter"), an exact filter with a position servo's affine bias and a clamped integrator; a fourth, plain
    motor shares the model.  The residual copies the whole state [qpos, qvel, act] (TASK_COPYSTATE)."""
    b = ModelBuilder(timestep=timestep, gravity=(0, 0, -9.81), contact=False)
    parent = 0
    for k, (axis, ln) in enumerate((((0, 1, 0), 0.3), ((0, 1, 0), 0.25), ((1, 0, 0), 0.2), ((0, 1, 0), 0.15))):
        body = b.body(f"l{k}", parent, pos=(0, 0, 0) if k == 0 else (0, 0, -(0.3, 0.25, 0.2)[k - 1]))
        b.joint(body, f"j{k}", HINGE, axis=axis, damping=0.05, armature=0.005, limited=True, range=(-2.0, 2.0))
        b.geom(body, f"g{k}", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0, 0, -ln), mass=0.4 - 0.08 * k)
        parent = body
    tip = b.site(parent, "tip", pos=(0, 0, -0.15))
    b.actuator("a_filter", "j0", gainprm=(3.0, 0, 0), ctrlrange=(-1, 1), dyntype=2, dynprm=0.06)
    b.actuator("a_exact", "j1", gainprm=(4.0, 0, 0), biastype=1, biasprm=(0, -4.0, -0.1), ctrlrange=(-1.5, 1.5), dyntype=3, dynprm=0.03)
    b.actuator("a_motor", "j2", gear=0.8, ctrlrange=(-1, 1))
    b.actuator("a_integrator", "j3", gainprm=(1.5, 0, 0), ctrlrange=(-1, 1), dyntype=1, actlimited=True, actrange=(-0.06, 0.06))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(4, 0, 1.0), (4, 0, 0.1), (3, 0, 0.5)], traces=[(OBJ_SITE, tip)])
    q = np.array([0.4, -0.3, 0.2, 0.1]); v = np.array([0.5, -1.0, 0.3, 0.0]); act = np.array([0.2, -0.4, 0.05])
    defaults = dict(N=6, P=4, sigma=(0.4, 0.0), interp=2, horizon=80, state=np.concatenate([q, v, act]), mocap=np.zeros(0))
    return m, task, defaults


# ----------------------------------------------------------------------------------- cartpole
def cartpole(timestep=0.01):
    b = ModelBuilder(timestep=timestep, contact=False)
    b.geom(0, "floor", PLANE, pos=(0, 0, -0.05), size=(4, 4, 0.2))
    cart = b.body("cart", 0, pos=(0, 0, 1))
    b.joint(cart, "slider", SLIDE, axis=(1, 0, 0), limited=True, range=(-1.8, 1.8), solreflimit=(0.08, 1.0), damping=1e-4)
    b.geom(cart, "cart", BOX, size=(0.2, 0.15, 0.1), mass=1.0)
    pole = b.body("pole_1", cart)
    b.joint(pole, "hinge_1", HINGE, axis=(0, 1, 0), damping=1e-4)
    b.geom(pole, "pole_1", CAPSULE, size=(0.045, 0), fromto=(0, 0, 0, 0, 0, 1), mass=0.1)
    tip = b.site(pole, "tip", pos=(0, 0, 1))
    b.actuator("slide", "slider", gear=10.0, ctrlrange=(-1, 1))
    b.key("home", [1.0, 0.0])
    m = b.compile()
    task = make_task(TASK_CARTPOLE,
                     [(1, 6, 10.0, [0.01]), (1, 6, 10.0, [0.1]), (1, 0, 0.1), (1, 0, 0.1)],
                     parameters=[0.0], traces=[(OBJ_SITE, tip)])
    defaults = dict(N=10, P=10, sigma=(0.5, 0.0), interp=2, horizon=101, state=np.array([1.0, 0, 0, 0]),
                    mocap=np.zeros(0))
    return m, task, defaults


# ----------------------------------------------------------------------------------- quadruped (Unitree A1, flat)
_A1_LEGS = [
    # body prefix, joint prefix, foot geom name, hip pos, side (+1 left / -1 right), front
    ("FR", "FR", "FR", (0.183, -0.047, 0), -1),
    ("FL", "FL", "FL", (0.183, 0.047, 0), +1),
    ("HR", "RR", "HR", (-0.183, -0.047, 0), -1),
    ("HL", "RL", "HL", (-0.183, 0.047, 0), +1),
]
_HIP_INERTIAL = {
    "FR": dict(mass=0.696, pos=(-0.003311, -0.000635, 3.1e-05), quat=(0.507528, 0.506268, 0.491507, 0.494499), diaginertia=(0.000807752, 0.00055293, 0.000468983)),
    "FL": dict(mass=0.696, pos=(-0.003311, 0.000635, 3.1e-05), quat=(0.494499, 0.491507, 0.506268, 0.507528), diaginertia=(0.000807752, 0.00055293, 0.000468983)),
    "HR": dict(mass=0.696, pos=(0.003311, -0.000635, 3.1e-05), quat=(0.491507, 0.494499, 0.507528, 0.506268), diaginertia=(0.000807752, 0.00055293, 0.000468983)),
    "HL": dict(mass=0.696, pos=(0.003311, 0.000635, 3.1e-05), quat=(0.506268, 0.507528, 0.494499, 0.491507), diaginertia=(0.000807752, 0.00055293, 0.000468983)),
}


def _thigh_inertial(side):
    s = -side   # right legs: +y com / mirrored quat signs
    return dict(mass=1.013, pos=(-0.003237, 0.022327 * s, -0.027326),
                quat=(0.999125, -0.00256393 * s, -0.0409531, -0.00806091 * s),
                diaginertia=(0.00555739, 0.00513936, 0.00133944))


_CALF_INERTIAL = dict(mass=0.226, pos=(0.00472659, 0, -0.131975), quat=(0.706886, 0.017653, 0.017653, 0.706886),
                      diaginertia=(0.00340344, 0.00339393, 3.54834e-05))


def _a1_robot(b, pos=(0, 0, 0.5)):
    """The A1 of a1.xml.patch on builder b: returns (trunk body, head site, {foot name: geom id})."""
    # A1 (a1.xml.patch); class a1: friction 0.6 margin 0.001 condim 1; class collision: capsule, group 3
    col = dict(friction=(0.6, 0.005, 0.0001), margin=0.001, condim=1, group=3)
    jdef = dict(damping=2.0, armature=0.01, frictionloss=0.2, limited=True)
    trunk = b.body("trunk", 0, pos=pos,
                   inertial=dict(mass=4.713, pos=(0, 0.0041, -0.0005),
                                 fullinertia=(0.0158533, 0.0377999, 0.0456542, -3.66e-05, -6.11e-05, -2.75e-05)))
    b.site(trunk, "torso")
    head = b.site(trunk, "head", pos=(0.3, 0, 0))
    b.joint(trunk, "root", FREE)
    b.geom(trunk, "", BOX, size=(0.125, 0.04, 0.057), **col)
    b.geom(trunk, "", CYLINDER, quat=(1, 0, 1, 0), pos=(0, -0.04, 0), size=(0.058, 0.125), **col)
    b.geom(trunk, "", CYLINDER, quat=(1, 0, 1, 0), pos=(0, 0.04, 0), size=(0.058, 0.125), **col)
    b.geom(trunk, "", BOX, pos=(0.25, 0, 0), size=(0.005, 0.06, 0.05), **col)
    b.geom(trunk, "", CAPSULE, pos=(0.25, 0.06, -0.01), size=(0.009, 0.035), **col)
    b.geom(trunk, "", CAPSULE, pos=(0.25, -0.06, -0.01), size=(0.009, 0.035), **col)
    b.geom(trunk, "", CAPSULE, pos=(0.25, 0, -0.05), size=(0.005, 0.06), quat=(1, 1, 0, 0), **col)
    b.geom(trunk, "", CAPSULE, pos=(0.255, 0, 0.0355), size=(0.021, 0.052), quat=(1, 1, 0, 0), **col)
    foot_geom = {}
    for pref, jpref, foot, hip_pos, side in _A1_LEGS:
        hip = b.body(f"{pref}_hip", trunk, pos=hip_pos, inertial=_HIP_INERTIAL[pref])
        b.joint(hip, f"{jpref}_hip_joint", HINGE, axis=(1, 0, 0), range=(-0.802851, 0.802851), **{**jdef, "damping": 1.0})
        b.geom(hip, "", CYLINDER, size=(0.04, 0.04), quat=(1, 1, 0, 0), pos=(0, 0.055 * side, 0), **col)
        if pref == "FL":   # extra collision cylinder present on FL_hip only (a1.xml.patch context)
            b.geom(hip, "", CYLINDER, size=(0.04, 0.04), quat=(1, 1, 0, 0), pos=(0, 0.055, 0), **col)
        thigh = b.body(f"{jpref}_thigh", hip, pos=(0, 0.08505 * side, 0), inertial=_thigh_inertial(side))
        b.joint(thigh, f"{jpref}_thigh_joint", HINGE, axis=(0, 1, 0), range=(-1.9472, 3.28879), ref=-0.9, **jdef)
        b.geom(thigh, "", CAPSULE, size=(0.015, 0), fromto=(-0.02, 0, 0, -0.02, 0, -0.16), **col)
        b.geom(thigh, "", CAPSULE, size=(0.015, 0), fromto=(0, 0, 0, -0.02, 0, -0.1), **col)
        b.geom(thigh, "", CAPSULE, size=(0.015, 0), fromto=(-0.02, 0, -0.16, 0, 0, -0.2), **col)
        calf = b.body(f"{jpref}_calf", thigh, pos=(0, 0, -0.2), inertial=_CALF_INERTIAL)
        b.joint(calf, f"{jpref}_calf_joint", HINGE, axis=(0, 1, 0), range=(-0.89653, 0.883702), ref=1.8, **jdef)
        b.geom(calf, "", CAPSULE, size=(0.01, 0), fromto=(0, 0, 0, 0.02, 0, -0.13), **col)
        b.geom(calf, "", CAPSULE, size=(0.01, 0), fromto=(0.02, 0, -0.13, 0, 0, -0.2), **col)
        foot_geom[foot] = b.geom(calf, foot, SPHERE, size=(0.02,), pos=(0, 0, -0.2), priority=1,
                                 solimp=(0.015, 1, 0.031, 0.5, 2), condim=6, friction=(0.8, 0.02, 0.01),
                                 margin=0.001, group=3)
        b.site(calf, jpref, pos=(0, 0, -0.2))
    return trunk, head, foot_geom


def quadruped(timestep=0.01, transitioned=True):
    b = ModelBuilder(timestep=timestep, cone=1, impratio=10.0, contact=True)
    b.nconmax = 32
    b.nefcmax = 128
    # world geoms (task_flat.xml:52-61)
    b.geom(0, "floor", PLANE, pos=(0, 0, -0.01), size=(0, 0, 0.1))
    b.geom(0, "ramp", BOX, pos=(3.13, 2.5, -0.18), size=(1.6, 1, 0.5), euler=(0, -0.2, 0))
    b.geom(0, "hill", SPHERE, pos=(6, 6, -5.5), size=(6,))
    goal = b.body("goal", 0, pos=(0.3, 0, 0.26), mocap=True)
    b.geom(goal, "goal", SPHERE, size=(0.12,), contype=0, conaffinity=0, group=2)
    box = b.body("box", 0, pos=(-2.5, 0, 0), mocap=True)
    b.geom(box, "box", BOX, size=(1, 1, 0.3))
    trunk, head, foot_geom = _a1_robot(b)
    for jpref in ("FR", "FL", "RR", "RL"):
        for part in ("hip", "thigh", "calf"):
            b.actuator(f"{jpref}_{part}", f"{jpref}_{part}_joint", gainprm=(40, 0, 0), ctrlrange=(-1, 1))
    home = [0, 0, 0.26, 1, 0, 0, 0,
            -0.000341931, 0.0181576, -0.0268335, 0.00160968, 0.0247957, -0.0270045,
            0.00191398, -0.033048, -0.0675298, -0.00199489, -0.0374747, -0.0681862]
    crouch = [-0.0501827, 0.00107117, 0.143925, 1, 0, 0, 0, 0, 0, -0.5, 0, 0, -0.5, 0, 0, -0.5, 0, 0, -0.5]
    b.key("home", home)
    b.key("crouch", crouch)
    m = b.compile()
    # residual parameters, numerics order of task_flat.xml:17-32
    params = [select_value(0), select_value(1), 2.0, 0.06, 0.0, 0.0, 0.0, select_value(0), select_value(0), 0.0]
    P_GAIT, P_SWITCH, P_CAD, P_AMP, P_DUTY, P_WSPEED, P_WTURN, P_FLIP, P_BIPED, P_HEADING = range(10)
    ints = [trunk, head, m["body_mocapid"][goal], foot_geom["FL"], foot_geom["HL"], foot_geom["FR"], foot_geom["HR"],
            P_GAIT, P_SWITCH, P_FLIP, P_BIPED, P_CAD, P_AMP, P_DUTY, P_HEADING, 0, 1, 0]
    # flip kinematics (quadruped.cc:552-597)
    g = 9.81
    kMaxHeight, kLeapHeight, kCrouchHeight, kHeightQuadruped = 0.8, 0.5, 0.15, 0.25
    jump_vel = math.sqrt(2 * g * (kMaxHeight - kLeapHeight))
    flight_time = 2 * jump_vel / g
    jump_acc = jump_vel * jump_vel / (2 * (kLeapHeight - kCrouchHeight))
    crouch_time = math.sqrt(2 * (kHeightQuadruped - kCrouchHeight) / jump_acc)
    leap_time = jump_vel / jump_acc
    jump_time = crouch_time + leap_time
    crouch_vel = -jump_acc * crouch_time
    land_time = 2 * (kLeapHeight - kHeightQuadruped) / jump_vel
    land_acc = jump_vel / land_time
    flight_rot_vel = 1.25 * math.pi / flight_time
    jump_rot_vel = math.pi / leap_time - flight_rot_vel
    jump_rot_acc = (flight_rot_vel - jump_rot_vel) / leap_time
    land_rot_acc = 2 * (flight_rot_vel * land_time - math.pi / 4) / (land_time * land_time)
    phase_velocity = 2 * math.pi * params[P_CAD] if transitioned else 0.0
    dbl = [0.0, 0, 0, 0, 0, 0, 0.0, 0.0, 0.0, 0, 0, 0, 0, 0.0, 0.0, 0.0, phase_velocity,
           g, jump_vel, flight_time, jump_acc, crouch_time, leap_time, jump_time, crouch_vel, land_time, land_acc,
           flight_rot_vel, jump_rot_vel, jump_rot_acc, land_rot_acc]
    terms = [(3, 6, 1.0, [0.05]), (1, 6, 1.0, [0.04]), (3, 2, 0.2, [0.1]), (4, 6, 2.0, [0.03]), (2, 2, 0.2, [0.1]),
             (12, 0, 0.03), (12, 0, 0.02), (2, 0, 0.0), (3, 0, 0.0)]
    task = make_task(TASK_QUADRUPED, terms, parameters=params, traces=[(OBJ_SITE, head)], int_data=ints, dbl_data=dbl)
    state = np.concatenate([np.array(home, float), np.zeros(m["nv"])])
    mocap = np.array([0.3, 0, 0.26, 1, 0, 0, 0, -2.5, 0, 0, 1, 0, 0, 0.0])
    defaults = dict(N=60, P=3, sigma=(0.04, 0.0), interp=2, horizon=36, state=state, mocap=mocap)
    return m, task, defaults


# ----------------------------------------------------------------------------------- humanoid tracking
_TRACK_NAMES = ["pelvis", "head", "ltoe", "rtoe", "lheel", "rheel", "lknee", "rknee", "lhand", "rhand", "lelbow", "relbow",
                "lshoulder", "rshoulder", "lhip", "rhip"]          # tracking.cc:59-63


def _humanoid_model(timestep, mocap_bodies):
    """mjpc/tasks/humanoid/humanoid.xml.patch (whole-file hunk): the modified dm_control humanoid shared by the tracking, stand
    and walk tasks.  MJCF defaults apply: angles in degrees, pyramidal cones, body geoms condim 1 vs floor condim 3 -> condim 3.
    Returns (builder, tracking-site ids, torso body id)."""
    D = math.pi / 180.0
    b = ModelBuilder(timestep=timestep, cone=0, impratio=1.0, contact=True)
    b.nconmax = 24
    b.nefcmax = 96
    b.geom(0, "floor", PLANE, size=(50, 50, 0.05))
    if mocap_bodies:
        for n in _TRACK_NAMES:                      # tracking/task.xml:30-77 (mocap bodies carry sites only)
            mb = b.body(f"mocap[{n}]", 0, mocap=True)
            b.site(mb, f"mocap[{n}]")
    G = dict(condim=1, friction=(0.7, 0.005, 0.0001), solimp=(0.9, 0.99, 0.003, 0.5, 2), solref=(0.015, 1))
    J = dict(damping=0.2, stiffness=1.0, armature=0.01, limited=True, solimplimit=(0, 0.99, 0.01, 0.5, 2))
    big = {**J, "damping": 5.0, "stiffness": 10.0}
    stiff = {**big, "stiffness": 20.0}

    def rng(lo, hi):
        return (lo * D, hi * D)
    torso = b.body("torso", 0, pos=(0, 0, 1.282))
    b.joint(torso, "root", FREE)
    b.geom(torso, "torso", CAPSULE, size=(0.07, 0), fromto=(0, -0.07, 0, 0, 0.07, 0), **G)
    b.geom(torso, "waist_upper", CAPSULE, size=(0.06, 0), fromto=(-0.01, -0.06, -0.12, -0.01, 0.06, -0.12), **G)
    head = b.body("head", torso, pos=(0, 0, 0.19))
    b.geom(head, "head", SPHERE, size=(0.09,), **G)
    sites = {"head": b.site(head, "tracking[head]", pos=(0.09, 0, 0))}
    wl = b.body("waist_lower", torso, pos=(-0.01, 0, -0.26))
    b.geom(wl, "waist_lower", CAPSULE, size=(0.06, 0), fromto=(0, -0.06, 0, 0, 0.06, 0), **G)
    b.joint(wl, "abdomen_z", HINGE, pos=(0, 0, 0.065), axis=(0, 0, 1), range=rng(-45, 45), **stiff)
    b.joint(wl, "abdomen_y", HINGE, pos=(0, 0, 0.065), axis=(0, 1, 0), range=rng(-75, 30), **big)
    pelvis = b.body("pelvis", wl, pos=(0, 0, -0.165))
    sites["pelvis"] = b.site(pelvis, "tracking[pelvis]", pos=(0, 0, 0.075))
    b.joint(pelvis, "abdomen_x", HINGE, pos=(0, 0, 0.1), axis=(1, 0, 0), range=rng(-35, 35), **big)
    b.geom(pelvis, "butt", CAPSULE, size=(0.09, 0), fromto=(-0.02, -0.07, 0, -0.02, 0.07, 0), **G)
    for side, sg in (("right", -1.0), ("left", 1.0)):
        s = side[0]
        th = b.body(f"thigh_{side}", pelvis, pos=(0, 0.1 * sg, -0.04))
        sites[f"{s}hip"] = b.site(th, f"tracking[{s}hip]", pos=(0, -0.025 * sg, 0.025))
        b.joint(th, f"hip_x_{side}", HINGE, axis=(-sg, 0, 0), range=rng(-30, 10), **big)
        b.joint(th, f"hip_z_{side}", HINGE, axis=(0, 0, -sg), range=rng(-60, 35), **big)
        b.joint(th, f"hip_y_{side}", HINGE, axis=(0, 1, 0), range=rng(-150, 20), **big)
        b.geom(th, f"thigh_{side}", CAPSULE, size=(0.06, 0), fromto=(0, 0, 0, 0, -0.01 * sg, -0.34), **G)
        sh = b.body(f"shin_{side}", th, pos=(0, -0.01 * sg, -0.4))
        b.joint(sh, f"knee_{side}", HINGE, pos=(0, 0, 0.02), axis=(0, -1, 0), range=rng(-160, 2), **J)
        sites[f"{s}knee"] = b.site(sh, f"tracking[{s}knee]", pos=(0, 0, 0.05))
        b.geom(sh, f"shin_{side}", CAPSULE, size=(0.049, 0), fromto=(0, 0, 0, 0, 0, -0.3), **G)
        ft = b.body(f"foot_{side}", sh, pos=(0, 0, -0.39))
        b.joint(ft, f"ankle_y_{side}", HINGE, pos=(0, 0, 0.08), axis=(0, 1, 0), range=rng(-50, 50), **{**J, "stiffness": 6.0})
        b.joint(ft, f"ankle_x_{side}", HINGE, pos=(0, 0, 0.04), axis=(-sg, 0, -0.5 * sg), range=rng(-50, 50), **{**J, "stiffness": 3.0})
        b.geom(ft, f"foot1_{side}", CAPSULE, size=(0.027, 0), fromto=(-0.07, -0.01, 0, 0.14, -0.03, 0), **G)
        b.geom(ft, f"foot2_{side}", CAPSULE, size=(0.027, 0), fromto=(-0.07, 0.01, 0, 0.14, 0.03, 0), **G)
        b.site(ft, f"foot_{side}", pos=(0.05, -0.03 * sg, 0))                          # humanoid.xml.patch:169-171, 216-218
        b.site(ft, "sp2" if side == "right" else "sp0", pos=(-0.07, 0, 0))
        b.site(ft, "sp3" if side == "right" else "sp1", pos=(0.14, 0, 0))
        heel = b.body(f"heel_{side}", ft, pos=(-0.05, 0, 0.04))
        sites[f"{s}heel"] = b.site(heel, f"tracking[{s}heel]")
        toe = b.body(f"toe_{side}", ft, pos=(0.07, 0, -0.01))
        sites[f"{s}toe"] = b.site(toe, f"tracking[{s}toe]")
    for side, sg in (("right", -1.0), ("left", 1.0)):
        s = side[0]
        ua = b.body(f"upper_arm_{side}", torso, pos=(0, 0.17 * sg, 0.06))
        sites[f"{s}shoulder"] = b.site(ua, f"tracking[{s}shoulder]")
        b.joint(ua, f"shoulder1_{side}", HINGE, axis=(-2 * sg, 1, -sg), range=rng(-85, 60), **J)
        b.joint(ua, f"shoulder2_{side}", HINGE, axis=(0, -1, -sg), range=rng(-85, 60), **J)
        b.geom(ua, f"upper_arm_{side}", CAPSULE, size=(0.04, 0), fromto=(0, 0, 0, 0.16, 0.16 * sg, -0.16), **G)
        la = b.body(f"lower_arm_{side}", ua, pos=(0.18, 0.18 * sg, -0.18))
        b.joint(la, f"elbow_{side}", HINGE, axis=(0, -1, -sg), range=rng(-100, 50), **{**J, "stiffness": 0.0})
        sites[f"{s}elbow"] = b.site(la, f"tracking[{s}elbow]")
        sites[f"{s}hand"] = b.site(la, f"tracking[{s}hand]", pos=(0.13, -0.13 * sg, 0.13))
        b.geom(la, f"lower_arm_{side}", CAPSULE, size=(0.031, 0), fromto=(0.01, -0.01 * sg, 0.01, 0.17, -0.17 * sg, 0.17), **G)
        hd = b.body(f"hand_{side}", la, pos=(0.18, -0.18 * sg, 0.18))
        b.geom(hd, f"hand_{side}", SPHERE, size=(0.04,), **G)
    b.exclude(b.body_id("waist_lower"), b.body_id("thigh_right"))
    b.exclude(b.body_id("waist_lower"), b.body_id("thigh_left"))
    b.tendon("hamstring_right", ["hip_y_right", "knee_right"], [0.5, -0.5], limited=True, range=(-0.3, 2))
    b.tendon("hamstring_left", ["hip_y_left", "knee_left"], [0.5, -0.5], limited=True, range=(-0.3, 2))
    for name, gear in [("abdomen_y", 40), ("abdomen_z", 40), ("abdomen_x", 40), ("hip_x_right", 40), ("hip_z_right", 40),
                       ("hip_y_right", 120), ("knee_right", 100), ("ankle_x_right", 20), ("ankle_y_right", 20),
                       ("hip_x_left", 40), ("hip_z_left", 40), ("hip_y_left", 120), ("knee_left", 100), ("ankle_x_left", 20),
                       ("ankle_y_left", 20), ("shoulder1_right", 20), ("shoulder2_right", 20), ("elbow_right", 40),
                       ("shoulder1_left", 20), ("shoulder2_left", 20), ("elbow_left", 40)]:
        b.actuator(name, name, gear=float(gear), ctrlrange=(-1, 1))
    return b, sites, torso


HUMANOID_MOTIONS = ("Jump", "Kick Spin", "Spin Kick", "Cartwheel (1)", "Crouch Flip", "Cartwheel (2)", "Monkey Flip", "Dance", "Run", "Walk")


def humanoid_track(timestep=0.005, motion=0):
    """humanoid model + tracking/task.xml; `motion` = the task's mode (tracking.cc:43-66): 0 "Jump" (121 keys) ... 9 "Walk" (510
    keys).  The model carries the key frames of all ten motions (data/humanoid_motion_keys.npz, made from the reference's key-frame
    data files by data/make_humanoid_keys.py); the frozen residual state is [mode, first key of the motion, its length, ...] and the
    default state is the motion's first key, the state Transition resets to on a motion switch (tracking.cc:231-238)."""
    import os
    b, sites, torso = _humanoid_model(timestep, True)
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "humanoid_motion_keys.npz"))
    mpos, lengths = data["mpos"], [int(x) for x in data["lengths"]]
    first = int(sum(lengths[:motion]))
    for k in range(mpos.shape[0]):
        b.key(f"key_{k + 1}", data["qpos0"][0] if k == 0 else [])
    b.key_mpos = mpos
    m = b.compile()
    ids_site = [sites[n] for n in _TRACK_NAMES]
    ids_mocap = [int(m["body_mocapid"][m["names"]["body"][f"mocap[{n}]"]]) for n in _TRACK_NAMES]
    ints = [motion, first, lengths[motion]] + ids_site + ids_mocap
    terms = [(21, 0, 0.001), (21, 3, 0.1, [0.3]), (3, 6, 100.0, [0.1]), (3, 6, 30.0, [0.1]), (3, 6, 0.0, [0.1]),
             (6, 7, 30.0, [0.2, 4]), (6, 7, 30.0, [0.2, 4]), (6, 6, 30.0, [0.1]), (6, 6, 30.0, [0.1]), (6, 7, 30.0, [0.2, 4]),
             (6, 6, 30.0, [0.1]), (6, 6, 30.0, [0.1]),
             (3, 6, 0.1, [0.3]), (3, 6, 0.0, [0.3])] + [(6, 6, 0.1, [0.3])] * 7
    task = make_task(TASK_HUMANOID_TRACK, terms, traces=[(OBJ_XBODY, torso)], int_data=ints, dbl_data=[0.0])
    state = np.concatenate([data["qpos0"][motion], data["qvel0"][motion]])
    mocap = np.concatenate([np.concatenate([mpos[first, 3 * i:3 * i + 3], [1, 0, 0, 0]]) for i in range(16)])
    defaults = dict(N=32, P=16, sigma=(0.15, 0.0), interp=2, horizon=101, state=state, mocap=mocap)
    return m, task, defaults


def humanoid_stand(timestep=0.015):
    """humanoid model + stand/task.xml:8-37 (agent_timestep 0.015, horizon 0.35 s, 3 spline points, sigma 0.05)."""
    b, sites, torso = _humanoid_model(timestep, False)
    m = b.compile()
    sid = m["names"]["site"]; bid = m["names"]["body"]
    ints = [sid["sp0"], sid["sp1"], sid["sp2"], sid["sp3"], bid["head"], bid["torso"]]
    terms = [(1, 6, 100.0, [0.1]), (1, 6, 50.0, [0.1]), (2, 0, 10.0), (21, 0, 0.01), (21, 3, 0.025, [0.3])]
    task = make_task(TASK_HUMANOID_STAND, terms, parameters=[1.4], traces=[(OBJ_BODY, torso)], int_data=ints)
    state = np.concatenate([m["qpos0"], np.zeros(m["nv"])])
    return m, task, dict(N=10, P=3, sigma=(0.05, 0.0), interp=0, horizon=24, state=state, mocap=np.zeros(0))


def humanoid_walk(timestep=0.015):
    """humanoid model + walk/task.xml:8-35 (Torso height goal 1.35, Speed 0.5).  The cost table follows the XML; the residual
    writes its 3 velocity numbers as (walk-forward, move-feet x, move-feet y), i.e. shifted by one against the term names."""
    b, sites, torso = _humanoid_model(timestep, False)
    m = b.compile()
    bid = m["names"]["body"]
    ints = [bid["torso"], bid["pelvis"], bid["foot_right"], bid["foot_left"], bid["waist_lower"]]
    terms = [(1, 7, 5.0, [0.1, 4.0]), (1, 8, 1.0, [0.05]), (2, 1, 5.0, [0.02, 4.0]), (8, 2, 5.0, [0.01]), (21, 0, 0.025),
             (2, 7, 0.625, [0.2, 4.0]), (1, 7, 1.0, [0.5, 3.0]), (21, 3, 0.1, [0.3])]
    task = make_task(TASK_HUMANOID_WALK, terms, parameters=[1.35, 0.5], traces=[(OBJ_BODY, torso)], int_data=ints)
    state = np.concatenate([m["qpos0"], np.zeros(m["nv"])])
    return m, task, dict(N=10, P=3, sigma=(0.05, 0.0), interp=0, horizon=24, state=state, mocap=np.zeros(0))


TASK_HUMANOID_INTERACT = 15
INTERACT_MODES = ("Sit Down", "Stand Up", "Relax", "Stay Still")
# default_weights of interact.h:40-45 (one row per mode; Transition copies the row on a mode change, interact.cc:191-197)
INTERACT_WEIGHTS = [[10, 10, 5, 5, 0, 20, 30, 0, 0, 0, 0.01, .1, 80.], [10, 0, 1, 1, 80, 0, 0, 100, 0, 0, 0.01, 0.025, 0.],
                    [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0.01, .8, 80.], [0, 0, 0, 0, 0, 0, 0, 0, 0, 50, 20, .025, 80.]]
_INTERACT_HOME = [-0.3729, -0.0358, 0.9018, 0.99555, 0.05669, -0.05346, 0.05304, 0.070186, 0.28932, 0.04422, -0.07611, -0.40237, -1.80319, -1.75152,
                  -0.02876, -0.02198, 0.179219, 0.093576, -1.67581, -1.49138, 0.01844, -0.06467, 0.677519, -0.82367, -0.94849, 0.859786, -0.88692, -1.14922]


def humanoid_interact(timestep=0.015, contact_pairs=(), facing_target=None):
    """mjpc/tasks/humanoid/interact (interact.cc:31-186, task.xml, scenes/armchair.xml): the modified dm_control humanoid next to an
    armchair (a static body of five boxes, scenes/armchair.xml:3-9), 13 cost terms over 68 residuals: up-vectors of torso / pelvis /
    feet, head and torso height goals, knee-feet and com-feet distances in the plane, facing direction, com velocity, joint
    velocities, controls, and the distances of up to five user-chosen contact pairs (body, local point) x 2 - none by default, as in
    the reference (ContactKeyframe() is empty until the GUI fills it); the `home` key of the scene (humanoid seated in front of
    the chair... standing pose of armchair.xml:28-58) is the default state.  The XML's cost weights are the start-up weights; the
    per-mode rows INTERACT_WEIGHTS are what the host Transition installs on a mode change."""
    b, sites, torso = _humanoid_model(timestep, False)
    chair = b.body("chair", 0, pos=(-0.35, 0, 0.2))
    box = dict(condim=3, friction=(1, 0.005, 0.0001))
    b.geom(chair, "seat", BOX, size=(.25, .35, .05), pos=(0, 0, 0.12), mass=10.0, **box)
    b.geom(chair, "seat_base", BOX, size=(.30, .35, .15), pos=(-0.12, 0, -0.05), mass=10.0, **box)
    b.geom(chair, "back", BOX, size=(.05, .35, .3), pos=(-0.35, 0, 0.35), quat=(0.984, 0, -0.178, 0), mass=10.0, **box)
    b.geom(chair, "chair_leg_1", BOX, size=(.3, .05, .3), pos=(-0.12, 0.37, 0.1), mass=5.0, **box)
    b.geom(chair, "chair_leg_2", BOX, size=(.3, .05, .3), pos=(-0.12, -0.37, 0.1), mass=5.0, **box)
    b.key("home", _INTERACT_HOME)
    m = b.compile()
    bid = m["names"]["body"]
    pairs = list(contact_pairs)[:5]
    ints = [bid["torso"], bid["pelvis"], bid["foot_right"], bid["foot_left"], bid["head"], bid["shin_right"], bid["shin_left"],
            1 if facing_target is not None else 0]
    dbls = list(facing_target) if facing_target is not None else [0.0, 0.0]
    for k in range(5):
        if k < len(pairs):
            b1, p1, b2, p2 = pairs[k]
            ints += [bid[b1] if isinstance(b1, str) else int(b1), bid[b2] if isinstance(b2, str) else int(b2)]
            dbls += list(p1) + list(p2)
        else:
            ints += [-1, -1]
            dbls += [0.0] * 6
    terms = [(1, 6, 10.0, [0.1]), (1, 6, 10.0, [0.1]), (1, 6, 5.0, [0.1]), (1, 6, 5.0, [0.1]), (1, 6, 0.0, [0.1]), (1, 6, 20.0, [0.1]),
             (1, 6, 30.0, [0.1]), (1, 6, 0.0, [0.1]), (1, 6, 0.0, [0.1]), (2, 0, 10.0), (21, 0, 0.01), (21, 3, 0.8, [0.05]), (15, 6, 100.0, [0.1])]
    task = make_task(TASK_HUMANOID_INTERACT, terms, parameters=[1.4, 1.3], traces=[(OBJ_BODY, torso)], int_data=ints, dbl_data=dbls)
    state = np.concatenate([np.array(_INTERACT_HOME), np.zeros(m["nv"])])
    return m, task, dict(N=10, P=3, sigma=(0.05, 0.0), interp=0, horizon=24, state=state, mocap=np.zeros(0))


# ----------------------------------------------------------------------------------- Shadow hand + cube (BASELINE configs[4])
# mjpc/tasks/shadow_reorient/task.xml is local (floor, goal body, cost table, planner numerics, the 35-value `grasp` key), so is
# the cube (common_assets/reorientation_cube.xml + cube.xml.patch: half size 0.022, 0.126 kg, at 0.325 0 0.075,
# quat 0.707 0.707 0 0).  The hand itself is menagerie's shadow_hand/right_hand.xml, fetched by CMake and NOT in the reference
# tree.  What follows is therefore a SYNTHETIC hand, documented as such: the published kinematic tree
# of the Shadow E3M5 right hand (forearm welded to the world; WRJ2, WRJ1; FF/MF/RF J4..J1; LF J5..J1; TH J5..J1 = 24 hinges),
# joint ranges / damping / armature / frictionloss / position-servo gains and the four J2+J1 coupling tendons as recalled from
# that file, link masses and primitive collision shapes of the same dimensions; the finger-tip meshes of the original are
# replaced by capsules and the forearm is oriented so that the `grasp` key of task.xml makes sense (palm up, cube in the palm).
# Sizes match the reference's task exactly: nq 35, nv 33, nu 20, 81 residuals, 6 cost terms.
_HAND_PLASTIC = dict(solimp=(0.5, 0.99, 0.0001, 0.5, 2.0), solref=(0.005, 1.0), friction=(0.6, 0.005, 0.0001))
_GRASP_KEY = [1, 0, 0, 0, 0.33326, -0.00362331, 0.0375343, 0.707635, 0.70405, 0.0500937, -0.0325089, 5.55212e-10, -0.235248,
              -0.178041, 0.480484, 0.730515, 0.6284, -0.059347, 0.535468, 0.746225, 0.56556, -0.03491, 0.544632, 0.53414, 0.793355,
              0.384846, -0.254843, 0.178072, 0.761935, 0.746225, -0.90042, 0.06721, 0.01047, 0.6981, 0.4255]     # task.xml:56


def shadow_hand(timestep=0.01, cone=0, nconmax=32, nefcmax=128):
    """Shadow-hand cube reorientation (mjpc/tasks/shadow_reorient/task.xml + hand.cc) on the synthetic hand described above.
    cone: 0 pyramidal (MuJoCo's default), 1 elliptic with impratio 10."""
    b = ModelBuilder(timestep=timestep, cone=cone, impratio=10.0 if cone == 1 else 1.0)
    b.nconmax = nconmax; b.nefcmax = nefcmax
    fr = (0.6, 0.005, 0.0001)                                       # task.xml:25-27 <default><geom friction=".6"/>
    b.geom(0, "floor", PLANE, pos=(0, 0, -0.2), size=(0, 0, 0.05), friction=fr)
    goal = b.body("goal", 0, pos=(0.325, 0.17, 0.0475))            # task.xml:32-35
    b.joint(goal, "goal_ball", BALL, damping=0.01)
    b.geom(goal, "goal", BOX, size=(0.022, 0.022, 0.022), mass=0.126, contype=0, conaffinity=0, friction=fr)
    cube = b.body("cube", 0, pos=(0.325, 0.0, 0.075), quat=(0.707, 0.707, 0, 0))      # cube.xml.patch
    b.joint(cube, "cube_free", FREE)
    b.geom(cube, "cube", BOX, size=(0.022, 0.022, 0.022), mass=0.126, friction=fr)
    # ---- the hand: local z = along the fingers -> world +x, local -y = palm side -> world +z
    P = _HAND_PLASTIC
    fore = b.body("rh_forearm", 0, pos=(0, 0, 0), quat=(0.5, -0.5, 0.5, -0.5),
                  inertial=dict(mass=3.0, pos=(0, 0, 0.09), diaginertia=(0.0138, 0.0138, 0.00744)))
    b.geom(fore, "forearm", CAPSULE, size=(0.04, 0.07), pos=(0, 0, 0.08), **P)
    b.geom(fore, "forearm_mount", BOX, size=(0.035, 0.035, 0.035), pos=(0.01, 0, 0.181), quat=(0.380188, 0.924909, 0, 0), **P)

    def hinge(body, name, axis, rng, damping=0.05):
        b.joint(body, name, HINGE, axis=axis, limited=True, range=rng, damping=damping, armature=0.0002, frictionloss=0.01)

    wrist = b.body("rh_wrist", fore, pos=(0.01, 0, 0.21348), inertial=dict(mass=0.1, pos=(0, 0, 0.029), diaginertia=(6.4e-5, 4.38e-5, 3.5e-5)))
    hinge(wrist, "rh_WRJ2", (0, 1, 0), (-0.523599, 0.174533), damping=0.5)
    b.geom(wrist, "wrist", CAPSULE, size=(0.0135, 0.015), quat=(0.5, 0.5, 0.5, -0.5), **P)
    palm = b.body("rh_palm", wrist, pos=(0, 0, 0.034), inertial=dict(mass=0.3, pos=(0, 0, 0.035), diaginertia=(0.0005287, 0.0003581, 0.000191)))
    hinge(palm, "rh_WRJ1", (1, 0, 0), (-0.698132, 0.488692), damping=0.5)
    grasp_site = b.site(palm, "grasp_site", pos=(0, -0.035, 0.09))
    for i, (size, pos) in enumerate([((0.031, 0.0035, 0.049), (0.011, 0.0085, 0.038)), ((0.018, 0.0085, 0.049), (-0.002, -0.0035, 0.038)),
                                     ((0.013, 0.0085, 0.005), (0.029, -0.0035, 0.082)), ((0.013, 0.007, 0.009), (0.0265, -0.001, 0.07)),
                                     ((0.0105, 0.0135, 0.012), (0.0315, -0.0085, 0.001)), ((0.009, 0.012, 0.002), (0.011, 0, 0.089)),
                                     ((0.01, 0.012, 0.02), (-0.03, 0, 0.009))]):
        b.geom(palm, f"palm{i}", BOX, size=size, pos=pos, **P)

    def finger(prefix, parent, pos, jnames):
        kn = b.body(f"rh_{prefix}knuckle", parent, pos=pos, inertial=dict(mass=0.008, pos=(0, 0, 0), diaginertia=(3.2e-7, 2.6e-7, 2.6e-7)))
        # abduction axis: -y for the first and middle finger, +y for ring and little finger (positive angle spreads the fingers)
        hinge(kn, jnames[0], (0, -1, 0) if prefix in ("ff", "mf") else (0, 1, 0), (-0.349066, 0.349066))
        pr = b.body(f"rh_{prefix}proximal", kn, inertial=dict(mass=0.03, pos=(0, 0, 0.0225), diaginertia=(1e-5, 9.8e-6, 1.8e-6)))
        hinge(pr, jnames[1], (1, 0, 0), (-0.261799, 1.5708))
        b.geom(pr, f"{prefix}proximal", CAPSULE, size=(0.009, 0.02), pos=(0, 0, 0.025), **P)
        mi = b.body(f"rh_{prefix}middle", pr, pos=(0, 0, 0.045), inertial=dict(mass=0.017, pos=(0, 0, 0.0125), diaginertia=(2.7e-6, 2.6e-6, 8.7e-7)))
        hinge(mi, jnames[2], (1, 0, 0), (0, 1.5708))
        b.geom(mi, f"{prefix}middle", CAPSULE, size=(0.009, 0.0125), pos=(0, 0, 0.0125), **P)
        di = b.body(f"rh_{prefix}distal", mi, pos=(0, 0, 0.025), inertial=dict(mass=0.013, pos=(0, 0, 0.0130769), diaginertia=(1.28092e-6, 1.12092e-6, 5.3e-7)))
        hinge(di, jnames[3], (1, 0, 0), (0, 1.5708))
        b.geom(di, f"{prefix}distal", CAPSULE, size=(0.0075, 0.009), pos=(0, 0, 0.013), **P)      # stands in for the finger-tip mesh
        b.tendon(f"rh_{prefix.upper()}T1", [jnames[2], jnames[3]], [1.0, 1.0])

    finger("ff", palm, (0.033, 0, 0.095), ["rh_FFJ4", "rh_FFJ3", "rh_FFJ2", "rh_FFJ1"])
    finger("mf", palm, (0.011, 0, 0.099), ["rh_MFJ4", "rh_MFJ3", "rh_MFJ2", "rh_MFJ1"])
    finger("rf", palm, (-0.011, 0, 0.095), ["rh_RFJ4", "rh_RFJ3", "rh_RFJ2", "rh_RFJ1"])
    lfm = b.body("rh_lfmetacarpal", palm, pos=(-0.033, 0, 0.02071), inertial=dict(mass=0.03, pos=(0, 0, 0.04), diaginertia=(1.638e-5, 1.45e-5, 4.272e-6)))
    hinge(lfm, "rh_LFJ5", (0.573576, 0, 0.819152), (0, 0.785398))
    b.geom(lfm, "lfmetacarpal", BOX, size=(0.011, 0.012, 0.025), pos=(0.002, 0, 0.033), **P)
    finger("lf", lfm, (0, 0, 0.06579), ["rh_LFJ4", "rh_LFJ3", "rh_LFJ2", "rh_LFJ1"])
    thb = b.body("rh_thbase", palm, pos=(0.034, -0.00858, 0.029), quat=(0.92388, 0, 0.382683, 0), inertial=dict(mass=0.01, pos=(0, 0, 0), diaginertia=(1.6e-7, 1.6e-7, 1.6e-7)))
    hinge(thb, "rh_THJ5", (0, 0, -1), (-1.0472, 1.0472))
    thp = b.body("rh_thproximal", thb, inertial=dict(mass=0.04, pos=(0, 0, 0.019), diaginertia=(1.36e-5, 1.36e-5, 3.13e-6)))
    hinge(thp, "rh_THJ4", (1, 0, 0), (0, 1.22173))
    b.geom(thp, "thproximal", CAPSULE, size=(0.013, 0.019), pos=(0, 0, 0.019), **P)
    thh = b.body("rh_thhub", thp, pos=(0, 0, 0.038), inertial=dict(mass=0.005, pos=(0, 0, 0), diaginertia=(1e-6, 1e-6, 3e-7)))
    hinge(thh, "rh_THJ3", (1, 0, 0), (-0.20944, 0.20944))
    thm = b.body("rh_thmiddle", thh, inertial=dict(mass=0.02, pos=(0, 0, 0.016), diaginertia=(5.1e-6, 5.1e-6, 1.21e-6)))
    hinge(thm, "rh_THJ2", (0, -1, 0), (-0.698132, 0.698132))
    b.geom(thm, "thmiddle", CAPSULE, size=(0.011, 0.009), pos=(0, 0, 0.012), **P)
    b.geom(thm, "thmiddle_tip", SPHERE, size=(0.01,), pos=(0, 0, 0.03), **P)
    thd = b.body("rh_thdistal", thm, pos=(0, 0, 0.032), quat=(0.707107, 0, 0, -0.707107), inertial=dict(mass=0.017, pos=(0, 0, 0.0145588), diaginertia=(2.37794e-6, 2.27794e-6, 1e-6)))
    hinge(thd, "rh_THJ1", (1, 0, 0), (-0.261799, 1.5708))
    b.geom(thd, "thdistal", CAPSULE, size=(0.009, 0.01), pos=(0, 0, 0.0145), **P)                 # stands in for the thumb-tip mesh
    b.exclude(wrist, fore)
    b.exclude(thp, thm)
    # position servos in the order of the original's <actuator> block
    b.position("rh_A_WRJ2", "rh_WRJ2", kp=10, ctrlrange=(-0.523599, 0.174533), forcerange=(-10, 10))
    b.position("rh_A_WRJ1", "rh_WRJ1", kp=8, ctrlrange=(-0.698132, 0.488692), forcerange=(-5, 5))
    b.position("rh_A_THJ5", "rh_THJ5", kp=0.4, ctrlrange=(-1.0472, 1.0472), forcerange=(-3, 3))
    b.position("rh_A_THJ4", "rh_THJ4", kp=1, ctrlrange=(0, 1.22173), forcerange=(-2, 2))
    b.position("rh_A_THJ3", "rh_THJ3", kp=0.5, ctrlrange=(-0.20944, 0.20944), forcerange=(-1, 1))
    b.position("rh_A_THJ2", "rh_THJ2", kp=1.5, ctrlrange=(-0.698132, 0.698132), forcerange=(-1, 1))
    b.position("rh_A_THJ1", "rh_THJ1", kp=1, ctrlrange=(-0.261799, 1.5708), forcerange=(-1, 1))
    for f in ("FF", "MF", "RF", "LF"):
        if f == "LF":
            b.position("rh_A_LFJ5", "rh_LFJ5", kp=1, ctrlrange=(0, 0.785398), forcerange=(-1, 1))
        b.position(f"rh_A_{f}J4", f"rh_{f}J4", kp=1, ctrlrange=(-0.349066, 0.349066), forcerange=(-1, 1))
        b.position(f"rh_A_{f}J3", f"rh_{f}J3", kp=1, ctrlrange=(-0.261799, 1.5708), forcerange=(-1, 1))
        b.position(f"rh_A_{f}J0", tendon=f"rh_{f}T1", kp=0.5, ctrlrange=(0, 3.1415), forcerange=(-1, 1))
    b.key("grasp", _GRASP_KEY)
    m = b.compile()
    assert (m["nq"], m["nv"], m["nu"]) == (35, 33, 20)
    bid = m["names"]["body"]
    # cost table: task.xml:39-44 (user = norm, weight, lo, hi, params...)
    terms = [(3, 1, 20.0, [0.02, 2.0]), (3, 0, 5.0), (3, 0, 10.0), (20, 0, 0.1), (26, 0, 2.5), (26, 0, 1.0e-4)]
    task = make_task(TASK_SHADOW_REORIENT, terms, traces=[(OBJ_BODY, bid["cube"])],          # trace0: framepos objtype="body" cube
                     int_data=[grasp_site, bid["cube"], bid["goal"], 0])
    key = np.array(_GRASP_KEY, float)
    # position targets that hold the grasp posture: joint angles of the key (J2 + J1 for the coupled pairs)
    qj = {n: key[m["jnt_qposadr"][j]] for n, j in m["names"]["joint"].items()}
    ctrl0 = []
    for a in b.actuators:
        ctrl0.append(qj[b.tendons[a["trnid"]]["joints"][0]] + qj[b.tendons[a["trnid"]]["joints"][1]] if a["trntype"] == 3 else key[m["jnt_qposadr"][a["trnid"]]])
    state = np.concatenate([key, np.zeros(m["nv"])])
    defaults = dict(N=10, P=5, sigma=(0.1, 0.0), interp=0, horizon=26, state=state, mocap=np.zeros(0), ctrl0=np.array(ctrl0))
    return m, task, defaults


# ----------------------------------------------------------------------------------- quadruped on the fractal terrain
TASK_QUADRUPED_HILL = 10


def quadruped_hill(timestep=0.01, stage=0):
    """mjpc/tasks/quadruped "Quadruped Hill" (task_hill.xml, quadruped.cc:716-812): the A1 with position servos (a1.xml.patch:17-35:
    kp 50, force range +-33.5) on the fractal height field (assets/fractal.xml: size 5 5 1 2, 100 x 100 samples derived from the
    reference's image by data/make_fractal.py), goal = a mocap body that Transition moves through the 19 stage keys."""
    import os
    data = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "quadruped_hill_terrain.npz"))
    b = ModelBuilder(timestep=timestep, cone=1, impratio=10.0, contact=True)
    b.nconmax = 32
    b.nefcmax = 128
    goal = b.body("goal", 0, pos=tuple(data["stages"][0][:3]), quat=tuple(data["stages"][0][3:]), mocap=True)
    b.geom(goal, "goal", CAPSULE, size=(0.1, 0), fromto=(-0.1, 0, 0, 0.1, 0, 0), contype=0, conaffinity=0, group=2)
    b.geom(0, "floor", PLANE, pos=(0, 0, -0.01), size=(50, 50, 0.05))
    b.geom(0, "terrain", HFIELD, hfield=dict(size=tuple(data["size"]), data=data["data"]))
    trunk, head, foot_geom = _a1_robot(b, pos=(0, 0, 0.5))
    cr = dict(hip=(-0.802851, 0.802851), thigh=(-1.9472, 3.28879), calf=(-0.89653, 0.883702))
    for jpref in ("FR", "FL", "RR", "RL"):
        for part in ("hip", "thigh", "calf"):
            b.position(f"{jpref}_{part}", joint=f"{jpref}_{part}_joint", kp=50.0, ctrlrange=cr[part], forcerange=(-33.5, 33.5))
    b.key("home", list(data["home"]))
    m = b.compile()
    sid = m["names"]["site"]
    stages = np.asarray(data["stages"], float)
    task = make_task(TASK_QUADRUPED_HILL, [(1, 0, 1.0), (3, 0, 5.0), (9, 0, 1.0), (12, 0, 0.25)], parameters=[0.25],
                     traces=[(OBJ_BODY, trunk)], int_data=[trunk, sid["FR"], sid["FL"], sid["RR"], sid["RL"], int(stage)],
                     dbl_data=list(stages.ravel()))
    state = np.concatenate([np.asarray(data["home"], float), np.zeros(m["nv"])])
    defaults = dict(N=10, P=5, sigma=(0.3, 0.0), interp=2, horizon=26, state=state, mocap=stages[stage].copy())
    return m, task, defaults


# ----------------------------------------------------------------------------------- walker, acrobot (registry tasks)
def walker(timestep=0.01):
    """mjpc/tasks/walker (task.xml:10-33, walker.cc:39-57).  The planar walker of dm_control's walker.xml is fetched and patched by
    CMake (walker.xml.patch), not in the tree: SYNTHETIC restatement of its published structure (torso on rootz / rootx / rooty,
    two legs of thigh / leg / foot capsules, hinge axes -y, motors with gears 100 / 50 / 20), numbers recalled.  Cost table,
    parameters and agent settings are task.xml's: horizon 0.8 s at 0.01 s, 3 spline points, exploration 0.5."""
    b = ModelBuilder(timestep=timestep, cone=0, contact=True)
    b.geom(0, "floor", PLANE, pos=(998, 0, 0), size=(1000, 0.8, 0.2), friction=(0.7, 0.1, 0.1), contype=0, conaffinity=1)
    kw = dict(contype=1, conaffinity=0, friction=(0.7, 0.1, 0.1), density=1000.0)
    torso = b.body("torso", 0, pos=(0, 0, 1.3))
    b.joint(torso, "rootz", SLIDE, axis=(0, 0, 1))
    b.joint(torso, "rootx", SLIDE, axis=(1, 0, 0))
    b.joint(torso, "rooty", HINGE, axis=(0, 1, 0))
    b.geom(torso, "torso", CAPSULE, size=(0.07, 0.3), **kw)
    site = b.site(torso, "torso_site")
    jkw = dict(axis=(0, -1, 0), damping=0.1, armature=0.01, limited=True, solimplimit=(0, 0.99, 0.01, 0.5, 2))
    for side, y in (("right", -0.05), ("left", 0.05)):
        thigh = b.body(side + "_thigh", torso, pos=(0, y, -0.3))
        b.joint(thigh, side + "_hip", HINGE, range=(math.radians(-20), math.radians(100)), **jkw)
        b.geom(thigh, side + "_thigh", CAPSULE, size=(0.05, 0.225), pos=(0, 0, -0.225), **kw)
        leg = b.body(side + "_leg", thigh, pos=(0, 0, -0.7))
        b.joint(leg, side + "_knee", HINGE, pos=(0, 0, 0.25), range=(math.radians(-150), 0.0), **jkw)
        b.geom(leg, side + "_leg", CAPSULE, size=(0.04, 0.25), **kw)
        foot = b.body(side + "_foot", leg, pos=(0.06, 0, -0.25))
        b.joint(foot, side + "_ankle", HINGE, pos=(-0.06, 0, 0), range=(math.radians(-45), math.radians(45)), **jkw)
        b.geom(foot, side + "_foot", CAPSULE, size=(0.05, 0.1), zaxis=(1, 0, 0), **kw)
    for side in ("right", "left"):
        for jn, gear in (("hip", 100.0), ("knee", 50.0), ("ankle", 20.0)):
            b.actuator(f"{side}_{jn}", f"{side}_{jn}", gear=gear, ctrlrange=(-1, 1))
    m = b.compile()
    task = make_task(TASK_WALKER, [(6, 0, 0.1), (1, 0, 10.0), (1, 0, 3.0), (1, 0, 1.0)], parameters=[1.2, 0.0],
                     traces=[(OBJ_SITE, site)], int_data=[torso])
    st = np.concatenate([m["qpos0"], np.zeros(m["nv"])])
    defaults = dict(N=10, P=3, sigma=(0.5, 0.0), interp=2, horizon=80, state=st, mocap=np.zeros(0))
    return m, task, defaults


def acrobot(timestep=0.01):
    """mjpc/tasks/acrobot (task.xml:8-31, acrobot.cc:34-49).  dm_control's acrobot.xml is fetched and patched by CMake, not in the
    tree: SYNTHETIC restatement (two unit-mass 1 m capsule links on y hinges, damping 0.05, motor on the elbow with gear 2,
    constraints disabled, target site 4 m up), numbers recalled.  Cost table and agent settings are task.xml's: horizon 2 s at
    0.01 s, 10 spline points, exploration 0.05; start = key "home" (hanging)."""
    b = ModelBuilder(timestep=timestep, contact=False)
    b.geom(0, "floor", PLANE, size=(3, 3, 0.2), contype=0, conaffinity=0)
    target = b.site(0, "target", pos=(0, 0, 4))
    upper = b.body("upper_arm", 0, pos=(0, 0, 2))
    b.joint(upper, "shoulder", HINGE, axis=(0, 1, 0), damping=0.05)
    b.geom(upper, "upper_arm", CAPSULE, size=(0.051, 0), fromto=(0, 0, 0, 0, 0, 1), mass=1.0, contype=0, conaffinity=0)
    lower = b.body("lower_arm", upper, pos=(0, 0, 1))
    b.joint(lower, "elbow", HINGE, axis=(0, 1, 0), damping=0.05)
    b.geom(lower, "lower_arm", CAPSULE, size=(0.049, 0), fromto=(0, 0, 0, 0, 0, 1), mass=1.0, contype=0, conaffinity=0)
    tip = b.site(lower, "tip", pos=(0, 0, 1))
    b.actuator("elbow", "elbow", gear=2.0, ctrlrange=(-1, 1))
    b.key("home", [3.142, 0.0])
    b.disableflags = 1                                   # <flag constraint="disable"/> (acrobot.xml.patch:17)
    m = b.compile()
    task = make_task(TASK_ACROBOT, [(2, 0, 50.0), (2, 0, 1.0), (1, 0, 0.05)], parameters=[0.0], traces=[(OBJ_SITE, tip)],
                     int_data=[target, tip])
    defaults = dict(N=10, P=10, sigma=(0.05, 0.0), interp=2, horizon=200, state=np.array([3.142, 0.0, 0.0, 0.0]), mocap=np.zeros(0))
    return m, task, defaults


# ----------------------------------------------------------------------------------- a7 features off the BASELINE models
def ball_chain(timestep=0.005, tendon_frictionloss=0.0):
    """Small test model for the mj_step features no BASELINE model has: limited ball joints and a fixed tendon with a spring, a
    damper and a limit that couples joints on DIFFERENT branches (its limit row lies outside M's sparsity pattern).  The residual
    copies the state (TASK_COPYSTATE).  tendon_frictionloss > 0: that tendon also has friction loss, and a second, single-joint tendon
    (inside the pattern) has half of it."""
    b = ModelBuilder(timestep=timestep, gravity=(0, 0, -9.81), contact=True)
    b.geom(0, "floor", PLANE, pos=(0, 0, -1.2), size=(2, 2, 0.1))
    l1 = b.body("l1", 0, pos=(0, 0, 0))
    b.joint(l1, "b1", BALL, limited=True, range=(0, 0.5), damping=0.05)
    b.geom(l1, "g1", CAPSULE, size=(0.04, 0), fromto=(0, 0, 0, 0, 0, -0.4), mass=1.0)
    l2 = b.body("l2", l1, pos=(0, 0, -0.4))
    b.joint(l2, "b2", BALL, limited=True, range=(0, 0.7), margin=0.02, damping=0.05)
    b.geom(l2, "g2", CAPSULE, size=(0.035, 0), fromto=(0, 0, 0, 0, 0, -0.35), mass=0.6)
    arms = []
    for name, y in (("ra", -0.1), ("la", 0.1)):
        a = b.body(name, l1, pos=(0, y, -0.2))
        b.joint(a, name + "_j", HINGE, axis=(1, 0, 0), damping=0.02, armature=0.01)
        b.geom(a, name + "_g", CAPSULE, size=(0.02, 0), fromto=(0, 0, 0, 0, 1.5 * y, -0.2), mass=0.2)
        arms.append(name + "_j")
    tip = b.site(l2, "tip", pos=(0, 0, -0.35))
    b.tendon("couple", arms, [1.0, 1.0], limited=True, range=(-0.4, 0.4), stiffness=4.0, damping=0.1, springlength=(-0.05, 0.05),
             frictionloss=tendon_frictionloss)
    if tendon_frictionloss > 0:
        b.tendon("ra_drag", arms[:1], [1.5], frictionloss=0.5 * tendon_frictionloss, solreffriction=(0.03, 1.0))
    b.actuator("ra_m", "ra_j", gear=1.0, ctrlrange=(-1, 1))
    b.actuator("la_m", "la_j", gear=1.0, ctrlrange=(-1, 1))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(10, 0, 1.0), (8, 0, 0.1)], traces=[(OBJ_SITE, tip)])
    q = np.zeros(m["nq"]); q[0] = 1.0; q[4] = 1.0
    v = np.zeros(m["nv"]); v[0:3] = [5.0, 2.0, 0.6]; v[3:6] = [-2.0, 4.0, 1.0]; v[6] = 6.0; v[7] = 4.0
    defaults = dict(N=6, P=4, sigma=(0.4, 0.0), interp=2, horizon=60, state=np.concatenate([q, v]), mocap=np.zeros(0))
    return m, task, defaults


def cylinder_pile(timestep=0.004):
    """Test model for the convex pairs that go through the portal-refinement collider: a cylinder standing on a box, a second
    one lying across it (cylinder-box and cylinder-cylinder contacts), a motor pushing the lower one sideways; an ellipsoid on
    the box and a tilted one on the floor plane."""
    b = ModelBuilder(timestep=timestep, cone=1, impratio=1.0, contact=True)
    b.geom(0, "floor", PLANE, pos=(0, 0, 0), size=(2, 2, 0.1))
    b.geom(0, "table", BOX, pos=(0, 0, 0.1), size=(0.3, 0.3, 0.1), friction=(0.8, 0.005, 0.0001))
    c1 = b.body("c1", 0, pos=(0.0, 0.0, 0.2 + 0.06 + 0.002))
    b.joint(c1, "c1_free", FREE)
    b.geom(c1, "c1_g", CYLINDER, size=(0.05, 0.06), mass=0.5, friction=(0.8, 0.005, 0.0001))
    c2 = b.body("c2", 0, pos=(0.01, 0.0, 0.2 + 0.12 + 0.03 + 0.006), quat=(math.cos(math.pi / 4), math.sin(math.pi / 4), 0, 0))
    b.joint(c2, "c2_free", FREE)
    b.geom(c2, "c2_g", CYLINDER, size=(0.03, 0.08), mass=0.2, friction=(0.8, 0.005, 0.0001))
    e1 = b.body("e1", 0, pos=(0.18, 0.1, 0.2 + 0.03 + 0.002))
    b.joint(e1, "e1_free", FREE)
    b.geom(e1, "e1_g", ELLIPSOID, size=(0.06, 0.04, 0.03), mass=0.3, friction=(0.8, 0.005, 0.0001))        # on the table: ellipsoid-box
    e2 = b.body("e2", 0, pos=(0.6, 0.0, 0.05 + 0.002), quat=(math.cos(0.3), 0, math.sin(0.3), 0))
    b.joint(e2, "e2_free", FREE)
    b.geom(e2, "e2_g", ELLIPSOID, size=(0.05, 0.08, 0.04), mass=0.3, friction=(0.8, 0.005, 0.0001))        # on the floor: plane-ellipsoid
    pusher = b.body("pusher", 0, pos=(-0.12, 0, 0.26))
    b.joint(pusher, "push", SLIDE, axis=(1, 0, 0), limited=True, range=(-0.02, 0.1), damping=2.0)
    b.geom(pusher, "push_g", SPHERE, size=(0.02,), mass=0.1)
    top = b.site(c2, "top", pos=(0, 0, 0))
    b.actuator("push_m", "push", gear=4.0, ctrlrange=(-1, 1))
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(m["nq"], 0, 1.0), (m["nv"], 0, 0.1)], traces=[(OBJ_SITE, top)])
    st = np.concatenate([m["qpos0"], np.zeros(m["nv"])])
    defaults = dict(N=6, P=3, sigma=(0.5, 0.0), interp=2, horizon=50, state=st, mocap=np.zeros(0))
    return m, task, defaults


def terrain_balls(timestep=0.004):
    """Test model for height-field collisions: a bumpy terrain, two spheres, a capsule and an ellipsoid dropped on it, a motor
    pushing one sphere uphill."""
    b = ModelBuilder(timestep=timestep, cone=1, impratio=1.0, contact=True)
    n = 17
    xs = np.linspace(-1, 1, n)
    data = 0.5 + 0.25 * np.sin(2.1 * xs)[None, :] * np.cos(1.7 * xs)[:, None] + 0.15 * xs[None, :]
    data = (data - data.min()) / (data.max() - data.min())
    b.geom(0, "terrain", HFIELD, hfield=dict(size=(1.0, 1.0, 0.3, 0.1), data=data), friction=(0.8, 0.005, 0.0001), condim=3)
    def height(x, y):       # bilinear is not what the collider uses (triangles), good enough to place things above the ground
        i = min(max(int((y + 1) / 2 * (n - 1)), 0), n - 2); j = min(max(int((x + 1) / 2 * (n - 1)), 0), n - 2)
        return 0.3 * max(data[i, j], data[i, j + 1], data[i + 1, j], data[i + 1, j + 1])
    for k, (ty, size, xy) in enumerate([(SPHERE, (0.06,), (-0.3, 0.2)), (SPHERE, (0.05,), (0.35, -0.25)), (CAPSULE, (0.04, 0.08), (0.1, 0.45)),
                                        (ELLIPSOID, (0.07, 0.05, 0.04), (-0.45, -0.4))]):
        bid = b.body(f"o{k}", 0, pos=(xy[0], xy[1], height(*xy) + 0.1), quat=(math.cos(0.2 * k), math.sin(0.2 * k), 0, 0))
        b.joint(bid, f"o{k}_free", FREE)
        b.geom(bid, f"o{k}_g", ty, size=size, mass=0.3, friction=(0.8, 0.005, 0.0001))
    cart = b.body("cart", 0, pos=(0.0, 0.0, height(0, 0) + 0.05))
    b.joint(cart, "cx", SLIDE, axis=(1, 0, 0), damping=1.0)
    b.joint(cart, "cz", SLIDE, axis=(0, 0, 1), damping=0.5)
    b.geom(cart, "cart_g", SPHERE, size=(0.05,), mass=0.4, friction=(0.8, 0.005, 0.0001))
    site = b.site(cart, "cart_site")
    b.actuator("push", "cx", gear=3.0, ctrlrange=(-1, 1))
    b.nconmax = 24; b.nefcmax = 96
    m = b.compile()
    task = make_task(TASK_COPYSTATE, [(m["nq"], 0, 1.0), (m["nv"], 0, 0.1)], traces=[(OBJ_SITE, site)])
    st = np.concatenate([m["qpos0"], np.zeros(m["nv"])])
    defaults = dict(N=6, P=3, sigma=(0.5, 0.0), interp=2, horizon=60, state=st, mocap=np.zeros(0))
    return m, task, defaults


REGISTRY = {"humanoid_interact": humanoid_interact, "fingers": fingers, "site_servo": site_servo, "noslip_elliptic3": lambda: noslip_mix(1, 3), "noslip_elliptic4": lambda: noslip_mix(1, 4), "noslip_elliptic6": lambda: noslip_mix(1, 6), "noslip_pyramidal3": lambda: noslip_mix(0, 3), "noslip_pyramidal6": lambda: noslip_mix(0, 6), "fingers_grasp": lambda: fingers(grasp=True), "welded": welded, "swimmer": swimmer, "quadrotor": quadrotor, "linkage": linkage, "servo_arm": servo_arm, "particle_timevarying": particle_task, "particle_fixed": lambda: particle_task(fixed=True), "filter_arm": filter_arm, "ball_chain_friction": lambda: ball_chain(tendon_frictionloss=0.3), "quadruped_hill": quadruped_hill, "terrain_balls": terrain_balls, "walker": walker, "acrobot": acrobot, "ball_chain": ball_chain, "cylinder_pile": cylinder_pile, "humanoid_stand": humanoid_stand, "humanoid_walk": humanoid_walk, "particle": particle, "cartpole": cartpole, "quadruped": quadruped, "humanoid_track": humanoid_track, "shadow_hand": shadow_hand}
