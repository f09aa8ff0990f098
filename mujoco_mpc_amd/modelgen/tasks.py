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

from .builder import (BOX, CAPSULE, CYLINDER, FREE, HINGE, PLANE, SLIDE, SPHERE, ModelBuilder)

TASK_PARTICLE, TASK_CARTPOLE, TASK_QUADRUPED, TASK_COPYSTATE = 0, 1, 2, 3
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
    # A1 (a1.xml.patch); class a1: friction 0.6 margin 0.001 condim 1; class collision: capsule, group 3
    col = dict(friction=(0.6, 0.005, 0.0001), margin=0.001, condim=1, group=3)
    jdef = dict(damping=2.0, armature=0.01, frictionloss=0.2, limited=True)
    trunk = b.body("trunk", 0, pos=(0, 0, 0.5),
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


REGISTRY = {"particle": particle, "cartpole": cartpole, "quadruped": quadruped}
