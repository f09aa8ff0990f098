"""Minimal MJCF-like model compiler (numpy) -> flat arrays mirroring `mjModel` field names.

The reference loads its models with MuJoCo's MJCF compiler (`mj_loadXML`, mjpc/agent.cc:242), which
is a third-party dependency absent from this image.  The engine's ABI (include/mjpc_hip.h) takes the
*compiled* model, so in a real MJPC checkout the shim copies pointers out of `mjModel`.  This module
exists so that tests / bench can author the BASELINE models without MuJoCo: it restates the parts of
the compiler those models need (frames, fromto, inertia from geoms, inertial-frame diagonalisation,
dof tree, qpos0, rbound, invweight0, meaninertia).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

FREE, BALL, SLIDE, HINGE = 0, 1, 2, 3
PLANE, HFIELD, SPHERE, CAPSULE, ELLIPSOID, CYLINDER, BOX, MESH = range(8)
MINVAL = 1e-15


# ---------------------------------------------------------------------------- quaternion helpers
def quat_mul(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return np.array([
        a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3],
        a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2],
        a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1],
        a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]])


def quat2mat(q):
    q = np.asarray(q, float)
    w, x, y, z = q
    return np.array([
        [w*w + x*x - y*y - z*z, 2*(x*y - w*z), 2*(x*z + w*y)],
        [2*(x*y + w*z), w*w - x*x + y*y - z*z, 2*(y*z - w*x)],
        [2*(x*z - w*y), 2*(y*z + w*x), w*w - x*x - y*y + z*z]])


def mat2quat(m):
    m = np.asarray(m, float)
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    if tr > 0:
        s = math.sqrt(tr + 1.0) * 2
        q = [0.25 * s, (m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [(m[2, 1] - m[1, 2]) / s, 0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 2] - m[2, 0]) / s, (m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[1, 0] - m[0, 1]) / s, (m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s]
    q = np.array(q)
    return q / np.linalg.norm(q)


def axisangle2quat(axis, angle):
    axis = np.asarray(axis, float)
    return np.concatenate([[math.cos(angle / 2)], axis * math.sin(angle / 2)])


def z2quat(vec):
    """quaternion rotating the z axis onto `vec` (MJCF fromto / zaxis)."""
    v = np.asarray(vec, float)
    v = v / np.linalg.norm(v)
    z = np.array([0.0, 0.0, 1.0])
    axis = np.cross(z, v)
    s = np.linalg.norm(axis)
    if s < 1e-10:
        return np.array([1.0, 0, 0, 0]) if v[2] > 0 else np.array([0.0, 1.0, 0, 0])
    axis /= s
    ang = math.atan2(s, v[2])
    return axisangle2quat(axis, ang)


def euler2quat(e):
    """MJCF default eulerseq 'xyz' (intrinsic)."""
    q = np.array([1.0, 0, 0, 0])
    for i, a in enumerate(e):
        ax = np.zeros(3); ax[i] = 1
        q = quat_mul(q, axisangle2quat(ax, a))
    return q


def normq(q):
    q = np.asarray(q, float)
    return q / np.linalg.norm(q)


# ---------------------------------------------------------------------------- spec records
DEF_SOLREF = (0.02, 1.0)
DEF_SOLIMP = (0.9, 0.95, 0.001, 0.5, 2.0)


@dataclass
class _Joint:
    name: str; body: int; type: int; axis: np.ndarray; pos: np.ndarray
    limited: bool; range: tuple; damping: float; armature: float; frictionloss: float
    stiffness: float; ref: float; springref: float; margin: float
    solreflimit: tuple; solimplimit: tuple; solreffriction: tuple; solimpfriction: tuple


@dataclass
class _Geom:
    name: str; body: int; type: int; size: np.ndarray; pos: np.ndarray; quat: np.ndarray
    mass: float | None; density: float; contype: int; conaffinity: int; condim: int
    friction: tuple; priority: int; margin: float; gap: float; solmix: float
    solref: tuple; solimp: tuple; group: int


@dataclass
class _Body:
    name: str; parent: int; pos: np.ndarray; quat: np.ndarray; mocap: bool
    inertial: dict | None = None
    joints: list = field(default_factory=list)
    geoms: list = field(default_factory=list)


class ModelBuilder:
    def __init__(self, timestep=0.002, gravity=(0, 0, -9.81), cone=0, impratio=1.0,
                 contact=True, tolerance=1e-8, iterations=100, ls_iterations=50, ls_tolerance=0.01, integrator=0, density=0.0, viscosity=0.0, wind=(0, 0, 0)):
        self.integrator = integrator      # 0 Euler, 2 implicit, 3 implicitfast (mjtIntegrator)
        self.fluid = (float(density), float(viscosity), tuple(float(x) for x in wind))      # inertia-box fluid model (mjOption)
        self.opt = dict(timestep=timestep, gravity=np.array(gravity, float), cone=cone, impratio=impratio,
                        contact=contact, tolerance=tolerance, iterations=iterations,
                        ls_iterations=ls_iterations, ls_tolerance=ls_tolerance)
        self.bodies = [_Body("world", 0, np.zeros(3), np.array([1.0, 0, 0, 0]), False)]
        self.joints: list[_Joint] = []
        self.geoms: list[_Geom] = []
        self.sites: list[tuple] = []
        self.actuators: list[dict] = []
        self.keys: list[tuple] = []
        self.tendons: list[dict] = []
        self.equalities: list[dict] = []
        self.gravcomp: dict[int, float] = {}
        self.actfrc: dict[str, tuple] = {}
        self.key_mpos = None          # optional [nkey, 3*nmocap]
        self.excludes: list[tuple] = []
        self.nuserdata = 0
        self.nconmax = 0
        self.nefcmax = 0

    # ---- authoring API
    def body(self, name, parent=0, pos=(0, 0, 0), quat=(1, 0, 0, 0), mocap=False, inertial=None, gravcomp=0.0):
        if isinstance(parent, str):
            parent = self.body_id(parent)
        self.bodies.append(_Body(name, parent, np.array(pos, float), normq(quat), mocap, inertial))
        if gravcomp:
            self.gravcomp[len(self.bodies) - 1] = float(gravcomp)
        return len(self.bodies) - 1

    def body_id(self, name):
        for i, b in enumerate(self.bodies):
            if b.name == name:
                return i
        raise KeyError(name)

    def joint(self, body, name, type=HINGE, axis=(0, 0, 1), pos=(0, 0, 0), limited=False, range=(0, 0),
              damping=0.0, armature=0.0, frictionloss=0.0, stiffness=0.0, ref=0.0, springref=0.0, margin=0.0,
              solreflimit=DEF_SOLREF, solimplimit=DEF_SOLIMP, solreffriction=DEF_SOLREF, solimpfriction=DEF_SOLIMP, actuatorfrcrange=None):
        if actuatorfrcrange is not None:      # MJCF actuatorfrcrange: clamp of the total actuator force on this (scalar) joint
            self.actfrc[name] = tuple(actuatorfrcrange)
        ax = np.array(axis, float)
        if type in (HINGE, SLIDE):
            ax = ax / np.linalg.norm(ax)
        j = _Joint(name, body, type, ax, np.array(pos, float), limited, tuple(range), damping, armature,
                   frictionloss, stiffness, ref, springref, margin, tuple(solreflimit), tuple(solimplimit),
                   tuple(solreffriction), tuple(solimpfriction))
        # joints of a body must be contiguous: insert after the last joint of this body
        self.joints.append(j)
        self.bodies[body].joints.append(len(self.joints) - 1)
        return len(self.joints) - 1

    def geom(self, body, name="", type=SPHERE, size=(0, 0, 0), pos=(0, 0, 0), quat=(1, 0, 0, 0), fromto=None,
             mass=None, density=1000.0, contype=1, conaffinity=1, condim=3, friction=(1, 0.005, 0.0001),
             priority=0, margin=0.0, gap=0.0, solmix=1.0, solref=DEF_SOLREF, solimp=DEF_SOLIMP, group=0,
             zaxis=None, euler=None, mesh=None, hfield=None):
        """mesh: (nvert, 3) vertices of a MESH geom in the geom frame (collision uses their convex hull, as MuJoCo does); its
        size becomes the half extents of the vertex cloud (mass properties of that box: model authoring for tests only)."""
        if mesh is not None:
            mesh = np.asarray(mesh, float).reshape(-1, 3)
            size = tuple(np.abs(mesh).max(axis=0))
        if hfield is not None:       # dict(size=(radius_x, radius_y, elevation_z, base_z), data=[nrow][ncol] in [0, 1]); static geoms only
            hfield = dict(size=tuple(float(x) for x in hfield["size"]), data=np.asarray(hfield["data"], float))
            size = hfield["size"][:3]; mass = 0.0 if mass is None else mass
        size = list(size) + [0.0] * (3 - len(size))
        pos = np.array(pos, float); quat = normq(quat)
        if zaxis is not None:
            quat = z2quat(zaxis)
        if euler is not None:
            quat = euler2quat(euler)
        if fromto is not None:
            a = np.array(fromto[:3], float); b = np.array(fromto[3:], float)
            pos = 0.5 * (a + b)
            quat = z2quat(b - a)
            size[1] = 0.5 * np.linalg.norm(b - a)
        g = _Geom(name, body, type, np.array(size, float), pos, quat, mass, density, contype, conaffinity, condim,
                  tuple(friction), priority, margin, gap, solmix, tuple(solref), tuple(solimp), group)
        g.mesh = mesh; g.hfield = hfield
        self.geoms.append(g)
        self.bodies[body].geoms.append(len(self.geoms) - 1)
        return len(self.geoms) - 1

    def site(self, body, name, pos=(0, 0, 0), quat=(1, 0, 0, 0)):
        self.sites.append((name, body, np.array(pos, float), normq(quat)))
        return len(self.sites) - 1

    def actuator(self, name, joint=None, gainprm=(1, 0, 0), biastype=0, biasprm=(0, 0, 0), gear=1.0,
                 ctrllimited=True, ctrlrange=(-1, 1), forcelimited=False, forcerange=(0, 0), tendon=None,
                 dyntype=0, dynprm=1.0, actlimited=False, actrange=(0, 0), site=None, gear6=None, refsite=None):
        """joint transmission (joint=name) or fixed-tendon transmission (tendon=name); dyntype 1 integrator / 2 filter / 3 filterexact
        gives the actuator one activation state (time constant dynprm)"""
        if gear6 is not None:
            gear = float(gear6[0])             # (mjModel.actuator_gear[6 * i]; site transmissions read all six)
        self.actuators.append(dict(name=name, joint=joint, tendon=tendon, gainprm=tuple(gainprm), biastype=biastype,
                                   biasprm=tuple(biasprm), gear=gear, ctrllimited=ctrllimited,
                                   ctrlrange=tuple(ctrlrange), forcelimited=forcelimited, forcerange=tuple(forcerange),
                                   dyntype=int(dyntype), dynprm=float(dynprm), actlimited=bool(actlimited), actrange=tuple(actrange),
                                   site=site, gear6=None if gear6 is None else tuple(gear6), refsite=refsite))

    def position(self, name, joint=None, tendon=None, kp=1.0, ctrlrange=(-1, 1), forcerange=None, gear=1.0):
        """MJCF <position>: gain kp, affine bias (0, -kp, 0)."""
        self.actuator(name, joint=joint, tendon=tendon, gainprm=(kp, 0, 0), biastype=1, biasprm=(0, -kp, 0), gear=gear,
                      ctrlrange=ctrlrange, forcelimited=forcerange is not None, forcerange=forcerange or (0, 0))

    def tendon(self, name, joints, coefs, limited=False, range=(0, 0), margin=0.0, solreflimit=DEF_SOLREF, solimplimit=DEF_SOLIMP,
               stiffness=0.0, damping=0.0, springlength=None, frictionloss=0.0,
               solreffriction=DEF_SOLREF, solimpfriction=DEF_SOLIMP):
        """fixed tendon: length = sum coef * qpos[joint]; passive spring (dead band `springlength` = value or (lo, hi); None = the
        length at qpos0, MJCF's springlength="-1"), damper and friction loss (one friction row along the tendon)"""
        self.tendons.append(dict(name=name, joints=list(joints), coefs=list(coefs), limited=limited, range=tuple(range), margin=margin,
                                 solref=tuple(solreflimit), solimp=tuple(solimplimit), stiffness=float(stiffness), damping=float(damping),
                                 springlength=springlength, frictionloss=float(frictionloss), solref_fri=tuple(solreffriction), solimp_fri=tuple(solimpfriction)))

    def key(self, name, qpos):
        self.keys.append((name, np.array(qpos, float)))

    def connect(self, body1, body2, anchor, solref=DEF_SOLREF, solimp=DEF_SOLIMP, active=True):
        """MJCF <connect>: the point `anchor` (body1 frame) stays where it is relative to body2 (0 = the world) at qpos0"""
        self.equalities.append(dict(type=0, obj1=body1, obj2=body2, anchor=tuple(anchor), solref=tuple(solref), solimp=tuple(solimp), active=active))

    def weld(self, body1, body2=0, anchor=(0, 0, 0), relpose=None, torquescale=1.0, solref=DEF_SOLREF, solimp=DEF_SOLIMP, active=True):
        """MJCF <weld>: body1 keeps its pose relative to body2 (0 = the world); `anchor` (body2 frame) is the point the position rows
        are taken at; relpose = (pos 3, quat 4) of body2 in body1's frame, None = the pose at qpos0 (mj_setConst)"""
        self.equalities.append(dict(type=1, obj1=body1, obj2=body2, anchor=tuple(anchor), relpose=None if relpose is None else tuple(relpose),
                                    torquescale=float(torquescale), solref=tuple(solref), solimp=tuple(solimp), active=active))

    def joint_equality(self, joint1, joint2=None, polycoef=(0, 1, 0, 0, 0), solref=DEF_SOLREF, solimp=DEF_SOLIMP, active=True):
        """MJCF <equality><joint>: q1 - q1_0 = poly(q2 - q2_0) (joint names; joint2 None: q1 - q1_0 = polycoef[0])"""
        self.equalities.append(dict(type=2, obj1=joint1, obj2=joint2, polycoef=tuple(polycoef), solref=tuple(solref), solimp=tuple(solimp), active=active))

    def tendon_equality(self, tendon1, tendon2=None, polycoef=(0, 1, 0, 0, 0), solref=DEF_SOLREF, solimp=DEF_SOLIMP, active=True):
        """MJCF <equality><tendon>: L1 - L1_0 = poly(L2 - L2_0) (tendon names, lengths relative to qpos0)"""
        self.equalities.append(dict(type=3, obj1=tendon1, obj2=tendon2, polycoef=tuple(polycoef), solref=tuple(solref), solimp=tuple(solimp), active=active))

    def exclude(self, body1, body2):
        self.excludes.append((body1, body2))

    # ---- geometry helpers
    @staticmethod
    def _geom_volume_inertia(g: _Geom):
        s = g.size
        t = g.type
        if t == SPHERE:
            vol = 4.0 / 3.0 * math.pi * s[0] ** 3
            unit = np.array([2 / 5 * s[0] ** 2] * 3)          # per unit mass
        elif t == CAPSULE:
            h = 2 * s[1]; r = s[0]
            vol = math.pi * r * r * (4 * r / 3 + h)
            ms = 4 * r / (4 * r + 3 * h); mc = 1 - ms
            ixx = mc * (3 * r * r + h * h) / 12 + 2 * ms * r * r / 5 + ms * h * (3 * r + 2 * h) / 8
            izz = mc * r * r / 2 + 2 * ms * r * r / 5
            unit = np.array([ixx, ixx, izz])
        elif t == CYLINDER:
            h = 2 * s[1]; r = s[0]
            vol = math.pi * r * r * h
            unit = np.array([(3 * r * r + h * h) / 12, (3 * r * r + h * h) / 12, r * r / 2])
        elif t == BOX:
            vol = 8 * s[0] * s[1] * s[2]
            unit = np.array([(s[1] ** 2 + s[2] ** 2) / 3, (s[0] ** 2 + s[2] ** 2) / 3, (s[0] ** 2 + s[1] ** 2) / 3])
        elif t == MESH:            # box of the vertex cloud's half extents
            vol = 8 * s[0] * s[1] * s[2]
            unit = np.array([(s[1] ** 2 + s[2] ** 2) / 3, (s[0] ** 2 + s[2] ** 2) / 3, (s[0] ** 2 + s[1] ** 2) / 3])
        elif t == ELLIPSOID:
            vol = 4.0 / 3.0 * math.pi * s[0] * s[1] * s[2]
            unit = np.array([(s[1] ** 2 + s[2] ** 2) / 5, (s[0] ** 2 + s[2] ** 2) / 5, (s[0] ** 2 + s[1] ** 2) / 5])
        else:
            vol = 0.0; unit = np.zeros(3)
        mass = g.mass if g.mass is not None else g.density * vol
        return mass, unit * mass

    @staticmethod
    def _rbound(g: _Geom):
        s = g.size
        return {PLANE: 0.0, SPHERE: s[0], CAPSULE: s[0] + s[1], CYLINDER: math.hypot(s[0], s[1]),
                BOX: float(np.linalg.norm(s)), ELLIPSOID: float(max(s)),
                MESH: float(np.linalg.norm(getattr(g, "mesh", np.zeros((1, 3))), axis=1).max()) if getattr(g, "mesh", None) is not None else 0.0,
                HFIELD: (math.sqrt(g.hfield["size"][0] ** 2 + g.hfield["size"][1] ** 2 + max(g.hfield["size"][2], g.hfield["size"][3]) ** 2)
                         if getattr(g, "hfield", None) is not None else 0.0)}.get(g.type, 0.0)

    # ---- compile
    def compile(self):
        nb = len(self.bodies)
        # joints must be ordered by body (MuJoCo orders joints in body order)
        order = [j for b in self.bodies for j in b.joints]
        remap = {old: new for new, old in enumerate(order)}
        joints = [self.joints[j] for j in order]
        for b in self.bodies:
            b.joints = [remap[j] for j in b.joints]
        for a in self.actuators:
            if a.get("site") is not None:             # mjTRN_SITE: the 6-vector gear is a wrench in the site frame
                a["trntype"] = 4; a["trnid"] = int(a["site"])
                continue
            if a.get("tendon") is not None:
                a["trntype"] = 3
                a["trnid"] = [t["name"] for t in self.tendons].index(a["tendon"]) if isinstance(a["tendon"], str) else int(a["tendon"])
                continue
            if isinstance(a["joint"], str):
                a["joint"] = [j.name for j in joints].index(a["joint"])
            else:
                a["joint"] = remap[a["joint"]]
            a["trntype"] = 0; a["trnid"] = a["joint"]
        self.joints = joints
        nj = len(joints)
        M = {}
        # ---- bodies
        parent = np.array([b.parent for b in self.bodies], np.int32)
        M["body_parentid"] = parent
        rootid = np.zeros(nb, np.int32); weldid = np.zeros(nb, np.int32)
        for i in range(1, nb):
            rootid[i] = i if parent[i] == 0 else rootid[parent[i]]
            weldid[i] = i if self.bodies[i].joints else weldid[parent[i]]
        M["body_rootid"] = rootid; M["body_weldid"] = weldid
        mocapid = -np.ones(nb, np.int32); nmocap = 0
        for i, b in enumerate(self.bodies):
            if b.mocap:
                mocapid[i] = nmocap; nmocap += 1
        M["body_mocapid"] = mocapid
        M["body_pos"] = np.array([b.pos for b in self.bodies]); M["body_quat"] = np.array([b.quat for b in self.bodies])
        # joints / dofs
        qadr = 0; dadr = 0
        jnt_qposadr = np.zeros(nj, np.int32); jnt_dofadr = np.zeros(nj, np.int32)
        body_jntnum = np.zeros(nb, np.int32); body_jntadr = -np.ones(nb, np.int32)
        body_dofnum = np.zeros(nb, np.int32); body_dofadr = -np.ones(nb, np.int32)
        dof_bodyid = []; dof_jntid = []; dof_parentid = []
        qpos0 = []; qpos_spring = []
        last_dof_of_body = -np.ones(nb, np.int32)
        for bi, b in enumerate(self.bodies):
            body_jntnum[bi] = len(b.joints)
            if b.joints:
                body_jntadr[bi] = b.joints[0]; body_dofadr[bi] = dadr
            # nearest ancestor dof
            anc = parent[bi] if bi > 0 else 0
            prev = -1
            a = anc
            while bi > 0 and True:
                if last_dof_of_body[a] >= 0:
                    prev = last_dof_of_body[a]; break
                if a == 0:
                    break
                a = parent[a]
            for j in b.joints:
                jt = joints[j]
                jnt_qposadr[j] = qadr; jnt_dofadr[j] = dadr
                nqj, nvj = {FREE: (7, 6), BALL: (4, 3), SLIDE: (1, 1), HINGE: (1, 1)}[jt.type]
                if jt.type == FREE:
                    qpos0 += list(b.pos) + list(b.quat); qpos_spring += list(b.pos) + list(b.quat)
                elif jt.type == BALL:
                    qpos0 += [1, 0, 0, 0]; qpos_spring += [1, 0, 0, 0]
                else:
                    qpos0.append(jt.ref); qpos_spring.append(jt.springref)
                for k in range(nvj):
                    dof_bodyid.append(bi); dof_jntid.append(j); dof_parentid.append(prev)
                    prev = dadr + k
                qadr += nqj; dadr += nvj
            if b.joints:
                body_dofnum[bi] = dadr - body_dofadr[bi]
                last_dof_of_body[bi] = dadr - 1
        nq, nv = qadr, dadr
        M.update(body_jntnum=body_jntnum, body_jntadr=body_jntadr, body_dofnum=body_dofnum, body_dofadr=body_dofadr)
        M["jnt_type"] = np.array([j.type for j in joints], np.int32)
        M["jnt_qposadr"] = jnt_qposadr; M["jnt_dofadr"] = jnt_dofadr
        M["jnt_bodyid"] = np.array([j.body for j in joints], np.int32)
        M["jnt_limited"] = np.array([int(j.limited) for j in joints], np.int32)
        M["jnt_pos"] = np.array([j.pos for j in joints]).reshape(nj, 3)
        M["jnt_axis"] = np.array([j.axis for j in joints]).reshape(nj, 3)
        M["jnt_stiffness"] = np.array([j.stiffness for j in joints], float)
        M["jnt_range"] = np.array([j.range for j in joints], float).reshape(nj, 2)
        M["jnt_margin"] = np.array([j.margin for j in joints], float)
        M["jnt_solref"] = np.array([j.solreflimit for j in joints], float).reshape(nj, 2)
        M["jnt_solimp"] = np.array([j.solimplimit for j in joints], float).reshape(nj, 5)
        M["qpos0"] = np.array(qpos0, float); M["qpos_spring"] = np.array(qpos_spring, float)
        M["dof_bodyid"] = np.array(dof_bodyid, np.int32); M["dof_jntid"] = np.array(dof_jntid, np.int32)
        M["dof_parentid"] = np.array(dof_parentid, np.int32)
        dj = [joints[j] for j in dof_jntid]
        M["dof_armature"] = np.array([j.armature for j in dj], float)
        M["dof_damping"] = np.array([j.damping for j in dj], float)
        M["dof_frictionloss"] = np.array([j.frictionloss for j in dj], float)
        M["dof_solref"] = np.array([j.solreffriction for j in dj], float).reshape(nv, 2)
        M["dof_solimp"] = np.array([j.solimpfriction for j in dj], float).reshape(nv, 5)
        # ---- geoms
        ng = len(self.geoms)
        G = self.geoms
        M["geom_type"] = np.array([g.type for g in G], np.int32)
        M["geom_contype"] = np.array([g.contype for g in G], np.int32)
        M["geom_conaffinity"] = np.array([g.conaffinity for g in G], np.int32)
        M["geom_condim"] = np.array([g.condim for g in G], np.int32)
        M["geom_bodyid"] = np.array([g.body for g in G], np.int32)
        M["geom_group"] = np.array([g.group for g in G], np.int32)
        M["geom_priority"] = np.array([g.priority for g in G], np.int32)
        M["geom_size"] = np.array([g.size for g in G], float).reshape(ng, 3)
        M["geom_pos"] = np.array([g.pos for g in G], float).reshape(ng, 3)
        M["geom_quat"] = np.array([g.quat for g in G], float).reshape(ng, 4)
        M["geom_friction"] = np.array([g.friction for g in G], float).reshape(ng, 3)
        M["geom_solmix"] = np.array([g.solmix for g in G], float)
        M["geom_solref"] = np.array([g.solref for g in G], float).reshape(ng, 2)
        M["geom_solimp"] = np.array([g.solimp for g in G], float).reshape(ng, 5)
        M["geom_margin"] = np.array([g.margin for g in G], float)
        M["geom_gap"] = np.array([g.gap for g in G], float)
        M["geom_rbound"] = np.array([self._rbound(g) for g in G], float)
        # convex meshes: vertex pools (mjModel mesh_vertadr / mesh_vertnum / mesh_vert, geom_dataid)
        dataid, vadr, vnum, verts = [], [], [], []
        for g in G:
            if g.type == MESH and getattr(g, "mesh", None) is not None:
                dataid.append(len(vadr)); vadr.append(sum(vnum)); vnum.append(len(g.mesh)); verts.append(g.mesh)
            else:
                dataid.append(-1)
        hn, hc, ha, hs, hd = [], [], [], [], []
        for gi, g in enumerate(G):
            if g.type == HFIELD and getattr(g, "hfield", None) is not None:
                dataid[gi] = len(hn); hn.append(g.hfield["data"].shape[0]); hc.append(g.hfield["data"].shape[1]); ha.append(sum(len(x) for x in hd))
                hs.append(g.hfield["size"]); hd.append(g.hfield["data"].ravel())
        M["hfield_nrow"] = np.array(hn, np.int32); M["hfield_ncol"] = np.array(hc, np.int32); M["hfield_adr"] = np.array(ha, np.int32)
        M["hfield_size"] = np.array(hs, float).reshape(-1, 4) if hs else np.zeros((0, 4))
        M["hfield_data"] = np.concatenate(hd) if hd else np.zeros(0)
        M["geom_dataid"] = np.array(dataid, np.int32)
        M["mesh_vertadr"] = np.array(vadr, np.int32); M["mesh_vertnum"] = np.array(vnum, np.int32)
        M["mesh_vert"] = np.concatenate(verts).reshape(-1, 3) if verts else np.zeros((0, 3))
        # ---- inertial frames
        body_mass = np.zeros(nb); body_ipos = np.zeros((nb, 3)); body_iquat = np.tile([1.0, 0, 0, 0], (nb, 1))
        body_inertia = np.zeros((nb, 3))
        for bi, b in enumerate(self.bodies):
            if bi == 0:
                continue
            if b.inertial is not None:
                ine = b.inertial
                body_mass[bi] = ine["mass"]; body_ipos[bi] = np.array(ine.get("pos", (0, 0, 0)), float)
                if "fullinertia" in ine:
                    f = ine["fullinertia"]
                    I = np.array([[f[0], f[3], f[4]], [f[3], f[1], f[5]], [f[4], f[5], f[2]]], float)
                    body_inertia[bi], body_iquat[bi] = self._diagonalize(I)
                else:
                    body_inertia[bi] = np.array(ine["diaginertia"], float)
                    body_iquat[bi] = normq(ine.get("quat", (1, 0, 0, 0)))
                continue
            masses = []; coms = []; tensors = []
            for gi in b.geoms:
                g = G[gi]
                m_, diag = self._geom_volume_inertia(g)
                if m_ <= 0:
                    continue
                R = quat2mat(g.quat)
                masses.append(m_); coms.append(g.pos); tensors.append(R @ np.diag(diag) @ R.T)
            if not masses:
                continue
            mt = sum(masses)
            com = sum(m_ * c for m_, c in zip(masses, coms)) / mt
            I = np.zeros((3, 3))
            for m_, c, T in zip(masses, coms, tensors):
                dlt = c - com
                I += T + m_ * (dlt @ dlt * np.eye(3) - np.outer(dlt, dlt))
            body_mass[bi] = mt; body_ipos[bi] = com
            body_inertia[bi], body_iquat[bi] = self._diagonalize(I)
        M.update(body_mass=body_mass, body_ipos=body_ipos, body_iquat=body_iquat, body_inertia=body_inertia)
        sub = body_mass.copy()
        for i in range(nb - 1, 0, -1):
            sub[parent[i]] += sub[i]
        M["body_subtreemass"] = sub
        M["body_gravcomp"] = np.array([self.gravcomp.get(i, 0.0) for i in range(nb)], float)
        # ---- sites, actuators, keys, excludes
        ns = len(self.sites)
        M["site_bodyid"] = np.array([s[1] for s in self.sites], np.int32)
        M["site_pos"] = np.array([s[2] for s in self.sites], float).reshape(ns, 3)
        M["site_quat"] = np.array([s[3] for s in self.sites], float).reshape(ns, 4)
        nu = len(self.actuators); A = self.actuators
        M["actuator_trntype"] = np.array([a["trntype"] for a in A], np.int32)
        M["actuator_trnid"] = np.array([a["trnid"] for a in A], np.int32)
        M["actuator_ctrllimited"] = np.array([int(a["ctrllimited"]) for a in A], np.int32)
        M["actuator_forcelimited"] = np.array([int(a["forcelimited"]) for a in A], np.int32)
        M["actuator_biastype"] = np.array([a["biastype"] for a in A], np.int32)
        M["actuator_gainprm"] = np.array([a["gainprm"] for a in A], float).reshape(nu, 3)
        M["actuator_biasprm"] = np.array([a["biasprm"] for a in A], float).reshape(nu, 3)
        M["actuator_gear"] = np.array([a["gear"] for a in A], float)
        M["actuator_ctrlrange"] = np.array([a["ctrlrange"] for a in A], float).reshape(nu, 2)
        M["actuator_forcerange"] = np.array([a["forcerange"] for a in A], float).reshape(nu, 2)
        M["actuator_refsite"] = np.array([-1 if a.get("refsite") is None else int(a["refsite"]) for a in A], np.int32)
        M["actuator_gear6"] = np.array([a["gear6"] if a.get("gear6") is not None else (a["gear"], 0, 0, 0, 0, 0) for a in A], float).reshape(nu, 6)
        M["jnt_actfrclimited"] = np.array([int(j.name in self.actfrc) for j in joints], np.int32)
        M["jnt_actfrcrange"] = np.array([self.actfrc.get(j.name, (0.0, 0.0)) for j in joints], float).reshape(len(joints), 2)
        M["actuator_dyntype"] = np.array([a.get("dyntype", 0) for a in A], np.int32)
        adr = []; na = 0
        for a in A:
            adr.append(na if a.get("dyntype", 0) else -1); na += 1 if a.get("dyntype", 0) else 0
        M["actuator_actadr"] = np.array(adr, np.int32)
        M["actuator_actlimited"] = np.array([int(a.get("actlimited", False)) for a in A], np.int32)
        M["actuator_dynprm"] = np.array([a.get("dynprm", 1.0) for a in A], float)
        M["actuator_actrange"] = np.array([a.get("actrange", (0, 0)) for a in A], float).reshape(nu, 2)
        nkey = len(self.keys)
        kq = np.tile(M["qpos0"], (max(nkey, 1), 1))
        for k, (_, qp) in enumerate(self.keys):
            kq[k, :len(qp)] = qp
        M["key_qpos"] = kq[:nkey].reshape(nkey, nq) if nkey else np.zeros((0, nq))
        M["exclude_signature"] = np.array([(min(a, b) << 16) + max(a, b) for a, b in self.excludes], np.int32)
        # fixed tendons
        jnames = [j.name for j in joints]
        T = self.tendons
        wrap_objid = []; wrap_prm = []; tadr = []; tnum = []
        for t in T:
            tadr.append(len(wrap_objid)); tnum.append(len(t["joints"]))
            for jn, cf in zip(t["joints"], t["coefs"]):
                wrap_objid.append(jnames.index(jn) if isinstance(jn, str) else remap[jn]); wrap_prm.append(cf)
        M["tendon_adr"] = np.array(tadr, np.int32); M["tendon_num"] = np.array(tnum, np.int32)
        M["tendon_limited"] = np.array([int(t["limited"]) for t in T], np.int32)
        M["wrap_objid"] = np.array(wrap_objid, np.int32); M["wrap_prm"] = np.array(wrap_prm, float)
        M["tendon_range"] = np.array([t["range"] for t in T], float).reshape(len(T), 2)
        M["tendon_margin"] = np.array([t["margin"] for t in T], float)
        M["tendon_solref_lim"] = np.array([t["solref"] for t in T], float).reshape(len(T), 2)
        M["tendon_solimp_lim"] = np.array([t["solimp"] for t in T], float).reshape(len(T), 5)
        M["tendon_stiffness"] = np.array([t.get("stiffness", 0.0) for t in T], float)
        M["tendon_damping"] = np.array([t.get("damping", 0.0) for t in T], float)
        M["tendon_frictionloss"] = np.array([t.get("frictionloss", 0.0) for t in T], float)
        M["tendon_solref_fri"] = np.array([t.get("solref_fri", DEF_SOLREF) for t in T], float).reshape(len(T), 2)
        M["tendon_solimp_fri"] = np.array([t.get("solimp_fri", DEF_SOLIMP) for t in T], float).reshape(len(T), 5)
        ls = np.zeros((len(T), 2))
        for ti, t in enumerate(T):
            sl = t.get("springlength")
            if sl is None:            # resting length = the tendon's length at qpos0 (mjModel set0)
                l0 = sum(cf * M["qpos0"][jnt_qposadr[wrap_objid[tadr[ti] + k]]] for k, cf in enumerate(t["coefs"]))
                ls[ti] = (l0, l0)
            else:
                ls[ti] = (sl, sl) if np.isscalar(sl) else tuple(sl)
        M["tendon_lengthspring"] = ls
        sizes = dict(nq=nq, nv=nv, nu=nu, na=na, nbody=nb, njnt=nj, ngeom=ng, nsite=ns, nmocap=nmocap,
                     nuserdata=self.nuserdata, nkey=nkey, nexclude=len(self.excludes), ntendon=len(self.tendons),
                     nwrap=len(wrap_objid), nmesh=len(M["mesh_vertadr"]), nmeshvert=len(M["mesh_vert"]), nhfield=len(M["hfield_nrow"]), nhfielddata=len(M["hfield_data"]))
        M.update(sizes)
        o = self.opt
        M.update(timestep=o["timestep"], gravity=o["gravity"], impratio=o["impratio"], tolerance=o["tolerance"],
                 ls_tolerance=o["ls_tolerance"], cone=o["cone"], iterations=o["iterations"],
                 ls_iterations=o["ls_iterations"], disableflags=(0 if o["contact"] else (1 << 4)) | int(getattr(self, "disableflags", 0)),
                 enableflags=0, solver=2, integrator=int(getattr(self, "integrator", 0)), noslip_iterations=int(getattr(self, "noslip_iterations", 0)),
                 noslip_tolerance=float(getattr(self, "noslip_tolerance", 1e-6)), neq=0, unsupported=0,
                 nconmax=self.nconmax, nefcmax=self.nefcmax, density=self.fluid[0], viscosity=self.fluid[1], wind=np.array(self.fluid[2]))
        # ---- quantities evaluated at qpos0 (mjModel "set0")
        Mq, Jp, Jr = mass_matrix(M, M["qpos0"])
        Minv = np.linalg.inv(Mq) if nv else np.zeros((0, 0))
        binv = np.zeros((nb, 2))
        for bi in range(1, nb):
            if nv == 0:
                break
            At = Jp[bi] @ Minv @ Jp[bi].T; Ar = Jr[bi] @ Minv @ Jr[bi].T
            binv[bi] = [np.trace(At) / 3, np.trace(Ar) / 3]
        dinv = np.zeros(nv)
        for j, jt in enumerate(joints):
            da = jnt_dofadr[j]
            if jt.type == FREE:
                dinv[da:da + 3] = np.mean(np.diag(Minv)[da:da + 3]); dinv[da + 3:da + 6] = np.mean(np.diag(Minv)[da + 3:da + 6])
            elif jt.type == BALL:
                dinv[da:da + 3] = np.mean(np.diag(Minv)[da:da + 3])
            else:
                dinv[da] = Minv[da, da]
        M["body_invweight0"] = binv; M["dof_invweight0"] = dinv
        tinv = np.zeros(len(T))
        for ti, t in enumerate(T):
            Jt = np.zeros(nv)
            for k in range(tnum[ti]):
                Jt[jnt_dofadr[wrap_objid[tadr[ti] + k]]] = wrap_prm[tadr[ti] + k]
            tinv[ti] = Jt @ Minv @ Jt
        M["tendon_invweight0"] = tinv
        # ---- equality constraints (the second anchor of a connect is where the first one sits at qpos0)
        E = self.equalities
        M["neq"] = len(E)
        M["eq_type"] = np.array([e["type"] for e in E], np.int32); M["eq_active0"] = np.array([int(e["active"]) for e in E], np.int32)
        o1, o2 = [], []; data = np.zeros((len(E), 11))
        xpos0, xquat0, xmat0, _, _ = kinematics(M, M["qpos0"])
        for k, e in enumerate(E):
            if e["type"] == 0:
                b1, b2 = e["obj1"], e["obj2"]
                o1.append(b1); o2.append(b2)
                world = xpos0[b1] + xmat0[b1] @ np.array(e["anchor"], float)
                data[k, :3] = e["anchor"]; data[k, 3:6] = xmat0[b2].T @ (world - xpos0[b2])
            elif e["type"] == 1:
                # [anchor in body2's frame, the same point in body1's frame, quat of body2 in body1's frame, torquescale]
                b1, b2 = e["obj1"], e["obj2"]
                o1.append(b1); o2.append(b2)
                data[k, :3] = e["anchor"]; data[k, 10] = e["torquescale"]
                rp = e["relpose"]
                if rp is None or not np.any(np.asarray(rp[3:7], float)):
                    world = xpos0[b2] + xmat0[b2] @ np.array(e["anchor"], float)
                    data[k, 3:6] = xmat0[b1].T @ (world - xpos0[b1])
                    data[k, 6:10] = quat_mul(xquat0[b1] * np.array([1.0, -1, -1, -1]), xquat0[b2])
                else:
                    data[k, 3:6] = rp[:3]; data[k, 6:10] = normq(np.asarray(rp[3:7], float))
            elif e["type"] == 3:
                tn = [t["name"] for t in self.tendons]
                o1.append(tn.index(e["obj1"])); o2.append(-1 if e["obj2"] is None else tn.index(e["obj2"]))
                data[k, :5] = e["polycoef"]
            else:
                o1.append(jnames.index(e["obj1"])); o2.append(-1 if e["obj2"] is None else jnames.index(e["obj2"]))
                data[k, :5] = e["polycoef"]
        M["eq_obj1id"] = np.array(o1, np.int32); M["eq_obj2id"] = np.array(o2, np.int32); M["eq_data"] = data
        M["eq_solref"] = np.array([e["solref"] for e in E], float).reshape(len(E), 2)
        M["eq_solimp"] = np.array([e["solimp"] for e in E], float).reshape(len(E), 5)
        km = self.key_mpos if self.key_mpos is not None else np.zeros((nkey, 3 * nmocap))
        M["key_mpos"] = np.asarray(km, float).reshape(nkey, 3 * nmocap) if nkey else np.zeros((0, 3 * nmocap))
        M["meaninertia"] = max(float(np.mean(np.diag(Mq))) if nv else 1.0, MINVAL)
        # names
        M["names"] = dict(
            body={b.name: i for i, b in enumerate(self.bodies)},
            joint={j.name: i for i, j in enumerate(joints)},
            geom={g.name: i for i, g in enumerate(G) if g.name},
            site={s[0]: i for i, s in enumerate(self.sites)},
            actuator={a["name"]: i for i, a in enumerate(A)},
            key={k[0]: i for i, k in enumerate(self.keys)})
        return M

    @staticmethod
    def _diagonalize(I):
        w, V = np.linalg.eigh(I)
        idx = np.argsort(-w)              # decreasing, like mju_eig3
        w = w[idx]; V = V[:, idx]
        if np.linalg.det(V) < 0:
            V[:, 2] = -V[:, 2]
        return w, mat2quat(V)


# ---------------------------------------------------------------------------- numpy kinematics (qpos0 quantities + test cross-check)
def kinematics(M, qpos):
    nb = M["nbody"]
    xpos = np.zeros((nb, 3)); xquat = np.tile([1.0, 0, 0, 0], (nb, 1)); xmat = np.tile(np.eye(3), (nb, 1, 1))
    nj = M["njnt"]
    xanchor = np.zeros((nj, 3)); xaxis = np.zeros((nj, 3))
    for i in range(1, nb):
        p = M["body_parentid"][i]
        jn, ja = M["body_jntnum"][i], M["body_jntadr"][i]
        if jn == 1 and M["jnt_type"][ja] == FREE:
            qa = M["jnt_qposadr"][ja]
            pos = qpos[qa:qa + 3].copy(); quat = normq(qpos[qa + 3:qa + 7])
            xanchor[ja] = pos; xaxis[ja] = M["jnt_axis"][ja]
        else:
            pos = xpos[p] + xmat[p] @ M["body_pos"][i]
            quat = quat_mul(xquat[p], M["body_quat"][i])
            for j in range(ja, ja + jn):
                qa = M["jnt_qposadr"][j]
                R = quat2mat(quat)
                xaxis[j] = R @ M["jnt_axis"][j]
                xanchor[j] = R @ M["jnt_pos"][j] + pos
                t = M["jnt_type"][j]
                if t == SLIDE:
                    pos = pos + xaxis[j] * (qpos[qa] - M["qpos0"][qa])
                else:
                    qloc = normq(qpos[qa:qa + 4]) if t == BALL else axisangle2quat(M["jnt_axis"][j], qpos[qa] - M["qpos0"][qa])
                    quat = quat_mul(quat, qloc)
                    pos = xanchor[j] - quat2mat(quat) @ M["jnt_pos"][j]
        quat = normq(quat)
        xpos[i] = pos; xquat[i] = quat; xmat[i] = quat2mat(quat)
    return xpos, xquat, xmat, xanchor, xaxis


def mass_matrix(M, qpos):
    """Dense joint-space inertia by summing J^T I J over bodies; also per-body COM Jacobians."""
    nb, nv = M["nbody"], M["nv"]
    xpos, xquat, xmat, xanchor, xaxis = kinematics(M, qpos)
    Jp = np.zeros((nb, 3, nv)); Jr = np.zeros((nb, 3, nv))
    Mq = np.diag(M["dof_armature"].astype(float)) if nv else np.zeros((0, 0))
    for b in range(1, nb):
        com = xpos[b] + xmat[b] @ M["body_ipos"][b]
        # chain of dofs
        a = b
        while a > 0:
            for j in range(M["body_jntadr"][a], M["body_jntadr"][a] + M["body_jntnum"][a]):
                t = M["jnt_type"][j]; da = M["jnt_dofadr"][j]
                if t == FREE:
                    for k in range(3):
                        Jp[b, k, da + k] = 1.0
                    for k in range(3):
                        ax = xmat[a][:, k]
                        Jr[b, :, da + 3 + k] = ax; Jp[b, :, da + 3 + k] = np.cross(ax, com - xpos[a])
                elif t == BALL:
                    for k in range(3):
                        ax = xmat[a][:, k]
                        Jr[b, :, da + k] = ax; Jp[b, :, da + k] = np.cross(ax, com - xanchor[j])
                elif t == SLIDE:
                    Jp[b, :, da] = xaxis[j]
                else:
                    Jr[b, :, da] = xaxis[j]; Jp[b, :, da] = np.cross(xaxis[j], com - xanchor[j])
            a = M["body_parentid"][a]
        if nv:
            R = xmat[b] @ quat2mat(M["body_iquat"][b])
            Iw = R @ np.diag(M["body_inertia"][b]) @ R.T
            Mq = Mq + M["body_mass"][b] * Jp[b].T @ Jp[b] + Jr[b].T @ Iw @ Jr[b]
    return Mq, Jp, Jr
