"""MJCF subset -> ModelBuilder (and from there the MjpcHipModel arrays).

Covers what the task files of the reference that are complete without MuJoCo's model zoo use
(mjpc/test/testdata/particle*.xml, mjpc/tasks/humanoid/humanoid.xml.patch + task.xml files): <include>, <compiler angle>,
<option> (+ <flag contact>), nested <default> classes with `childclass` / `class`, the body tree with <joint>/<freejoint>/
<geom>/<site>/<inertial>, <actuator> motor / position / velocity / intvelocity / general (incl. filter dynamics; joint or site transmission with an optional reference site), fixed <tendon>s, <equality> connect / weld / joint / tendon, <contact><exclude>, <keyframe>,
<custom><numeric>, and the MJPC cost table in <sensor><user> (mjpc/task.cc:203-238).  Anything visual is ignored.

MuJoCo semantics restated here (compile-time only, no simulation): defaults inherit along the class tree and apply per
element type; `childclass` sets the default class of a body's subtree; `fromto` on capsule/cylinder/box geoms; angles in
degrees unless <compiler angle="radian">; `limited`/`ctrllimited`/`forcelimited` "auto" follow the presence of a range
(compiler autolimits, MuJoCo >= 2.3).  The loader is used by the CPU tests to cross-check the hand-authored generators in
tasks.py against the reference's own model files; the product never needs it at run time.
"""
from __future__ import annotations

import math
import os
import re
import xml.etree.ElementTree as ET

import numpy as np

from .builder import BALL, BOX, CAPSULE, CYLINDER, FREE, HINGE, PLANE, SLIDE, SPHERE, ModelBuilder

_GEOM_TYPES = {"plane": PLANE, "sphere": SPHERE, "capsule": CAPSULE, "cylinder": CYLINDER, "box": BOX}
_JNT_TYPES = {"free": FREE, "ball": BALL, "slide": SLIDE, "hinge": HINGE}
_ELEMENT_DEFAULT_KINDS = ("joint", "geom", "site", "motor", "position", "general", "velocity", "intvelocity", "tendon")


def _floats(s):
    return [float(x) for x in s.split()]


class _Defaults:
    """class name -> {element kind -> attribute dict}, inheritance resolved at parse time"""

    def __init__(self):
        self.classes = {"main": {k: {} for k in _ELEMENT_DEFAULT_KINDS}}

    def parse(self, node, parent="main"):
        name = node.get("class", "main") if parent is not None else "main"
        if name not in self.classes:
            self.classes[name] = {k: dict(v) for k, v in self.classes[parent].items()}
        cur = self.classes[name]
        for ch in node:
            if ch.tag == "default":
                continue
            kind = "general" if ch.tag == "general" else ch.tag
            if kind in cur:
                cur[kind].update(ch.attrib)
        for ch in node:
            if ch.tag == "default":
                self.parse(ch, name)

    def get(self, cls, kind):
        return self.classes.get(cls or "main", self.classes["main"]).get(kind, {})


def _clean(text):
    """drop backslash-escaped comment markers (`<!-\\- ... -\\->`, as tracking/task.xml:143-152 carries them)"""
    return re.sub(r"<!-\\-.*?-\\->", "", text, flags=re.S)


def _expand_includes(root, base_dir, reader):
    """in-place <include file=...> expansion (the included file's <mujoco> children are spliced in)"""
    changed = True
    while changed:
        changed = False
        for parent in root.iter():
            for i, ch in enumerate(list(parent)):
                if ch.tag == "include":
                    sub = ET.fromstring(reader(os.path.join(base_dir, ch.get("file"))))
                    parent.remove(ch)
                    for k, g in enumerate(list(sub)):
                        parent.insert(i + k, g)
                    changed = True
                    break
            if changed:
                break


def parse_mjcf(path_or_text, base_dir=None, reader=None, missing_ok=()):
    """Returns (ModelBuilder, info) with info = dict(numeric={name: [floats]}, cost_terms=[(dim, norm, weight, params)],
    keys=[dict(name, qpos, qvel, mpos, ctrl)], names of trace sensors ...).  `reader(path) -> text` lets the caller supply files
    that exist only as patches; includes listed in `missing_ok` (basenames) are dropped."""
    def default_reader(p):
        with open(p) as f:
            return f.read()
    reader = reader or default_reader
    if os.path.exists(path_or_text):
        base_dir = base_dir or os.path.dirname(os.path.abspath(path_or_text))
        text = reader(path_or_text)
    else:
        text = path_or_text
    root = ET.fromstring(_clean(text))

    def guarded_reader(p):
        if os.path.basename(p) in missing_ok:
            return "<mujoco/>"
        return _clean(reader(p))
    _expand_includes(root, base_dir or ".", guarded_reader)

    degree = True
    for c in root.iter("compiler"):
        if c.get("angle") == "radian":
            degree = False
    ang = math.pi / 180.0 if degree else 1.0

    opt = dict(timestep=0.002, cone=0, impratio=1.0, contact=True, tolerance=1e-8, iterations=100, ls_iterations=50, ls_tolerance=0.01,
               gravity=(0, 0, -9.81))
    noslip = {}
    for o in root.iter("option"):
        for k in ("timestep", "impratio", "tolerance", "ls_tolerance"):
            if o.get(k) is not None:
                opt[k] = float(o.get(k))
        for k in ("iterations", "ls_iterations"):
            if o.get(k) is not None:
                opt[k] = int(o.get(k))
        if o.get("noslip_iterations") is not None:
            noslip["iterations"] = int(o.get("noslip_iterations"))
        if o.get("noslip_tolerance") is not None:
            noslip["tolerance"] = float(o.get("noslip_tolerance"))
        if o.get("cone") is not None:
            opt["cone"] = 1 if o.get("cone") == "elliptic" else 0
        if o.get("gravity") is not None:
            opt["gravity"] = tuple(_floats(o.get("gravity")))
        for k in ("density", "viscosity"):          # inertia-box fluid model
            if o.get(k) is not None:
                opt[k] = float(o.get(k))
        if o.get("wind") is not None:
            opt["wind"] = tuple(_floats(o.get("wind")))
        if o.get("integrator") is not None:
            integ = o.get("integrator")
            if integ not in ("Euler", "implicit", "implicitfast"):
                raise ValueError(f"integrator {integ} not in the supported subset (Euler, implicit, implicitfast)")
            opt["integrator"] = {"Euler": 0, "implicit": 2, "implicitfast": 3}[integ]
        for fl in o.iter("flag"):
            if fl.get("contact") == "disable":
                opt["contact"] = False
    b = ModelBuilder(**opt)
    if "iterations" in noslip:
        b.noslip_iterations = noslip["iterations"]
    if "tolerance" in noslip:
        b.noslip_tolerance = noslip["tolerance"]

    defaults = _Defaults()
    for dnode in root.findall("default"):
        defaults.parse(dnode, None)

    def attrs(node, kind, childclass):
        cls = node.get("class") or childclass
        a = dict(defaults.get(cls, kind))
        a.update(node.attrib)
        return a

    def quat_of(a):
        if "quat" in a:
            return tuple(_floats(a["quat"]))
        return (1, 0, 0, 0)

    info = dict(numeric={}, text={}, cost_terms=[], traces=[], keys=[], sensors=[])

    def add_body(node, parent, childclass):
        for ch in node:
            if ch.tag == "body":
                cc = ch.get("childclass") or childclass
                inertial = None
                ine = ch.find("inertial")
                if ine is not None:
                    inertial = dict(pos=tuple(_floats(ine.get("pos", "0 0 0"))), quat=tuple(_floats(ine.get("quat", "1 0 0 0"))),
                                    mass=float(ine.get("mass")), diaginertia=tuple(_floats(ine.get("diaginertia", "0 0 0"))))
                bid = b.body(ch.get("name", f"body{len(b.bodies)}"), parent, pos=tuple(_floats(ch.get("pos", "0 0 0"))), quat=quat_of(ch.attrib),
                             mocap=ch.get("mocap") == "true", inertial=inertial, gravcomp=float(ch.get("gravcomp", 0)))
                add_body(ch, bid, cc)
            elif ch.tag == "freejoint":
                b.joint(parent, ch.get("name", f"joint{len(b.joints)}"), FREE)
            elif ch.tag == "joint":
                a = attrs(ch, "joint", childclass)
                jt = _JNT_TYPES[a.get("type", "hinge")]
                rng = _floats(a["range"]) if "range" in a else [0.0, 0.0]
                limited = a.get("limited", "auto")
                lim = (limited == "true") or (limited == "auto" and "range" in a)
                scale = ang if jt in (HINGE, BALL) else 1.0
                kw = dict(type=jt, axis=tuple(_floats(a.get("axis", "0 0 1"))), pos=tuple(_floats(a.get("pos", "0 0 0"))), limited=lim,
                          range=(rng[0] * scale, rng[1] * scale), damping=float(a.get("damping", 0)), armature=float(a.get("armature", 0)),
                          frictionloss=float(a.get("frictionloss", 0)), stiffness=float(a.get("stiffness", 0)),
                          ref=float(a.get("ref", 0)) * scale, springref=float(a.get("springref", 0)) * scale, margin=float(a.get("margin", 0)))
                if "solreflimit" in a:
                    kw["solreflimit"] = tuple(_floats(a["solreflimit"]))
                if "solimplimit" in a:
                    v = _floats(a["solimplimit"]); kw["solimplimit"] = tuple(v + [0.9, 0.95, 0.001, 0.5, 2][len(v):])
                if "actuatorfrcrange" in a and a.get("actuatorfrclimited", "auto") in ("true", "auto"):
                    kw["actuatorfrcrange"] = tuple(_floats(a["actuatorfrcrange"]))
                b.joint(parent, a.get("name", f"joint{len(b.joints)}"), **kw)
            elif ch.tag == "geom":
                a = attrs(ch, "geom", childclass)
                gt = _GEOM_TYPES.get(a.get("type", "sphere"))
                if gt is None:
                    raise ValueError(f"geom type {a.get('type')} not in the supported subset")
                kw = dict(type=gt, size=tuple(_floats(a.get("size", "0 0 0"))), pos=tuple(_floats(a.get("pos", "0 0 0"))), quat=quat_of(a),
                          contype=int(a.get("contype", 1)), conaffinity=int(a.get("conaffinity", 1)), condim=int(a.get("condim", 3)),
                          priority=int(a.get("priority", 0)), margin=float(a.get("margin", 0)), gap=float(a.get("gap", 0)),
                          solmix=float(a.get("solmix", 1)), group=int(a.get("group", 0)), density=float(a.get("density", 1000)))
                if "fromto" in a:
                    kw["fromto"] = tuple(_floats(a["fromto"]))
                if "zaxis" in a:
                    kw["zaxis"] = tuple(_floats(a["zaxis"]))
                if "euler" in a:
                    kw["euler"] = tuple(x * ang for x in _floats(a["euler"]))
                if "mass" in a:
                    kw["mass"] = float(a["mass"])
                if "friction" in a:
                    v = _floats(a["friction"]); kw["friction"] = tuple(v + [1, 0.005, 0.0001][len(v):])
                if "solref" in a:
                    kw["solref"] = tuple(_floats(a["solref"]))
                if "solimp" in a:
                    v = _floats(a["solimp"]); kw["solimp"] = tuple(v + [0.9, 0.95, 0.001, 0.5, 2][len(v):])
                b.geom(parent, a.get("name", ""), **kw)
            elif ch.tag == "site":
                a = attrs(ch, "site", childclass)
                b.site(parent, a.get("name", f"site{len(b.sites)}"), pos=tuple(_floats(a.get("pos", "0 0 0"))), quat=quat_of(a))

    for wb in root.findall("worldbody"):
        add_body(wb, 0, None)

    for act in root.findall("actuator"):
        for ch in act:
            kind = ch.tag
            if kind not in ("motor", "position", "velocity", "intvelocity", "general"):
                raise ValueError(f"actuator <{kind}> not in the supported subset")
            cls = ch.get("class")
            a = dict(defaults.get(cls, kind)); a.update(ch.attrib)
            gear = _floats(a.get("gear", "1"))[0]
            kw = dict(gear=gear)
            if "site" in a:                      # site transmission: the whole 6-vector gear, an optional reference site
                g6 = _floats(a.get("gear", "1")); site_names = [sv[0] for sv in b.sites]
                kw.update(site=site_names.index(a["site"]), gear6=tuple(g6 + [0.0] * (6 - len(g6))))
                if "refsite" in a:
                    kw["refsite"] = site_names.index(a["refsite"])
            if "ctrlrange" in a:
                kw["ctrlrange"] = tuple(_floats(a["ctrlrange"]))
            cl = a.get("ctrllimited", "auto")
            kw["ctrllimited"] = (cl == "true") or (cl == "auto" and "ctrlrange" in a)
            if "forcerange" in a:
                kw["forcerange"] = tuple(_floats(a["forcerange"]))
            fl = a.get("forcelimited", "auto")
            kw["forcelimited"] = (fl == "true") or (fl == "auto" and "forcerange" in a)
            if kind == "velocity":               # MJCF <velocity kv>: gain kv, bias (0, 0, -kv)
                kvv = float(a.get("kv", 1)); kw.update(gainprm=(kvv, 0, 0), biastype=1, biasprm=(0, 0, -kvv))
            elif kind == "position":
                kp = float(a.get("kp", 1)); kv = float(a.get("kv", 0))
                kw.update(gainprm=(kp, 0, 0), biastype=1, biasprm=(0, -kp, -kv))
            elif kind == "intvelocity":          # MJCF <intvelocity kp kv actrange>: a position servo on the integral of the control
                kp = float(a.get("kp", 1)); kv = float(a.get("kv", 0))
                kw.update(gainprm=(kp, 0, 0), biastype=1, biasprm=(0, -kp, -kv), dyntype=1, actlimited=True, actrange=tuple(_floats(a["actrange"])))
            elif kind == "general":
                g = _floats(a.get("gainprm", "1")); kw["gainprm"] = tuple(g + [0, 0, 0][len(g):3])[:3]
                if a.get("biastype", "none") == "affine":
                    bp = _floats(a.get("biasprm", "0 0 0")); kw.update(biastype=1, biasprm=tuple(bp + [0, 0, 0][len(bp):3])[:3])
                dyn = a.get("dyntype", "none")
                if dyn not in ("none", "integrator", "filter", "filterexact"):
                    raise ValueError(f"actuator dyntype {dyn} not in the supported subset")
                if dyn != "none":
                    kw.update(dyntype=("none", "integrator", "filter", "filterexact").index(dyn), dynprm=_floats(a.get("dynprm", "1"))[0])
                    if "actrange" in a:
                        kw.update(actrange=tuple(_floats(a["actrange"])), actlimited=a.get("actlimited", "auto") in ("true", "auto"))
            b.actuator(a.get("name", f"actuator{len(b.actuators)}"), a.get("joint"), **kw)

    for tn in root.findall("tendon"):
        for fx in tn.findall("fixed"):
            a = dict(defaults.get(fx.get("class"), "tendon")); a.update(fx.attrib)
            joints = [j.get("joint") for j in fx.findall("joint")]; coefs = [float(j.get("coef")) for j in fx.findall("joint")]
            kw = {}
            if "range" in a:
                kw["range"] = tuple(_floats(a["range"]))
            lim = a.get("limited", "auto")
            kw["limited"] = (lim == "true") or (lim == "auto" and "range" in a)
            if "solreflimit" in a:
                kw["solreflimit"] = tuple(_floats(a["solreflimit"]))
            if "solimplimit" in a:
                v = _floats(a["solimplimit"]); kw["solimplimit"] = tuple(v + [0.9, 0.95, 0.001, 0.5, 2][len(v):])
            if "solreffriction" in a:
                kw["solreffriction"] = tuple(_floats(a["solreffriction"]))
            if "solimpfriction" in a:
                v = _floats(a["solimpfriction"]); kw["solimpfriction"] = tuple(v + [0.9, 0.95, 0.001, 0.5, 2][len(v):])
            for k in ("stiffness", "damping", "frictionloss", "margin"):
                if k in a:
                    kw[k] = float(a[k])
            if "springlength" in a:
                sl = _floats(a["springlength"])
                kw["springlength"] = None if sl[0] < 0 else (sl[0] if len(sl) == 1 else (sl[0], sl[1]))
            b.tendon(a.get("name", ""), joints, coefs, **kw)

    for eq in root.findall("equality"):
        for ch in eq:
            kw = {}
            if ch.get("solref") is not None:
                kw["solref"] = tuple(_floats(ch.get("solref")))
            if ch.get("solimp") is not None:
                v = _floats(ch.get("solimp")); kw["solimp"] = tuple(v + [0.9, 0.95, 0.001, 0.5, 2][len(v):])
            if ch.get("active") is not None:
                kw["active"] = ch.get("active") == "true"
            poly = _floats(ch.get("polycoef", "0 1 0 0 0")); poly = tuple(poly + [0.0] * (5 - len(poly)))
            if ch.tag == "connect":
                b.connect(b.body_id(ch.get("body1")), b.body_id(ch.get("body2")) if ch.get("body2") else 0, tuple(_floats(ch.get("anchor"))), **kw)
            elif ch.tag == "weld":
                rp = _floats(ch.get("relpose", "0 1 0 0 0 0 0"))
                b.weld(b.body_id(ch.get("body1")), b.body_id(ch.get("body2")) if ch.get("body2") else 0, tuple(_floats(ch.get("anchor", "0 0 0"))),
                       relpose=tuple(rp), torquescale=float(ch.get("torquescale", "1")), **kw)
            elif ch.tag == "joint":
                b.joint_equality(ch.get("joint1"), ch.get("joint2"), polycoef=poly, **kw)
            elif ch.tag == "tendon":
                b.tendon_equality(ch.get("tendon1"), ch.get("tendon2"), polycoef=poly, **kw)
            else:
                raise ValueError(f"equality <{ch.tag}> not in the supported subset (connect, weld, joint, tendon)")

    for ct in root.findall("contact"):
        for ex in ct.findall("exclude"):
            b.exclude(b.body_id(ex.get("body1")), b.body_id(ex.get("body2")))

    for cu in root.findall("custom"):
        for nu in cu.findall("numeric"):
            info["numeric"][nu.get("name")] = _floats(nu.get("data"))
        for tx in cu.findall("text"):
            info["text"][tx.get("name")] = tx.get("data")

    for sn in root.findall("sensor"):
        for ch in sn:
            if ch.tag == "user":
                u = _floats(ch.get("user"))          # norm, weight, lo, hi, norm params... (mjpc/task.cc:203-238)
                info["cost_terms"].append((int(ch.get("dim")), int(u[0]), u[1], u[4:], ch.get("name")))
            else:
                info["sensors"].append(dict(kind=ch.tag, **ch.attrib))
                if ch.get("name", "").startswith("trace"):
                    info["traces"].append((ch.get("objtype"), ch.get("objname")))

    for kf in root.findall("keyframe"):
        for k in kf.findall("key"):
            info["keys"].append({a: (_floats(k.get(a)) if a != "name" else k.get(a)) for a in k.attrib})
    return b, info


def read_patch_new_file(patch_path):
    """the "+++" side of a unified diff whose hunks cover the whole file (tasks/humanoid/humanoid.xml.patch): context and added
    lines, in order"""
    out = []
    with open(patch_path) as f:
        for line in f:
            if line.startswith("+++") or line.startswith("---") or line.startswith("@@") or line.startswith("diff ") or line.startswith("index "):
                continue
            if line.startswith("+") or line.startswith(" "):
                out.append(line[1:])
    return "".join(out)
