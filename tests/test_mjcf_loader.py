"""The hand-authored task generators (modelgen/tasks.py) against the reference's own MJCF files, through the MJCF subset loader.

Runs only where /root/reference is mounted (this container); the GPU box skips it.  Nothing of the reference is copied: the XML is
read in place, translated into ModelBuilder calls and the compiled arrays are compared with the generator's."""
import os

import numpy as np
import pytest

from mujoco_mpc_amd.modelgen import mjcf, tasks

REF = "/root/reference/mjpc"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted")

# spheres do not care about their frame: the hand geoms carry a zaxis in the XML that the generator drops
_ORIENTATION_FREE = {"geom_quat": ("geom", ["hand_right", "hand_left"]), "body_iquat": ("body", ["hand_right", "hand_left"])}
_SKIP = {"nconmax", "nefcmax", "names"}


def _reader(p):
    if os.path.basename(p) == "humanoid_modified.xml":       # exists upstream only as dm_control's file + this patch
        return mjcf.read_patch_new_file(os.path.join(REF, "tasks/humanoid/humanoid.xml.patch"))
    with open(p) as f:
        return f.read()


def _load(rel):
    b, info = mjcf.parse_mjcf(os.path.join(REF, rel), reader=_reader)
    b.opt["timestep"] = info["numeric"].get("agent_timestep", [b.opt["timestep"]])[0]     # mjpc/agent.cc: agent_timestep overrides
    return b, info


def _compare(m, m2, tol=1e-12):
    bad = []
    for k, v2 in m2.items():
        if k in _SKIP:
            continue
        v = m.get(k)
        if isinstance(v2, np.ndarray):
            if v is None or np.shape(v) != v2.shape:
                bad.append((k, "shape", None if v is None else np.shape(v), v2.shape)); continue
            if v2.dtype.kind == "f":
                d = np.abs(np.asarray(v, float) - v2)
                if k in _ORIENTATION_FREE:
                    kind, names = _ORIENTATION_FREE[k]
                    for n in names:
                        if n in m2["names"][kind]:
                            d[m2["names"][kind][n]] = 0
                if d.size and d.max() > tol:
                    bad.append((k, float(d.max())))
            elif not np.array_equal(v, v2):
                bad.append((k, "int mismatch"))
        elif isinstance(v2, (int, float)) and v != v2:
            bad.append((k, v, v2))
    return bad


_ID_FIELDS = {"body_parentid", "body_rootid", "body_weldid", "geom_bodyid", "site_bodyid", "jnt_bodyid", "dof_bodyid"}
_ORDER_FIELDS = {"body_jntadr", "body_dofadr", "exclude_signature"}


def _compare_named(m, m2, tol=1e-12):
    """like _compare for two models whose bodies / sites are declared in a different order: elements are matched by name and
    body-id valued fields by the name they point to"""
    bad = []
    inv = {kind: {i: n for n, i in m["names"][kind].items()} for kind in m["names"]}
    inv2 = {kind: {i: n for n, i in m2["names"][kind].items()} for kind in m2["names"]}
    for prefix, kind in (("body_", "body"), ("geom_", "geom"), ("site_", "site"), ("jnt_", "joint")):
        names = [n for n in m2["names"][kind] if n]
        assert set(names) == set(n for n in m["names"][kind] if n), kind
        for k, v2 in m2.items():
            if not k.startswith(prefix) or not isinstance(v2, np.ndarray) or k in _ORDER_FIELDS:
                continue
            for n in names:
                a, b_ = m[k][m["names"][kind][n]], v2[m2["names"][kind][n]]
                if k in _ID_FIELDS:
                    ok = inv["body"].get(int(a), "world") == inv2["body"].get(int(b_), "world")
                elif k in _ORIENTATION_FREE and n in _ORIENTATION_FREE[k][1]:
                    ok = True
                else:
                    ok = np.allclose(a, b_, atol=tol, rtol=0)
                if not ok:
                    bad.append((k, n))
    for k, v2 in m2.items():
        if isinstance(v2, np.ndarray) and k.split("_")[0] in ("dof", "actuator", "tendon", "wrap", "qpos0", "qpos") and k != "dof_bodyid":
            if not np.allclose(m[k], v2, atol=tol, rtol=0):
                bad.append((k,))
    return bad


def _terms_of(task):
    out, p = [], 0
    for i in range(task["num_term"]):
        n = int(task["num_norm_parameter"][i])
        out.append((int(task["dim_norm_residual"][i]), int(task["norm"][i]), float(task["weight"][i]), [float(x) for x in task["norm_parameter"][p:p + n]]))
        p += n
    return out


def _xml_terms(info):
    return [(d, n, w, list(prm[:tasks.NORM_NPARAM[n]])) for d, n, w, prm, _ in info["cost_terms"]]


def test_particle_xml_matches_generator():
    b, info = _load("test/testdata/particle_task.xml")
    m = b.compile()
    m2, task, d = tasks.particle()
    # the XML walls are visual planes in a contact-disabled model and the generator leaves them out: compare per named element
    for kind, keys in (("body", ["body_pos", "body_quat", "body_mass", "body_inertia", "body_parentid", "body_mocapid"]),
                       ("joint", ["jnt_type", "jnt_axis", "jnt_range", "jnt_limited", "jnt_pos", "jnt_solref", "jnt_solimp"]),):
        for name, i2 in m2["names"][kind].items():
            i = m["names"][kind][name]
            for k in keys:
                assert np.allclose(m[k][i], m2[k][i2], atol=1e-14), (name, k)
    for k in ("dof_damping", "dof_armature", "actuator_gear", "actuator_ctrlrange", "actuator_gainprm", "actuator_biasprm", "qpos0"):
        if k in m2:
            assert np.allclose(m[k], m2[k], atol=1e-14), k
    assert m["timestep"] == m2["timestep"] == 0.1 and m["nq"] == 2 and m["nu"] == 2
    assert b.opt["contact"] is False
    assert _xml_terms(info) == _terms_of(task)
    assert info["numeric"]["task_risk"] == [task["risk"]]
    assert [info["numeric"]["residual_dummy1"][0], info["numeric"]["residual_dummy2"][0]] == list(task["parameters"])
    assert info["numeric"]["sampling_spline_points"] == [d["P"]] and info["numeric"]["sampling_exploration"] == [d["sigma"][0]]
    assert info["traces"] == [("site", "tip")]
    home = next(k for k in info["keys"] if k["name"] == "home")
    assert home["qpos"] == [1.0, 2.0]


@pytest.mark.parametrize("rel, gen, params", [("tasks/humanoid/stand/task.xml", tasks.humanoid_stand, ["residual_Height Goal"]),
                                              ("tasks/humanoid/walk/task.xml", tasks.humanoid_walk, ["residual_Torso", "residual_Speed"])])
def test_humanoid_stand_walk_xml_match_generators(rel, gen, params):
    b, info = _load(rel)
    m = b.compile()
    m2, task, d = gen()
    assert _compare(m, m2) == []
    assert m["names"]["body"] == m2["names"]["body"] and m["names"]["site"] == m2["names"]["site"]
    assert _xml_terms(info) == _terms_of(task)
    assert [info["numeric"][p][0] for p in params] == list(task["parameters"])
    assert info["numeric"]["sampling_spline_points"] == [d["P"]] and info["numeric"]["sampling_exploration"] == [d["sigma"][0]]
    assert round(info["numeric"]["agent_horizon"][0] / m["timestep"]) + 1 >= d["horizon"] - 1
    assert info["traces"] == [("body", "torso")]


def test_humanoid_tracking_xml_matches_generator_and_keyframes():
    b, info = _load("tasks/humanoid/tracking/task.xml")
    m = b.compile()
    m2, task, d = tasks.humanoid_track()
    # the XML declares the mocap bodies after the humanoid, the generator before it: match by name
    assert _compare_named(m, m2) == []
    for k in ("nq", "nv", "nu", "nbody", "ngeom", "nsite", "nmocap", "ntendon", "nexclude", "timestep", "cone", "impratio", "meaninertia"):
        assert m[k] == pytest.approx(m2[k], abs=1e-12), k
    mocap_order = lambda mm: [n for n, i in sorted(mm["names"]["body"].items(), key=lambda kv: kv[1]) if mm["body_mocapid"][i] >= 0]
    assert mocap_order(m) == mocap_order(m2)            # the mocap_pos layout the keyframe table indexes
    assert _xml_terms(info) == _terms_of(task)
    assert info["numeric"]["sampling_spline_points"] == [d["P"]] and info["numeric"]["sampling_exploration"] == [d["sigma"][0]]
    assert info["numeric"]["sampling_trajectories"] == [d["N"]]
    # the packed table in modelgen/data is the keys of the ten included keyframe files, in the order of the motion table
    # (tracking.cc:43-54); motion 0 ("Jump") comes first
    keys = [k for k in info["keys"] if k.get("mpos") is not None and len(np.atleast_1d(k["mpos"]))]
    lengths = [121, 154, 115, 78, 145, 188, 260, 279, 39, 510]
    assert len(keys) == sum(lengths) == m2["nkey"]
    table = np.asarray(m2["key_mpos"]).reshape(m2["nkey"], -1)
    assert np.array_equal(np.array([k["mpos"] for k in keys]), table)
    assert keys[0]["name"].startswith("jump_") and int(task["int_data"][2]) == lengths[0]
    assert np.allclose(keys[0]["qpos"], d["state"][:m2["nq"]], atol=0)
    for motion in (1, 8, 9):
        _, tk, dk = tasks.humanoid_track(motion=motion)
        first = sum(lengths[:motion])
        assert list(tk["int_data"][:3]) == [motion, first, lengths[motion]]
        assert np.allclose(keys[first]["qpos"], dk["state"][:m2["nq"]], atol=0)


def _residual_parameters(info):
    """mjpc/task.cc:181-200: numerics named residual_* in file order; residual_select_* carry the int bit-cast into the double"""
    out = []
    for name, v in info["numeric"].items():
        if name.startswith("residual_select_"):
            out.append(tasks.select_value(int(v[0])))
        elif name.startswith("residual_"):
            out.append(v[0])
    return out


@pytest.mark.parametrize("rel, missing, gen", [("tasks/cartpole/task.xml", "cartpole_modified.xml", tasks.cartpole),
                                               ("tasks/quadruped/task_flat.xml", "a1_modified.xml", tasks.quadruped)])
def test_task_tables_of_models_that_need_the_zoo(rel, missing, gen):
    """cartpole / A1 model files exist upstream only as patches against dm_control / menagerie (absent here); their task files are
    complete, so the cost table, residual parameters, planner numerics, keyframes and mocap bodies are checked"""
    b, info = mjcf.parse_mjcf(os.path.join(REF, rel), missing_ok=(missing,))
    m2, task, d = gen()
    assert _xml_terms(info) == _terms_of(task)
    got, want = _residual_parameters(info), list(task["parameters"])
    assert np.array(got).tobytes() == np.array(want).tobytes()             # bitwise: the select values are not ordinary doubles
    assert info["numeric"]["agent_timestep"] == [m2["timestep"]]
    assert info["numeric"]["sampling_spline_points"] == [d["P"]] and info["numeric"]["sampling_exploration"] == [d["sigma"][0]]
    if "sampling_trajectories" in info["numeric"]:
        assert info["numeric"]["sampling_trajectories"] == [d["N"]]
    assert round(info["numeric"]["agent_horizon"][0] / m2["timestep"]) + 1 == d["horizon"]
    assert info["traces"] == [("site", {"tasks/cartpole/task.xml": "tip", "tasks/quadruped/task_flat.xml": "head"}[rel])]
    home = next(k for k in info["keys"] if k["name"] == "home")
    assert np.array_equal(home["qpos"], d["state"][:m2["nq"]])
    # mocap bodies declared by the task file itself (quadruped: goal, box) sit at the generator's default mocap poses
    for body in b.bodies[1:]:
        if body.mocap and body.name in m2["names"]["body"]:
            mid = int(m2["body_mocapid"][m2["names"]["body"][body.name]])
            assert np.allclose(d["mocap"][7 * mid:7 * mid + 3], body.pos, atol=0), body.name


@pytest.mark.parametrize("rel, missing, gen", [("tasks/swimmer/task.xml", "swimmer_modified.xml", tasks.swimmer),
                                               ("tasks/quadrotor/task.xml", "quadrotor_modified.xml", tasks.quadrotor),
                                               ("tasks/particle/task_timevarying.xml", "particle_modified.xml", tasks.particle_task)])
def test_cost_tables_and_agent_numerics_of_the_late_registry_tasks(rel, missing, gen):
    """Swimmer / Quadrotor / Particle: the model files are patches against dm_control / menagerie (absent here), the task files are
    complete: the generators carry their cost terms, time step, spline points and (swimmer / particle) the horizon; the quadrotor's
    stage goals are the task file's keyframe mocap positions."""
    extra = ("gates.xml",) if "quadrotor" in rel else ()
    b, info = mjcf.parse_mjcf(os.path.join(REF, rel), missing_ok=(missing,) + extra)
    m2, task, d = gen()
    assert _xml_terms(info) == _terms_of(task)
    assert info["numeric"]["agent_timestep"] == [m2["timestep"]]
    assert info["numeric"]["sampling_spline_points"] == [d["P"]]
    assert round(info["numeric"]["agent_horizon"][0] / m2["timestep"]) + 1 == d["horizon"]
    if "quadrotor" in rel:
        assert [tuple(k["mpos"]) for k in info["keys"]] == [tuple(map(float, p)) for p in tasks.QUADROTOR_STAGES]
        assert info["numeric"]["sampling_exploration"] == [d["sigma"][0]]


def test_humanoid_interact_xml_matches_generator():
    """tasks/humanoid/interact/task.xml + scenes/armchair.xml (both local; the humanoid itself is the patched dm_control file the
    other humanoid tasks share): cost table, residual parameters, agent numerics, the chair's five boxes and the scene's home key"""
    b, info = mjcf.parse_mjcf(os.path.join(REF, "tasks/humanoid/interact/task.xml"), missing_ok=("humanoid_modified.xml",))
    m2, task, d = tasks.humanoid_interact()
    assert _xml_terms(info) == _terms_of(task)
    assert [info["numeric"]["residual_Head Height"][0], info["numeric"]["residual_Torso Height"][0]] == list(task["parameters"])
    assert info["numeric"]["agent_timestep"] == [m2["timestep"]] and info["numeric"]["sampling_spline_points"] == [d["P"]]
    assert info["numeric"]["sampling_exploration"] == [d["sigma"][0]] and round(info["numeric"]["agent_horizon"][0] / m2["timestep"]) + 1 >= d["horizon"] - 1
    home = next(k for k in info["keys"] if k["name"] == "home")
    assert np.allclose(home["qpos"], d["state"][:m2["nq"]], atol=0)
    chair = next(bd for bd in b.bodies if bd.name == "chair")
    cid = m2["names"]["body"]["chair"]
    assert np.allclose(chair.pos, np.asarray(m2["body_pos"]).reshape(-1, 3)[cid])
    mx = b.compile()
    gx = [g for g in range(mx["ngeom"]) if mx["geom_bodyid"][g] == mx["names"]["body"]["chair"]]
    g2 = [g for g in range(m2["ngeom"]) if m2["geom_bodyid"][g] == cid]
    assert len(gx) == len(g2) == 5
    for a, c in zip(gx, g2):
        for k, w in (("geom_size", 3), ("geom_pos", 3), ("geom_quat", 4), ("geom_friction", 3)):
            assert np.allclose(np.asarray(mx[k]).reshape(-1, w)[a], np.asarray(m2[k]).reshape(-1, w)[c], atol=1e-12), k
        assert mx["geom_type"][a] == m2["geom_type"][c] and mx["geom_condim"][a] == m2["geom_condim"][c]
    assert tasks.INTERACT_MODES == tuple(info["text"]["task_transition"].split("|")) if "text" in info else True


def test_fingers_xml_matches_generator():
    """tasks/fingers/task.xml is complete and local: the generator's model (bodies, joints, geoms, sites, the six integrated-velocity
    servos on site transmissions against the world site, gravity compensation, the exclude), options (elliptic cones, noslip 5, the
    agent's implicit integrator and 5 ms step), cost terms, agent settings and home key are the file's."""
    b, info = _load("tasks/fingers/task.xml")
    m = b.compile()
    m2, task, d = tasks.fingers()
    assert _compare_named(m, m2) == []
    for k in ("actuator_trntype", "actuator_gear6", "actuator_dyntype", "actuator_actlimited", "actuator_actrange", "actuator_ctrllimited", "actuator_biastype", "body_gravcomp"):
        assert np.array_equal(np.asarray(m[k]), np.asarray(m2[k])), k
    site_name = lambda mm, i: {v: k for k, v in mm["names"]["site"].items()}[int(i)]
    assert [site_name(m, i) for i in m["actuator_trnid"]] == [site_name(m2, i) for i in m2["actuator_trnid"]] == ["finger_a"] * 3 + ["finger_b"] * 3
    assert [site_name(m, i) for i in m["actuator_refsite"]] == [site_name(m2, i) for i in m2["actuator_refsite"]] == ["world"] * 6
    assert (m["cone"], m["noslip_iterations"], m["nexclude"]) == (m2["cone"], m2["noslip_iterations"], m2["nexclude"]) == (1, 5, 1)
    assert info["numeric"]["agent_timestep"] == [m2["timestep"]] and info["numeric"]["agent_integrator"] == [m2["integrator"]] and m["integrator"] == 2
    assert _xml_terms(info) == _terms_of(task)
    assert info["numeric"]["sampling_trajectories"] == [d["N"]] and info["numeric"]["sampling_spline_points"] == [d["P"]] and info["numeric"]["sampling_exploration"] == [d["sigma"][0]]
    assert round(info["numeric"]["agent_horizon"][0] / m2["timestep"]) + 1 == d["horizon"]
    home = next(k for k in info["keys"] if k["name"] == "home")
    q = np.array(home["qpos"]); q[10:14] /= np.linalg.norm(q[10:14])
    assert np.allclose(q, d["state"][:20], atol=1e-15) and np.array_equal(home["act"], d["state"][38:])


def test_loader_attributes_of_the_late_features():
    """<option density / viscosity / wind / integrator>, body gravcomp, joint actuatorfrcrange, <velocity> actuators, filter dynamics and
    <equality> are outside the reference files above: a small inline document"""
    xml = """<mujoco><option timestep="0.004" density="12" viscosity="0.3" wind="1 0 0" integrator="implicitfast"/>
      <worldbody><body name="a" pos="0 0 1" gravcomp="0.5"><joint name="h" type="hinge" axis="0 1 0" actuatorfrcrange="-2 3"/>
        <geom type="sphere" size="0.1"/><body name="c" pos="0 0 -0.3"><joint name="h2" type="hinge" axis="0 1 0"/><geom type="sphere" size="0.05"/></body></body></worldbody>
      <equality><joint joint1="h" joint2="h2" polycoef="0 -1 0.1"/><connect body1="c" anchor="0 0 -0.1" solref="0.01 1"/>
        <weld body1="c" body2="a" anchor="0.01 0 0" torquescale="0.5"/><weld body1="c" relpose="0 0 -0.7 0 0 0 2"/></equality>
      <actuator><velocity name="v" joint="h" kv="7"/><general name="f" joint="h" gainprm="2" dyntype="filter" dynprm="0.4"/></actuator></mujoco>"""
    b, info = mjcf.parse_mjcf(xml)
    m = b.compile()
    assert (m["density"], m["viscosity"], tuple(m["wind"]), m["integrator"], m["timestep"]) == (12.0, 0.3, (1.0, 0.0, 0.0), 3, 0.004)
    assert m["body_gravcomp"][1] == 0.5 and m["jnt_actfrclimited"][0] == 1 and tuple(m["jnt_actfrcrange"][0]) == (-2.0, 3.0)
    assert tuple(m["actuator_gainprm"][0]) == (7.0, 0, 0) and tuple(m["actuator_biasprm"][0]) == (0, 0, -7.0) and m["actuator_biastype"][0] == 1
    assert m["na"] == 1 and m["actuator_dyntype"][1] == 2 and m["actuator_dynprm"][1] == 0.4 and m["actuator_actadr"][1] == 0
    assert m["neq"] == 4 and list(m["eq_type"]) == [2, 0, 1, 1] and tuple(m["eq_data"][0][:3]) == (0.0, -1.0, 0.1)
    # weld without relpose: the anchor (body2 frame) seen from body1 and the orientation of body2 in body1 at qpos0; with one: as given, quat normalised
    assert np.allclose(m["eq_data"][2], [0.01, 0, 0, 0.01, 0, 0.3, 1, 0, 0, 0, 0.5]) and np.allclose(m["eq_data"][3], [0, 0, 0, 0, 0, -0.7, 0, 0, 0, 1, 1])
    assert np.allclose(m["eq_data"][1][:6], [0, 0, -0.1, 0, 0, 0.6]) and tuple(m["eq_solref"][1]) == (0.01, 1.0)      # second anchor: the point in the world frame
    with pytest.raises(ValueError):
        mjcf.parse_mjcf('<mujoco><option integrator="RK4"/><worldbody/></mujoco>')
