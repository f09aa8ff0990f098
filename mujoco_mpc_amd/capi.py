"""ctypes mirror of include/mjpc_hip.h and loader for the in-tree HIP engine `libmjpc_hip.so`.

The engine is the product path: `load_engine()` raises if the shared library is missing — there is
no CPU fallback (the CPU oracle under oracle/ is test infrastructure and is never imported here).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ENGINE_PATH = os.path.join(_HERE, "csrc", "libmjpc_hip.so")

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)

_MODEL_INT_SIZES = ["nq", "nv", "nu", "na", "nbody", "njnt", "ngeom", "nsite", "nmocap", "nuserdata", "nkey", "nexclude", "ntendon", "nwrap", "nmesh", "nmeshvert", "nhfield", "nhfielddata"]
_OPTIONAL_TENDON = ("tendon_stiffness", "tendon_damping", "tendon_lengthspring", "tendon_frictionloss", "tendon_solref_fri", "tendon_solimp_fri")
_OPTIONAL_EQ = ("eq_type", "eq_obj1id", "eq_obj2id", "eq_active0", "eq_data", "eq_solref", "eq_solimp")
_OPTIONAL_ACT = ("actuator_dyntype", "actuator_actadr", "actuator_actlimited", "actuator_dynprm", "actuator_actrange")
_OPTION_DEFAULTS = dict(enableflags=0, solver=2, integrator=0, noslip_iterations=0, neq=0, unsupported=0)
_OPTIONAL_MESH = ("nmesh", "nmeshvert", "geom_dataid", "mesh_vertadr", "mesh_vertnum", "mesh_vert", "nhfield", "nhfielddata", "hfield_nrow",
                  "hfield_ncol", "hfield_adr", "hfield_size", "hfield_data")
_MODEL_INT_ARRAYS_BODY = ["body_parentid", "body_rootid", "body_weldid", "body_mocapid", "body_jntnum", "body_jntadr",
                          "body_dofnum", "body_dofadr"]
_MODEL_DBL_ARRAYS_BODY = ["body_pos", "body_quat", "body_ipos", "body_iquat", "body_mass", "body_subtreemass",
                          "body_inertia", "body_invweight0"]


class MjpcHipModel(C.Structure):
    _fields_ = (
        [("struct_size", C.c_int)] + [(n, C.c_int) for n in _MODEL_INT_SIZES]
        + [("timestep", C.c_double), ("gravity", C.c_double * 3), ("impratio", C.c_double),
           ("tolerance", C.c_double), ("ls_tolerance", C.c_double), ("cone", C.c_int), ("iterations", C.c_int),
           ("ls_iterations", C.c_int), ("disableflags", C.c_int), ("enableflags", C.c_int), ("solver", C.c_int), ("integrator", C.c_int),
           ("noslip_iterations", C.c_int), ("neq", C.c_int), ("unsupported", C.c_int), ("meaninertia", C.c_double),
           ("density", C.c_double), ("viscosity", C.c_double), ("wind", C.c_double * 3),
           ("nconmax", C.c_int), ("nefcmax", C.c_int)]
        + [(n, c_int_p) for n in _MODEL_INT_ARRAYS_BODY]
        + [(n, c_double_p) for n in _MODEL_DBL_ARRAYS_BODY]
        + [("jnt_actfrclimited", c_int_p), ("jnt_actfrcrange", c_double_p), ("body_gravcomp", c_double_p)]
        + [(n, c_int_p) for n in ["jnt_type", "jnt_qposadr", "jnt_dofadr", "jnt_bodyid", "jnt_limited"]]
        + [(n, c_double_p) for n in ["jnt_pos", "jnt_axis", "jnt_stiffness", "jnt_range", "jnt_margin",
                                     "jnt_solref", "jnt_solimp", "qpos0", "qpos_spring"]]
        + [(n, c_int_p) for n in ["dof_bodyid", "dof_jntid", "dof_parentid"]]
        + [(n, c_double_p) for n in ["dof_armature", "dof_damping", "dof_frictionloss", "dof_invweight0",
                                     "dof_solref", "dof_solimp"]]
        + [(n, c_int_p) for n in ["geom_type", "geom_contype", "geom_conaffinity", "geom_condim", "geom_bodyid",
                                  "geom_group", "geom_priority"]]
        + [(n, c_double_p) for n in ["geom_size", "geom_pos", "geom_quat", "geom_friction", "geom_solmix",
                                     "geom_solref", "geom_solimp", "geom_margin", "geom_gap", "geom_rbound"]]
        + [("exclude_signature", c_int_p)]
        + [(n, c_int_p) for n in ["eq_type", "eq_obj1id", "eq_obj2id", "eq_active0"]] + [(n, c_double_p) for n in ["eq_data", "eq_solref", "eq_solimp"]]
        + [("site_bodyid", c_int_p), ("site_pos", c_double_p), ("site_quat", c_double_p)]
        + [(n, c_int_p) for n in ["actuator_trntype", "actuator_trnid", "actuator_ctrllimited", "actuator_forcelimited", "actuator_biastype"]]
        + [(n, c_double_p) for n in ["actuator_gainprm", "actuator_biasprm", "actuator_gear", "actuator_gear6", "actuator_ctrlrange",
                                     "actuator_forcerange"]]
        + [(n, c_int_p) for n in ["actuator_dyntype", "actuator_actadr", "actuator_actlimited"]]
        + [(n, c_double_p) for n in ["actuator_dynprm", "actuator_actrange"]]
        + [(n, c_int_p) for n in ["tendon_adr", "tendon_num", "tendon_limited", "wrap_objid"]]
        + [(n, c_double_p) for n in ["wrap_prm", "tendon_range", "tendon_margin", "tendon_solref_lim", "tendon_solimp_lim",
                                     "tendon_invweight0", "tendon_stiffness", "tendon_damping", "tendon_lengthspring",
                                     "tendon_frictionloss", "tendon_solref_fri", "tendon_solimp_fri"]]
        + [(n, c_int_p) for n in ["geom_dataid", "mesh_vertadr", "mesh_vertnum"]] + [("mesh_vert", c_double_p)]
        + [(n, c_int_p) for n in ["hfield_nrow", "hfield_ncol", "hfield_adr"]] + [("hfield_size", c_double_p), ("hfield_data", c_double_p)]
        + [("key_qpos", c_double_p), ("key_mpos", c_double_p)]
        + [("actuator_refsite", c_int_p), ("noslip_tolerance", C.c_double)]
    )


class MjpcHipTask(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int), ("task_id", C.c_int), ("num_residual", C.c_int), ("num_term", C.c_int), ("num_trace", C.c_int),
        ("dim_norm_residual", c_int_p), ("norm", c_int_p), ("num_norm_parameter", c_int_p),
        ("weight", c_double_p), ("norm_parameter", c_double_p), ("risk", C.c_double),
        ("num_parameter", C.c_int), ("parameters", c_double_p),
        ("trace_objtype", c_int_p), ("trace_objid", c_int_p),
        ("num_int", C.c_int), ("int_data", c_int_p), ("num_dbl", C.c_int), ("dbl_data", c_double_p),
    ]


class MjpcHipPlanInput(C.Structure):
    _fields_ = [
        ("state", c_double_p), ("mocap", c_double_p), ("userdata", c_double_p), ("time", C.c_double),
        ("knot_times", c_double_p), ("knot_values", c_double_p), ("num_spline_points", C.c_int),
        ("interpolation", C.c_int), ("num_trajectory", C.c_int), ("horizon", C.c_int),
        ("candidate_offset", C.c_int), ("num_local", C.c_int), ("noise_exploration", C.c_double * 2),
        ("noise_eps", c_double_p), ("noise_sel", c_int_p), ("seed", C.c_uint64), ("stream", C.c_uint64),
        ("noise_std", c_double_p), ("nominal_index", C.c_int),
        ("candidate_knots", c_double_p), ("xfrc_std", C.c_double), ("xfrc_rate", C.c_double),
    ]


class MjpcHipPlanOutput(C.Structure):
    _fields_ = [
        ("returns", c_double_p), ("failure", c_int_p), ("winner", C.c_int), ("winner_return", C.c_double),
        ("states", c_double_p), ("actions", c_double_p), ("times", c_double_p), ("residual", c_double_p),
        ("costs", c_double_p), ("trace", c_double_p), ("winner_knots", c_double_p),
        ("noise_compute_time_us", C.c_double), ("rollouts_compute_time_us", C.c_double),
    ]


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(c_int_p) if a is not None else None


class CModel:
    """Owns contiguous numpy copies and the ctypes structs that point into them."""

    def __init__(self, model: dict, task: dict):
        self.model = model
        self.task = task
        self._keep = []
        m = MjpcHipModel()
        for name, ctype in MjpcHipModel._fields_:
            if name == "struct_size":
                m.struct_size = C.sizeof(MjpcHipModel)
                continue
            if name in _OPTION_DEFAULTS and name not in model:        # mjOption fields added later: MuJoCo's defaults
                v = _OPTION_DEFAULTS[name]
            elif name in _OPTIONAL_TENDON and name not in model:       # models built before these fields existed: no passive tendon forces
                nt = int(model["ntendon"])
                v = (np.tile([0.02, 1.0], nt) if name == "tendon_solref_fri" else np.tile([0.9, 0.95, 0.001, 0.5, 2.0], nt) if name == "tendon_solimp_fri"
                     else np.zeros(nt * (2 if name == "tendon_lengthspring" else 1)))
            elif name in ("density", "viscosity", "wind") and name not in model:      # models built before fluid forces existed
                v = (0.0, 0.0, 0.0) if name == "wind" else 0.0
            elif name in ("jnt_actfrclimited", "jnt_actfrcrange") and name not in model:      # models built before the joint-level force clamp existed
                v = np.zeros(int(model["njnt"]) * (2 if name == "jnt_actfrcrange" else 1))
            elif name == "body_gravcomp" and name not in model:       # models built before gravity compensation existed
                v = np.zeros(int(model["nbody"]))
            elif name == "actuator_gear6" and name not in model:      # models built before site transmissions existed
                v = np.zeros(6 * int(model["nu"]))
            elif name == "actuator_refsite" and name not in model:    # models built before reference sites existed: none
                v = -np.ones(int(model["nu"]))
            elif name == "noslip_tolerance" and name not in model:
                v = 1e-6
            elif name in _OPTIONAL_EQ and name not in model:            # models built before equality constraints existed: none
                v = np.zeros(0)
            elif name in _OPTIONAL_ACT and name not in model:           # models built before activation states existed: none
                nu_ = int(model["nu"])
                v = -np.ones(nu_) if name == "actuator_actadr" else np.zeros(nu_ * (2 if name == "actuator_actrange" else 1))
            elif name in _OPTIONAL_MESH and name not in model:        # ... no meshes
                v = -np.ones(int(model["ngeom"])) if name == "geom_dataid" else (0 if name in ("nmesh", "nmeshvert", "nhfield", "nhfielddata") else np.zeros(0))
            else:
                v = model[name]
            if ctype is c_double_p:
                arr = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel())
                if arr.size == 0:
                    arr = np.zeros(1)
                self._keep.append(arr); setattr(m, name, _dp(arr))
            elif ctype is c_int_p:
                arr = np.ascontiguousarray(np.asarray(v, dtype=np.int32).ravel())
                if arr.size == 0:
                    arr = np.zeros(1, np.int32)
                self._keep.append(arr); setattr(m, name, _ip(arr))
            elif name in ("gravity", "wind"):
                setattr(m, name, (C.c_double * 3)(*[float(x) for x in v]))
            elif ctype is C.c_int:
                setattr(m, name, int(v))
            else:
                setattr(m, name, float(v))
        self.c_model = m
        self.c_task = self.make_task(task)

    def make_task(self, task: dict) -> MjpcHipTask:
        t = MjpcHipTask()
        for name, ctype in MjpcHipTask._fields_:
            if name == "struct_size":
                t.struct_size = C.sizeof(MjpcHipTask)
                continue
            v = task[name]
            if ctype is c_double_p:
                arr = np.ascontiguousarray(np.asarray(v, dtype=np.float64).ravel())
                if arr.size == 0:
                    arr = np.zeros(1)
                self._keep.append(arr); setattr(t, name, _dp(arr))
            elif ctype is c_int_p:
                arr = np.ascontiguousarray(np.asarray(v, dtype=np.int32).ravel())
                if arr.size == 0:
                    arr = np.zeros(1, np.int32)
                self._keep.append(arr); setattr(t, name, _ip(arr))
            elif ctype is C.c_int:
                setattr(t, name, int(v))
            else:
                setattr(t, name, float(v))
        return t

    @property
    def dim_state(self):
        return self.model["nq"] + self.model["nv"] + self.model["na"]


def make_plan_input(cm: CModel, state, mocap, time, knot_times, knot_values, interpolation, num_trajectory, horizon,
                    sigma=(0.1, 0.0), noise_eps=None, noise_sel=None, seed=0, stream=0, candidate_offset=0,
                    num_local=None, userdata=None, noise_std=None, nominal_index=0, candidate_knots=None, xfrc_std=0.0, xfrc_rate=0.0):
    keep = []

    def arr(x, n=None):
        a = np.ascontiguousarray(np.asarray(x if x is not None else np.zeros(max(n or 1, 1)), dtype=np.float64).ravel())
        if a.size == 0:
            a = np.zeros(1)
        keep.append(a)
        return a
    inp = MjpcHipPlanInput()
    inp.state = _dp(arr(state)); inp.mocap = _dp(arr(mocap, 7 * cm.model["nmocap"]))
    inp.userdata = _dp(arr(userdata, cm.model["nuserdata"])); inp.time = float(time)
    kt = arr(knot_times); kv = arr(knot_values)
    inp.knot_times = _dp(kt); inp.knot_values = _dp(kv)
    inp.num_spline_points = int(len(np.asarray(knot_times).ravel())); inp.interpolation = int(interpolation)
    inp.num_trajectory = int(num_trajectory); inp.horizon = int(horizon)
    inp.candidate_offset = int(candidate_offset)
    inp.num_local = int(num_trajectory if num_local is None else num_local)
    inp.noise_exploration = (C.c_double * 2)(float(sigma[0]), float(sigma[1]))
    if noise_eps is not None:
        inp.noise_eps = _dp(arr(noise_eps))
    if noise_sel is not None:
        s = np.ascontiguousarray(np.asarray(noise_sel, dtype=np.int32)); keep.append(s); inp.noise_sel = _ip(s)
    inp.seed = int(seed); inp.stream = int(stream)
    if noise_std is not None:
        inp.noise_std = _dp(arr(noise_std))
    inp.nominal_index = int(nominal_index)
    if candidate_knots is not None:
        inp.candidate_knots = _dp(arr(candidate_knots))
    inp.xfrc_std = float(xfrc_std); inp.xfrc_rate = float(xfrc_rate)
    inp._keep = keep
    return inp


_engine = None


def load_engine():
    """Load libmjpc_hip.so (HIP engine).  Fails loudly when it has not been built."""
    global _engine
    if _engine is not None:
        return _engine
    if not os.path.exists(ENGINE_PATH):
        raise RuntimeError(f"HIP engine not built: {ENGINE_PATH} missing (run `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = C.CDLL(ENGINE_PATH)
    lib.mjpc_hip_create.restype = C.c_void_p
    lib.mjpc_hip_create.argtypes = [C.POINTER(MjpcHipModel), C.POINTER(MjpcHipTask), C.c_int, C.c_int, C.c_int]
    lib.mjpc_hip_destroy.argtypes = [C.c_void_p]
    lib.mjpc_hip_destroy.restype = None
    lib.mjpc_hip_set_task.argtypes = [C.c_void_p, C.POINTER(MjpcHipTask)]
    lib.mjpc_hip_plan.argtypes = [C.c_void_p, C.POINTER(MjpcHipPlanInput), C.POINTER(MjpcHipPlanOutput)]
    lib.mjpc_hip_plan_async.argtypes = [C.c_void_p, C.POINTER(MjpcHipPlanInput)]
    lib.mjpc_hip_plan_fetch.argtypes = [C.c_void_p, C.POINTER(MjpcHipPlanOutput)]
    lib.mjpc_hip_get_candidate.argtypes = [C.c_void_p, C.c_int, C.POINTER(MjpcHipPlanOutput)]
    lib.mjpc_hip_get_knots.argtypes = [C.c_void_p, c_double_p]
    lib.mjpc_hip_get_frame.argtypes = [C.c_void_p] + [c_double_p] * 5
    lib.mjpc_hip_kernel_time.argtypes = [C.c_void_p, c_double_p, c_double_p]
    lib.mjpc_hip_device_ptrs.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.mjpc_hip_get_all_candidates.argtypes = [C.c_void_p] + [c_double_p] * 7 + [c_int_p]
    lib.mjpc_hip_get_traces.argtypes = [C.c_void_p, c_double_p]
    lib.mjpc_hip_lds_bytes.argtypes = [C.c_void_p]
    lib.mjpc_hip_layout_bytes.argtypes = [C.POINTER(MjpcHipModel), C.POINTER(MjpcHipTask), C.c_int]
    lib.mjpc_hip_set_fetch_mode.argtypes = [C.c_void_p, C.c_int]
    lib.mjpc_hip_dense_tier.argtypes = [C.c_void_p, c_int_p]
    lib.mjpc_hip_debug_dense_capacity.argtypes = [C.c_void_p, c_int_p, c_int_p, c_int_p]; lib.mjpc_hip_debug_dense_capacity.restype = None
    lib.mjpc_hip_multi_create.restype = C.c_void_p
    lib.mjpc_hip_multi_create.argtypes = [C.POINTER(MjpcHipModel), C.POINTER(MjpcHipTask), C.c_int, C.c_int, C.c_int, c_int_p]
    lib.mjpc_hip_multi_destroy.argtypes = [C.c_void_p]
    lib.mjpc_hip_multi_destroy.restype = None
    lib.mjpc_hip_multi_set_task.argtypes = [C.c_void_p, C.POINTER(MjpcHipTask)]
    lib.mjpc_hip_multi_plan.argtypes = [C.c_void_p, C.POINTER(MjpcHipPlanInput), C.POINTER(MjpcHipPlanOutput)]
    lib.mjpc_hip_multi_get_candidate.argtypes = [C.c_void_p, C.c_int, C.POINTER(MjpcHipPlanOutput)]
    lib.mjpc_hip_multi_get_knots.argtypes = [C.c_void_p, c_double_p]
    lib.mjpc_hip_multi_get_traces.argtypes = [C.c_void_p, c_double_p]
    lib.mjpc_hip_multi_num_devices.argtypes = [C.c_void_p]
    lib.mjpc_hip_multi_engine.restype = C.c_void_p
    lib.mjpc_hip_multi_engine.argtypes = [C.c_void_p, C.c_int]
    lib.mjpc_hip_last_error.restype = C.c_char_p
    lib.mjpc_hip_version.restype = C.c_int
    # a stale library next to a newer ctypes layout must not get as far as reading pointers out of the wrong offsets
    if lib.mjpc_hip_version() != ABI_VERSION:
        raise RuntimeError(f"{ENGINE_PATH}: ABI revision {lib.mjpc_hip_version()}, these bindings are revision {ABI_VERSION}: rebuild the engine")
    for what, ctype in (("model", MjpcHipModel), ("task", MjpcHipTask), ("plan_input", MjpcHipPlanInput), ("plan_output", MjpcHipPlanOutput)):
        n = getattr(lib, "mjpc_hip_sizeof_" + what)()
        if n != C.sizeof(ctype):
            raise RuntimeError(f"{ENGINE_PATH}: sizeof {what} struct is {n} in the library, {C.sizeof(ctype)} in capi.py: rebuild the engine")
    lib.mjpc_hip_debug_set.argtypes = [C.c_char_p, C.c_char_p]
    lib.mjpc_hip_debug_set.restype = None
    _engine = lib
    return lib


def debug_set(name: str, value=None):
    """Engine diagnostics knob (include/mjpc_hip_debug.h), read when an engine is created; value None clears it."""
    load_engine().mjpc_hip_debug_set(name.encode(), None if value is None else str(value).encode())


ABI_VERSION = 4            # MJPC_HIP_ABI_VERSION of include/mjpc_hip.h

EXPORTED_SYMBOLS = [
    "mjpc_hip_sizeof_model", "mjpc_hip_sizeof_task", "mjpc_hip_sizeof_plan_input", "mjpc_hip_sizeof_plan_output", "mjpc_hip_debug_set",
    "mjpc_hip_create", "mjpc_hip_destroy", "mjpc_hip_set_task", "mjpc_hip_plan", "mjpc_hip_plan_async",
    "mjpc_hip_plan_fetch", "mjpc_hip_get_candidate", "mjpc_hip_kernel_time", "mjpc_hip_device_ptrs",
    "mjpc_hip_last_error", "mjpc_hip_version", "mjpc_hip_get_knots", "mjpc_hip_get_frame",
    "mjpc_hip_get_traces", "mjpc_hip_get_all_candidates", "mjpc_hip_lds_bytes", "mjpc_hip_layout_bytes", "mjpc_hip_set_fetch_mode", "mjpc_hip_dense_tier", "mjpc_hip_debug_dense_capacity",
    "mjpc_hip_multi_create", "mjpc_hip_multi_destroy", "mjpc_hip_multi_set_task", "mjpc_hip_multi_plan", "mjpc_hip_multi_get_candidate",
    "mjpc_hip_multi_get_knots", "mjpc_hip_multi_get_traces", "mjpc_hip_multi_num_devices", "mjpc_hip_multi_engine",
]
