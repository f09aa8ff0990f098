"""ctypes wrapper around oracle/_build/liboracle.so — TEST INFRASTRUCTURE (checker only)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from mujoco_mpc_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.environ.get("MJPC_ORACLE_SO") or os.path.join(ORACLE_DIR, "_build", "liboracle.so")   # override: the asan build

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class OPlanOutput(C.Structure):
    _fields_ = [("returns", c_double_p), ("failure", c_int_p), ("states", c_double_p), ("actions", c_double_p),
                ("times", c_double_p), ("residual", c_double_p), ("costs", c_double_p), ("trace", c_double_p),
                ("knots", c_double_p), ("winner", C.c_int), ("unsupported", C.c_int), ("solver_iter_total", C.c_int)]


_lib = None
FAST = False      # bench.py's cpu_baseline sets this before the first use: -O3 -march=native build, rebuilt locally


def lib():
    global _lib
    if _lib is not None:
        return _lib
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".c", ".h"))]
    srcs.append(os.path.join(ROOT, "include", "mjpc_hip.h"))
    so = ORACLE_SO
    if FAST:
        so = os.path.join(ORACLE_DIR, "_build", "liboracle_fast.so")
        subprocess.check_call(["make", "-B", "-C", ORACLE_DIR, "fast"], stdout=subprocess.DEVNULL)   # -march=native: build where it runs
    else:
        # parallel test workers: one of them rebuilds, the others wait for the lock and find the library fresh (the Makefile renames
        # the finished file into place, so a reader never sees a half-written one)
        import fcntl
        os.makedirs(os.path.join(ORACLE_DIR, "_build"), exist_ok=True)
        with open(os.path.join(ORACLE_DIR, "_build", ".lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            if (not os.path.exists(ORACLE_SO)) or any(os.path.getmtime(s) > os.path.getmtime(ORACLE_SO) for s in srcs):
                subprocess.check_call(["make", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)
    L = C.CDLL(so)
    L.oracle_create.restype = C.c_void_p
    L.oracle_create.argtypes = [C.POINTER(capi.MjpcHipModel), C.POINTER(capi.MjpcHipTask)]
    L.oracle_destroy.argtypes = [C.c_void_p]
    L.oracle_set_task.argtypes = [C.c_void_p, C.POINTER(capi.MjpcHipTask)]
    L.oracle_plan.argtypes = [C.c_void_p, C.POINTER(capi.MjpcHipPlanInput), C.POINTER(OPlanOutput), C.c_int]
    L.oracle_pool_create.restype = C.c_void_p
    L.oracle_pool_create.argtypes = [C.c_void_p, C.c_int]
    L.oracle_pool_destroy.argtypes = [C.c_void_p]
    L.oracle_pool_plan.argtypes = [C.c_void_p, C.POINTER(capi.MjpcHipPlanInput), C.POINTER(OPlanOutput)]
    L.oracle_spline_sample.argtypes = [c_double_p, c_double_p, C.c_int, C.c_int, C.c_int, C.c_double, c_double_p]
    L.oracle_norm.restype = C.c_double
    L.oracle_norm.argtypes = [c_double_p, c_double_p, C.c_int, C.c_int]
    L.oracle_cost_value.restype = C.c_double
    L.oracle_cost_value.argtypes = [C.POINTER(capi.MjpcHipTask), c_double_p, c_double_p]
    L.oracle_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, c_double_p, c_int_p]
    L.oracle_philox.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.oracle_debug_forward.argtypes = [C.c_void_p] + [c_double_p] * 4 + [C.c_double] + [c_double_p] * 5 + [c_int_p, c_int_p, c_double_p, c_double_p, c_double_p]
    L.oracle_debug_step.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.c_int, c_double_p]
    L.oracle_debug_vel_derivatives.argtypes = [C.c_void_p] + [c_double_p] * 6
    L.oracle_debug_actuation.argtypes = [C.c_void_p] + [c_double_p] * 6
    L.oracle_debug_constraints.argtypes = [C.c_void_p, c_double_p, c_double_p, c_double_p, C.c_int] + [c_double_p] * 5; L.oracle_debug_constraints.restype = C.c_int
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(c_double_p) if a is not None else None


def spline_sample(times, values, interp, t):
    times = np.ascontiguousarray(times, float); values = np.ascontiguousarray(values, float)
    P = len(times); dim = values.size // max(P, 1) if P else values.shape[-1] if values.ndim else 1
    out = np.zeros(dim)
    lib().oracle_spline_sample(_dp(times), _dp(values), P, dim, interp, float(t), _dp(out))
    return out


def norm(x, params, typ):
    x = np.ascontiguousarray(x, float)
    p = np.ascontiguousarray(list(params) + [0.0, 0.0], float)
    return lib().oracle_norm(_dp(x), _dp(p), len(x), int(typ))


def noise(seed, stream, i0, n, P, nu, sigma2=0.0):
    eps = np.zeros((n, P, nu)); sel = np.zeros(n, np.int32)
    lib().oracle_noise(seed, stream, i0, n, P, nu, sigma2, _dp(eps), sel.ctypes.data_as(c_int_p))
    return eps, sel


class Oracle:
    def __init__(self, model: dict, task: dict):
        self.cm = capi.CModel(model, task)
        self.h = lib().oracle_create(C.byref(self.cm.c_model), C.byref(self.cm.c_task))
        self.model = model; self.task = task

    def __del__(self):
        self.close_pool()
        if getattr(self, "h", None):
            lib().oracle_destroy(self.h); self.h = None

    def open_pool(self, nthreads):
        """Persistent FIFO worker pool (threads + per-worker data live across plan() calls, like the reference's
        ThreadPool); plan() uses it instead of a one-shot pool until close_pool()."""
        self.close_pool()
        self._pool = lib().oracle_pool_create(self.h, int(nthreads))
        return self

    def close_pool(self):
        if getattr(self, "_pool", None):
            lib().oracle_pool_destroy(self._pool)
        self._pool = None

    def set_task(self, task: dict):
        self.task = task
        t = self.cm.make_task(task)
        lib().oracle_set_task(self.h, C.byref(t))

    def cost(self, residual):
        r = np.ascontiguousarray(residual, float)
        terms = np.zeros(self.task["num_term"])
        t = self.cm.make_task(self.task)
        v = lib().oracle_cost_value(C.byref(t), _dp(r), _dp(terms))
        return v, terms

    def plan(self, state, mocap, time, knot_times, knot_values, interp, N, H, sigma=(0.1, 0.0), noise_eps=None,
             noise_sel=None, seed=0, stream=0, nthreads=1, candidate_offset=0, num_local=None, noise_std=None, nominal_index=0, candidate_knots=None, xfrc_std=0.0, xfrc_rate=0.0):
        m = self.model; t = self.task
        inp = capi.make_plan_input(self.cm, state, mocap, time, knot_times, knot_values, interp, N, H, sigma,
                                   noise_eps, noise_sel, seed, stream, candidate_offset, num_local,
                                   noise_std=noise_std, nominal_index=nominal_index, candidate_knots=candidate_knots,
                                   xfrc_std=xfrc_std, xfrc_rate=xfrc_rate)
        nl = inp.num_local
        ds = m["nq"] + m["nv"] + m["na"]; nu = m["nu"]; nr = t["num_residual"]; ntr = 3 * t["num_trace"]
        P = inp.num_spline_points
        out = dict(returns=np.zeros(nl), failure=np.zeros(nl, np.int32), states=np.zeros((nl, H, ds)),
                   actions=np.zeros((nl, H, nu)), times=np.zeros((nl, H)), residual=np.zeros((nl, H, nr)),
                   costs=np.zeros((nl, H)), trace=np.zeros((nl, H, max(ntr, 1))), knots=np.zeros((nl, P, nu)))
        o = OPlanOutput()
        for k in ["returns", "states", "actions", "times", "residual", "costs", "trace", "knots"]:
            setattr(o, k, _dp(out[k]))
        o.failure = out["failure"].ctypes.data_as(c_int_p)
        if getattr(self, "_pool", None):
            lib().oracle_pool_plan(self._pool, C.byref(inp), C.byref(o))
        else:
            lib().oracle_plan(self.h, C.byref(inp), C.byref(o), int(nthreads))
        out["winner"] = o.winner; out["unsupported"] = o.unsupported
        out["trace"] = out["trace"][:, :, :ntr]
        return out

    def forward(self, qpos, qvel=None, ctrl=None, mocap=None, time=0.0):
        m = self.model
        nv, nb = m["nv"], m["nbody"]
        qpos = np.ascontiguousarray(qpos, float)
        qvel = np.ascontiguousarray(qvel if qvel is not None else np.zeros(nv), float)
        ctrl = np.ascontiguousarray(ctrl if ctrl is not None else np.zeros(max(m["nu"], 1)), float)
        mocap_a = np.ascontiguousarray(mocap, float) if mocap is not None else None
        res = dict(qacc=np.zeros(nv), qM=np.zeros((nv, nv)), xpos=np.zeros((nb, 3)),
                   sensordata=np.zeros(self.task["num_residual"]), contact_dist=np.zeros(256),
                   efc_force=np.zeros(1024), geom_xpos=np.zeros((m["ngeom"], 3)),
                   extra=np.zeros(3 * nv + 2 + 6 * nb))
        ncon = C.c_int(0); nefc = C.c_int(0)
        w = lib().oracle_debug_forward(self.h, _dp(qpos), _dp(qvel), _dp(ctrl), _dp(mocap_a), float(time),
                                       _dp(res["qacc"]), _dp(res["qM"]), _dp(res["xpos"]), _dp(res["sensordata"]),
                                       _dp(res["contact_dist"]), C.byref(ncon), C.byref(nefc), _dp(res["efc_force"]),
                                       _dp(res["geom_xpos"]), _dp(res["extra"]))
        res["ncon"] = ncon.value; res["nefc"] = nefc.value; res["warning"] = w
        res["contact_dist"] = res["contact_dist"][:ncon.value]; res["efc_force"] = res["efc_force"][:nefc.value]
        e = res.pop("extra")
        res["qacc_smooth"] = e[:nv]; res["qfrc_bias"] = e[nv:2 * nv]; res["qfrc_constraint"] = e[2 * nv:3 * nv]
        res["solver_iter"] = int(e[3 * nv]); res["unsupported"] = int(e[3 * nv + 1])
        res["subtree_com"] = e[3 * nv + 2:3 * nv + 2 + 3 * nb].reshape(nb, 3)
        res["subtree_linvel"] = e[3 * nv + 2 + 3 * nb:].reshape(nb, 3)
        return res

    def vel_derivatives(self, qpos, qvel):
        """d qfrc_bias / d qvel, -d qfrc_fluid / d qvel (what the implicit integrators add to M, per unit h), qfrc_bias, qfrc_passive"""
        nv = self.model["nv"]
        qpos = np.ascontiguousarray(qpos, float); qvel = np.ascontiguousarray(qvel, float)
        out = dict(dbias=np.zeros((nv, nv)), dfluid=np.zeros((nv, nv)), qfrc_bias=np.zeros(nv), qfrc_passive=np.zeros(nv))
        out["warning"] = lib().oracle_debug_vel_derivatives(self.h, _dp(qpos), _dp(qvel), _dp(out["dbias"]), _dp(out["dfluid"]),
                                                            _dp(out["qfrc_bias"]), _dp(out["qfrc_passive"]))
        return out

    def actuation(self, qpos, qvel=None, ctrl=None, act=None):
        """actuator_force [nu] and qfrc_actuator [nv] at the given state"""
        m = self.model
        qpos = np.ascontiguousarray(qpos, float); qvel = np.ascontiguousarray(qvel if qvel is not None else np.zeros(m["nv"]), float)
        ctrl = np.ascontiguousarray(ctrl if ctrl is not None else np.zeros(max(m["nu"], 1)), float)
        act_a = np.ascontiguousarray(act, float) if act is not None else None
        force, qfrc = np.zeros(max(m["nu"], 1)), np.zeros(m["nv"])
        lib().oracle_debug_actuation(self.h, _dp(qpos), _dp(qvel), _dp(ctrl), _dp(act_a), _dp(force), _dp(qfrc))
        return force[:m["nu"]], qfrc

    def constraints(self, qpos, qvel=None, mocap=None, cap=256):
        """the constraint rows at (qpos, qvel): efc_J, efc_pos, efc_diagApprox, efc_R, efc_aref"""
        nv = self.model["nv"]
        qpos = np.ascontiguousarray(qpos, float); qvel = np.ascontiguousarray(qvel if qvel is not None else np.zeros(nv), float)
        out = dict(J=np.zeros((cap, nv)), pos=np.zeros(cap), diag=np.zeros(cap), R=np.zeros(cap), aref=np.zeros(cap))
        mocap_a = np.ascontiguousarray(mocap, float) if mocap is not None else None
        n = lib().oracle_debug_constraints(self.h, _dp(qpos), _dp(qvel), _dp(mocap_a), cap, _dp(out["J"]), _dp(out["pos"]), _dp(out["diag"]), _dp(out["R"]), _dp(out["aref"]))
        assert n <= cap
        return {k: v[:n] for k, v in out.items()}

    def step(self, qpos, qvel, ctrl=None, mocap=None, time=0.0, nstep=1):
        m = self.model
        qpos = np.ascontiguousarray(qpos, float).copy(); qvel = np.ascontiguousarray(qvel, float).copy()
        ctrl = np.ascontiguousarray(ctrl if ctrl is not None else np.zeros(max(m["nu"], 1)), float)
        mocap_a = np.ascontiguousarray(mocap, float) if mocap is not None else None
        t = C.c_double(time)
        energy = np.zeros((nstep, 2))
        w = lib().oracle_debug_step(self.h, _dp(qpos), _dp(qvel), _dp(ctrl), _dp(mocap_a), C.byref(t), nstep, _dp(energy))
        return qpos, qvel, t.value, energy, w
