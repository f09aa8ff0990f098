"""ctypes view of the C++ host planner (include/mjpc_hip_planner.h via include/mjpc_hip_planner_c.h).

The C++ classes `mjpc_hip::TimeSpline` / `mjpc_hip::SamplingPlanner` in libmjpc_hip.so are the product's host side
(the reference's planner is compiled C++: mjpc/planners/sampling/planner.cc); this module only lets Python drive
them.  There is no Python fall-back: a missing library raises in capi.load_engine().
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
_ERR_CB = C.CFUNCTYPE(None, C.c_char_p)

PLANNER_C_SYMBOLS = [
    "mjpc_planner_set_error_handler",
    "mjpc_spline_create", "mjpc_spline_destroy", "mjpc_spline_size", "mjpc_spline_add_node", "mjpc_spline_sample",
    "mjpc_spline_discard_before", "mjpc_spline_clear", "mjpc_spline_set_interpolation",
    "mjpc_planner_create", "mjpc_planner_create_sharded", "mjpc_planner_destroy", "mjpc_planner_reset", "mjpc_planner_set_state", "mjpc_planner_set_task",
    "mjpc_planner_optimize_policy", "mjpc_planner_nominal_trajectory", "mjpc_planner_action_from_policy",
    "mjpc_planner_optimize_policy_candidates", "mjpc_planner_candidate_score", "mjpc_planner_action_from_candidate_policy",
    "mjpc_planner_copy_candidate_to_policy", "mjpc_planner_winner", "mjpc_planner_improvement", "mjpc_planner_num_parameters",
    "mjpc_planner_set_seed", "mjpc_planner_set_num_trajectory", "mjpc_planner_set_noise", "mjpc_planner_returns",
    "mjpc_planner_policy", "mjpc_planner_best_trajectory", "mjpc_planner_timings",
    "mjpc_cem_create", "mjpc_cem_destroy", "mjpc_cem_reset", "mjpc_cem_set_state", "mjpc_cem_set_seed", "mjpc_cem_set_noise",
    "mjpc_cem_optimize_policy", "mjpc_cem_nominal_trajectory", "mjpc_cem_action_from_policy", "mjpc_cem_improvement",
    "mjpc_cem_returns", "mjpc_cem_variance", "mjpc_cem_policy", "mjpc_cem_best_trajectory",
    "mjpc_testspeed_run",
    "mjpc_robust_create", "mjpc_robust_destroy", "mjpc_robust_reset", "mjpc_robust_set_state", "mjpc_robust_set_seed",
    "mjpc_robust_optimize_policy", "mjpc_robust_action_from_policy", "mjpc_robust_last", "mjpc_robust_delegate",
]


class PlannerError(RuntimeError):
    """Raised where the reference would abort through mju_error (planner.cc:69-72, spline.cc:205-208)."""


_lib = None
_pending: list[str] = []


@_ERR_CB
def _on_error(msg):
    _pending.append(msg.decode() if msg else "unknown")


def _check():
    if _pending:
        msg = _pending[0]
        _pending.clear()
        raise PlannerError(msg)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_double_p)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = capi.load_engine()
    vp, d, i = C.c_void_p, C.c_double, C.c_int
    sig = {
        "mjpc_planner_set_error_handler": (None, [_ERR_CB]),
        "mjpc_spline_create": (vp, [i, i]), "mjpc_spline_destroy": (None, [vp]), "mjpc_spline_size": (i, [vp]),
        "mjpc_spline_add_node": (None, [vp, d, c_double_p]), "mjpc_spline_sample": (None, [vp, d, c_double_p]),
        "mjpc_spline_discard_before": (i, [vp, d]), "mjpc_spline_clear": (None, [vp]),
        "mjpc_spline_set_interpolation": (None, [vp, i]),
        "mjpc_planner_create": (vp, [C.POINTER(capi.MjpcHipModel), C.POINTER(capi.MjpcHipTask), c_double_p, i, i, i, i, i, i, i]),
        "mjpc_planner_create_sharded": (vp, [C.POINTER(capi.MjpcHipModel), C.POINTER(capi.MjpcHipTask), c_double_p, i, i, i, i, i, i, i, C.POINTER(C.c_int)]),
        "mjpc_planner_destroy": (None, [vp]), "mjpc_planner_reset": (None, [vp, i, c_double_p]),
        "mjpc_planner_set_state": (None, [vp, c_double_p, c_double_p, c_double_p, d]),
        "mjpc_planner_set_task": (None, [vp, C.POINTER(capi.MjpcHipTask)]),
        "mjpc_planner_optimize_policy": (None, [vp, i]), "mjpc_planner_nominal_trajectory": (None, [vp, i]),
        "mjpc_planner_action_from_policy": (None, [vp, c_double_p, d, i]),
        "mjpc_planner_optimize_policy_candidates": (i, [vp, i, i]), "mjpc_planner_candidate_score": (d, [vp, i]),
        "mjpc_planner_action_from_candidate_policy": (None, [vp, c_double_p, i, d]),
        "mjpc_planner_copy_candidate_to_policy": (None, [vp, i]), "mjpc_planner_winner": (i, [vp]),
        "mjpc_planner_improvement": (d, [vp]), "mjpc_planner_num_parameters": (i, [vp]),
        "mjpc_planner_set_seed": (None, [vp, C.c_ulonglong, C.c_ulonglong]), "mjpc_planner_set_num_trajectory": (None, [vp, i]),
        "mjpc_planner_set_noise": (None, [vp, c_double_p, c_int_p]), "mjpc_planner_returns": (None, [vp, c_double_p, i]),
        "mjpc_planner_policy": (i, [vp, i, c_double_p, c_double_p]),
        "mjpc_planner_best_trajectory": (i, [vp] + [c_double_p] * 7 + [c_int_p]),
        "mjpc_planner_timings": (None, [vp, c_double_p, c_double_p, c_double_p]),
        "mjpc_cem_create": (vp, [C.POINTER(capi.MjpcHipModel), C.POINTER(capi.MjpcHipTask), d, d, i, i, i, i, i, i, i]),
        "mjpc_cem_destroy": (None, [vp]), "mjpc_cem_reset": (None, [vp, i, c_double_p]),
        "mjpc_cem_set_state": (None, [vp, c_double_p, c_double_p, c_double_p, d]),
        "mjpc_cem_set_seed": (None, [vp, C.c_ulonglong, C.c_ulonglong]), "mjpc_cem_set_noise": (None, [vp, c_double_p]),
        "mjpc_cem_optimize_policy": (None, [vp, i]), "mjpc_cem_nominal_trajectory": (None, [vp, i]),
        "mjpc_cem_action_from_policy": (None, [vp, c_double_p, d, i]), "mjpc_cem_improvement": (d, [vp]),
        "mjpc_cem_returns": (None, [vp, c_double_p, i]), "mjpc_cem_variance": (None, [vp, c_double_p, i]),
        "mjpc_cem_policy": (i, [vp, c_double_p, c_double_p]),
        "mjpc_cem_best_trajectory": (i, [vp, c_double_p, c_double_p, c_double_p, c_double_p]),
        "mjpc_robust_create": (vp, [C.POINTER(capi.MjpcHipModel), C.POINTER(capi.MjpcHipTask), c_double_p, i, i, i, i, i, d, d, i, i, i]),
        "mjpc_robust_destroy": (None, [vp]), "mjpc_robust_reset": (None, [vp, i]),
        "mjpc_robust_set_state": (None, [vp, c_double_p, c_double_p, c_double_p, d]),
        "mjpc_robust_set_seed": (None, [vp, C.c_ulonglong, C.c_ulonglong, C.c_ulonglong]),
        "mjpc_robust_optimize_policy": (None, [vp, i]), "mjpc_robust_action_from_policy": (None, [vp, c_double_p, d]),
        "mjpc_robust_last": (None, [vp, c_int_p, c_double_p, c_double_p]), "mjpc_robust_delegate": (vp, [vp]),
        "mjpc_testspeed_run": (d, [C.POINTER(capi.MjpcHipModel), C.POINTER(capi.MjpcHipTask), vp, i, c_double_p, c_double_p, d, i, i, d, i,
                                   c_double_p, c_double_p, i, d, c_double_p]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res; f.argtypes = args
    L.mjpc_planner_set_error_handler(_on_error)
    _lib = L
    return L


class TimeSpline:
    """mjpc_hip::TimeSpline (C++) — same surface as mjpc::spline::TimeSpline (spline.h:41-276)."""

    def __init__(self, dim=0, interpolation=0):
        self._L = lib()
        self.dim_ = int(dim)
        self._h = C.c_void_p(self._L.mjpc_spline_create(self.dim_, int(interpolation)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.mjpc_spline_destroy(self._h); self._h = None

    def Size(self): return self._L.mjpc_spline_size(self._h)
    def Dim(self): return self.dim_
    def SetInterpolation(self, interpolation): self._L.mjpc_spline_set_interpolation(self._h, int(interpolation))
    def Clear(self): self._L.mjpc_spline_clear(self._h)

    def AddNode(self, time, values=None):
        v = None if values is None else np.ascontiguousarray(values, dtype=np.float64)
        self._L.mjpc_spline_add_node(self._h, float(time), _dp(v))
        _check()

    def DiscardBefore(self, time): return self._L.mjpc_spline_discard_before(self._h, float(time))

    def Sample(self, time):
        out = np.zeros(self.dim_)
        self._L.mjpc_spline_sample(self._h, float(time), _dp(out))
        _check()
        return out


class Trajectory:
    pass


class SamplingPlanner:
    """mjpc_hip::SamplingPlanner (C++) driven from Python; method names follow planners/sampling/planner.h:51-112."""

    def __init__(self):
        self._L = lib()
        self._h = None
        self._noise = None

    def Initialize(self, model: dict, task: dict, numerics: dict | None = None, max_samples=128, max_horizon=512, device=0, devices=None):
        """devices: list of HIP ordinals to shard every plan step's candidate batch over (one engine each); None = `device` only."""
        numerics = numerics or {}
        self.cm = capi.CModel(model, task)
        se = numerics.get("sampling_exploration", 0.1)
        se = list(se) if isinstance(se, (list, tuple)) else [se]
        self._expl = np.array([float(se[0]), float(se[1]) if len(se) > 1 else 0.0])
        self.nu = int(model["nu"]); self.ns = int(model["nq"] + model["nv"] + model["na"])
        self.nr = int(task["num_residual"]); self.ntrace = int(task["num_trace"])
        self.max_samples, self.max_horizon = int(max_samples), int(max_horizon)
        self.close()
        args = (C.byref(self.cm.c_model), C.byref(self.cm.c_task), _dp(self._expl),
                int(numerics.get("sampling_trajectories", 10)), int(numerics.get("sampling_representation", 2)),
                int(numerics.get("sampling_sliding_plan", 0)), int(numerics.get("sampling_spline_points", 512)),
                self.max_samples, self.max_horizon)
        if devices is None:
            h = self._L.mjpc_planner_create(*args, int(device))
        else:
            dv = (C.c_int * len(devices))(*[int(x) for x in devices])
            h = self._L.mjpc_planner_create_sharded(*args, len(devices), dv)
        self._h = C.c_void_p(h)
        _check()

    def Allocate(self):        # done inside mjpc_planner_create (Initialize + Allocate)
        pass

    def close(self):
        if self._h:
            self._L.mjpc_planner_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Reset(self, horizon=0, initial_repeated_action=None):
        a = None if initial_repeated_action is None else np.ascontiguousarray(initial_repeated_action, dtype=np.float64)
        self._L.mjpc_planner_reset(self._h, int(horizon or 0), _dp(a))

    def SetState(self, state, mocap=None, userdata=None, time=0.0):
        s = np.ascontiguousarray(state, dtype=np.float64)
        m = None if mocap is None else np.ascontiguousarray(mocap, dtype=np.float64)
        u = None if userdata is None else np.ascontiguousarray(userdata, dtype=np.float64)
        self._L.mjpc_planner_set_state(self._h, _dp(s), _dp(m), _dp(u), float(time))

    def SetTask(self, task: dict):
        t = self.cm.make_task(task)
        self._L.mjpc_planner_set_task(self._h, C.byref(t)); _check()

    def set_seed(self, seed, plan_iter=0): self._L.mjpc_planner_set_seed(self._h, int(seed), int(plan_iter))
    def set_num_trajectory(self, n): self._L.mjpc_planner_set_num_trajectory(self._h, int(n))

    def set_noise(self, eps, sel):
        if eps is None:
            self._noise = None
            self._L.mjpc_planner_set_noise(self._h, None, None)
            return
        e = np.ascontiguousarray(eps, dtype=np.float64); s = np.ascontiguousarray(sel, dtype=np.int32)
        self._noise = (e, s)
        self._L.mjpc_planner_set_noise(self._h, _dp(e), s.ctypes.data_as(c_int_p))

    def OptimizePolicy(self, horizon): self._L.mjpc_planner_optimize_policy(self._h, int(horizon)); _check()
    def NominalTrajectory(self, horizon): self._L.mjpc_planner_nominal_trajectory(self._h, int(horizon)); _check()

    def OptimizePolicyCandidates(self, ncandidates, horizon):
        n = self._L.mjpc_planner_optimize_policy_candidates(self._h, int(ncandidates), int(horizon)); _check()
        return n

    def CandidateScore(self, candidate): return self._L.mjpc_planner_candidate_score(self._h, int(candidate))
    def CopyCandidateToPolicy(self, candidate): self._L.mjpc_planner_copy_candidate_to_policy(self._h, int(candidate)); _check()

    def ActionFromCandidatePolicy(self, candidate, time):
        a = np.zeros(self.nu)
        self._L.mjpc_planner_action_from_candidate_policy(self._h, _dp(a), int(candidate), float(time)); _check()
        return a

    def ActionFromPolicy(self, time, use_previous=False):
        a = np.zeros(self.nu)
        self._L.mjpc_planner_action_from_policy(self._h, _dp(a), float(time), int(bool(use_previous))); _check()
        return a

    @property
    def winner(self): return self._L.mjpc_planner_winner(self._h)
    @property
    def improvement(self): return self._L.mjpc_planner_improvement(self._h)
    def NumParameters(self): return self._L.mjpc_planner_num_parameters(self._h)

    def returns(self, n):
        out = np.zeros(int(n)); self._L.mjpc_planner_returns(self._h, _dp(out), int(n)); return out

    def policy_knots(self, previous=False):
        P = self._L.mjpc_planner_policy(self._h, int(previous), None, None)
        t = np.zeros(max(P, 1)); v = np.zeros((max(P, 1), self.nu))
        self._L.mjpc_planner_policy(self._h, int(previous), _dp(t), _dp(v))
        return t[:P], v[:P]

    def timings(self):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        self._L.mjpc_planner_timings(self._h, C.byref(a), C.byref(b), C.byref(c))
        return dict(noise_us=a.value, rollouts_us=b.value, policy_update_us=c.value)

    def BestTrajectory(self):
        Hm = self.max_horizon
        st = np.zeros((Hm, self.ns)); ac = np.zeros((Hm, self.nu)); ti = np.zeros(Hm); re = np.zeros((Hm, max(self.nr, 1)))
        co = np.zeros(Hm); tr = np.zeros((Hm, 3 * max(self.ntrace, 1))); tot = C.c_double(); fail = C.c_int()
        H = self._L.mjpc_planner_best_trajectory(self._h, _dp(st), _dp(ac), _dp(ti), _dp(re), _dp(co), _dp(tr), C.byref(tot), C.byref(fail))
        if H == 0:
            return None
        t = Trajectory()
        t.horizon = H; t.states = st.ravel()[:H * self.ns].reshape(H, self.ns); t.actions = ac.ravel()[:H * self.nu].reshape(H, self.nu)
        t.times = ti[:H]; t.residual = re.ravel()[:H * self.nr].reshape(H, self.nr); t.costs = co[:H]
        t.trace = tr.ravel()[:H * 3 * self.ntrace].reshape(H, 3 * self.ntrace)
        t.total_return = tot.value; t.failure = bool(fail.value)
        return t


class CrossEntropyPlanner:
    """mjpc_hip::CrossEntropyPlanner (C++) driven from Python; names follow planners/cross_entropy/planner.h:32-147."""

    def __init__(self):
        self._L = lib()
        self._h = None
        self._noise = None

    def Initialize(self, model: dict, task: dict, numerics: dict | None = None, max_samples=128, max_horizon=512, device=0):
        numerics = numerics or {}
        self.cm = capi.CModel(model, task)
        self.nu = int(model["nu"]); self.ns = int(model["nq"] + model["nv"] + model["na"])
        self.max_samples, self.max_horizon = int(max_samples), int(max_horizon)
        self.P = int(numerics.get("sampling_spline_points", 512))
        self.close()
        h = self._L.mjpc_cem_create(C.byref(self.cm.c_model), C.byref(self.cm.c_task), float(numerics.get("sampling_exploration", 0.1)),
                                    float(numerics.get("std_min", 0.1)), int(numerics.get("sampling_trajectories", 10)),
                                    int(numerics.get("n_elite", -1)), int(numerics.get("sampling_representation", 0)), self.P,
                                    self.max_samples, self.max_horizon, int(device))
        self._h = C.c_void_p(h)
        _check()

    def Allocate(self):
        pass

    def close(self):
        if self._h:
            self._L.mjpc_cem_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Reset(self, horizon=0, initial_repeated_action=None):
        a = None if initial_repeated_action is None else np.ascontiguousarray(initial_repeated_action, dtype=np.float64)
        self._L.mjpc_cem_reset(self._h, int(horizon or 0), _dp(a))

    def SetState(self, state, mocap=None, userdata=None, time=0.0):
        s = np.ascontiguousarray(state, dtype=np.float64)
        m = None if mocap is None else np.ascontiguousarray(mocap, dtype=np.float64)
        u = None if userdata is None else np.ascontiguousarray(userdata, dtype=np.float64)
        self._L.mjpc_cem_set_state(self._h, _dp(s), _dp(m), _dp(u), float(time))

    def set_seed(self, seed, plan_iter=0): self._L.mjpc_cem_set_seed(self._h, int(seed), int(plan_iter))

    def set_noise(self, eps):
        self._noise = None if eps is None else np.ascontiguousarray(eps, dtype=np.float64)
        self._L.mjpc_cem_set_noise(self._h, _dp(self._noise))

    def OptimizePolicy(self, horizon): self._L.mjpc_cem_optimize_policy(self._h, int(horizon)); _check()
    def NominalTrajectory(self, horizon): self._L.mjpc_cem_nominal_trajectory(self._h, int(horizon)); _check()

    def ActionFromPolicy(self, time, use_previous=False):
        a = np.zeros(self.nu)
        self._L.mjpc_cem_action_from_policy(self._h, _dp(a), float(time), int(bool(use_previous))); _check()
        return a

    @property
    def improvement(self): return self._L.mjpc_cem_improvement(self._h)

    def returns(self, n):
        out = np.zeros(int(n)); self._L.mjpc_cem_returns(self._h, _dp(out), int(n)); return out

    def variance(self):
        out = np.zeros(self.P * self.nu); self._L.mjpc_cem_variance(self._h, _dp(out), out.size); return out

    def policy_knots(self):
        P = self._L.mjpc_cem_policy(self._h, None, None)
        t = np.zeros(max(P, 1)); v = np.zeros((max(P, 1), self.nu))
        self._L.mjpc_cem_policy(self._h, _dp(t), _dp(v))
        return t[:P], v[:P]

    def BestTrajectory(self):
        Hm = self.max_horizon
        st = np.zeros((Hm, self.ns)); ac = np.zeros((Hm, self.nu)); co = np.zeros(Hm); tot = C.c_double()
        H = self._L.mjpc_cem_best_trajectory(self._h, _dp(st), _dp(ac), _dp(co), C.byref(tot))
        t = Trajectory()
        t.horizon = H; t.states = st.ravel()[:H * self.ns].reshape(H, self.ns); t.actions = ac.ravel()[:H * self.nu].reshape(H, self.nu)
        t.costs = co[:H]; t.total_return = tot.value
        return t


def testspeed(planner, state, mocap=None, time0=0.0, horizon=None, steps_per_planning_iteration=1, total_time=1.0, device=0, mode=0, mode_time=0.0):
    """mjpc/testspeed.cc:44-129 (`SynchronousPlanningCost`) through the C++ harness: `planner` is a cplanner.SamplingPlanner or
    cplanner.CrossEntropyPlanner that has been Initialize()d / Reset(); the world is stepped on the HIP engine as well."""
    L = lib()
    cm = planner.cm
    m = cm.model
    nsteps = int(np.ceil(total_time / m["timestep"]))
    st = np.ascontiguousarray(state, dtype=np.float64).copy()
    mc = None if (mocap is None or m["nmocap"] == 0) else np.ascontiguousarray(mocap, dtype=np.float64).copy()
    costs = np.zeros(nsteps); out = np.zeros(6); params = np.zeros(max(int(cm.task["num_parameter"]), 1))
    kind = 1 if isinstance(planner, CrossEntropyPlanner) else 0
    total = L.mjpc_testspeed_run(C.byref(cm.c_model), C.byref(cm.c_task), planner._h, kind, _dp(st), _dp(mc), float(time0), int(horizon),
                                 int(steps_per_planning_iteration), float(total_time), int(device), _dp(costs), _dp(out), int(mode), float(mode_time), _dp(params))
    _check()
    return dict(total_cost=total, average_cost=out[0], wall_seconds=out[1], realtime_factor=out[2], plan_seconds=out[3],
                plan_steps=int(out[4]), failure=bool(out[5]), cost_per_step=costs, state=st, mocap=mc, parameters=params)


class RobustPlanner:
    """mjpc_hip::RobustPlanner (C++) driven from Python; mirrors planners/robust/robust_planner.h:31-80 over a SamplingPlanner."""

    def __init__(self):
        self._L = lib()
        self._h = None

    def Initialize(self, model: dict, task: dict, numerics: dict | None = None, max_samples=128, max_horizon=512, device=0):
        numerics = numerics or {}
        self.cm = capi.CModel(model, task)
        se = numerics.get("sampling_exploration", 0.1)
        se = list(se) if isinstance(se, (list, tuple)) else [se]
        self._expl = np.array([float(se[0]), float(se[1]) if len(se) > 1 else 0.0])
        self.nu = int(model["nu"]); self.ns = int(model["nq"] + model["nv"] + model["na"])
        self.close()
        h = self._L.mjpc_robust_create(C.byref(self.cm.c_model), C.byref(self.cm.c_task), _dp(self._expl),
                                       int(numerics.get("sampling_trajectories", 10)), int(numerics.get("sampling_representation", 2)),
                                       int(numerics.get("sampling_spline_points", 512)), int(numerics.get("robust_repetitions", 5)),
                                       int(numerics.get("robust_candidates", -1)), float(numerics.get("robust_xfrc", 0.1)),
                                       float(numerics.get("robust_xfrc_rate", 0.1)), int(max_samples), int(max_horizon), int(device))
        self._h = C.c_void_p(h)
        _check()
        # a non-owning view of the delegate for the mjpc_planner_* accessors
        self.delegate = SamplingPlanner.__new__(SamplingPlanner)
        self.delegate._L = self._L; self.delegate._h = C.c_void_p(self._L.mjpc_robust_delegate(self._h)); self.delegate._noise = None
        self.delegate.cm = self.cm; self.delegate.nu = self.nu; self.delegate.ns = self.ns
        self.delegate.nr = int(task["num_residual"]); self.delegate.ntrace = int(task["num_trace"])
        self.delegate.max_samples = int(max_samples); self.delegate.max_horizon = int(max_horizon)
        self.delegate.close = lambda: None

    def close(self):
        if self._h:
            self._L.mjpc_robust_destroy(self._h); self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def Reset(self, horizon=0): self._L.mjpc_robust_reset(self._h, int(horizon or 0))

    def SetState(self, state, mocap=None, userdata=None, time=0.0):
        s = np.ascontiguousarray(state, dtype=np.float64)
        m = None if mocap is None else np.ascontiguousarray(mocap, dtype=np.float64)
        u = None if userdata is None else np.ascontiguousarray(userdata, dtype=np.float64)
        self._L.mjpc_robust_set_state(self._h, _dp(s), _dp(m), _dp(u), float(time))

    def set_seed(self, delegate_seed, robust_seed, plan_iter=0):
        self._L.mjpc_robust_set_seed(self._h, int(delegate_seed), int(robust_seed), int(plan_iter))

    def OptimizePolicy(self, horizon): self._L.mjpc_robust_optimize_policy(self._h, int(horizon)); _check()

    def ActionFromPolicy(self, time):
        a = np.zeros(self.nu)
        self._L.mjpc_robust_action_from_policy(self._h, _dp(a), float(time)); _check()
        return a

    def last(self):
        o = np.zeros(3, np.int32)
        self._L.mjpc_robust_last(self._h, o.ctypes.data_as(c_int_p), None, None)
        nc, rep = int(o[1]), int(o[2])
        scores = np.zeros(max(nc, 1)); noisy = np.zeros(max(nc * rep, 1))
        self._L.mjpc_robust_last(self._h, o.ctypes.data_as(c_int_p), _dp(scores), _dp(noisy))
        return dict(best_candidate=int(o[0]), scores=scores[:nc], noisy_returns=noisy[:nc * rep].reshape(nc, rep) if nc else noisy[:0])
