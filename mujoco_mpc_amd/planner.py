"""ctypes wrappers of the rollout engine's C ABI (include/mjpc_hip.h): `HipBackend` = one engine (mjpc_hip_plan and friends),
`HipMultiBackend` = one planner process with one engine per GPU (mjpc_hip_multi_*).

The rollouts of mjpc/planners/sampling/planner.cc:342-380 run on the GPU behind that ABI; the host logic of the reference's
SamplingPlanner lives in C++ (csrc/planner.cc, driven from Python through cplanner.py).  There is no CPU fallback: a missing
libmjpc_hip.so raises in capi.load_engine().  (The Python restatement of the host logic that the tests cross-check the C++ planner
against is test infrastructure: tests/host_mirror.py.)
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

kZeroSpline, kLinearSpline, kCubicSpline = 0, 1, 2
kMaxTrajectoryHorizon = 512          # mjpc/trajectory.h:27
kMaxTrajectoryReference = 128        # mjpc/planners/planner.h:28 (lifted here)


class HipBackend:
    """Thin object wrapper over the C ABI (include/mjpc_hip.h)."""

    def __init__(self, model: dict, task: dict, max_samples=128, max_horizon=kMaxTrajectoryHorizon, device=0):
        self.lib = capi.load_engine()
        self.cm = capi.CModel(model, task)
        self.model = model; self.task = task
        self.h = self.lib.mjpc_hip_create(C.byref(self.cm.c_model), C.byref(self.cm.c_task), int(max_samples),
                                          int(max_horizon), int(device))
        if not self.h:
            raise RuntimeError("mjpc_hip_create failed: " + self.lib.mjpc_hip_last_error().decode())
        self.max_samples = max_samples

    def close(self):
        if getattr(self, "h", None):
            self.lib.mjpc_hip_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_task(self, task: dict):
        self.task = task
        t = self.cm.make_task(task)
        if self.lib.mjpc_hip_set_task(self.h, C.byref(t)) != 0:
            raise RuntimeError(self.lib.mjpc_hip_last_error().decode())

    def set_fetch_mode(self, summary_only: bool):
        """summary_only: plan() brings back returns / failure flags / local elite (index, return, knots) only; the winner's
        trajectory rows stay on the device until candidate() asks for them (shards of a multi-GPU plan)."""
        if self.lib.mjpc_hip_set_fetch_mode(self.h, 1 if summary_only else 0) != 0:
            raise RuntimeError(self.lib.mjpc_hip_last_error().decode())

    def _alloc_out(self, nl, H, P):
        m, t = self.model, self.task
        ds = m["nq"] + m["nv"] + m["na"]; nu = m["nu"]; nr = t["num_residual"]; ntr = 3 * t["num_trace"]
        o = dict(returns=np.zeros(nl), failure=np.zeros(nl, np.int32), states=np.zeros((H, ds)), actions=np.zeros((H, nu)),
                 times=np.zeros(H), residual=np.zeros((H, nr)), costs=np.zeros(H), trace=np.zeros((H, max(ntr, 1))),
                 winner_knots=np.zeros((P, nu)))
        c = capi.MjpcHipPlanOutput()
        for k in ["returns", "states", "actions", "times", "residual", "costs", "trace", "winner_knots"]:
            setattr(c, k, o[k].ctypes.data_as(capi.c_double_p))
        c.failure = o["failure"].ctypes.data_as(capi.c_int_p)
        return o, c, ntr

    def make_input(self, **kw):
        return capi.make_plan_input(self.cm, **kw)

    def plan(self, **kw):
        inp = self.make_input(**kw)
        return self.plan_input(inp)

    def plan_input(self, inp):
        o, c, ntr = self._alloc_out(inp.num_local, inp.horizon, inp.num_spline_points)
        rc = self.lib.mjpc_hip_plan(self.h, C.byref(inp), C.byref(c))
        if rc != 0:
            raise RuntimeError("mjpc_hip_plan failed: " + self.lib.mjpc_hip_last_error().decode())
        o["winner"] = c.winner; o["winner_return"] = c.winner_return
        o["noise_compute_time_us"] = c.noise_compute_time_us; o["rollouts_compute_time_us"] = c.rollouts_compute_time_us
        o["trace"] = o["trace"][:, :ntr]
        return o

    def plan_async(self, inp):
        if self.lib.mjpc_hip_plan_async(self.h, C.byref(inp)) != 0:
            raise RuntimeError("mjpc_hip_plan_async failed: " + self.lib.mjpc_hip_last_error().decode())

    def plan_fetch(self, inp):
        o, c, ntr = self._alloc_out(inp.num_local, inp.horizon, inp.num_spline_points)
        if self.lib.mjpc_hip_plan_fetch(self.h, C.byref(c)) != 0:
            raise RuntimeError("mjpc_hip_plan_fetch failed: " + self.lib.mjpc_hip_last_error().decode())
        o["winner"] = c.winner; o["winner_return"] = c.winner_return
        o["trace"] = o["trace"][:, :ntr]
        return o

    def candidate(self, local_index, H, P):
        o, c, ntr = self._alloc_out(1, H, P)
        if self.lib.mjpc_hip_get_candidate(self.h, int(local_index), C.byref(c)) != 0:
            raise RuntimeError(self.lib.mjpc_hip_last_error().decode())
        o["trace"] = o["trace"][:, :ntr]
        return o

    def fetch_all(self, nl, H, P):
        """Every local candidate's Trajectory arrays of the last plan (tests / GUI traces)."""
        m, t = self.model, self.task
        ds = m["nq"] + m["nv"] + m["na"]; nu = m["nu"]; nr = t["num_residual"]; ntr = 3 * t["num_trace"]
        o = dict(states=np.zeros((nl, H, ds)), actions=np.zeros((nl, H, nu)), times=np.zeros((nl, H)),
                 residual=np.zeros((nl, H, nr)), costs=np.zeros((nl, H)), trace=np.zeros((nl, H, max(ntr, 1))),
                 knots=np.zeros((nl, P, nu)), diag=np.zeros((nl, 4), np.int32))
        rc = self.lib.mjpc_hip_get_all_candidates(self.h, *[o[k].ctypes.data_as(capi.c_double_p) for k in
                                                         ["states", "actions", "times", "residual", "costs", "trace", "knots"]],
                                               o["diag"].ctypes.data_as(capi.c_int_p))
        if rc != 0:
            raise RuntimeError(self.lib.mjpc_hip_last_error().decode())
        o["trace"] = o["trace"][:, :, :ntr]
        return o

    def lds_bytes(self):
        return self.lib.mjpc_hip_lds_bytes(self.h)

    def dense_tier(self):
        """(LDS bytes of the two-candidates-per-CU tier or 0, whether the last plan ran on it)."""
        used = C.c_int(0)
        n = self.lib.mjpc_hip_dense_tier(self.h, C.byref(used))
        return n, bool(used.value)

    def dense_capacity(self):
        """(rows, contacts, hot tables in LDS) of the dense tier; (0, 0, False) without one.  Diagnostics."""
        a = C.c_int(0); b = C.c_int(0); h = C.c_int(0)
        self.lib.mjpc_hip_debug_dense_capacity(self.h, C.byref(a), C.byref(b), C.byref(h))
        return a.value, b.value, bool(h.value)

    def kernel_time(self):
        a = C.c_double(0); b = C.c_double(0)
        n = self.lib.mjpc_hip_kernel_time(self.h, C.byref(a), C.byref(b))
        return n, a.value, b.value


class HipMultiBackend:
    """One planner process, one rollout engine per GPU (mjpc_hip_multi_*, include/mjpc_hip.h): plan() block-partitions the
    global candidate batch over the engines, picks the elite across them and copies the winner's trajectory from its owner.
    devices: HIP ordinals, repeats allowed (several engines on one GPU: 1-GPU rehearsal)."""

    def __init__(self, model: dict, task: dict, devices, max_samples=128, max_horizon=kMaxTrajectoryHorizon):
        self.lib = capi.load_engine()
        self.cm = capi.CModel(model, task)
        self.model = model; self.task = task
        dv = (C.c_int * len(devices))(*[int(x) for x in devices])
        self.h = self.lib.mjpc_hip_multi_create(C.byref(self.cm.c_model), C.byref(self.cm.c_task), int(max_samples), int(max_horizon),
                                                len(devices), dv)
        if not self.h:
            raise RuntimeError("mjpc_hip_multi_create failed: " + self.lib.mjpc_hip_last_error().decode())
        self.h = C.c_void_p(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mjpc_hip_multi_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _alloc_out = HipBackend._alloc_out

    def plan(self, **kw):
        inp = capi.make_plan_input(self.cm, **kw)
        o, c, ntr = self._alloc_out(inp.num_trajectory, inp.horizon, inp.num_spline_points)
        if self.lib.mjpc_hip_multi_plan(self.h, C.byref(inp), C.byref(c)) != 0:
            raise RuntimeError("mjpc_hip_multi_plan failed: " + self.lib.mjpc_hip_last_error().decode())
        o["winner"] = c.winner; o["winner_return"] = c.winner_return
        o["noise_compute_time_us"] = c.noise_compute_time_us; o["rollouts_compute_time_us"] = c.rollouts_compute_time_us
        o["trace"] = o["trace"][:, :ntr]
        return o

    def candidate(self, index, H, P):
        o, c, ntr = self._alloc_out(1, H, P)
        if self.lib.mjpc_hip_multi_get_candidate(self.h, int(index), C.byref(c)) != 0:
            raise RuntimeError(self.lib.mjpc_hip_last_error().decode())
        o["trace"] = o["trace"][:, :ntr]
        return o

    def knots(self, N, P):
        k = np.zeros((N, P, self.model["nu"]))
        if self.lib.mjpc_hip_multi_get_knots(self.h, k.ctypes.data_as(capi.c_double_p)) != 0:
            raise RuntimeError(self.lib.mjpc_hip_last_error().decode())
        return k

    def traces(self, N, H):
        ntr = 3 * self.task["num_trace"]
        t = np.zeros((N, H, max(ntr, 1)))
        if self.lib.mjpc_hip_multi_get_traces(self.h, t.ctypes.data_as(capi.c_double_p)) != 0:
            raise RuntimeError(self.lib.mjpc_hip_last_error().decode())
        return t[:, :, :ntr] if ntr else t[:, :, :0]
