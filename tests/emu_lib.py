"""Loader for the 1-lane emulation build of the kernel source (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

from mujoco_mpc_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "libmjpc_emu.so")
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class EmuOut(C.Structure):
    _fields_ = [("returns", c_double_p), ("failure", c_int_p), ("states", c_double_p), ("actions", c_double_p),
                ("times", c_double_p), ("residual", c_double_p), ("costs", c_double_p), ("trace", c_double_p),
                ("knots", c_double_p), ("diag", c_int_p)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        csrc = os.path.join(ROOT, "mujoco_mpc_amd", "csrc")
        srcs = [os.path.join(EMU_DIR, "emu.cpp")] + [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(".h")]
        asan = bool(os.environ.get("MJPC_EMU_ASAN"))      # memory-checked build of the kernel source: LD_PRELOAD=$(gcc -print-file-name=libasan.so) MJPC_EMU_ASAN=1 pytest ...
        so = EMU_SO[:-3] + "_asan.so" if asan else EMU_SO
        flags = ["-O1", "-g", "-fsanitize=address"] if asan else ["-O2"]
        import fcntl
        with open(os.path.join(EMU_DIR, ".build.lock"), "w") as lock:      # pytest-xdist workers: one builds, the others wait
            fcntl.flock(lock, fcntl.LOCK_EX)
            if (not os.path.exists(so)) or any(os.path.getmtime(s_) > os.path.getmtime(so) for s_ in srcs):
                tmp = so + f".{os.getpid()}.tmp"
                subprocess.check_call(["g++"] + flags + ["-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-o", tmp, os.path.join(EMU_DIR, "emu.cpp")])
                os.replace(tmp, so)
        _lib = C.CDLL(so)
        _lib.emu_plan.argtypes = [C.POINTER(capi.MjpcHipModel), C.POINTER(capi.MjpcHipTask),
                                  C.POINTER(capi.MjpcHipPlanInput), C.POINTER(EmuOut)]
    return _lib


def plan(model, task, state, mocap, time, knot_times, knot_values, interp, N, H, sigma=(0.1, 0.0), noise_eps=None,
         noise_sel=None, candidate_offset=0, num_local=None, noise_std=None, nominal_index=0, candidate_knots=None, xfrc_std=0.0,
         xfrc_rate=0.0, seed=0, stream=0):
    cm = capi.CModel(model, task)
    inp = capi.make_plan_input(cm, state, mocap, time, knot_times, knot_values, interp, N, H, sigma, noise_eps, noise_sel,
                               seed, stream, candidate_offset, num_local, noise_std=noise_std, nominal_index=nominal_index,
                               candidate_knots=candidate_knots, xfrc_std=xfrc_std, xfrc_rate=xfrc_rate)
    nl = inp.num_local
    ds = model["nq"] + model["nv"] + model["na"]; nu = model["nu"]; nr = task["num_residual"]; ntr = 3 * task["num_trace"]
    P = inp.num_spline_points
    out = dict(returns=np.zeros(nl), failure=np.zeros(nl, np.int32), states=np.zeros((nl, H, ds)),
               actions=np.zeros((nl, H, nu)), times=np.zeros((nl, H)), residual=np.zeros((nl, H, nr)),
               costs=np.zeros((nl, H)), trace=np.zeros((nl, H, max(ntr, 1))), knots=np.zeros((nl, P, nu)),
               diag=np.zeros((nl, 4), np.int32))
    o = EmuOut()
    for k in ["returns", "states", "actions", "times", "residual", "costs", "trace", "knots"]:
        setattr(o, k, out[k].ctypes.data_as(c_double_p))
    o.failure = out["failure"].ctypes.data_as(c_int_p); o.diag = out["diag"].ctypes.data_as(c_int_p)
    rc = lib().emu_plan(C.byref(cm.c_model), C.byref(cm.c_task), C.byref(inp), C.byref(o))
    assert rc > 0
    out["lds_doubles"] = rc
    out["trace"] = out["trace"][:, :, :ntr]
    return out
