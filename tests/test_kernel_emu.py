"""CPU tier: the product kernel source (mujoco_mpc_amd/csrc/core.h) compiled in 1-lane emulation mode vs the
oracle, plus ABI checks that need no GPU.  The emulation library is test infrastructure only."""
import ctypes
import os
import re

import numpy as np
import pytest

import emu_lib
import oracle_lib as ol
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.modelgen import cartpole, humanoid_track, particle, quadruped

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


@pytest.mark.parametrize("name,P,H,N,sigma,tol", [("particle", 11, 26, 6, (0.3, 0.0), 1e-12), ("cartpole", 10, 50, 8, (0.5, 0.0), 1e-12),
                                                   ("quadruped", 3, 30, 6, (0.04, 0.0), 1e-5),
                                                   ("humanoid_track", 16, 30, 4, (0.15, 0.0), 1e-5)])
def test_kernel_source_matches_oracle(name, P, H, N, sigma, tol):
    m, task, d = {"particle": particle, "cartpole": cartpole, "quadruped": quadruped, "humanoid_track": humanoid_track}[name]()
    o = ol.Oracle(m, task)
    kt = np.linspace(0, (H - 1) * m["timestep"], P); kv = np.random.default_rng(0).uniform(-0.3, 0.3, (P, m["nu"]))
    eps, sel = ol.noise(1, 0, 0, N, P, m["nu"])
    mocap = d["mocap"] if len(d["mocap"]) else None
    a = o.plan(d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=sigma, noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], mocap, 0.0, kt, kv, 2, N, H, sigma=sigma, noise_eps=eps, noise_sel=sel)
    assert np.array_equal(a["knots"], b["knots"]) and np.array_equal(a["times"], b["times"]) and np.array_equal(a["actions"], b["actions"])
    for k in ("states", "residual", "costs", "trace", "returns"):
        assert _rel(b[k], a[k]) < tol, k
    assert int(np.argmin(b["returns"])) == a["winner"]
    assert b["lds_doubles"] * 8 <= 160 * 1024          # per-candidate state must fit one CU's LDS


def test_cross_entropy_noise_mode_in_oracle_and_kernel_source():
    """ABI extension for the Cross-Entropy planner (cross_entropy/planner.cc:340-415): absolute per-parameter std, every
    candidate perturbed except `nominal_index`."""
    m, task, d = cartpole()
    P, H, N = 4, 20, 7
    kt = np.linspace(0, 0.19, P); kv = np.random.default_rng(3).uniform(-0.2, 0.2, (P, 1))
    eps, _ = ol.noise(5, 0, 0, N, P, 1)
    std = np.array([0.05, 0.4, 3.0, 0.0])
    o = ol.Oracle(m, task)
    a = o.plan(d["state"], None, 0.0, kt, kv, 2, N, H, noise_eps=eps, noise_std=std, nominal_index=N - 1)
    expect = np.clip(kv[None] + std[None, :, None] * eps.reshape(N, P, 1), -1.0, 1.0)
    expect[N - 1] = kv
    assert np.array_equal(a["knots"], expect)
    assert np.any(a["knots"][0] != kv) and np.any(np.abs(a["knots"]) == 1.0)         # candidate 0 is perturbed; clamping is hit
    b = emu_lib.plan(m, task, d["state"], None, 0.0, kt, kv, 2, N, H, noise_eps=eps, noise_std=std, nominal_index=N - 1)
    assert np.array_equal(a["knots"], b["knots"])
    assert _rel(b["returns"], a["returns"]) < 1e-12


def test_engine_library_exports_every_declared_symbol():
    """libmjpc_hip.so loads without a GPU and exports exactly what include/mjpc_hip.h declares."""
    import __graft_entry__ as g
    so = g.build_engine()
    lib = ctypes.CDLL(so)
    hdr = open(os.path.join(ROOT, "include", "mjpc_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(mjpc_hip_[a-z_]+)\s*\(", hdr)))
    assert declared == sorted(capi.EXPORTED_SYMBOLS)
    for sym in declared:
        assert hasattr(lib, sym), sym
    lib.mjpc_hip_version.restype = ctypes.c_int
    assert lib.mjpc_hip_version() == 1


def test_ctypes_structs_match_header_field_order():
    hdr = open(os.path.join(ROOT, "include", "mjpc_hip.h")).read()
    body = hdr[hdr.index("typedef struct MjpcHipModel {"):hdr.index("} MjpcHipModel;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split(";"):
        stmt = stmt.replace("typedef struct MjpcHipModel {", "").strip()
        if not stmt:
            continue
        stmt = re.sub(r"^(const\s+)?(int|double)\s+", "", stmt)
        for part in stmt.split(","):
            n = part.strip().lstrip("*").strip()
            n = re.sub(r"\[\d+\]", "", n)
            if n:
                names.append(n)
    assert names == [f[0] for f in capi.MjpcHipModel._fields_]


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mujoco_mpc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace("the CPU oracle under oracle/", ""), f
