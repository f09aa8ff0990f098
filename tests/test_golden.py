"""Plan-step fixtures of tests/golden/ (drop-in Trajectory dump format) against the oracle and the
kernel source on the CPU tier and against the HIP engine on the GPU tier.  The committed outputs are oracle-generated
(regression data); a real MuJoCo dump for the same inputs can replace them without touching this file."""
import glob
import os

import numpy as np
import pytest

import emu_lib
import oracle_lib as ol
from golden.make_golden import CONFIGS, inputs

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = sorted(CONFIGS)
TOL = {"c1_cartpole_16x50": 1e-9, "particle_10x11": 1e-9}      # contact-free; everything else: the north star's 1e-5 relative


def _load(name):
    z = np.load(os.path.join(HERE, "golden", name + ".npz"))
    return {k[3:]: z[k] for k in z.files if k.startswith("in_")}, {k[4:]: z[k] for k in z.files if k.startswith("out_")}


def _rel(a, b):
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-300)


def _check(name, got_returns, got_failure, got_winner, rows, knots, tol):
    _, out = _load(name)
    assert np.array_equal(got_failure != 0, out["failure"] != 0)
    assert _rel(got_returns, out["total_return"]) < tol
    assert int(got_winner) == int(out["winner"])                                        # argmin index: exact
    assert np.array_equal(knots, out["candidate_knots"])                                # candidate policies: bit-exact
    assert np.array_equal(rows["times"], out["times"]) and _rel(rows["actions"], out["actions"]) < 1e-14
    for k in ("states", "residual", "costs", "trace"):
        if out[k].size:
            assert _rel(rows[k], out[k]) < tol, k


def test_fixture_set_is_complete_and_inputs_are_reproducible():
    files = sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(HERE, "golden", "*.npz")))
    assert files == NAMES
    for name in NAMES:
        m, task, inp = inputs(name)
        stored, _ = _load(name)
        for k, v in inp.items():
            assert np.array_equal(np.asarray(v), stored[k]), (name, k)                  # the generator script made these files


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_the_fixture(name):
    m, task, inp = inputs(name)
    inp, _ = _load(name)
    N, H = int(inp["num_trajectory"]), int(inp["horizon"])
    r = ol.Oracle(m, task).plan(inp["state"], inp["mocap"] if inp["mocap"].size else None, float(inp["time"]), inp["knot_times"],
                                inp["knot_values"], int(inp["interpolation"]), N, H, sigma=tuple(inp["noise_exploration"]),
                                noise_eps=inp["noise_eps"], noise_sel=inp["noise_sel"], nthreads=4)
    w = r["winner"]
    _check(name, r["returns"], r["failure"], w, {k: r[k][w] for k in ("states", "actions", "times", "residual", "costs", "trace")}, r["knots"], 1e-12)


@pytest.mark.parametrize("name", NAMES)
def test_kernel_source_reproduces_the_fixture(name):
    m, task, _ = inputs(name)
    inp, _ = _load(name)
    N, H = int(inp["num_trajectory"]), int(inp["horizon"])
    r = emu_lib.plan(m, task, inp["state"], inp["mocap"] if inp["mocap"].size else None, float(inp["time"]), inp["knot_times"], inp["knot_values"],
                     int(inp["interpolation"]), N, H, sigma=tuple(inp["noise_exploration"]), noise_eps=inp["noise_eps"], noise_sel=inp["noise_sel"])
    w = int(np.argmin(r["returns"]))
    _check(name, r["returns"], r["failure"], w, {k: r[k][w] for k in ("states", "actions", "times", "residual", "costs", "trace")}, r["knots"],
           TOL.get(name, 1e-5))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_engine_reproduces_the_fixture(name):
    from mujoco_mpc_amd.planner import HipBackend
    m, task, _ = inputs(name)
    inp, _ = _load(name)
    N, H, P = int(inp["num_trajectory"]), int(inp["horizon"]), len(inp["knot_times"])
    be = HipBackend(m, task, max_samples=N, max_horizon=H)
    out = be.plan(state=inp["state"], mocap=inp["mocap"] if inp["mocap"].size else None, time=float(inp["time"]), knot_times=inp["knot_times"],
                  knot_values=inp["knot_values"], interpolation=int(inp["interpolation"]), num_trajectory=N, horizon=H,
                  sigma=tuple(inp["noise_exploration"]), noise_eps=inp["noise_eps"], noise_sel=inp["noise_sel"])
    knots = be.fetch_all(N, H, P)["knots"]
    _check(name, out["returns"], out["failure"], out["winner"], out, knots, TOL.get(name, 1e-5))
    be.close()
