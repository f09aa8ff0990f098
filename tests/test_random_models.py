"""Fuzz parity: random articulated models (tests/random_models.py) through the CPU oracle vs the product kernel source (1-lane
emulation, CPU tier) and vs the HIP engine through the C ABI (GPU tier).  Same candidates, same failure codes, trajectories and
returns within the north star's 1e-5."""
import numpy as np
import pytest

import oracle_lib as ol
from random_models import random_model

SEEDS = list(range(24))


def _plan_inputs(m, seed):
    P, H, N = 4, 40, 4
    kt = np.linspace(0, (H - 1) * m["timestep"], P)
    kv = np.random.default_rng(1000 + seed).uniform(-0.5, 0.5, (P, m["nu"]))
    eps, sel = ol.noise(seed, 0, 0, N, P, m["nu"])
    return P, H, N, kt, kv, eps, sel


def _check(a, b, tol=1e-5, steps=None):
    """steps: compare only the first `steps` rows of the trajectories (and neither failure flags nor returns): for rollouts whose
    contacts are rounding-sensitive by construction, before the differences have been amplified by the dynamics"""
    assert np.array_equal(a["knots"], b["knots"])
    if steps is None:
        assert np.array_equal(a["failure"], b["failure"])
    ok = (a["failure"] == 0) & (b["failure"] == 0)
    for k in ("states", "residual", "costs", "trace"):
        if ok.any():
            x, y = b[k][ok][:, :steps], a[k][ok][:, :steps]
            assert np.abs(x - y).max() / (np.abs(y).max() + 1e-300) < tol, k
    if steps is None:
        assert np.abs(a["returns"] - b["returns"]).max() / (np.abs(a["returns"]).max() + 1e-300) < tol


# portal=True: loose cylinders / ellipsoids against everything.  The portal-refinement collider (like libccd's MPR) finds depth and
# position to its tolerance, but on some configurations its contact NORMAL moves at the 1e-4 level with a last-bit change of the
# inputs (the final portal triangle depends on the refinement path).  That is a property of the rollout, not of who computes it,
# and it is MEASURED here instead of assumed: the oracle is run a second time from a state whose qpos is moved by one ulp.  Where
# the oracle reproduces itself to 1e-6 (11 of the 12 portal models, and every analytic one), the kernel is held to the north star's
# 1e-5 on everything - failure flags, trajectories, returns, winner.  Where it does not (it then differs from ITSELF by 1e-3 and
# more), no implementation can be asked for 1e-5: those rollouts are compared over their first 20 steps at 1e-2, flags included.
def _oracle_is_reproducible(m, task, d, a, kt, kv, N, H, eps, sel, nthreads=4):
    st = d["state"].copy()
    st[:m["nq"]] = np.nextafter(st[:m["nq"]], np.inf)
    a2 = ol.Oracle(m, task).plan(st, None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=nthreads)
    if not np.array_equal(a["failure"], a2["failure"]):
        return False
    ok = a["failure"] == 0
    return (not ok.any()) or np.abs(a2["states"][ok] - a["states"][ok]).max() / (np.abs(a["states"][ok]).max() + 1e-300) < 1e-6


@pytest.mark.parametrize("seed,portal", [(s, False) for s in SEEDS] + [(s, True) for s in SEEDS[:12]])
def test_random_model_kernel_source_matches_oracle(seed, portal):
    import emu_lib
    m, task, d = random_model(seed, portal)
    P, H, N, kt, kv, eps, sel = _plan_inputs(m, seed)
    a = ol.Oracle(m, task).plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
    b = emu_lib.plan(m, task, d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
    assert a["unsupported"] == 0
    strict = (not portal) or _oracle_is_reproducible(m, task, d, a, kt, kv, N, H, eps, sel)
    _check(a, b, *((1e-5, None) if strict else (1e-2, 20)))
    if strict:
        assert int(np.argmin(b["returns"])) == a["winner"]
    else:
        assert np.array_equal(a["failure"], b["failure"])


def test_most_portal_models_are_held_to_the_strict_bar():
    """the loose bar above is the exception: at most 2 of the 12 portal models may be ill-conditioned in the oracle itself"""
    loose = 0
    for seed in SEEDS[:12]:
        m, task, d = random_model(seed, True)
        P, H, N, kt, kv, eps, sel = _plan_inputs(m, seed)
        a = ol.Oracle(m, task).plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=4)
        loose += not _oracle_is_reproducible(m, task, d, a, kt, kv, N, H, eps, sel)
    assert loose <= 2, loose


@pytest.mark.gpu
def test_random_models_hip_engine_matches_oracle():
    from mujoco_mpc_amd.planner import HipBackend
    active = nloose = 0
    for seed, portal in [(s, False) for s in SEEDS] + [(s, True) for s in SEEDS[:12]]:
        m, task, d = random_model(seed, portal)
        P, H, N, kt, kv, eps, sel = _plan_inputs(m, seed)
        a = ol.Oracle(m, task).plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=8)
        be = HipBackend(m, task, max_samples=N, max_horizon=H)
        out = be.plan(state=d["state"], mocap=None, time=0.0, knot_times=kt, knot_values=kv, interpolation=2, num_trajectory=N, horizon=H,
                      sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
        b = be.fetch_all(N, H, P)
        b["returns"] = out["returns"]; b["failure"] = out["failure"]
        be.close()
        strict = (not portal) or _oracle_is_reproducible(m, task, d, a, kt, kv, N, H, eps, sel, nthreads=8)
        _check(a, b, *((1e-5, None) if strict else (1e-2, 20)))
        if strict:
            assert out["winner"] == a["winner"]
        else:
            assert np.array_equal(a["failure"], b["failure"])
        nloose += not strict
        active += int(b["diag"][:, 2].max() > 0)
    assert active >= len(SEEDS)                 # most random models really exercise constraints
    assert nloose <= 2, nloose                  # the loose bar stays the exception
