"""Oracle vs the reference's own golden vectors for the MJPC-side arithmetic (CPU only)."""
import math

import numpy as np
import pytest

import oracle_lib as ol
from mujoco_mpc_amd.modelgen import particle
from host_mirror import TimeSpline, kCubicSpline, kLinearSpline, kZeroSpline


# ---- mjpc/test/spline/spline_test.cc ---------------------------------------------------
def test_spline_empty_samples_zero():                      # spline_test.cc:40-49
    out = ol.spline_sample(np.zeros(0), np.zeros((0, 10)), kCubicSpline, 2.0)
    assert out.shape == (10,) and np.all(out == 0.0)
    s = TimeSpline(10)
    assert s.Size() == 0 and s.Dim() == 10 and np.all(s.Sample(2.0) == 0.0)


@pytest.mark.parametrize("interp", [kZeroSpline, kLinearSpline, kCubicSpline])
def test_spline_one_and_two_nodes(interp):                 # spline_test.cc:51-80
    for t in (0.0, 2.0, 4.0):
        assert list(ol.spline_sample([1.0], [[1.0, 2.0]], interp, t)) == [1.0, 2.0]
    times, vals = [1.0, 2.0], [[1.0, 2.0], [3.0, 4.0]]
    assert list(ol.spline_sample(times, vals, interp, 0)) == [1.0, 2.0]
    assert list(ol.spline_sample(times, vals, interp, 1)) == [1.0, 2.0]
    assert list(ol.spline_sample(times, vals, interp, 2)) == [3.0, 4.0]
    assert list(ol.spline_sample(times, vals, interp, 3)) == [3.0, 4.0]
    s = TimeSpline(2, interp)
    s.AddNode(1.0, [1.0, 2.0]); n = s.AddNode(2.0); n[0] = 3.0; n[1] = 4.0
    assert s.Size() == 2
    for t, e in ((0, [1, 2]), (1, [1, 2]), (2, [3, 4]), (3, [3, 4])):
        assert list(s.Sample(t)) == e


def test_spline_zero_linear_cubic_goldens():               # spline_test.cc:115-159
    times, vals = [1.0, 2.0], [[1.0, 2.0], [3.0, 4.0]]
    assert list(ol.spline_sample(times, vals, kZeroSpline, 1.5)) == [1.0, 2.0]
    assert list(ol.spline_sample(times, vals, kLinearSpline, 1.5)) == [2.0, 3.0]
    assert list(ol.spline_sample(times, vals, kCubicSpline, 1.5)) == [2.0, 3.0]
    t4, v4 = [0.0, 1.0, 2.0, 3.0], [[1.0, 2.0], [1.0, 2.0], [3.0, 4.0], [3.0, 4.0]]
    assert list(ol.spline_sample(t4, v4, kCubicSpline, 1.5)) == [2.0, 3.0]
    t3, v3 = [-1.0, 0.0, 1.0], [[1.0], [0.0], [1.0]]
    s = TimeSpline(1, kCubicSpline)
    for t, v in zip(t3, v3):
        s.AddNode(t, v)
    x = 0.0
    while x <= 1.0:
        y = -math.pow(x, 3) + 2 * math.pow(x, 2)           # known closed form of this spline
        assert ol.spline_sample(t3, v3, kCubicSpline, x)[0] == y
        assert s.Sample(x)[0] == y
        x += 0.125


def test_spline_add_before_start_and_discard():            # spline_test.cc:82-98,161-231
    s = TimeSpline(2)
    s.AddNode(2.0, [2.0, 3.0]); s.AddNode(1.0, [1.0, 2.0]); s.AddNode(3.0, [3.0, 4.0]); s.AddNode(0.0, [0.0, 1.0])
    for t in range(4):
        assert list(s.Sample(t)) == [float(t), float(t + 1)]
    for interp in (kZeroSpline, kLinearSpline, kCubicSpline):
        s = TimeSpline(2, interp)
        for k in range(1, 5):
            s.AddNode(float(k), [float(k), float(k + 1)])
        assert s.DiscardBefore(0.9) == 0 and s.Size() == 4
        assert list(s.Sample(0.0)) == [1.0, 2.0]
        discarded = s.DiscardBefore(3.0)
        if interp == kCubicSpline:
            assert discarded == 1 and s.Size() == 3 and list(s.Sample(1.0)) == [2.0, 3.0]
        else:
            assert discarded == 2 and s.Size() == 2 and list(s.Sample(1.0)) == [3.0, 4.0]
        assert s.DiscardBefore(3.9) == 0
    s = TimeSpline(1)
    for k in range(1, 5):
        s.AddNode(float(k), [float(k)])
    assert s.DiscardBefore(3) == 2 and s.Size() == 2
    s.AddNode(5.0, [5.0]); s.AddNode(6.0, [6.0])
    assert s.DiscardBefore(6.0) == 3 and s.Size() == 1 and s.Sample(1.0)[0] == 6.0


def test_python_spline_matches_oracle_bitwise():
    rng = np.random.default_rng(0)
    for interp in (kZeroSpline, kLinearSpline, kCubicSpline):
        for P in (1, 2, 3, 7):
            times = np.cumsum(rng.uniform(0.05, 0.3, P)); vals = rng.normal(size=(P, 3))
            s = TimeSpline(3, interp)
            for t, v in zip(times, vals):
                s.AddNode(t, v)
            for t in rng.uniform(times[0] - 0.2, times[-1] + 0.2, 40):
                assert np.array_equal(s.Sample(t), ol.spline_sample(times, vals, interp, t))


# ---- mjpc/test/tasks/task_test.cc:58-95 ------------------------------------------------
def test_cost_terms_and_risk():
    m, task, _ = particle()
    assert task["num_residual"] == 4 and task["num_term"] == 2
    assert list(task["dim_norm_residual"]) == [2, 2] and list(task["num_norm_parameter"]) == [0, 0]
    assert list(task["norm"]) == [0, 0] and abs(task["weight"][0] - 5.0) < 1e-5 and abs(task["weight"][1] - 0.1) < 1e-5
    assert abs(task["risk"] - 1.0) < 1e-5 and list(task["parameters"]) == [0.05, -0.1]
    o = ol.Oracle(m, task)
    residual = np.array([1.0e-3, 2.0e-3, 3.0e-3, 4.0e-3])
    c = 5.0 * 0.5 * residual[:2] @ residual[:2] + 0.1 * 0.5 * residual[2:] @ residual[2:]
    _, terms = o.cost(residual)
    assert abs(terms.sum() - c) < 1e-5
    task2 = dict(task); task2["risk"] = 0.2
    o.set_task(task2)
    tc, _ = o.cost(residual)
    assert abs(tc - (math.exp(0.2 * c) - 1.0) / 0.2) < 1e-5
    task3 = dict(task); task3["risk"] = 0.0
    o.set_task(task3)
    assert o.cost(residual)[0] == terms.sum()


# ---- norm known answers (formulae of mjpc/norm.cc:50-210 evaluated independently) ------
NORM_POINTS = [np.array([0.1, 0.2, 0.3]), np.array([-0.5, 1.0, 2.0]), np.array([0.0, 0.0, 0.0]), np.array([3.0]), np.array([-1e-3, 2e-3])]


@pytest.mark.parametrize("x", NORM_POINTS)
def test_norm_values(x):
    p, q = 0.3, 1.7
    expect = {
        0: 0.5 * float(x @ x),
        1: (float(x @ x) ** (q / 2) + p ** q) ** (1 / q) - p,
        2: math.sqrt(float(x @ x) + p * p) - p,
        3: sum(p * p * (math.cosh(v / p) - 1.0) for v in x),
        5: sum(abs(v) ** p for v in x),
        6: sum(math.sqrt(v * v + p * p) - p for v in x),
        7: sum((abs(v) ** q + p ** q) ** (1 / q) - p for v in x),
        8: sum(p * math.log(1 + math.exp(v / p)) for v in x),
    }
    for typ, e in expect.items():
        got = ol.norm(x, [p, q], typ)
        assert got == pytest.approx(e, rel=1e-13, abs=1e-15), typ
    assert ol.norm(x[:1], [], -1) == x[0]
    assert [ol.lib().oracle_norm_parameter_dimension(t) for t in (-1, 0, 1, 2, 3, 5, 6, 7, 8)] == [0, 0, 2, 1, 1, 1, 1, 2, 1]


# ---- noise: Philox4x32-10 known-answer vectors (Random123 kat_vectors) -----------------
def test_philox_kat():
    import ctypes as C
    out = (C.c_uint32 * 4)()
    L = ol.lib()
    # counter = (c0, c1, stream_lo, stream_hi), key = (seed_lo, seed_hi)
    L.oracle_philox(0, 0, 0, 0, out)
    assert [hex(v) for v in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    L.oracle_philox(0xffffffffffffffff, 0xffffffffffffffff, 0xffffffff, 0xffffffff, out)
    assert [hex(v) for v in out] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    L.oracle_philox(0x299f31d0a4093822, 0x0370734413198a2e, 0x243f6a88, 0x85a308d3, out)
    assert [hex(v) for v in out] == ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_noise_statistics_and_determinism():
    eps, sel = ol.noise(0x5EED, 3, 0, 512, 4, 6, sigma2=0.5)
    eps2, sel2 = ol.noise(0x5EED, 3, 0, 512, 4, 6, sigma2=0.5)
    assert np.array_equal(eps, eps2) and np.array_equal(sel, sel2)
    assert abs(eps.mean()) < 0.03 and abs(eps.std() - 1.0) < 0.03
    assert 0.1 < sel.mean() < 0.3
    part, _ = ol.noise(0x5EED, 3, 100, 10, 4, 6)
    assert np.array_equal(part, eps[100:110])             # shard-invariant: indexed by global candidate id
    other, _ = ol.noise(0x5EED, 4, 0, 8, 4, 6)
    assert not np.array_equal(other, eps[:8])
