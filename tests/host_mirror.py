"""TEST INFRASTRUCTURE: a Python restatement of the reference's host-side planner logic
  mjpc/planners/sampling/planner.{h,cc}   SamplingPlanner
  mjpc/planners/sampling/policy.{h,cc}    SamplingPolicy
  mjpc/spline/spline.{h,cc}               TimeSpline
with the reference's names, so that the parity tests read like the reference's own tests.  The product's host logic is the C++
planner (mujoco_mpc_amd/csrc/planner.cc); the tests hold the two against each other and run this one on the CPU oracle
(tests/oracle_backend.py).  Nothing under mujoco_mpc_amd/ imports this file.
"""
from __future__ import annotations

import bisect
import time as _time

import numpy as np

from mujoco_mpc_amd.planner import HipBackend, kCubicSpline, kLinearSpline, kMaxTrajectoryHorizon, kMaxTrajectoryReference, kZeroSpline  # noqa: F401


class TimeSpline:
    """mjpc/spline/spline.h:41-276 (value semantics; ring buffer replaced by python lists)."""

    def __init__(self, dim=0, interpolation=kZeroSpline):
        self.dim_ = dim
        self.interpolation_ = interpolation
        self.times_: list[float] = []
        self.values_: list[np.ndarray] = []

    def Size(self): return len(self.times_)
    def Dim(self): return self.dim_
    def Interpolation(self): return self.interpolation_
    def SetInterpolation(self, interpolation): self.interpolation_ = interpolation
    def Reserve(self, num_nodes): pass

    def Clear(self):
        self.times_ = []; self.values_ = []

    def copy(self):
        s = TimeSpline(self.dim_, self.interpolation_)
        s.times_ = list(self.times_); s.values_ = [v.copy() for v in self.values_]
        return s

    def AddNode(self, time, values=None):
        """spline.cc:203-238: only before the first or after the last node."""
        v = np.zeros(self.dim_) if values is None else np.array(values, float)
        assert v.shape == (self.dim_,)
        if not self.times_ or time > self.times_[-1]:
            self.times_.append(float(time)); self.values_.append(v)
        else:
            assert time < self.times_[0], "Adding nodes to the middle of the spline isn't supported."
            self.times_.insert(0, float(time)); self.values_.insert(0, v)
        return v

    def DiscardBefore(self, time):
        """spline.cc:164-188."""
        last = bisect.bisect_right(self.times_, time)
        if last == 0:
            return 0
        keep = 1 if self.interpolation_ == kCubicSpline else 0
        last -= 1
        while last != 0 and keep:
            last -= 1; keep -= 1
        del self.times_[:last]; del self.values_[:last]
        return last

    def _slope(self, node, k):
        """spline.cc:259-277."""
        t, v = self.times_, self.values_
        if node == 0:
            return (v[1][k] - v[0][k]) / (t[1] - t[0])
        if node == len(t) - 1:
            return (v[node][k] - v[node - 1][k]) / (t[node] - t[node - 1])
        return (0.5 * (v[node + 1][k] - v[node][k]) / (t[node + 1] - t[node]) +
                0.5 * (v[node][k] - v[node - 1][k]) / (t[node] - t[node - 1]))

    def Sample(self, time):
        """spline.cc:103-156."""
        out = np.zeros(self.dim_)
        if not self.times_:
            return out
        upper = bisect.bisect_right(self.times_, time)
        if upper == len(self.times_):
            return self.values_[-1].copy()
        if upper == 0:
            return self.values_[0].copy()
        lower = upper - 1
        lo, up = self.times_[lower], self.times_[upper]
        t = (time - lo) / (up - lo)
        if self.interpolation_ == kZeroSpline:
            return self.values_[lower].copy()
        if self.interpolation_ == kLinearSpline:
            for i in range(self.dim_):
                out[i] = self.values_[lower][i] * (1 - t) + self.values_[upper][i] * t
            return out
        c0 = 2.0 * t*t*t - 3.0 * t*t + 1.0
        c1 = (t*t*t - 2.0 * t*t + t) * (up - lo)
        c2 = -2.0 * t*t*t + 3 * t*t
        c3 = (t*t*t - t*t) * (up - lo)
        for i in range(self.dim_):
            p0 = self.values_[lower][i]; m0 = self._slope(lower, i)
            m1 = self._slope(upper, i); p1 = self.values_[upper][i]
            out[i] = c0 * p0 + c1 * m0 + c2 * p1 + c3 * m1
        return out

    def arrays(self):
        if not self.times_:
            return np.zeros(0), np.zeros((0, self.dim_))
        return np.array(self.times_, float), np.array(self.values_, float).reshape(len(self.times_), self.dim_)


class SamplingPolicy:
    """mjpc/planners/sampling/policy.cc:30-78."""

    def __init__(self, model: dict, num_spline_points: int):
        self.model = model
        self.num_spline_points = num_spline_points
        self.plan = TimeSpline(model["nu"])

    def Reset(self, horizon=None, initial_repeated_action=None):
        self.plan.Clear()
        if initial_repeated_action is not None:
            self.plan.AddNode(0, initial_repeated_action)

    def Action(self, time):
        a = self.plan.Sample(time)
        r = self.model["actuator_ctrlrange"].reshape(-1, 2)
        return np.minimum(np.maximum(a, r[:, 0]), r[:, 1])      # Clamp, utilities.cc:94-98

    def CopyFrom(self, other):
        self.plan = other.plan.copy(); self.num_spline_points = other.num_spline_points


class Trajectory:
    """mjpc/trajectory.h:74-86 public arrays (winner only is materialised on the host)."""

    def __init__(self):
        self.horizon = 0
        self.states = self.actions = self.times = self.residual = self.costs = self.trace = None
        self.total_return = 0.0
        self.failure = False


class SamplingPlanner:
    """Public surface of mjpc::SamplingPlanner (planners/sampling/planner.h:51-162)."""

    def __init__(self, backend=None):
        self.backend = backend
        self.model = None
        self.task = None
        self.noise_exploration = [0.1, 0.0]
        self.num_trajectory_ = 10
        self.interpolation_ = kCubicSpline
        self.sliding_plan_ = 0
        self.winner = 0
        self.time = 0.0
        self.improvement = 0.0
        self.noise_compute_time = 0.0
        self.rollouts_compute_time = 0.0
        self.policy_update_compute_time = 0.0
        self.seed = 0x5EED
        self.plan_iter = 0
        self.injected_noise = None     # optional (eps[N,P,nu], sel[N]) — reproducible-noise hook

    # --- Initialize / Allocate / Reset (planner.cc:40-143)
    def Initialize(self, model: dict, task: dict, numerics: dict | None = None):
        numerics = numerics or {}
        self.model = model; self.task = task
        se = numerics.get("sampling_exploration", 0.1)
        se = list(se) if isinstance(se, (list, tuple)) else [se]
        self.noise_exploration = [float(se[0]), float(se[1]) if len(se) > 1 else 0.0]
        self.num_trajectory_ = int(numerics.get("sampling_trajectories", 10))
        self.interpolation_ = int(numerics.get("sampling_representation", kCubicSpline))
        self.sliding_plan_ = int(numerics.get("sampling_sliding_plan", 0))
        self.num_spline_points = int(numerics.get("sampling_spline_points", kMaxTrajectoryHorizon))
        self.winner = 0

    def Allocate(self):
        m = self.model
        self.state = np.zeros(m["nq"] + m["nv"] + m["na"])
        self.mocap = np.zeros(7 * m["nmocap"]); self.userdata = np.zeros(m["nuserdata"])
        self.policy = SamplingPolicy(m, self.num_spline_points)
        self.previous_policy = SamplingPolicy(m, self.num_spline_points)
        self.winner_policy = SamplingPolicy(m, self.num_spline_points)   # candidate_policy[winner]
        self.trajectory_order = []
        self.returns = np.zeros(0)
        self.best = None
        self.winner = -1

    def Reset(self, horizon=None, initial_repeated_action=None):
        self.state[:] = 0; self.mocap[:] = 0; self.userdata[:] = 0; self.time = 0.0
        self.policy.Reset(horizon, initial_repeated_action)
        self.previous_policy.Reset(horizon, initial_repeated_action)
        self.winner_policy.Reset(horizon, initial_repeated_action)
        self.improvement = 0.0
        self.winner = 0
        self.best = None

    def SetState(self, state, mocap=None, userdata=None, time=0.0):
        """planner.cc:146-149 (State::CopyTo)."""
        self.state = np.array(state, float)
        if mocap is not None:
            self.mocap = np.array(mocap, float)
        if userdata is not None:
            self.userdata = np.array(userdata, float)
        self.time = float(time)

    # --- UpdateNominalPolicy (planner.cc:236-310)
    def UpdateNominalPolicy(self, horizon):
        num_spline_points = self.winner_policy.num_spline_points
        nominal_time = self.time
        time_horizon = (horizon - 1) * self.model["timestep"]
        if self.sliding_plan_:
            extra_points = {kZeroSpline: 1, kLinearSpline: 2, kCubicSpline: 4}[self.interpolation_]
            if num_spline_points > extra_points:
                time_shift = max(time_horizon / (num_spline_points - extra_points), 1.0e-5)
            else:
                time_shift = time_horizon
            self.policy.plan.DiscardBefore(nominal_time)
            if self.policy.plan.Size() == 0:
                self.policy.plan.AddNode(self.time)
            while self.policy.plan.Size() < num_spline_points:
                new_time = self.policy.plan.times_[-1] + time_shift
                self.policy.plan.AddNode(new_time, self.policy.plan.values_[-1].copy())
        else:
            if self.interpolation_ == kZeroSpline:
                time_shift = max(time_horizon / num_spline_points, 1.0e-5)
            else:
                time_shift = max(time_horizon / (num_spline_points - 1), 1.0e-5)
            scratch = TimeSpline(self.model["nu"], self.interpolation_)
            for _ in range(num_spline_points):
                scratch.AddNode(nominal_time, self.winner_policy.Action(nominal_time))
                nominal_time += time_shift            # repeated addition, like the reference
            self.policy.plan = scratch

    # --- OptimizePolicyCandidates / OptimizePolicy (planner.cc:151-208)
    def OptimizePolicyCandidates(self, ncandidates, horizon):
        num_trajectory = self.num_trajectory_
        ncandidates = min(ncandidates, num_trajectory)
        t0 = _time.perf_counter()
        self.policy.plan.SetInterpolation(self.interpolation_)
        kt, kv = self.policy.plan.arrays()
        eps = sel = None
        if self.injected_noise is not None:
            eps, sel = self.injected_noise
        out = self.backend.plan(state=self.state, mocap=self.mocap, userdata=self.userdata, time=self.time,
                                knot_times=kt, knot_values=kv, interpolation=self.interpolation_,
                                num_trajectory=num_trajectory, horizon=horizon, sigma=self.noise_exploration,
                                noise_eps=eps, noise_sel=sel, seed=self.seed, stream=self.plan_iter)
        self.plan_iter += 1
        self._last = out; self._last_kt = kt; self._last_H = horizon
        self.returns = out["returns"]
        # std::partial_sort on total_return (planner.cc:177-181): stable argsort gives lowest index on ties
        self.trajectory_order = list(np.argsort(self.returns, kind="stable"))
        self.noise_compute_time = out.get("noise_compute_time_us", 0.0)
        self.rollouts_compute_time = (_time.perf_counter() - t0) * 1e6
        return ncandidates

    def OptimizePolicy(self, horizon):
        self.UpdateNominalPolicy(horizon)
        self.OptimizePolicyCandidates(1, horizon)
        t0 = _time.perf_counter()
        self.CopyCandidateToPolicy(0)
        best_return = self.returns[0]
        self.improvement = max(best_return - self.returns[self.winner], 0.0)
        self.policy_update_compute_time = (_time.perf_counter() - t0) * 1e6

    def NominalTrajectory(self, horizon):
        """planner.cc:211-222: one rollout of candidate 0 (no noise)."""
        self.policy.plan.SetInterpolation(self.interpolation_)
        kt, kv = self.policy.plan.arrays()
        if len(kt) == 0:
            kt = np.array([self.time]); kv = np.zeros((1, self.model["nu"]))
        out = self.backend.plan(state=self.state, mocap=self.mocap, userdata=self.userdata, time=self.time,
                                knot_times=kt, knot_values=kv, interpolation=self.interpolation_, num_trajectory=1,
                                horizon=horizon, sigma=(0.0, 0.0))
        self._set_best(out, horizon, 0)

    def _set_best(self, out, horizon, winner):
        tr = Trajectory()
        tr.horizon = horizon
        for k in ["states", "actions", "times", "residual", "costs", "trace"]:
            setattr(tr, k, out[k])
        tr.total_return = float(out["returns"][winner]); tr.failure = bool(out["failure"][winner])
        self.best = tr

    def CopyCandidateToPolicy(self, candidate):
        """planner.cc:525-534."""
        self.winner = int(self.trajectory_order[candidate])
        out = self._last
        if self.winner != out["winner"]:
            out2 = self.backend.candidate(self.winner, self._last_H, len(self._last_kt))
            for k in ["states", "actions", "times", "residual", "costs", "trace", "winner_knots"]:
                out[k] = out2[k]
        self.previous_policy.CopyFrom(self.policy)
        wp = SamplingPolicy(self.model, self.policy.num_spline_points)
        wp.plan = TimeSpline(self.model["nu"], self.interpolation_)
        for t, v in zip(self._last_kt, out["winner_knots"]):
            wp.plan.AddNode(t, v)
        self.winner_policy = wp
        self.policy.CopyFrom(wp)
        self._set_best(out, self._last_H, self.winner)

    def ActionFromPolicy(self, time, use_previous=False):
        """planner.cc:225-233."""
        return (self.previous_policy if use_previous else self.policy).Action(time)

    def BestTrajectory(self):
        return self.best if self.winner >= 0 else None

    def CandidateScore(self, candidate):
        return float(self.returns[self.trajectory_order[candidate]])

    def NumParameters(self):
        return self.policy.num_spline_points * self.model["nu"]
