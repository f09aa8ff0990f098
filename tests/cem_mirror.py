"""TEST INFRASTRUCTURE: a Python restatement of mjpc::CrossEntropyPlanner (planners/cross_entropy/planner.cc:164-338) over a
plan backend (the CPU oracle in the tests).  The product's Cross-Entropy planner is the C++ class in csrc/planner.cc; this mirror
only exists so that the GPU test can compare it against an independent implementation on the oracle."""
import numpy as np

from host_mirror import SamplingPolicy, TimeSpline


class CrossEntropyMirror:
    def __init__(self, backend, model, task, numerics):
        self.backend = backend; self.model = model; self.task = task
        self.std_initial = float(numerics.get("sampling_exploration", 0.1)); self.std_min = float(numerics.get("std_min", 0.1))
        self.N = int(numerics.get("sampling_trajectories", 10))
        self.n_elite = int(numerics.get("n_elite", max(self.N // 10, 2)))
        self.interp = int(numerics.get("sampling_representation", 0))
        self.P = int(numerics.get("sampling_spline_points", 512))
        self.nu = model["nu"]
        self.seed = 0x5EED; self.plan_iter = 0; self.injected_noise = None
        self.policy = SamplingPolicy(model, self.P); self.resampled = SamplingPolicy(model, self.P)

    def Reset(self, horizon):
        self.policy.Reset(horizon); self.resampled.Reset(horizon)
        self.variance = np.full(self.P * self.nu, self.std_initial * self.std_initial)
        self.time = 0.0; self.improvement = 0.0

    def SetState(self, state, mocap, userdata, time):
        self.state = np.array(state, float); self.mocap = mocap; self.time = float(time)

    def OptimizePolicy(self, H):
        m = self.model; nu = self.nu; P = self.P
        self.resampled.plan.SetInterpolation(self.interp)
        self.resampled.CopyFrom(self.policy)
        # ResamplePolicy (planner.cc:313-338)
        t = self.time; shift = max((H - 1) * m["timestep"] / (P - 1), 1.0e-5)
        times = np.zeros(P); params = np.zeros((P, nu))
        for k in range(P):
            times[k] = t; params[k] = self.resampled.Action(t); t += shift
        keep = self.policy.plan.Interpolation()
        self.resampled.plan = TimeSpline(nu, keep)
        for k in range(P):
            self.resampled.plan.AddNode(times[k], params[k])
        std = np.maximum(np.sqrt(self.variance), self.std_min)
        out = self.backend.plan(state=self.state, mocap=self.mocap, time=self.time, knot_times=times, knot_values=params,
                                interpolation=self.resampled.plan.Interpolation(), num_trajectory=self.N + 1, horizon=H, sigma=(0.0, 0.0),
                                noise_eps=self.injected_noise, seed=self.seed, stream=self.plan_iter, noise_std=std, nominal_index=self.N)
        self.plan_iter += 1
        allr = self.backend._all
        self.returns = out["returns"]; knots = allr["knots"].reshape(self.N + 1, P * nu)
        order = np.argsort(self.returns[:self.N], kind="stable")
        ne = min(self.n_elite, self.N)
        avg = np.zeros(P * nu); avg_ret = 0.0
        for i in range(ne):
            avg += knots[order[i]]; avg_ret += self.returns[order[i]]
        avg *= 1.0 / ne; avg_ret /= ne
        var = np.zeros(P * nu)
        best = knots[order[0]]
        for i in range(ne):                                   # the reference reads the best elite for every i
            diff = best - avg
            var += diff * diff / (ne - 1)
        self.variance = var
        self.policy.plan = TimeSpline(nu, self.interp)
        for k in range(P):
            self.policy.plan.AddNode(times[k], avg[k * nu:(k + 1) * nu])
        self.improvement = max(avg_ret - self.returns[order[0]], 0.0)
        self.nominal_states = allr["states"][self.N]; self.nominal_return = self.returns[self.N]
        self.order = order
