"""Adapter so that tests/host_mirror.py SamplingPlanner can run on the CPU oracle IN TESTS ONLY."""
import numpy as np

from oracle_lib import Oracle


class OracleBackend:
    def __init__(self, model, task, nthreads=1):
        self.o = Oracle(model, task)
        self.model = model; self.task = task
        self.nthreads = nthreads
        self._all = None
        self.summary_only = False

    def set_fetch_mode(self, summary_only):
        self.summary_only = bool(summary_only)

    def set_task(self, task):
        self.o.set_task(task); self.task = task

    def plan(self, state, mocap, time, knot_times, knot_values, interpolation, num_trajectory, horizon, sigma,
             noise_eps=None, noise_sel=None, seed=0, stream=0, userdata=None, candidate_offset=0, num_local=None,
             noise_std=None, nominal_index=0, candidate_knots=None, xfrc_std=0.0, xfrc_rate=0.0):
        r = self.o.plan(state, mocap, time, knot_times, knot_values, interpolation, num_trajectory, horizon, sigma,
                        noise_eps, noise_sel, seed, stream, self.nthreads, candidate_offset, num_local,
                        noise_std=noise_std, nominal_index=nominal_index, candidate_knots=candidate_knots,
                        xfrc_std=xfrc_std, xfrc_rate=xfrc_rate)
        self._all = r
        w = r["winner"] - candidate_offset
        out = dict(returns=r["returns"], failure=r["failure"], winner=r["winner"], winner_return=r["returns"][w])
        for k in ["states", "actions", "times", "residual", "costs", "trace"]:
            out[k] = np.zeros_like(r[k][w]) if self.summary_only else r[k][w]       # summary mode: rows stay "on the device"
        out["winner_knots"] = r["knots"][w]
        return out

    def candidate(self, i, H, P):
        r = self._all
        out = {k: r[k][i] for k in ["states", "actions", "times", "residual", "costs", "trace"]}
        out["winner_knots"] = r["knots"][i]
        out["returns"] = r["returns"][i:i + 1]; out["failure"] = r["failure"][i:i + 1]
        return out
