"""Candidate sharding across GPUs: one process per GPU, block partition of the candidate index, and ONE tiny
collective per plan step for the elite.

The reference has no distributed path (single process, std::thread pool, planner.cc:342-380); candidates are
independent given (x0, nominal spline, cost), so rank g rolls out candidates [g*n, (g+1)*n) with no data-path
collective.  The only exchange is the elite pick: every rank contributes (local min return, global argmin index,
winner knots) to one all_gather (RCCL over xGMI on GPUs; gloo in the CPU tests) and takes the identical
lexicographic minimum — lowest index wins ties, bit-exact with the single-GPU argmin.
"""
from __future__ import annotations

import numpy as np


class ShardedSampler:
    def __init__(self, backend, rank: int, world: int, samples_per_rank: int, dist=None, device=None):
        self.backend = backend
        self.rank = rank; self.world = world; self.nper = int(samples_per_rank)
        self.dist = dist            # torch.distributed (already initialised) or None for world == 1
        self.device = device        # torch device for the collective buffers ("cuda:k" with nccl, "cpu" with gloo)
        self._buf = None
        self.always_exchange = False      # tests: run the collective even with a single rank
        self.fetch_winner_rows = True     # the owner of the global elite copies its trajectory to the host after the exchange
        if world > 1 and hasattr(backend, "set_fetch_mode"):
            backend.set_fetch_mode(True)  # shards report summaries only; nobody but the owner moves a trajectory

    @property
    def num_trajectory(self):
        return self.nper * self.world

    def plan(self, **kw):
        """kw: the arguments of HipBackend.plan except num_trajectory / candidate_offset / num_local."""
        out = self.backend.plan(num_trajectory=self.num_trajectory, candidate_offset=self.rank * self.nper,
                                num_local=self.nper, **kw)
        return self.exchange(out)

    def exchange(self, out):
        """Pick the global elite.  Returns dict(winner, winner_return, winner_knots, owner) identical on all ranks."""
        knots = np.ascontiguousarray(out["winner_knots"], dtype=np.float64).ravel()
        if (self.world == 1 and not self.always_exchange) or self.dist is None:
            return dict(winner=int(out["winner"]), winner_return=float(out["winner_return"]), winner_knots=out["winner_knots"],
                        owner=0, local=out)
        import torch
        n = knots.size + 2
        on_gpu = str(self.device).startswith("cuda")
        if self._buf is None or self._buf.numel() != n * self.world:
            # persistent buffers: pinned host staging on GPUs so that the two tiny copies around the collective are asynchronous
            self._buf = torch.empty(n * self.world, dtype=torch.float64, device=self.device)
            self._mine_h = torch.empty(n, dtype=torch.float64, pin_memory=on_gpu)
            self._all_h = torch.empty(n * self.world, dtype=torch.float64, pin_memory=on_gpu)
            self._mine_d = torch.empty(n, dtype=torch.float64, device=self.device) if on_gpu else self._mine_h
        mh = self._mine_h.numpy()
        mh[0] = float(out["winner_return"]); mh[1] = float(out["winner"]); mh[2:] = knots
        if on_gpu:
            self._mine_d.copy_(self._mine_h, non_blocking=True)
            self.dist.all_gather_into_tensor(self._buf, self._mine_d)
            self._all_h.copy_(self._buf, non_blocking=True)
            torch.cuda.current_stream().synchronize()
            allv = self._all_h.numpy().reshape(self.world, n)
        else:
            self.dist.all_gather_into_tensor(self._buf, self._mine_h)
            allv = self._buf.view(self.world, n).numpy()
        best = 0
        for r in range(1, self.world):      # lexicographic (return, index): lowest index on ties
            if allv[r, 0] < allv[best, 0] or (allv[r, 0] == allv[best, 0] and allv[r, 1] < allv[best, 1]):
                best = r
        res = dict(winner=int(allv[best, 1]), winner_return=float(allv[best, 0]),
                   winner_knots=allv[best, 2:].reshape(np.asarray(out["winner_knots"]).shape).copy(), owner=best, local=out)
        if best == self.rank and self.fetch_winner_rows and hasattr(self.backend, "candidate"):
            H = np.asarray(out["times"]).shape[0]; P = np.asarray(out["winner_knots"]).shape[0]
            res["trajectory"] = self.backend.candidate(res["winner"] - self.rank * self.nper, H, P)      # owner-only D2H
        return res
