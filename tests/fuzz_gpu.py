"""Diagnostic (GPU box): the fuzz parity of tests/test_random_models.py over many more seeds (default 24..223), HIP engine vs oracle.
Test infrastructure only (uses oracle/)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from random_models import random_model
from test_random_models import _plan_inputs, _check, _oracle_is_reproducible
from mujoco_mpc_amd.planner import HipBackend
lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 224)
bad, refused, loose = [], [], []
for seed in range(lo, hi):
    m, task, d = random_model(seed)
    P, H, N, kt, kv, eps, sel = _plan_inputs(m, seed)
    a = ol.Oracle(m, task).plan(d["state"], None, 0.0, kt, kv, 2, N, H, sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel, nthreads=8)
    try:
        be = HipBackend(m, task, max_samples=N, max_horizon=H)
    except RuntimeError as e:          # a model larger than one CU's LDS is refused at create (loudly): not a parity case
        if "exceeds 160 KiB" in str(e):
            refused.append(seed); continue
        raise
    out = be.plan(state=d["state"], mocap=None, time=0.0, knot_times=kt, knot_values=kv, interpolation=2, num_trajectory=N, horizon=H,
                  sigma=(0.3, 0.0), noise_eps=eps, noise_sel=sel)
    b = be.fetch_all(N, H, P); b["returns"] = out["returns"]; b["failure"] = out["failure"]
    be.close()
    try:
        _check(a, b)
        assert out["winner"] == a["winner"]
    except AssertionError as e:
        # the same self-measured bar as the test: where the oracle does not reproduce ITSELF under a one-ulp change of qpos (a
        # rollout that blows up, an ill-conditioned contact), the first 20 steps are compared at 1e-2 instead
        if _oracle_is_reproducible(m, task, d, a, kt, kv, N, H, eps, sel, nthreads=8):
            bad.append((seed, str(e)[:60]))
        else:
            try:
                _check(a, b, 1e-2, 20); loose.append(seed)
            except AssertionError as e2:
                bad.append((seed, "loose bar: " + str(e2)[:60]))
print("seeds", lo, hi, "refused for size:", refused, "held to the loose bar (oracle irreproducible):", loose, "failed:", bad)
