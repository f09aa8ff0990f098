"""Diagnostic: the hand at its own capacity (tables from L2) vs a model authored with fewer rows / contacts (needs rollout_cached
instantiated for 33 dofs to show the LDS-cached flavour: measured 7.43 -> 7.10 ms at 88 / 24, but 3 % of the candidates then overflow)."""
import sys; sys.path.insert(0,'.')
import numpy as np
from mujoco_mpc_amd.modelgen import shadow_hand
from mujoco_mpc_amd.planner import HipBackend
for kw in (dict(), dict(nefcmax=88, nconmax=24)):
    m, task, d = shadow_hand(**kw)
    N,H,P=256,64,5
    be = HipBackend(m, task, max_samples=N, max_horizon=H)
    kt = np.arange(P) * ((H - 1) * m['timestep'] / P); kv = np.tile(d['ctrl0'], (P, 1))
    k = dict(state=d['state'], mocap=None, time=0.0, knot_times=kt, interpolation=0, num_trajectory=N, horizon=H, sigma=(0.1, 0.0), seed=0x5EED)
    kn = kv
    for i in range(3): kn = be.plan(knot_values=kn, stream=i, **k)['winner_knots']
    be.kernel_time()
    nf=0
    for i in range(20): r = be.plan(knot_values=kn, stream=3 + i, **k); kn = r['winner_knots']; nf+=int((r['failure']!=0).sum())
    nl, us, tot = be.kernel_time()
    print(kw, "lds", be.lds_bytes(), "kernel ms %.3f" % (us/1e3), "failures", nf, "winner", r['winner'], r['winner_return'])
    be.close()
