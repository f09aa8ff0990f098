import os, sys, subprocess, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "build", "variants", "lib_*.so")))
for lib in libs:
    code = f"""
import sys, os, time
sys.path.insert(0, {ROOT!r})
from mujoco_mpc_amd import capi
capi.ENGINE_PATH = {lib!r}
import numpy as np
from mujoco_mpc_amd.modelgen import quadruped, humanoid_track
from mujoco_mpc_amd.planner import HipBackend
def run(gen, N, H, P, sigma, reps):
    m, task, d = gen()
    be = HipBackend(m, task, max_samples=N, max_horizon=H)
    kt = np.linspace(0, (H - 1) * m['timestep'], P); kv = np.zeros((P, m['nu']))
    kw = dict(state=d['state'], mocap=d['mocap'], time=0.0, knot_times=kt, interpolation=2, num_trajectory=N, horizon=H, sigma=(sigma, 0.0), seed=0x5EED)
    kn = kv
    for i in range(3): kn = be.plan(knot_values=kn, stream=i, **kw)['winner_knots']
    be.kernel_time()
    t0 = time.perf_counter()
    for i in range(reps): r = be.plan(knot_values=kn, stream=3 + i, **kw); kn = r['winner_knots']
    dt = time.perf_counter() - t0
    nl, us, tot = be.kernel_time()
    be.close()
    return 1e3 * dt / reps, us / 1e3, r['winner'], r['winner_return']
print({os.path.basename(lib)!r}, 'C2 ms/plan %.3f kernel %.3f winner %d ret %.12g' % run(quadruped, 256, 100, 3, 0.04, 30), flush=True)
if {('Q512' in os.environ)!r}: print('   Q512 ms/plan %.3f kernel %.3f winner %d ret %.12g' % run(quadruped, 512, 100, 3, 0.04, 20), flush=True)
if {('HAND' in os.environ)!r}:
    from mujoco_mpc_amd.modelgen import shadow_hand
    def runh(N, H, P, reps):
        m, task, d = shadow_hand()
        be = HipBackend(m, task, max_samples=N, max_horizon=H)
        kt = np.arange(P) * ((H - 1) * m['timestep'] / P); kv = np.tile(d['ctrl0'], (P, 1))
        kw = dict(state=d['state'], mocap=None, time=0.0, knot_times=kt, interpolation=0, num_trajectory=N, horizon=H, sigma=(0.1, 0.0), seed=0x5EED)
        kn = kv
        for i in range(3): kn = be.plan(knot_values=kn, stream=i, **kw)['winner_knots']
        be.kernel_time()
        for i in range(reps): r = be.plan(knot_values=kn, stream=3 + i, **kw); kn = r['winner_knots']
        nl, us, tot = be.kernel_time()
        be.close()
        return us / 1e3, r['winner'], r['winner_return']
    print('   HAND kernel %.3f winner %d ret %.12g' % runh(256, 64, 5, 20), flush=True)
if {('C3' in os.environ)!r}: print('   C3 ms/plan %.3f kernel %.3f winner %d ret %.12g' % run(humanoid_track, 1024, 128, 16, 0.15, 5), flush=True)
"""
    subprocess.run([sys.executable, "-c", code], check=False, timeout=300)
