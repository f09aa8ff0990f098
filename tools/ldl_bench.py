import sys, os, ctypes as C, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from mujoco_mpc_amd import capi
from mujoco_mpc_amd.modelgen import quadruped, humanoid_track
import glob
import __graft_entry__ as g
lib = C.CDLL(sys.argv[1] if len(sys.argv) > 1 else g.TESTHOOKS_SO)
dp = C.POINTER(C.c_double)
lib.mjpc_hip_debug_ldl_bench.argtypes = [C.c_int] * 4 + [dp, dp, dp, C.c_int]
for n, gen in ((18, quadruped),):
    m = gen()[0]; par = [int(p) for p in m["dof_parentid"]]
    rng = np.random.default_rng(n); A = np.zeros((n, n))
    for i in range(n):
        a = i
        while a >= 0: A[i, a] = A[a, i] = rng.normal(); a = par[a]
    A[np.arange(n), np.arange(n)] = np.abs(A).sum(1) + 1.0
    b = rng.normal(size=n)
    for tree in (0, 1):
        for spin in (0,):
            out = np.zeros(2)
            lib.mjpc_hip_debug_ldl_bench(n, tree, 2000, spin, A.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp), 0)
            print(f"n={n} tree={tree} spinners={spin}: {out[0]:.0f} ticks per factor+solve", flush=True)
