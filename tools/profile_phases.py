"""Diagnostic: build the engine with in-kernel phase stamps (MJPC_PROFILE) and print where a step's cycles go.
Never used for timing claims (stamps perturb the schedule); only the shares matter."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "mujoco_mpc_amd", "csrc")
so = os.path.join(ROOT, "gpurun_out", "libmjpc_hip_prof.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
if os.environ.get("PROFILE_LIB"):      # prebuilt with tools/mkvariant.sh prof_wN "-DMJPC_PROFILE=1 -DMJPC_PROFILE_WAVE=N"
    so = os.path.abspath(os.environ["PROFILE_LIB"])
else:
    flags = ["-DMJPC_PROFILE=1"] + (["-DMJPC_WAVES=1"] if os.environ.get("PROFILE_ONE_WAVE") else []) + \
            ([f"-DMJPC_PROFILE_WAVE={int(os.environ['PROFILE_WAVE'])}"] if os.environ.get("PROFILE_WAVE") else [])
    objs, jobs = [], []
    for f in sorted(os.listdir(CSRC)):
        if f == "engine.hip" or (f.startswith("rollout_") and f.endswith(".hip")):
            o = os.path.join(ROOT, "gpurun_out", "prof_" + f[:-4] + ".o"); objs.append(o)
            jobs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value",
                                          "-Wno-unused-result"] + flags + ["-c", "-o", o, os.path.join(CSRC, f)], cwd=CSRC))
    for j in jobs:
        assert j.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", so] + objs +
                          [os.path.join(CSRC, "_obj", x) for x in ("planner.o", "testspeed.o", "multi.o")] + ["-lpthread"])
from mujoco_mpc_amd import capi
capi.ENGINE_PATH = so
from mujoco_mpc_amd.modelgen import quadruped, humanoid_track, shadow_hand
from mujoco_mpc_amd.planner import HipBackend
WL = os.environ.get("PROFILE_WORKLOAD", "quadruped")
HUM = WL == "humanoid"; HAND = WL == "hand"
m, task, d = humanoid_track() if HUM else (shadow_hand() if HAND else quadruped())
N, H, P = int(sys.argv[1]) if len(sys.argv) > 1 else 256, (128 if HUM else (64 if HAND else 100)), (16 if HUM else (5 if HAND else 3))
SIG = 0.15 if HUM else (0.1 if HAND else 0.04)
INTERP = 0 if HAND else 2
kt = np.arange(P) * ((H - 1) * m["timestep"] / P) if HAND else np.linspace(0, (H - 1) * m["timestep"], P)
kv = np.tile(d["ctrl0"], (P, 1)) if HAND else np.zeros((P, m["nu"]))
if not len(d["mocap"]): d = dict(d, mocap=None)
be = HipBackend(m, task, max_samples=N, max_horizon=H)
for i in range(2):
    out = be.plan(state=d["state"], mocap=d["mocap"], time=0.0, knot_times=kt, knot_values=kv, interpolation=INTERP, num_trajectory=N,
                  horizon=H, sigma=(SIG, 0.0), seed=0x5EED, stream=i)
prof = np.zeros((N, 24), np.int64)
be.lib.mjpc_hip_debug_fetch_prof.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
be.lib.mjpc_hip_debug_fetch_prof(be.h, prof.ctypes.data_as(C.POINTER(C.c_longlong)))
allc = be.fetch_all(N, H, P)
names = ["(loop overhead/record)", "kinematics", "com_pos + barrier wait after head", "BARRIER WAIT after presolve", "collision", "make_constraint", "BARRIER WAIT after solve", "impedance(+warm)",
         "solver tail", "gradient (J^T f)", "assemble: hq + diag", "integrate", "warm-start evals / ls commit", "newton misc (search, loop top)", "(alpha pick)", "hq init / rank-1 updates", "wait for the workers", "factor + solve H", "assemble: add cone partials", "assemble: qH store", "ls: Mv,jv", "ls: load", "ls: evals"]
W = int(os.environ.get("PROFILE_WAVE", "0"))
if W == 3:      # side wave
    names = ["(head: idle)", "smooth: com->vel/acc level sweep", "BARRIER WAIT after head", "BARRIER WAIT after presolve", "smooth: subtree sums", "smooth: actuation/bias/springs", "BARRIER WAIT after solve",
             "smooth: wait for factor(M)", "smooth: solve qacc_smooth", "residual", "cost+record", "prefactor M+hB", "worker: wait for a job", "worker: cone rows (column loop)", "worker: flag", "worker: zero the partial", "worker: row setup (t = B J)"] + ["-"] * 6
elif W == 1:    # helper 0
    names = ["-", "inertia: crb subtree sums", "BARRIER WAIT after head", "BARRIER WAIT after presolve", "inertia: M entries", "inertia: factor M", "BARRIER WAIT after solve (= solver helper loop)"] + ["-"] * 5 + ["worker: wait for a job", "worker: cone rows (column loop)", "worker: flag", "worker: zero the partial", "worker: row setup (t = B J)"] + ["-"] * 6
elif W == 2:    # helper 1
    names = ["-", "noncontact rows", "BARRIER WAIT after head", "BARRIER WAIT after presolve", "noncontact impedance", "-", "BARRIER WAIT after solve (= cost@smooth + solver helper loop)"] + ["-"] * 5 + ["worker: wait for a job", "worker: cone rows (column loop)", "worker: flag", "worker: zero the partial", "worker: row setup (t = B J)"] + ["-"] * 6
tot = prof[:, :(23 if W == 0 else 22)].sum(1).mean()
print(f"rollout us {out['rollouts_compute_time_us']:.0f}; mean stamped ticks/candidate {tot:.3e} ")
for i, n in enumerate(names):
    print(f"  {n:26s} {100*prof[:, i].mean()/tot:6.2f} %   {prof[:, i].mean()/tot*out['rollouts_compute_time_us']/H:8.2f} us/step")
print("ls evals per step (mean)", prof[:, 23].mean() / H) if W == 0 else print("worker: active (contact, row) pairs per job", prof[:, 23].sum() / max(prof[:, 22].sum(), 1), "jobs per step", prof[:, 22].mean() / H)
print("newton iters per step (mean)", allc["diag"][:, 0].mean() / H, "max ncon", allc["diag"][:, 1].max(), "max nefc", allc["diag"][:, 2].max())
it = allc["diag"][:, 0] / H
tk = prof[:, :23].sum(1)
print(f"newton iters/step per candidate: min {it.min():.2f} mean {it.mean():.2f} p90 {np.percentile(it, 90):.2f} max {it.max():.2f}")
print(f"stamped ticks per candidate: min {tk.min():.3e} mean {tk.mean():.3e} max {tk.max():.3e}  (max/mean {tk.max()/tk.mean():.2f})")
