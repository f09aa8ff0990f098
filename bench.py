#!/usr/bin/env python
"""bench.py — candidate rollouts/s of the Predictive-Sampling hot path on MI355X.

One "step" = one plan step = one pass of the hot path over one batch of synthetic candidates:
noise -> N rollouts of H steps (physics + residual + cost) -> argmin -> winner D2H (+ elite exchange when
sharded).  N=1 GPU runs BASELINE config C2 (Quadruped flat, 256 samples, horizon 100, 3 cubic knots, dt 0.01,
sigma 0.04, Philox(0x5EED) noise generated on device).  With --gpus G (launched by torch.distributed.run, one rank
per GPU) every rank rolls out its own 256 candidates of a global batch of 256*G ("weak") and the elite is picked
with a single all_gather over RCCL.

Prints ONE JSON line (rank 0) with `roofline` (rollout_kernel, HIP-event timed on the engine stream) and
`cpu_baseline` (the CPU oracle's ThreadPool-style plan on this box's host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)


def measured_traffic(N, H):
    """HBM bytes per rollout_kernel launch from the committed rocprofv3 PMC passes (profiles/r1/g_traffic.json:
    FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE); only valid for the workload it was measured on."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "r1", "g_traffic.json")))
        if t.get("workload") == f"C2 quadruped {N}x{H}" and N == 256:
            return t["traffic_bytes_per_launch"]
    except Exception:
        pass
    return None


def algorithmic_bytes_per_candidate_step(model, task):
    """the Trajectory I/O contract in fp64: state + action + residual + trace + time + cost."""
    ds = model["nq"] + model["nv"] + model["na"]
    return 8 * (ds + model["nu"] + task["num_residual"] + 3 * task["num_trace"] + 2)


def cpu_baseline(model, task, d, kt, kv, N, H, sigma, seconds_target=12.0):
    """Reference-shaped CPU path (oracle = our restatement; the reference cannot be built here: MuJoCo absent)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    g.build_oracle()
    import oracle_lib as ol
    ol.FAST = True                           # -O3 -march=native build of the oracle, compiled on this box
    ncpu = os.cpu_count() or 1
    threads = max(1, ncpu - 5)               # testspeed default: hw_threads - 5 (mjpc/testspeed_app.cc:24)
    o = ol.Oracle(model, task)
    n = min(N, 4 * threads)
    t0 = time.perf_counter()
    o.plan(d["state"], d["mocap"], 0.0, kt, kv, 2, n, H, sigma=(sigma, 0.0), seed=0x5EED, stream=0, nthreads=threads)
    probe = time.perf_counter() - t0
    per_rollout = probe / n                   # wall seconds per rollout at this thread count
    reps = max(1, int(seconds_target / max(per_rollout * N, 1e-9)))
    reps = min(reps, 20)
    t0 = time.perf_counter()
    for r in range(reps):
        o.plan(d["state"], d["mocap"], 0.0, kt, kv, 2, N, H, sigma=(sigma, 0.0), seed=0x5EED, stream=r, nthreads=threads)
    dt = time.perf_counter() - t0
    return dict(value=N * reps / dt, unit="rollouts/s", cores=threads, kind="port",
                sample=f"{reps} plan step(s) of the same workload (N={N}, H={H}) on the CPU oracle's FIFO pool (gcc -O3 -march=native), "
                       f"{threads} threads of {ncpu} host cpus; plan-step {1e3 * dt / reps:.1f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="quadruped", choices=["quadruped", "humanoid"],
                    help="quadruped = BASELINE configs[1] (the metric's config, default); humanoid = configs[2]")
    ap.add_argument("--samples", type=int, default=None, help="candidates per GPU (default 256 / 1024)")
    ap.add_argument("--global-samples", type=int, default=None,
                    help="strong-scaling mode: a fixed global batch (e.g. 4096 = BASELINE configs[3]) split evenly over the ranks")
    ap.add_argument("--horizon", type=int, default=None, help="default 100 / 128")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    dist = None
    # rehearsal knobs for a 1-GPU box (never set by the driver): all ranks on device 0, exchange over gloo
    rehearsal = os.environ.get("BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    from mujoco_mpc_amd.modelgen import humanoid_track, quadruped
    from mujoco_mpc_amd.planner import HipBackend
    from mujoco_mpc_amd.sharded import ShardedSampler

    if args.workload == "quadruped":
        model, task, d = quadruped()
        N, H, P, sigma, wname = args.samples or 256, args.horizon or 100, 3, 0.04, "Quadruped flat (A1)"
    else:
        model, task, d = humanoid_track()
        N, H, P, sigma, wname = args.samples or 1024, args.horizon or 128, 16, 0.15, "Humanoid tracking (Jump)"
    scaling = "weak"
    if args.global_samples:
        if args.global_samples % world:
            raise SystemExit(f"--global-samples {args.global_samples} is not divisible by {world} ranks")
        N, scaling = args.global_samples // world, "strong"
    dt_model = model["timestep"]
    kt = np.linspace(0.0, (H - 1) * dt_model, P)
    kv = np.zeros((P, model["nu"]))
    be = HipBackend(model, task, max_samples=N, max_horizon=H, device=dev_index if world > 1 else 0)
    sampler = ShardedSampler(be, rank, world, N, dist=dist, device=("cpu" if rehearsal else f"cuda:{local_rank}") if world > 1 else None)

    def step(i, knots):
        return sampler.plan(state=d["state"], mocap=d["mocap"], time=0.0, knot_times=kt, knot_values=knots,
                            interpolation=2, horizon=H, sigma=(sigma, 0.0), seed=0x5EED, stream=i)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    knots = kv
    for i in range(args.warmup):
        knots = step(i, knots)["winner_knots"]
    be.kernel_time()                       # reset the HIP-event accumulators
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        res = step(args.warmup + i, knots)
        knots = res["winner_knots"]
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    nlaunch, rollout_us, total_us = be.kernel_time()

    if rank == 0:
        b_step = algorithmic_bytes_per_candidate_step(model, task)
        bytes_per_launch = b_step * N * H
        achieved = bytes_per_launch / (rollout_us * 1e-6) / 1e9 if rollout_us > 0 else 0.0
        total_rollouts = N * world * args.steps
        out = {
            "metric": "candidate rollouts/s (horizon x samples)",
            "value": total_rollouts / elapsed,
            "unit": "rollouts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{wname}, SamplingPlanner rollouts, {N} samples/GPU x horizon {H}, "
                                   f"{P} cubic knots, dt {dt_model}, sigma {sigma}, Philox(0x5EED) noise "
                                   f"(BASELINE configs[{1 if args.workload == 'quadruped' else 2}])",
                       "samples_per_gpu": N, "global_samples": N * world, "horizon": H,
                       "candidate_steps_per_s": total_rollouts * H / elapsed,
                       "lds_bytes_per_candidate": be.lds_bytes(),
                       "winner": res["winner"], "winner_return": res["winner_return"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(N, H),
                         "kernel": "rollout_kernel", "avg_launch_us": rollout_us, "launches": nlaunch,
                         "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_per_candidate_step": b_step,
                         "plan_device_us": total_us},
        }
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(model, task, d, kt, kv, N, H, sigma)
        print(json.dumps(out))
    be.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
