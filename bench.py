#!/usr/bin/env python
"""bench.py — candidate rollouts/s of the Predictive-Sampling hot path on MI355X.

One "step" = one plan step = one pass of the hot path over one batch of synthetic candidates:
noise -> N rollouts of H steps (physics + residual + cost) -> argmin -> winner D2H (+ elite exchange when
sharded).

* 1 GPU (default): BASELINE configs[1] "C2" = Quadruped flat, 256 samples, horizon 100, 3 cubic knots, dt 0.01,
  sigma 0.04, Philox(0x5EED) noise generated on device.  The same JSON line also carries `strong_scaling_ref`, the
  4096-sample batch of configs[3] "C4" on this one GPU (the 1-GPU point of the strong-scaling curve).
* --gpus G > 1 (launched by torch.distributed.run, one rank per GPU over RCCL): the headline is BASELINE configs[3]
  "C4" = Quadruped flat, 4096 samples GLOBAL, horizon 100, block-partitioned over the ranks ("strong" scaling, the
  north star's scaling target) with one all_gather for the elite; the weak-scaling line (256 samples per GPU) of the
  same run sits beside it under `weak`.  `--mode weak` makes the weak line the headline instead.

Prints ONE JSON line (rank 0) with `roofline` (rollout_kernel, HIP-event timed on the engine stream) and
`cpu_baseline` (the CPU oracle on a PERSISTENT FIFO worker pool — the reference's ThreadPool shape — timed on this
box's host cores at T = 1, T = physical cores and T = hw-5 threads, bounded sample).
"""
import argparse
import gc
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md)
TRAFFIC_FILES = ["profiles/r3/traffic.json", "profiles/r2/traffic.json", "profiles/r1/g_traffic.json"]


def profiled_traffic(workload_key):
    """HBM bytes per rollout_kernel launch.  NOT measured inside this run: PMC counters need rocprofv3 around the
    process, so the number is read from the committed rocprofv3 --pmc passes of this same command (FETCH_SIZE /
    WRITE_SIZE with the gfx950 corrections of MI355X_MICROARCH.md) and labelled with its source file."""
    for rel in TRAFFIC_FILES:
        try:
            t = json.load(open(os.path.join(ROOT, rel)))
        except Exception:
            continue
        entries = t if isinstance(t, list) else [t]
        for e in entries:
            if e.get("workload") == workload_key:
                return e["traffic_bytes_per_launch"], rel
    return None, None


def algorithmic_bytes_per_candidate_step(model, task):
    """the Trajectory I/O contract in fp64: state + action + residual + trace + time + cost."""
    ds = model["nq"] + model["nv"] + model["na"]
    return 8 * (ds + model["nu"] + task["num_residual"] + 3 * task["num_trace"] + 2)


def physical_cores():
    try:
        seen = set()
        for d in os.listdir("/sys/devices/system/cpu"):
            p = f"/sys/devices/system/cpu/{d}/topology/thread_siblings_list"
            if d.startswith("cpu") and d[3:].isdigit() and os.path.exists(p):
                seen.add(open(p).read().strip())
        if seen:
            return len(seen)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_share():
    """CPUs this process may actually use: the smaller of its affinity mask and its cgroup CPU quota (the GPU boxes hand a
    container a share of the host's cores; os.cpu_count() still reports all of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]            # cgroup v2
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return n, quota


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(model, task, d, kt, kv, N, H, interp, sigma, budget_s=24.0):
    """Reference-shaped CPU path (oracle = our restatement; the reference cannot be built here: MuJoCo absent).
    Persistent worker pool + per-worker data (threadpool.cc:30-85, planners/planner.cc:23-33); median plan-step time."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import __graft_entry__ as g
    g.build_oracle()
    import oracle_lib as ol
    ol.FAST = True                           # -O3 -march=native build of the oracle, compiled on this box
    ncpu = os.cpu_count() or 1
    phys = min(physical_cores(), ncpu)
    affinity, quota = cpu_share()
    share = int(round(min(affinity, quota))) if quota else affinity
    o = ol.Oracle(model, task)

    def run(threads, n, max_plans, budget):
        o.open_pool(threads)
        times = []
        t_begin = time.perf_counter()
        warm = 2 if threads > 1 else 0
        for r in range(warm + max_plans):
            t0 = time.perf_counter()
            o.plan(d["state"], d["mocap"], 0.0, kt, kv, interp, n, H, sigma=(sigma, 0.0), seed=0x5EED, stream=r)
            if r >= warm:
                times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget and len(times) >= 3:
                break
        o.close_pool()
        med = statistics.median(times)
        return dict(threads=threads, samples=n, plans=len(times), median_plan_ms=1e3 * med, rollouts_per_s=n / med)

    sweep = []
    # T = 1: a per-core figure on a slice of the batch (a full 256-candidate plan is ~4 s on one core)
    n1 = max(4, min(N, 16))
    sweep.append(run(1, n1, 5, budget_s * 0.2))
    per_thread = sweep[0]["rollouts_per_s"]
    # thread counts: the CPU share this container really has, the physical cores and hw-5 (testspeed default,
    # mjpc/testspeed_app.cc:24) of the host; beyond the share the extra threads only time-slice
    counts = sorted({max(2, share), phys, max(1, ncpu - 5)} if share < ncpu else {phys, max(1, ncpu - 5)})
    for T in counts:
        if T > 1:
            sweep.append(run(T, N, 50, budget_s * 0.8 / len(counts)))
    for s in sweep:
        s["parallel_efficiency"] = s["rollouts_per_s"] / (per_thread * s["threads"])
    best = max(sweep, key=lambda s: s["rollouts_per_s"])
    return dict(value=best["rollouts_per_s"], unit="rollouts/s", cores=best["threads"], kind="port",
                sample=f"median of {best['plans']} plan steps of the same workload (N={best['samples']}, H={H}) after 2 warm-ups on the CPU "
                       f"oracle's persistent FIFO pool (gcc -O3 -march=native), best of the thread counts in `sweep`; "
                       f"{ncpu} host cpus / {phys} physical cores, {cpu_model_name()}",
                sweep=sweep, host_cpus=ncpu, physical_cores=phys, cpu_affinity=affinity, cgroup_cpu_quota=quota,
                note="parallel_efficiency is relative to the T=1 rate; thread counts above the container's CPU share "
                     "(cgroup quota / affinity) cannot speed up")


WORKLOADS = {
    # name: (generator, samples, horizon, knots, interpolation, sigma, label, BASELINE config index)
    "quadruped": ("quadruped", 256, 100, 3, 2, 0.04, "Quadruped flat (A1)", 1),
    "humanoid": ("humanoid_track", 1024, 128, 16, 2, 0.15, "Humanoid tracking (Jump)", 2),
    "hand": ("shadow_hand", 2048, 64, 5, 0, 0.1, "Shadow-hand cube reorientation (synthetic hand)", 4),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="quadruped", choices=sorted(WORKLOADS),
                    help="quadruped = BASELINE configs[1] (the metric's config, default); humanoid = configs[2]; hand = configs[4]")
    ap.add_argument("--mode", default=None, choices=["weak", "strong"],
                    help="N > 1 only: which line is the headline (default strong = configs[3], 4096 global samples)")
    ap.add_argument("--samples", type=int, default=None, help="candidates per GPU of the weak line (default 256 / 1024 / 256)")
    ap.add_argument("--global-samples", type=int, default=None, help="global batch of the strong line (default 4096; hand: 2048)")
    ap.add_argument("--horizon", type=int, default=None, help="default 100 / 128 / 64")
    ap.add_argument("--tier", default=None, choices=["A", "B"],
                    help="diagnostics (include/mjpc_hip_debug.h): A = full-capacity kernel only, B = always the dense tier first; default: the engine decides")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary line (strong_scaling_ref at N=1, weak at N>1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: start the N ranks ourselves as CHILD processes (the driver's command line, 127.0.0.1
        # rendezvous) before this process has touched torch or the GPU, and leave with their exit code — a `--gpus 8` command
        # never degrades to a one-rank line
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
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
    from mujoco_mpc_amd import modelgen
    from mujoco_mpc_amd.planner import HipBackend
    from mujoco_mpc_amd.sharded import ShardedSampler
    if args.tier:
        from mujoco_mpc_amd import capi
        capi.debug_set("tier", args.tier)

    gen, n_default, h_default, P, interp, sigma, wname, cfg_index = WORKLOADS[args.workload]
    model, task, d = getattr(modelgen, gen)()
    H = args.horizon or h_default
    dt_model = model["timestep"]
    shift = (H - 1) * dt_model / (P if interp == 0 else max(P - 1, 1))
    kt = np.arange(P) * shift if interp == 0 else np.linspace(0.0, (H - 1) * dt_model, P)
    kv = np.zeros((P, model["nu"]))
    if "ctrl0" in d:                             # hand: the nominal plan holds the grasp posture (position actuators)
        kv = np.tile(np.asarray(d["ctrl0"], float), (P, 1))
    n_weak = args.samples or (n_default if args.workload != "hand" else 256)
    n_global = args.global_samples or (4096 if args.workload == "quadruped" else n_default)
    if world > 1 and n_global % world:
        raise SystemExit(f"global batch {n_global} is not divisible by {world} ranks")
    mode = args.mode or ("strong" if world > 1 else "weak")
    if world == 1 and args.global_samples:
        mode = "strong"
    coll_device = ("cpu" if rehearsal else f"cuda:{local_rank}") if world > 1 else None

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(n_per_rank, steps, warmup):
        """steps timed plan steps of n_per_rank candidates on every rank; returns the result record (rank 0) or None."""
        be = HipBackend(model, task, max_samples=n_per_rank, max_horizon=H, device=dev_index if world > 1 else 0)
        sampler = ShardedSampler(be, rank, world, n_per_rank, dist=dist, device=coll_device)

        def step(i, knots):
            return sampler.plan(state=d["state"], mocap=d["mocap"], time=0.0, knot_times=kt, knot_values=knots,
                                interpolation=interp, horizon=H, sigma=(sigma, 0.0), seed=0x5EED, stream=i)
        knots = kv
        for i in range(warmup):
            knots = step(i, knots)["winner_knots"]
        be.kernel_time()                       # reset the HIP-event accumulators
        per_step = []; noise_us = []; roll_us = []
        # no cyclic garbage collection inside the timed region (as timeit does): with torch imported a full collection costs
        # ~40 ms of host time, and one used to land in the first ten plan steps of every run
        gc.collect(); gc.disable()
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            ts = time.perf_counter()
            res = step(warmup + i, knots)
            knots = res["winner_knots"]
            per_step.append(time.perf_counter() - ts)     # a plan step is blocking (winner D2H + elite exchange inside)
            noise_us.append(res["local"]["noise_compute_time_us"]); roll_us.append(res["local"]["rollouts_compute_time_us"])
        sync()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if world > 1:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        nlaunch, rollout_us, total_us = be.kernel_time()
        if os.environ.get("BENCH_DEBUG") and rank == 0:
            print("per-step ms:", " ".join(f"{1e3 * x:.2f}" for x in per_step), file=sys.stderr)
            print("rollout ms: ", " ".join(f"{1e-3 * x:.2f}" for x in roll_us), file=sys.stderr)
        rec = dict(n_per_rank=n_per_rank, elapsed=elapsed, steps=steps, rollout_us=rollout_us, total_us=total_us, launches=nlaunch,
                   median_ms=1e3 * statistics.median(per_step), lds=be.lds_bytes(), winner=res["winner"], winner_return=res["winner_return"],
                   phases_ms=dict(noise=1e-3 * statistics.median(noise_us), rollouts=1e-3 * statistics.median(roll_us),
                                  policy_update_and_exchange=max(0.0, 1e3 * statistics.median(per_step) - 1e-3 * (statistics.median(noise_us) + statistics.median(roll_us)))))
        be.close()
        return rec

    def measure_one_process(n_total, engines, steps, warmup):
        """The other multi-GPU front end (csrc/multi.cc: ONE planner process, one engine per GPU, host-side elite pick) timed the only
        way a one-GPU box allows: `engines` engines on this GPU sharing the batch of the headline (wall clock per mjpc_hip_multi_plan)."""
        from mujoco_mpc_amd.planner import HipMultiBackend
        mb = HipMultiBackend(model, task, [dev_index] * engines, max_samples=n_total, max_horizon=H)
        knots = kv; ts = []
        for i in range(warmup + steps):
            t0 = time.perf_counter()
            o = mb.plan(state=d["state"], mocap=d["mocap"], time=0.0, knot_times=kt, knot_values=knots, interpolation=interp,
                        num_trajectory=n_total, horizon=H, sigma=(sigma, 0.0), seed=0x5EED, stream=i)
            if i >= warmup:
                ts.append(time.perf_counter() - t0)
            knots = mb.knots(n_total, P)[o["winner"]]
        mb.close()
        med = statistics.median(ts)
        return dict(engines_on_this_gpu=engines, global_samples=n_total, steps=steps, median_ms_per_step=1e3 * med, value=n_total / med,
                    unit="rollouts/s", winner=int(o["winner"]), winner_return=float(o["winner_return"]),
                    note="mjpc_hip_multi_plan rehearsal: the engines share ONE GPU here, so this prices the one-process path's host "
                         "overhead (G async launches, G summary fetches, host elite pick, owner-only copy), not a speed-up")

    n_head = (n_global // world) if mode == "strong" else n_weak
    head = measure(n_head, args.steps, args.warmup)
    second = None
    if not args.no_secondary:
        if world == 1 and mode == "weak" and args.workload == "quadruped":
            second = ("strong_scaling_ref", measure(n_global, max(3, min(args.steps, 8)), 1))      # C4's batch on this one GPU
        elif world > 1:
            n2 = n_weak if mode == "strong" else n_global // world
            second = ("weak" if mode == "strong" else "strong", measure(n2, args.steps, args.warmup))

    if rank == 0:
        b_step = algorithmic_bytes_per_candidate_step(model, task)

        def line(rec):
            total_rollouts = rec["n_per_rank"] * world * rec["steps"]
            return dict(value=total_rollouts / rec["elapsed"], unit="rollouts/s", samples_per_gpu=rec["n_per_rank"],
                        global_samples=rec["n_per_rank"] * world, steps=rec["steps"], ms_per_step=1e3 * rec["elapsed"] / rec["steps"],
                        median_ms_per_step=rec["median_ms"], rollout_kernel_avg_us=rec["rollout_us"], phases_ms=rec["phases_ms"])
        hl = line(head)
        bytes_per_launch = b_step * head["n_per_rank"] * H
        achieved = bytes_per_launch / (head["rollout_us"] * 1e-6) / 1e9 if head["rollout_us"] > 0 else 0.0
        wkey = f"{args.workload} {head['n_per_rank']}x{H}"
        traffic, traffic_src = profiled_traffic(wkey)
        cfg_name = cfg_index if not (args.workload == "quadruped" and mode == "strong") else 3
        out = {
            "metric": "candidate rollouts/s (horizon x samples)",
            "value": hl["value"],
            "unit": "rollouts/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": hl["ms_per_step"],
            "higher_is_better": True,
            "scaling": mode,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{wname}, SamplingPlanner rollouts, {head['n_per_rank'] * world} samples global = {head['n_per_rank']} per GPU x "
                                   f"horizon {H}, {P} {'zero-order' if interp == 0 else 'cubic'} knots, dt {dt_model}, sigma {sigma}, "
                                   f"Philox(0x5EED) noise (BASELINE configs[{cfg_name}])",
                       "samples_per_gpu": head["n_per_rank"], "global_samples": head["n_per_rank"] * world, "horizon": H,
                       "candidate_steps_per_s": hl["value"] * H, "median_ms_per_step": hl["median_ms_per_step"],
                       "phases_ms": hl["phases_ms"], "lds_bytes_per_candidate": head["lds"],
                       "winner": head["winner"], "winner_return": head["winner_return"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": (f"{traffic_src} (rocprofv3 --pmc passes of this command, not measured in this run)" if traffic_src else None),
                         "kernel": "rollout_kernel", "avg_launch_us": head["rollout_us"], "launches": head["launches"],
                         "algorithmic_bytes_per_launch": bytes_per_launch, "bytes_per_candidate_step": b_step,
                         "plan_device_us": head["total_us"]},
        }
        if second is not None:
            out[second[0]] = line(second[1])
        if world == 1 and not args.no_secondary:
            out["one_process_multi_engine"] = measure_one_process(head["n_per_rank"], 2, max(3, min(args.steps, 20)), 2)
        if not args.no_cpu_baseline and world == 1:      # the CPU baseline is reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(model, task, d, kt, kv, head["n_per_rank"], H, interp, sigma)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
