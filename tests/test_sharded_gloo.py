"""N>1 path on CPU: world_size-2 gloo processes shard the candidates and pick the elite with one all_gather.
The rollouts come from the oracle backend here (no GPU in this tier); the code under test is
mujoco_mpc_amd/sharded.py, which is backend-agnostic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nper, outdir, equal_returns):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mujoco_mpc_amd.modelgen import cartpole
    from mujoco_mpc_amd.sharded import ShardedSampler
    from oracle_backend import OracleBackend
    m, task, d = cartpole()
    be = OracleBackend(m, task)
    sampler = ShardedSampler(be, rank, world, nper, dist=dist, device="cpu")
    kt = np.linspace(0, 0.29, 5); kv = np.zeros((5, 1))
    state = d["state"] if not equal_returns else np.array([0.0, 0.0, 1e11, 0.0])   # every rollout fails -> all returns 1e6
    res = sampler.plan(state=state, mocap=None, time=0.0, knot_times=kt, knot_values=kv, interpolation=2, horizon=30,
                       sigma=(0.5, 0.0), seed=3, stream=1)
    traj = res.get("trajectory")
    np.savez(os.path.join(outdir, f"r{rank}.npz"), winner=res["winner"], ret=res["winner_return"], knots=res["winner_knots"],
             owner=res["owner"], local_returns=res["local"]["returns"], has_traj=traj is not None,
             traj_states=traj["states"] if traj is not None else np.zeros(0), summary_mode=be.summary_only,
             local_rows_moved=bool(np.any(res["local"]["states"] != 0)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("equal_returns", [False, True])
def test_two_rank_sharding_matches_single_process(tmp_path, equal_returns):
    world, nper = 2, 8
    port = _free_port()
    mp.spawn(_worker, args=(world, port, nper, str(tmp_path), equal_returns), nprocs=world, join=True)
    r = [np.load(tmp_path / f"r{k}.npz") for k in range(world)]
    # both ranks agree on the elite
    assert int(r[0]["winner"]) == int(r[1]["winner"]) and float(r[0]["ret"]) == float(r[1]["ret"])
    assert np.array_equal(r[0]["knots"], r[1]["knots"]) and int(r[0]["owner"]) == int(r[1]["owner"])
    # and it is what one process computes over all 16 candidates
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    from mujoco_mpc_amd.modelgen import cartpole
    m, task, d = cartpole()
    state = d["state"] if not equal_returns else np.array([0.0, 0.0, 1e11, 0.0])
    full = ol.Oracle(m, task).plan(state, None, 0.0, np.linspace(0, 0.29, 5), np.zeros((5, 1)), 2, world * nper, 30,
                                   sigma=(0.5, 0.0), seed=3, stream=1)
    assert np.array_equal(np.concatenate([r[0]["local_returns"], r[1]["local_returns"]]), full["returns"])
    assert int(r[0]["winner"]) == full["winner"]
    assert float(r[0]["ret"]) == full["returns"][full["winner"]]
    assert np.array_equal(r[0]["knots"], full["knots"][full["winner"]])
    # shards report summaries; only the owner of the global elite copies a trajectory to the host
    owner = int(r[0]["owner"])
    for k in range(world):
        assert bool(r[k]["summary_mode"]) and not bool(r[k]["local_rows_moved"])
        assert bool(r[k]["has_traj"]) == (k == owner)
    assert np.array_equal(r[owner]["traj_states"], full["states"][full["winner"]])
    if equal_returns:
        assert int(r[0]["winner"]) == 0          # ties -> lowest global index, like the single-GPU argmin


def test_single_rank_passthrough():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mujoco_mpc_amd.modelgen import cartpole
    from mujoco_mpc_amd.sharded import ShardedSampler
    from oracle_backend import OracleBackend
    m, task, d = cartpole()
    s = ShardedSampler(OracleBackend(m, task), 0, 1, 6)
    res = s.plan(state=d["state"], mocap=None, time=0.0, knot_times=np.array([0.0, 0.1]), knot_values=np.zeros((2, 1)),
                 interpolation=1, horizon=10, sigma=(0.5, 0.0), seed=1, stream=0)
    assert res["owner"] == 0 and res["winner"] == res["local"]["winner"]


def test_bench_with_gpus_flag_and_no_launcher_starts_its_own_ranks_or_fails():
    """`python bench.py --gpus 2` without WORLD_SIZE must start 2 ranks itself (torch.distributed.run, 127.0.0.1) - on this GPU-less
    box they fail, so the command exits non-zero; what it must never do is fall back to one rank and print an `n_gpus: 1` line"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert '"n_gpus": 1' not in r.stdout
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
