"""The N > 1 path on CPU: two gloo ranks, each with its own shard of (emulated) envs, one
all-gather of episode summaries; the gathered table must equal a single-process run of the union
(placement invariance), row for row."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(num_executors=5, job_arrival_cap=8, job_arrival_rate=1.0e-4, moving_delay=1500.0, warmup_delay=500.0)
ENVS_PER_RANK, BASE_SEED, STEPS = 3, 500, 220


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_shard(rank: int, world: int):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from emu_util import load_emu
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.distributed import all_gather_episode_summaries, shard_seeds

    env = VecSparkSchedSimEnv(CFG, ENVS_PER_RANK, device="cpu", _lib=load_emu(), auto_reset=False)
    env.reset(seed=shard_seeds(BASE_SEED, ENVS_PER_RANK, rank))
    env.rollout("fair", STEPS)
    table = all_gather_episode_summaries(env)
    env.close()
    return table


def _worker(rank: int, world: int, port: int, out_path: str):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    table = _run_shard(rank, world)
    if rank == 0:
        torch.save(table, out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_all_gather(tmp_path):
    sys.path[:0] = [os.path.dirname(HERE), HERE]
    from emu_util import load_emu

    load_emu()  # build once, before forking workers
    out = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    gathered = torch.load(out)
    assert gathered.shape == (2 * ENVS_PER_RANK, 4)

    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.distributed import episode_summaries

    env = VecSparkSchedSimEnv(CFG, 2 * ENVS_PER_RANK, device="cpu", _lib=load_emu())
    env.reset(seed=BASE_SEED)
    env.rollout("fair", STEPS)
    single = episode_summaries(env)
    env.close()
    assert torch.equal(gathered, single)
    assert (single[:, 3] == 1).all(), "every env should have finished its episode"
