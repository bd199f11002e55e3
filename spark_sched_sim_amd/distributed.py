"""Multi-GPU use of the simulator: environments are independent (own RNG stream, own state), so
they shard across ranks with NO per-step traffic - one process per GPU, `torch.distributed` (RCCL
over xGMI on the GPU box; gloo in the CPU tests). The only exchange is one all-gather of per-env
episode summaries per rollout, which stands in for the reference's Pipe gather of rollout
statistics (reference trainers/trainer.py:113-121, rollout_worker.py:122-129). The message is
tiny (4 f64 per env), i.e. latency-bound on xGMI; there is nothing to tune.

Seeds are a function of the GLOBAL env id (rank * envs_per_rank + i), so results do not depend
on how envs are placed on GPUs.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist

SUMMARY_FIELDS = ("last_ep_return", "last_ep_steps", "last_ep_wall", "episodes")


def global_env_ids(envs_per_rank: int, rank: int | None = None) -> range:
    rank = dist.get_rank() if rank is None else rank
    return range(rank * envs_per_rank, (rank + 1) * envs_per_rank)


def shard_seeds(base_seed: int, envs_per_rank: int, rank: int | None = None) -> list[int]:
    return [base_seed + g for g in global_env_ids(envs_per_rank, rank)]


def episode_summaries(env) -> torch.Tensor:
    """f64[B, 4] device tensor: return, length and end time of each env's last finished episode,
    and its number of finished episodes"""
    cols = [env.header_field(f).to(torch.float64) for f in SUMMARY_FIELDS]
    return torch.stack(cols, dim=1).contiguous()


def all_gather_episode_summaries(env, group=None) -> torch.Tensor:
    """f64[world * B, 4] on every rank, rows ordered by global env id (ONE all-gather)"""
    local = episode_summaries(env)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    out = [torch.empty_like(local) for _ in range(dist.get_world_size(group))]
    dist.all_gather(out, local, group=group)
    return torch.cat(out, dim=0)


def launch_ranks(world_size: int, argv: list[str], port: int | None = None, env: dict | None = None,
                 timeout: float | None = None, grace: float = 5.0) -> int:
    """Starts `world_size` fresh Python processes running `argv` (script + arguments), one rank per
    GPU, with the torch.distributed environment set (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR =
    127.0.0.1, MASTER_PORT) - the stand-in for the reference's own worker start-up
    (trainers/trainer.py:264-293 spawns its rollout processes itself). The caller must not have
    touched the GPU: every rank is a new process, nothing is exec'ed over an initialised one.
    Rank 0 inherits stdout (it prints the result), the other ranks' stdout goes to stderr.

    All ranks are watched together: when one exits with a non-zero code, or `timeout` seconds pass,
    the ranks still running are terminated (SIGTERM, SIGKILL after `grace` seconds) - a rank that died
    at start-up would otherwise leave its siblings in `init_process_group` / a collective until the
    backend's own timeout. Only these children are signalled (by pid). Returns the first non-zero exit
    code seen, 124 after a timeout, else 0."""
    import time

    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    procs = []
    for rank in range(world_size):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world_size), LOCAL_WORLD_SIZE=str(world_size),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=None if rank == 0 else sys.stderr))

    def stop_rest() -> None:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            p.terminate()
        t_end = time.monotonic() + grace
        for p in live:
            try:
                p.wait(timeout=max(0.0, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()

    t0 = time.monotonic()
    rc = 0
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [c for c in codes if c not in (None, 0)]
            if bad:
                rc = bad[0]
                print(f"[launch_ranks] rank {codes.index(bad[0])} exited with code {bad[0]}: stopping the other ranks", file=sys.stderr)
                break
            if all(c == 0 for c in codes):
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                rc = 124
                print(f"[launch_ranks] {timeout:.0f} s passed: stopping {sum(c is None for c in codes)} running rank(s)", file=sys.stderr)
                break
            time.sleep(0.05)
    finally:
        stop_rest()
    return rc
