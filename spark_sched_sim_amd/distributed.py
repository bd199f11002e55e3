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
