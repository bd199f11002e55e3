"""Scheduler plugin surface, mirroring the reference's `schedulers/` package for the heuristics
(reference schedulers/scheduler.py:10-18, schedulers/heuristics/{round_robin,random_scheduler,utils}.py).

A plugin written for the reference - anything with `.schedule(obs) -> (action_dict, info_dict)` -
runs unchanged against `SparkSchedSimEnv` (env.py) or `VecSparkSchedSimEnv.obs_view(i)`: those
produce the reference's observation dict. The classes below are this repo's own implementations
of the reference's two heuristic plugins on that dict (host side, one env at a time); their
batched on-device counterparts are `VecSparkSchedSimEnv.policy_actions("fair" | "fifo" | "hash")`.
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any

import numpy as np


class Scheduler(ABC):
    """Interface for all schedulers (reference schedulers/scheduler.py:10-18)"""

    name: str
    env_wrapper_cls: Any | None

    @abstractmethod
    def schedule(self, obs: dict) -> tuple[dict, dict]:
        ...


def preprocess_obs(obs: dict[str, Any]) -> None:
    """adds `frontier_stages` (nodes without an incoming edge in the active subgraph) and
    `schedulable_stages` (node index -> position among schedulable nodes) to the observation;
    same keys as the reference helper (schedulers/heuristics/utils.py:5-14)."""
    nodes = obs["dag_batch"].nodes
    has_parent = np.zeros(nodes.shape[0], dtype=bool)
    has_parent[obs["dag_batch"].edge_links[:, 1]] = True
    sched_nodes = np.flatnonzero(nodes[:, 2] != 0)
    obs["frontier_stages"] = set(np.flatnonzero(~has_parent).tolist())
    obs["schedulable_stages"] = {int(n): i for i, n in enumerate(sched_nodes)}


def find_stage(obs: dict[str, Any], job_idx: int) -> int:
    """first schedulable frontier stage of the job, else its first schedulable stage, else -1
    (reference schedulers/heuristics/utils.py:17-37)"""
    lo, hi = obs["dag_ptr"][job_idx], obs["dag_ptr"][job_idx + 1]
    fallback = -1
    for node in range(lo, hi):
        i = obs["schedulable_stages"].get(node)
        if i is None:
            continue
        if node in obs["frontier_stages"]:
            return i
        if fallback == -1:
            fallback = i
    return fallback


class RoundRobinScheduler(Scheduler):
    """Spark's fair ("Fair", dynamic_partition=True) / FIFO scheduler as in the reference
    (schedulers/heuristics/round_robin.py:7-49)"""

    def __init__(self, num_executors: int, dynamic_partition: bool = True):
        self.name = "Fair" if dynamic_partition else "FIFO"
        self.num_executors = num_executors
        self.dynamic_partition = dynamic_partition
        self.env_wrapper_cls = None

    def schedule(self, obs: dict) -> tuple[dict, dict]:
        preprocess_obs(obs)
        n_jobs = len(obs["exec_supplies"])
        cap = int(np.ceil(self.num_executors / max(1, n_jobs))) if self.dynamic_partition else self.num_executors
        committable = obs["num_committable_execs"]
        src = obs["source_job_idx"]
        # the job that is releasing executors goes first
        if src < n_jobs:
            idx = find_stage(obs, src)
            if idx != -1:
                return {"stage_idx": idx, "num_exec": committable}, {}
        # then jobs in arrival order that are below their share
        for j in range(n_jobs):
            if j == src or obs["exec_supplies"][j] >= cap:
                continue
            idx = find_stage(obs, j)
            if idx != -1:
                return {"stage_idx": idx, "num_exec": min(committable, cap - obs["exec_supplies"][j])}, {}
        return {"stage_idx": -1, "num_exec": committable}, {}


class RandomScheduler(Scheduler):
    """The reference's random heuristic (schedulers/heuristics/random_scheduler.py:7-32): jobs are
    tried in a random order until one has a stage to offer, then a random executor count.

    What is pinned by tests/golden/c1_random.npz is the stream, not the text: the draws come from a
    legacy MT19937 `numpy.random.RandomState(seed)`; each job pick is one `choice` over the jobs not
    yet rejected in this call (in active order), and the executor count is one `randint` over
    [1, num_committable_execs] made after the job search, whether or not a stage was found."""

    name = "Random"
    env_wrapper_cls = None

    def __init__(self, seed: int = 42):
        self.set_seed(seed)

    def set_seed(self, seed: int) -> None:
        self.np_random = np.random.RandomState(seed)

    def _pick_stage(self, obs: dict) -> int:
        remaining = list(range(len(obs["exec_supplies"])))
        while remaining:
            job = self.np_random.choice(remaining)
            found = find_stage(obs, job)
            if found != -1:
                return found
            remaining.remove(job)
        return -1

    def schedule(self, obs: dict) -> tuple[dict, dict]:
        preprocess_obs(obs)
        stage = self._pick_stage(obs)
        count = self.np_random.randint(1, obs["num_committable_execs"] + 1)
        return {"stage_idx": stage, "num_exec": count}, {}


def make_scheduler(agent_cfg: dict) -> Scheduler:
    """by-name factory like the reference's (schedulers/__init__.py:17-21)"""
    cfg = dict(agent_cfg)
    cls = cfg.pop("agent_cls")
    table = {"RoundRobinScheduler": RoundRobinScheduler, "RandomScheduler": RandomScheduler}
    if cls == "DecimaScheduler":  # needs torch.nn; imported on demand like the reference's optional agents
        from .decima import DecimaScheduler
        table["DecimaScheduler"] = DecimaScheduler
    assert cls in table, f"'{cls}' is not a valid scheduler."
    return table[cls](**cfg)
