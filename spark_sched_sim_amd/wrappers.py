"""Episode time limits over the batched env, the rule of the reference's `StochasticTimeLimit`
wrapper (reference spark_sched_sim/wrappers/stochastic_time_limit.py:5-31, used by the trainers,
rollout_worker.py:83): every episode gets a limit drawn from an exponential distribution; the
limit is handed to `reset` as `options["time_limit"]` (it also ends the job arrival sequence,
tpch.py:63) and `truncated` is raised from the step on which `wall_time >= limit`.

`LimitStream` is the sampling rule for one env (a legacy `numpy.random.RandomState`, re-seeded by
a truthy reset seed and by nothing else - a seed of 0 keeps the stream running, :15).
`VecStochasticTimeLimit` applies it to every env of a `VecSparkSchedSimEnv` with the truncation
test on the device; `StochasticTimeLimit` is the same thing around the single-env facade.
"""
from __future__ import annotations

from typing import Any, Sequence

import numpy as np
import torch


class LimitStream:
    """the time limits of one env's successive episodes"""

    def __init__(self, mean_time_limit: float, seed: int = 42):
        self.mean = mean_time_limit
        self.rs = np.random.RandomState(seed)

    def next_limit(self, reset_seed: int | None) -> float:
        if reset_seed:  # None and 0 leave the stream where it is
            self.rs = np.random.RandomState(reset_seed)
        return float(self.rs.exponential(self.mean))


class VecStochasticTimeLimit:
    """one `LimitStream` per env of a `VecSparkSchedSimEnv`"""

    def __init__(self, env, mean_time_limit: float, seed: int = 42):
        if getattr(env, "auto_reset", False):
            # the kernels' auto-reset restarts an episode with the old limit and only on termination:
            # truncated episodes and fresh limits need the masked reset below
            raise ValueError("VecStochasticTimeLimit needs an env created with auto_reset=False")
        self.env = env
        self.mean_time_limit = mean_time_limit
        self._streams = [LimitStream(mean_time_limit, seed) for _ in range(env.num_envs)]
        self.time_limit = torch.full((env.num_envs,), float("inf"), dtype=torch.float64, device=env.device)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, *, seed: int | Sequence[int] | None = None, options: dict[str, Any] | None = None,
              mask: torch.Tensor | Sequence[bool] | None = None):
        """`mask` (bool per env): only those envs are reset and get a new limit (rollout collection
        resets envs one by one as their episodes end, rollout_worker.py:195-199)"""
        B = self.env.num_envs
        if seed is None:
            seeds: list[int | None] = [None] * B
        elif isinstance(seed, (int, np.integer)):
            seeds = [int(seed) + i for i in range(B)]
        else:
            seeds = [int(s) for s in seed]
        sel = np.ones(B, dtype=bool) if mask is None else np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask, dtype=bool)
        limits = self.time_limit.cpu().numpy().copy()
        for i in np.flatnonzero(sel):
            limits[i] = self._streams[i].next_limit(seeds[i])
        self.time_limit = torch.from_numpy(limits).to(self.env.device)
        options = dict(options or {})
        options["time_limit"] = limits
        kw = {} if mask is None else {"mask": torch.from_numpy(sel).to(self.env.device)}
        return self.env.reset(seed=None if seed is None else seeds, options=options, **kw)

    def step(self, actions):
        obs, rew, term, trunc, info = self.env.step(actions)
        return obs, rew, term, trunc | (info["wall_time"] >= self.time_limit), info

    def close(self):
        return self.env.close()


class StochasticTimeLimit:
    """the same rule around the single-env facade `SparkSchedSimEnv` (what the reference's
    `gymnasium.Wrapper` is to its env): `reset(seed, options)` / `step(action)` of the facade with
    the limit in between, attribute access forwarded to the wrapped env"""

    def __init__(self, env, mean_time_limit: float, seed: int = 42, verbose: bool = False):
        self.env = env
        self._stream = LimitStream(mean_time_limit, seed)
        self.verbose = verbose
        self.time_limit = float("inf")

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, seed=None, options=None):
        self.time_limit = self._stream.next_limit(seed)
        if self.verbose:
            print(f"new episode: seed {seed}, time limit {self.time_limit * 1e-3:.0f} s", flush=True)
        return self.env.reset(seed=seed, options=dict(options or {}, time_limit=self.time_limit))

    def step(self, action):
        obs, rew, term, trunc, info = self.env.step(action)
        return obs, rew, term, trunc or info["wall_time"] >= self.time_limit, info

    def close(self):
        return self.env.close()
