"""Episode time limits, mirroring the reference's `StochasticTimeLimit` wrapper
(reference spark_sched_sim/wrappers/stochastic_time_limit.py:5-31; used by the trainers,
rollout_worker.py:83): each episode's limit is drawn from an exponential distribution with the
legacy `numpy.random.RandomState` stream, passed to `reset` as `options["time_limit"]` (which also
bounds the job arrival sequence, tpch.py:63) and `truncated` is raised once `wall_time >= limit`.

`StochasticTimeLimit` wraps the single-env facade (`SparkSchedSimEnv`);
`VecStochasticTimeLimit` does the same for a `VecSparkSchedSimEnv`, one limit per env, with the
truncation test evaluated on the device.
"""
from __future__ import annotations

from typing import Any, Sequence

import numpy as np
import torch


class StochasticTimeLimit:
    """Samples each episode's time limit from an exponential distribution"""

    def __init__(self, env, mean_time_limit: float, seed: int = 42, verbose: bool = False):
        self.env = env
        self.mean_time_limit = mean_time_limit
        self.np_random = np.random.RandomState(seed)
        self.verbose = verbose
        self.time_limit = np.inf

    def __getattr__(self, name):  # gymnasium.Wrapper attribute forwarding
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def reset(self, seed=None, options=None):
        """samples a new time limit prior to resetting"""
        if seed:  # NB a seed of 0 does not re-seed, exactly like the reference (:15)
            self.np_random = np.random.RandomState(seed)
        self.time_limit = self.np_random.exponential(self.mean_time_limit)
        if self.verbose:  # the reference prints this line unconditionally (:18-20)
            print(f"resetting. seed={seed}, timelim={int(self.time_limit * 1e-3)}s", flush=True)
        if not options:
            options = {}
        options["time_limit"] = self.time_limit
        return self.env.reset(seed=seed, options=options)

    def step(self, act):
        """modifies `truncated` signal when time limit is reached"""
        obs, rew, term, trunc, info = self.env.step(act)
        if info["wall_time"] >= self.time_limit:
            trunc = True
        return obs, rew, term, trunc, info

    def close(self):
        return self.env.close()


class VecStochasticTimeLimit:
    """one exponential time limit per env of a `VecSparkSchedSimEnv` (same sampling rule per env:
    env i behaves like `StochasticTimeLimit(env_i, mean, seed)` reset with seed_i)"""

    def __init__(self, env, mean_time_limit: float, seed: int = 42):
        self.env = env
        self.mean_time_limit = mean_time_limit
        self._rs = [np.random.RandomState(seed) for _ in range(env.num_envs)]
        self.time_limit = torch.full((env.num_envs,), float("inf"), dtype=torch.float64, device=env.device)

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, *, seed: int | Sequence[int] | None = None, options: dict[str, Any] | None = None,
              mask: torch.Tensor | Sequence[bool] | None = None):
        """`mask` (bool per env): only those envs are reset / get a new limit (rollout collection
        resets envs one by one as their episodes end, rollout_worker.py:195-199)"""
        B = self.env.num_envs
        if seed is None:
            seeds = [None] * B
        elif isinstance(seed, (int, np.integer)):
            seeds = [int(seed) + i for i in range(B)]
        else:
            seeds = [int(s) for s in seed]
        sel = np.ones(B, dtype=bool) if mask is None else np.asarray(mask.cpu() if isinstance(mask, torch.Tensor) else mask, dtype=bool)
        limits = self.time_limit.cpu().numpy().copy()
        for i, s in enumerate(seeds):
            if not sel[i]:
                continue
            if s:
                self._rs[i] = np.random.RandomState(s)
            limits[i] = self._rs[i].exponential(self.mean_time_limit)
        self.time_limit = torch.from_numpy(limits).to(self.env.device)
        options = dict(options or {})
        options["time_limit"] = limits
        kw = {} if mask is None else {"mask": torch.from_numpy(sel).to(self.env.device)}
        return self.env.reset(seed=None if seed is None else seeds, options=options, **kw)

    def step(self, actions):
        obs, rew, term, trunc, info = self.env.step(actions)
        trunc = trunc | (info["wall_time"] >= self.time_limit)
        return obs, rew, term, trunc, info

    def close(self):
        return self.env.close()
