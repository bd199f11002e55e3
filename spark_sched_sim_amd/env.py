"""`SparkSchedSimEnv`: the reference's single-environment Gymnasium API (reference
spark_sched_sim/spark_sched_sim.py:29-245) as a facade over a 1-env `VecSparkSchedSimEnv`, i.e.
over the same HIP kernels. `reset`/`step` return the reference's observation dict, rewards and
flags, and raise the reference's exceptions, so harnesses like `examples.run_episode`
(reference examples.py:84-102) and `Scheduler.schedule(obs)` plugins work unchanged.

If gymnasium is importable the class derives from `gymnasium.Env` and is registered under the
reference's id "SparkSchedSimEnv-v0"; gymnasium itself is optional.
"""
from __future__ import annotations

from typing import Any

import numpy as np
import torch

from .binding import ERROR_NAMES
from .vec_env import VecSparkSchedSimEnv

try:  # pragma: no cover
    import gymnasium as _gym

    _Base = _gym.Env
except Exception:  # gymnasium is not installed in the target image
    _gym = None
    _Base = object


class JobView:
    """what reference metrics / trainers read from `env.jobs[j]` (metrics.py:6-9)"""

    def __init__(self, t_arrival: float, t_completed: float):
        self.t_arrival = t_arrival
        self.t_completed = t_completed


class SparkSchedSimEnv(_Base):  # type: ignore[misc]
    metadata = {"render_modes": [], "render_fps": 30}

    def __init__(self, env_cfg: dict[str, Any], device: str = "cuda:0", _lib=None) -> None:
        if env_cfg.get("render_mode"):
            raise ValueError("rendering is not available in the GPU build")  # pygame renderer is out of scope
        self.num_executors: int = env_cfg["num_executors"]
        self.moving_delay = env_cfg["moving_delay"]
        self.beta: float = env_cfg.get("beta", 0)
        self.job_arrival_cap = env_cfg.get("job_arrival_cap")
        self.render_mode = None
        from .spaces import make_action_space, make_observation_space
        self.action_space = make_action_space(self.num_executors)
        self.observation_space = make_observation_space(self.num_executors)
        self._vec = VecSparkSchedSimEnv(env_cfg, 1, device=device, _lib=_lib)
        self._act_s = torch.zeros(1, dtype=torch.int32, device=self._vec.device)
        self._act_n = torch.ones(1, dtype=torch.int32, device=self._vec.device)

    # ---- Gymnasium API ------------------------------------------------------------------

    def reset(self, seed: int | None = None, options: dict[str, Any] | None = None):
        time_limit = (options or {}).get("time_limit", np.inf)
        if time_limit is np.inf and not self.job_arrival_cap:
            raise ValueError("must either have a limit on job arrivals or time.")
        self._vec.reset(seed=None if seed is None else [seed], options=options)
        self._raise()
        self.job_arrival_cap = self._vec.header(0)["J"]  # reference overwrites it (spark_sched_sim.py:156)
        self.observation_space["source_job_idx"].n = self.job_arrival_cap + 1  # :157
        return self._observe(), self.info

    def step(self, action: dict):
        if not isinstance(action, dict) or set(action.keys()) != {"stage_idx", "num_exec"}:
            raise ValueError("invalid action: does not belong to the action space")
        for v in action.values():
            if not isinstance(v, (int, np.integer)):
                raise ValueError("invalid action: does not belong to the action space")
        self._act_s[0] = int(action["stage_idx"])
        self._act_n[0] = int(action["num_exec"])
        self._vec.step_async(self._act_s, self._act_n)
        self._raise()
        o = self._vec.obs_i32[0].cpu().numpy()
        reward = self._vec.obs_f64[0, 0].item()
        return self._observe(), reward, bool(o[6]), False, self.info

    def _observe(self) -> dict:
        obs = self._vec.obs_view(0)
        self.action_space["stage_idx"].n = len(obs["dag_batch"].nodes) + 1  # spark_sched_sim.py:403-404
        self.observation_space["dag_ptr"].feature_space.n = len(obs["dag_batch"].nodes) + 1
        return obs

    def close(self) -> None:
        self._vec.close()

    def _raise(self) -> None:
        code = int(self._vec.obs_i32[0, 7].item())
        if code == 0:
            return
        msg = ERROR_NAMES.get(code, str(code))
        if code in (5, 7):
            raise AssertionError(msg)
        if code == 2:
            raise KeyError(msg)
        raise ValueError(msg)

    # ---- attributes the reference's callers read ----------------------------------------------

    @property
    def unwrapped(self):
        return self

    @property
    def wall_time(self) -> float:
        return self._vec.header(0)["wall_time"]

    @property
    def info(self) -> dict:
        return {"wall_time": self.wall_time}

    @property
    def jobs(self) -> dict[int, JobView]:
        ta, tc, _, _ = self._vec.job_times(0)
        return {j: JobView(float(ta[j]), float(tc[j])) for j in range(len(ta))}

    @property
    def active_job_ids(self) -> list[int]:
        d = self._vec.dims
        n = self._vec.header(0)["n_active"]
        row = self._vec._env_view[0, d.off_active: d.off_active + 2 * n].cpu().numpy()
        return row.view(np.uint16).astype(int).tolist()

    @property
    def completed_job_ids(self) -> set[int]:
        _, _, order, _ = self._vec.job_times(0)
        done = [(int(o), j) for j, o in enumerate(order) if o >= 0]
        s: set[int] = set()
        for _, j in sorted(done):  # same insertion order as the reference's set (spark_sched_sim.py:695)
            s.add(j)
        return s

    @property
    def num_completed_jobs(self) -> int:
        return self._vec.header(0)["n_completed"]

    @property
    def num_active_jobs(self) -> int:
        return self._vec.header(0)["n_active"]

    @property
    def all_jobs_complete(self) -> bool:
        h = self._vec.header(0)
        return h["n_completed"] == h["J"]

    @property
    def job_duration_buff(self) -> list[float]:
        return self._vec.job_duration_buff(0)

    @property
    def avg_job_duration(self) -> float:
        return np.mean(self.job_duration_buff).item() * 1e-3  # spark_sched_sim.py:243-245


if _gym is not None:  # pragma: no cover
    try:
        from gymnasium.envs.registration import register

        register(id="SparkSchedSimEnv-v0", entry_point="spark_sched_sim_amd.env:SparkSchedSimEnv")
    except Exception:
        pass
