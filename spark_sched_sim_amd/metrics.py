"""episode metrics with the reference's definitions (reference spark_sched_sim/metrics.py:4-23),
over a `SparkSchedSimEnv` facade or env `i` of a `VecSparkSchedSimEnv`."""
from __future__ import annotations

import numpy as np


def job_durations(env, i: int | None = None):
    if i is None:
        e = env.unwrapped
        wall, jobs = e.wall_time, e.jobs
        ids = e.active_job_ids + list(e.completed_job_ids)
        return [min(jobs[j].t_completed, wall) - jobs[j].t_arrival for j in ids]
    ta, tc, _, _ = env.job_times(i)
    h = env.header(i)
    n = h["next_arrival"]
    return list(np.minimum(tc[:n], h["wall_time"]) - ta[:n])


def avg_job_duration(env, i: int | None = None):
    return np.mean(job_durations(env, i))


def avg_num_jobs(env, i: int | None = None):
    wall = env.unwrapped.wall_time if i is None else env.header(i)["wall_time"]
    return sum(job_durations(env, i)) / wall


def job_duration_percentiles(env, i: int | None = None):
    return np.percentile(job_durations(env, i), [25, 50, 75, 100])
