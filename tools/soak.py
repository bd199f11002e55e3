#!/usr/bin/env python3
"""GPU soak: long auto-reset runs of the fused rollout kernel at full batch size; reports the error
codes left in the envs (only 5 = the reference's own "[step]" stall is expected, and only with the
random policy), episodes finished and steps taken."""
import collections
import json
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from spark_sched_sim_amd import VecSparkSchedSimEnv  # noqa: E402

CASES = [
    ("c2_hash", dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash", 30, 20000, None),
    ("c3_fair", dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 12000, None),
    ("e64_fifo", dict(num_executors=64, job_arrival_cap=100, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0), "fifo", 0, 6000, None),
    ("timelimit_fair", dict(num_executors=10, job_arrival_cap=None, max_jobs=300, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 8000, 4.0e6),
    ("burst_hash0", dict(num_executors=20, job_arrival_cap=120, job_arrival_rate=4.0e-4, moving_delay=500.0, warmup_delay=100.0), "hash", 0, 8000, None),
]


def main():
    for name, cfg, policy, param, steps, tl in CASES:
        env = VecSparkSchedSimEnv(cfg, 4096, device="cuda:0", auto_reset=True)
        env.reset(seed=12345, options={"time_limit": tl} if tl else None)
        t0 = time.perf_counter()
        done = 0
        while done < steps:
            env.rollout(policy, 500, param=param)
            done += 500
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        err = collections.Counter(env.obs_i32[:, 7].cpu().tolist())
        c = env.counters()
        print(json.dumps({"case": name, "steps_per_env": steps, "seconds": round(dt, 2), "env_steps": c["n_steps"], "events": c["n_events"],
                          "episodes": int(env.header_field("episodes").sum()), "err_codes": {str(k): v for k, v in sorted(err.items())},
                          "max_active_jobs_now": int(env.header_field("n_active").max())}), flush=True)
        env.close()


if __name__ == "__main__":
    main()
