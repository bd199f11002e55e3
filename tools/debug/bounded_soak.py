#!/usr/bin/env python3
"""debug: a longer sss_step_bounded run against sss_step on the GPU (tests/bounded_util.check_bounded_steps at full config-3 sizing)"""
import os.path as osp
import sys

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, osp.join(ROOT, "tests"))
from bounded_util import check_bounded_steps  # noqa: E402
from spark_sched_sim_amd import workload  # noqa: E402

C3 = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
C2 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
pack = workload.default_pack()
for name, cfg, n_envs, n_steps, budgets, policy in (("c3", C3, 128, 1500, (16, 3), "fair"), ("c2", C2, 256, 450, (24, 5), "fair"), ("c2 hash", C2, 128, 450, (11,), "hash")):
    print(name, check_bounded_steps("cuda:0", None, cfg, list(range(7, 7 + n_envs)), policy, n_steps, budgets, pack=pack), flush=True)
