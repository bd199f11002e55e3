"""sss_step_bounded on the GPU: steps cut at an event budget continue in the next launch and leave what sss_step leaves"""
import pytest

from bounded_util import check_bounded_steps
from spark_sched_sim_amd import workload

C2 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
C3S = dict(num_executors=50, job_arrival_cap=60, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
E100 = dict(num_executors=100, job_arrival_cap=30, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,policy,n_envs,n_steps,budgets", [(C2, "fair", 96, 150, (2, 24)), (C3S, "fair", 64, 120, (5, 16)), (C2, "hash", 32, 150, (7,)),
                                                               (E100, "fair", 8, 60, (6,)), (C3S, "fair", 48, 150, ("mixed",))])
def test_bounded_steps_leave_what_steps_leave_gpu(cfg, policy, n_envs, n_steps, budgets):
    launches = check_bounded_steps("cuda:0", None, cfg, list(range(100, 100 + n_envs)), policy, n_steps, budgets, pack=workload.default_pack())
    small = min(budgets, key=lambda b: 0 if b == "mixed" else b)
    assert launches[small][1] > 0 and launches[small][0] > n_steps
