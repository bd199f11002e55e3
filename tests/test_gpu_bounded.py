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


@pytest.mark.gpu
@pytest.mark.parametrize("name,seeds,budget", [("c1_fair", [1234, 0, 1, 2], 4), ("c3_fair", [0, 1], 16), ("e100_fair", [0, 1], 9), ("c1_hash", [100, 101], "mixed")])
def test_bounded_entry_point_against_the_reference_recordings_gpu(name, seeds, budget):
    """the reference's recorded trajectories (whole episodes) with every step taken through `sss_step_bounded` alone"""
    from replay_util import replay_golden

    bad = replay_golden(name, seeds, workload.default_pack(), device="cuda:0", full_obs_steps=40, bounded=budget)
    assert not bad, "\n".join(bad[:10])


@pytest.mark.gpu
def test_bounded_entry_point_full_batch_against_the_oracle():
    """BASELINE config 3 at full size through the bounded entry point only: 4096 envs, the on-device fair policy, a different event
    budget every launch (envs drift apart: each launch some envs complete a step, others continue theirs) - every env's episode
    summary (steps, return, final wall time, jobs) equals the C oracle's, bit for bit"""
    import numpy as np
    import torch

    from golden_util import bits
    from replay_util import SKIP
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from test_gpu_fullsize_oracle import C3, oracle_episodes

    pack = workload.default_pack()
    B, base = 4096, 52000
    env = VecSparkSchedSimEnv(C3, B, device="cuda:0", pack=pack)  # no auto-reset: finished envs stay finished
    env.reset(seed=base)
    n = 0
    while True:
        a = env.policy_actions("fair")  # (for envs in the middle of a step the action is ignored)
        over = env.obs_i32[:, 6] != 0    # an env whose episode has ended sits the launches out
        env.step_bounded_async(torch.where(over, torch.full_like(a["stage_idx"], SKIP), a["stage_idx"]).contiguous(), a["num_exec"], (n * 7) % 23 + 8)
        n += 1
        if n % 500 == 0 and bool((env.header_field("terminated") != 0).all()):
            break
        assert n < 60000, "episodes did not finish"
    torch.cuda.synchronize()
    assert int((env.obs_i32[:, 7] != 0).sum()) == 0
    steps, ret = env.header_field("last_ep_steps").cpu().numpy(), env.header_field("last_ep_return").cpu().numpy()
    wall, J = env.header_field("last_ep_wall").cpu().numpy(), env.header_field("J").cpu().numpy()
    exp = oracle_episodes(pack, C3, 0, [base + i for i in range(B)])
    bad = [i for i in range(B) if (int(steps[i]), bits(ret[i]), bits(wall[i]), int(J[i])) != (exp[i][0], bits(exp[i][1]), bits(exp[i][2]), exp[i][3])]
    assert not bad, f"{len(bad)} of {B} envs differ from the oracle, first: env {bad[0]}"
    env.close()
