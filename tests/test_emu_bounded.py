"""sss_step_bounded on the wave emulator: steps cut at an event budget continue in the next launch and leave what sss_step leaves"""
import pytest

from bounded_util import check_bounded_steps
from emu_util import load_emu
from spark_sched_sim_amd import workload

C2 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
E50 = dict(num_executors=50, job_arrival_cap=40, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
E100 = dict(num_executors=100, job_arrival_cap=30, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)


@pytest.fixture(scope="module")
def pack():
    return workload.default_pack()


@pytest.mark.parametrize("cfg,policy,seeds,n_steps,budgets", [(C2, "fair", [1, 2, 3], 120, (1, 7, 64)), (E50, "fair", [4, 5], 60, (3, 16)),
                                                               (E50, "hash", [6], 80, (5,)), (E100, "fair", [7], 40, (4, 40)),
                                                               (E50, "fair", [8, 9], 60, ("mixed",))])
def test_bounded_steps_leave_what_steps_leave(cfg, policy, seeds, n_steps, budgets, pack):
    launches = check_bounded_steps("cpu", load_emu(), cfg, seeds, policy, n_steps, budgets, pack=pack)
    small = min(budgets, key=lambda b: 0 if b == "mixed" else b)
    assert launches[small][1] > 0, "no step was cut at the smallest budget: the test did not exercise the continuation"
    assert launches[small][0] > n_steps


def test_bounded_step_rejects_bad_arguments(pack):
    import torch

    from spark_sched_sim_amd import VecSparkSchedSimEnv

    env = VecSparkSchedSimEnv(C2, 2, device="cpu", pack=pack, _lib=load_emu())
    env.reset(seed=[1, 2])
    a = env.policy_actions("fair")
    with pytest.raises(ValueError):
        env.step_bounded_async(a["stage_idx"], a["num_exec"], 0)
    env.close()


@pytest.mark.parametrize("name,seeds,budget,max_steps", [("c1_fair", [1234, 3], 3, 200), ("c1_hash", [100], "mixed", 250), ("e100_fair", [0], 5, 120)])
def test_bounded_entry_point_against_the_reference_recordings(name, seeds, budget, max_steps, pack):
    """the golden trajectories recorded from the reference, every step taken through `sss_step_bounded` alone (no `sss_step` in
    between): rewards, wall times, flags and observation digests of every completed step, bit for bit"""
    from replay_util import replay_golden

    bad = replay_golden(name, seeds, pack, device="cpu", lib=load_emu(), full_obs_steps=10, max_steps=max_steps, bounded=budget)
    assert not bad, "\n".join(bad[:10])
