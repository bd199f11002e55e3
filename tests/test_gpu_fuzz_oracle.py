"""-m gpu: differential sweep over env configurations against the C oracle - executor counts from 1
to the 64-lane limit, job capacities across the LDS job-set size classes (<= 76, <= 306, above),
zero delays, arrival bursts, episodes bounded by a time limit instead of a job cap, discounted
rewards. 64 envs per configuration run whole episodes with the on-device fair / random policies;
steps, return, final wall time and job count of every env must equal the oracle's, bit for bit
(beta > 0: return within 1e-12 relative)."""
import numpy as np
import pytest
import torch

from golden_util import bits
from test_gpu_fullsize_oracle import oracle_episodes

pytestmark = pytest.mark.gpu


def _cfg(E, J, rate, md=2000.0, wd=1000.0, **kw):
    return dict(num_executors=E, job_arrival_cap=J, job_arrival_rate=rate, moving_delay=md, warmup_delay=wd, **kw)


CASES = [
    ("one_executor", _cfg(1, 6, 1.0e-4), "fair", None),
    ("two_executors_hash", _cfg(2, 12, 1.0e-4), "hash", None),
    ("three_executors", _cfg(3, 20, 8.0e-5), "fair", None),
    ("seventeen_executors", _cfg(17, 40, 1.0e-4), "hash", None),   # just past one DPP row
    ("sixty_three_executors", _cfg(63, 80, 1.0e-4), "fair", None),
    ("sixty_four_hash", _cfg(64, 60, 2.0e-4), "hash", None),
    ("jobset_512", _cfg(20, 300, 4.0e-4), "fair", None),           # job-set image class 2
    ("jobset_2048", _cfg(30, 400, 1.0e-3, md=100.0), "fair", None),  # class 3, hundreds of jobs active
    ("zero_delays", _cfg(10, 40, 1.0e-4, md=0.0, wd=0.0), "hash", None),
    ("all_at_once", _cfg(12, 60, 1.0e-1), "fair", None),           # every job arrives within a few ms
    ("time_limit_only", dict(num_executors=10, job_arrival_cap=None, max_jobs=400, job_arrival_rate=4.0e-5, moving_delay=2000.0,
                             warmup_delay=1000.0), "fair", 3.0e6),
    ("discounted", _cfg(10, 30, 4.0e-5, beta=5.0e-3), "fair", None),
    ("thousand_jobs", _cfg(40, 1000, 2.0e-3, md=200.0), "fair", None),   # the build's job-capacity limit
    ("hundred_executors", _cfg(100, 120, 1.0e-4), "fair", None),      # > 64 executors: the wide instantiation of the kernels
    ("e128_hash", _cfg(128, 60, 2.0e-4), "hash", None),
    ("e101_few_jobs", _cfg(101, 3, 2.0e-5), "hash", None),            # num_local_executors in 101 .. exec_cap (tpch.py:258-260)
    ("e65_fair", _cfg(65, 50, 1.0e-4), "fair", None),
    ("tiny_time_limit", dict(num_executors=5, job_arrival_cap=None, max_jobs=50, job_arrival_rate=1.0e-4, moving_delay=2000.0,
                             warmup_delay=1000.0), "fair", 5.0e3),        # mostly single-job episodes
]


@pytest.mark.parametrize("name,cfg,policy,time_limit", CASES, ids=[c[0] for c in CASES])
def test_config_sweep_matches_oracle(name, cfg, policy, time_limit, pack):
    from spark_sched_sim_amd import VecSparkSchedSimEnv

    B, base = 64, 7000
    env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=pack)
    env.reset(seed=base, options={"time_limit": time_limit} if time_limit else None)
    for _ in range(400):
        env.rollout(policy, 200)
        done = (env.header_field("terminated") != 0) | (env.obs_i32[:, 7] != 0)
        if bool(done.all()):
            break
    torch.cuda.synchronize()
    err = env.obs_i32[:, 7].cpu().numpy()
    term = env.header_field("terminated").cpu().numpy()
    steps = env.header_field("ep_steps").cpu().numpy()
    ret = env.header_field("ep_return").cpu().numpy()
    wall = env.header_field("wall_time").cpu().numpy()
    J = env.header_field("J").cpu().numpy()
    assert ((term != 0) | (err == 5)).all(), (name, np.unique(err))   # 5 = the reference's own [step] stall (random policy)
    exp = oracle_episodes(pack, cfg, 0 if policy == "fair" else 1, [base + i for i in range(B)], threads=8,
                          time_limit=time_limit if time_limit else float("inf"))
    for i in range(B):
        if err[i] == 5:   # the oracle must stop at the same stall (it reports -(code) - 100)
            assert exp[i][0] == -105, (name, i, exp[i])
            continue
        want = exp[i]
        assert int(steps[i]) == want[0] and bits(wall[i]) == bits(want[2]) and int(J[i]) == want[3], (name, i, steps[i], wall[i], want)
        if cfg.get("beta"):
            assert abs(ret[i] - want[1]) <= 1e-12 * abs(want[1]), (name, i)
        else:
            assert bits(ret[i]) == bits(want[1]), (name, i, ret[i], want[1])
    env.close()


def test_more_arrivals_than_max_jobs_is_reported(pack):
    """job_arrival_cap=None with a time limit: the arena holds `max_jobs` jobs; an episode whose
    Poisson sequence is longer fails loudly with code 10 instead of writing past the arena"""
    from spark_sched_sim_amd import VecSparkSchedSimEnv

    cfg = dict(num_executors=5, job_arrival_cap=None, max_jobs=8, job_arrival_rate=1.0e-3, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 32, device="cuda:0", pack=pack)
    env.reset(seed=1, options={"time_limit": 1.0e5})   # ~100 arrivals expected, 8 fit
    err = env.obs_i32[:, 7].cpu().numpy()
    assert (err == 10).all(), np.unique(err)
    with pytest.raises(ValueError, match="max_jobs"):
        env.raise_on_error()
    env.close()
