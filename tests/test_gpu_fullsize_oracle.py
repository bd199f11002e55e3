"""-m gpu: BASELINE-size batches (4096 envs) run to the end of their episodes on the GPU with the
on-device policies, every env's episode summary compared bit-for-bit with the C oracle playing
the same policy on the same seed: number of steps, episode return (same left-to-right f64
accumulation), final wall time, number of jobs. C2 sizing with the counter-based random policy,
C3 sizing (50 executors, 200 jobs) with the fair policy."""
import ctypes as C
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

from golden_util import bits
from oracle_binding import OracleEnv, SsoObsInfo

pytestmark = pytest.mark.gpu

C2 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
C3 = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)


def oracle_episodes(pack, cfg, policy_id, seeds, threads=16, time_limit=float("inf")):
    def work(chunk):
        env = OracleEnv(pack, cfg)
        out = []
        for s in chunk:
            r = C.c_double()
            n = env.lib.sso_run_episode_tl(env.h, int(s), time_limit, policy_id, 10**9, C.byref(r))  # releases the GIL
            info = SsoObsInfo()
            env.lib.sso_obs_sizes(env.h, C.byref(info))
            out.append((int(n), r.value, info.wall_time, info.num_jobs))
        env.close()
        return out

    chunks = [seeds[i::threads] for i in range(threads)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(work, chunks))
    res = {}
    for chunk, part in zip(chunks, parts):
        res.update(dict(zip(chunk, part)))
    return [res[s] for s in seeds]


E64 = dict(num_executors=64, job_arrival_cap=100, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0)  # the lane limit
BURST = dict(num_executors=20, job_arrival_cap=120, job_arrival_rate=4.0e-4, moving_delay=500.0, warmup_delay=100.0)  # ~100 jobs active at once


@pytest.mark.parametrize("cfg,policy,policy_id,B,max_steps", [(C2, "hash", 1, 4096, 1600), (C3, "fair", 0, 4096, 9000),
                                                              (E64, "fair", 0, 2048, 9000), (BURST, "fair", 0, 1024, 9000)])
def test_full_batch_episode_summaries_match_oracle(cfg, policy, policy_id, B, max_steps, pack):
    from spark_sched_sim_amd import VecSparkSchedSimEnv

    env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=pack)  # no auto-reset: finished envs stay finished
    base = 31000
    env.reset(seed=base)
    done = 0
    while done < max_steps:
        env.rollout(policy, 200)
        done += 200
        if bool((env.header_field("terminated") != 0).all()):
            break
    torch.cuda.synchronize()
    assert bool((env.header_field("terminated") != 0).all()), "episodes did not finish"
    assert int((env.obs_i32[:, 7] != 0).sum()) == 0
    steps = env.header_field("last_ep_steps").cpu().numpy()
    ret = env.header_field("last_ep_return").cpu().numpy()
    wall = env.header_field("last_ep_wall").cpu().numpy()
    J = env.header_field("J").cpu().numpy()
    seeds = [base + i for i in range(B)]
    exp = oracle_episodes(pack, cfg, policy_id, seeds)
    bad = [i for i in range(B)
           if (int(steps[i]), bits(ret[i]), bits(wall[i]), int(J[i])) != (exp[i][0], bits(exp[i][1]), bits(exp[i][2]), exp[i][3])]
    assert not bad, f"{len(bad)} of {B} envs differ from the oracle, first: env {bad[0]} got {(int(steps[bad[0]]), ret[bad[0]], wall[bad[0]])} expected {exp[bad[0]]}"
    env.close()
