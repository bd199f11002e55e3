"""`StochasticTimeLimit` (reference wrappers/stochastic_time_limit.py) over the facade and the
vector env, under the CPU wave emulator: limit sampling rule, `options["time_limit"]` reaching the
arrival sampler, truncation at `wall_time >= limit`."""
import numpy as np
import torch

from emu_util import load_emu
from golden_util import Golden, bits
from spark_sched_sim_amd import RoundRobinScheduler, SparkSchedSimEnv, VecSparkSchedSimEnv
from spark_sched_sim_amd.wrappers import StochasticTimeLimit, VecStochasticTimeLimit


def test_single_env_time_limit_matches_recorded_truncated_episode():
    g = Golden("tiny_fair_tlimit")  # recorded with options={"time_limit": 60000.0}
    cfg = dict(g.cfg, max_jobs=64)
    mean = 1.0e5
    env = StochasticTimeLimit(SparkSchedSimEnv(cfg, device="cpu", _lib=load_emu()), mean, seed=42)
    obs, _ = env.reset(seed=7)
    assert env.time_limit == np.random.RandomState(7).exponential(mean)  # reference rule (:15-17)
    env.reset(seed=0)  # seed 0 is falsy: the stream continues instead of re-seeding (:15)
    rs = np.random.RandomState(7)
    rs.exponential(mean)
    assert env.time_limit == rs.exponential(mean)
    # the wrapped episode == the plain facade reset with options={"time_limit": L}, truncated at wall >= L
    obs, _ = env.reset(seed=3)
    L = np.random.RandomState(3).exponential(mean)
    assert env.time_limit == L
    ref = SparkSchedSimEnv(cfg, device="cpu", _lib=load_emu())
    robs, _ = ref.reset(seed=3, options={"time_limit": L})
    assert env.unwrapped.job_arrival_cap == ref.job_arrival_cap  # arrivals stop at the limit (tpch.py:63)
    sched = RoundRobinScheduler(cfg["num_executors"])
    trunc = term = False
    n = 0
    while not (term or trunc):
        action, _ = sched.schedule(obs)
        obs, r, term, trunc, info = env.step(action)
        robs, rr, rterm, _, rinfo = ref.step(action)
        n += 1
        assert bits(r) == bits(rr) and bits(info["wall_time"]) == bits(rinfo["wall_time"]) and term == rterm
        assert trunc == (info["wall_time"] >= L)
    assert n > 3
    env.close()
    ref.close()


def test_vector_time_limit():
    g = Golden("tiny_fair_tlimit")
    cfg = dict(g.cfg, max_jobs=64)
    env = VecStochasticTimeLimit(VecSparkSchedSimEnv(cfg, 3, device="cpu", _lib=load_emu()), 5.0e4, seed=42)
    env.reset(seed=[5, 6, 7])
    exp = [np.random.RandomState(s).exponential(5.0e4) for s in (5, 6, 7)]
    assert env.time_limit.tolist() == exp
    done = torch.zeros(3, dtype=torch.bool)
    for _ in range(60):
        obs, rew, term, trunc, info = env.step(env.policy_actions("fair"))
        done |= trunc | term
        assert torch.equal(trunc, info["wall_time"] >= env.time_limit)
        if bool(done.all()):
            break
    assert bool(done.all())
    env.close()
