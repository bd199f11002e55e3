"""More than 64 executors: the WIDE instantiation of the kernel source (csrc/sss_sim.h compiled with SSS_WIDE - two executors per
lane in the queue's pop and the staging loops, 128-entry executor arrays; in the lane-parallel event machinery a lane speaks for the
one of its two executors whose event comes first, csrc/sss_sim.h lane_event;
tests/emu/emu_wide.cpp here, csrc/sss_hip_wide.hip on gfx950) under the CPU wave emulator, through the real C ABI, against the C
oracle step by step. The reference takes any `num_executors` (spark_sched_sim.py:37) and its level table reaches 100
(tpch.py:237-262); beyond 100 the interval table has the quirks of tpch.py:258-260 (rows 101 .. cap-1 are (100, 100), row cap
stays (0, 0) -> the stage's largest level). Fixtures recorded from the reference itself at 100 and 120 executors:
tests/test_emu_golden.py (e100_*, e120_hash)."""
import numpy as np
import pytest

from emu_util import load_emu
from golden_util import bits
from oracle_binding import OracleEnv
from spark_sched_sim_amd import VecSparkSchedSimEnv


def _cfg(E, J, rate=2.0e-4):
    return dict(num_executors=E, job_arrival_cap=J, job_arrival_rate=rate, moving_delay=2000.0, warmup_delay=1000.0)


@pytest.mark.parametrize("cfg,policy,seed", [
    (_cfg(65, 30), "fair", 2),        # the first count past one lane per executor
    (_cfg(100, 40), "hash", 11),      # the level table's top
    (_cfg(128, 40), "fair", 3),       # the build's limit: executor pools reach 1024-slot set images
    (_cfg(127, 30), "hash", 6),
    (_cfg(110, 25), "fifo", 7),
    (_cfg(128, 2, 2.0e-5), "fair", 3),    # all executors on one job: num_local_executors up to exec_cap
    (_cfg(105, 1), "hash", 2),
    (_cfg(120, 3, 2.0e-5), "fifo", 1),
])
def test_wide_instantiation_matches_oracle_step_by_step(cfg, policy, seed, pack):
    env = VecSparkSchedSimEnv(cfg, 1, device="cpu", pack=pack, _lib=load_emu())
    env.reset(seed=[seed])
    o = OracleEnv(pack, cfg)
    o.reset(seed)
    for t in range(4000):
        a = env.policy_actions(policy)
        s, n = int(a["stage_idx"][0]), int(a["num_exec"][0])
        _, rew, term, _, info = env.step(a)
        e, r, done = o.step(s, n)
        assert int(info["err"][0]) == e, (t, int(info["err"][0]), e)
        if e:   # (the counter-based random policy can run into the reference's own "[step]" stall: both stop there)
            assert e == 5
            break
        assert bits(float(rew[0])) == bits(r) and bits(float(info["wall_time"][0])) == bits(o.info().wall_time), t
        ov = env.obs_view(0)
        _, nodes, el, ptr, sup = o.obs()
        assert np.array_equal(ov["dag_batch"].nodes, nodes) and np.array_equal(np.asarray(ov["exec_supplies"], dtype=np.int32), sup), t
        assert ov["num_committable_execs"] == o.info().num_committable_execs and ov["source_job_idx"] == o.info().source_job_idx, t
        if bool(term[0]):
            assert done
            break
    else:
        raise AssertionError("episode did not finish")
    env.close()
    o.close()


def test_wide_fused_rollout_and_batch_of_envs_match_oracle_episodes(pack):
    """sss_rollout (policy -> step -> observe x n in one launch) on the wide instantiation, several envs: episode summaries equal
    to the oracle's; stepping the same seeds one call at a time lands in the same states"""
    import ctypes as C

    from oracle_binding import SsoObsInfo

    cfg = _cfg(100, 20, 1.0e-4)
    B, base = 3, 400
    envs = []
    for fused in (True, False):
        env = VecSparkSchedSimEnv(cfg, B, device="cpu", pack=pack, _lib=load_emu())
        env.reset(seed=base)
        for _ in range(40):
            if fused:
                env.rollout("fair", 100)
            else:
                for _ in range(100):
                    a = env.policy_actions("fair")
                    a["stage_idx"][env.header_field("terminated") != 0] = -(2 ** 31)   # SSS_SKIP_ENV: finished envs are not touched
                    env.step(a)
            if bool((env.header_field("terminated") != 0).all()):
                break
        assert bool((env.header_field("terminated") != 0).all()) and int((env.obs_i32[:, 7] != 0).sum()) == 0
        envs.append(env)
    for f in ("last_ep_steps", "last_ep_return", "last_ep_wall", "J"):
        assert np.array_equal(envs[0].header_field(f).numpy(), envs[1].header_field(f).numpy()), f
    o = OracleEnv(pack, cfg)
    for i in range(B):
        r = C.c_double()
        n = o.lib.sso_run_episode_tl(o.h, base + i, float("inf"), 0, 10**9, C.byref(r))
        info = SsoObsInfo()
        o.lib.sso_obs_sizes(o.h, C.byref(info))
        got = (int(envs[0].header_field("last_ep_steps")[i]), bits(float(envs[0].header_field("last_ep_return")[i])),
               bits(float(envs[0].header_field("last_ep_wall")[i])), int(envs[0].header_field("J")[i]))
        assert got == (int(n), bits(r.value), bits(info.wall_time), info.num_jobs), i
    o.close()
    for env in envs:
        env.close()


def test_executor_count_limits_of_the_c_abi(pack):
    """1..128 executors are accepted (the host picks the instantiation), 129 is refused with a message"""
    for E in (64, 65, 128):
        env = VecSparkSchedSimEnv(_cfg(E, 4), 1, device="cpu", pack=pack, _lib=load_emu())
        env.reset(seed=[1])
        assert int(env.obs_i32[0, 7]) == 0 and int(env.obs_i32[0, 4]) == E   # every executor committable at the start
        env.close()
    with pytest.raises(ValueError, match=r"num_executors must be in \[1, 128\]"):
        VecSparkSchedSimEnv(_cfg(129, 4), 1, device="cpu", pack=pack, _lib=load_emu())
