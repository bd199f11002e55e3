"""The fast run and the lane-parallel event batches of the event loop (csrc/sss_sim.h fast_run, batch_*_events) and the
pre-generated PCG64 stream they draw from (rng_refill), under the CPU wave emulator.

* whole episodes in long launches (the job cache is only rebuilt at launch boundaries, so slots are
  handed from job to job inside a launch) against the C oracle, env by env, bit for bit;
* the same kernel source compiled with every event going one at a time (-DSSS_NO_BATCH: no batches of any
  kind - task completions with tasks left, released executors, arriving executors - and no lane-parallel
  fulfilment chunks) must leave byte-identical env state after every launch;
* the batch path is really taken (device counters);
* the jump-ahead table against numpy's own PCG64.advance().
"""
import ctypes as C

import numpy as np
import pytest

from emu_util import load_emu
from golden_util import bits
from oracle_binding import OracleEnv, SsoObsInfo
from spark_sched_sim_amd import VecSparkSchedSimEnv

C2 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
E64 = dict(num_executors=64, job_arrival_cap=30, job_arrival_rate=8.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
E20 = dict(num_executors=20, job_arrival_cap=40, job_arrival_rate=2.0e-4, moving_delay=500.0, warmup_delay=100.0)
# BASELINE config 3's executor count (fewer jobs): whole fulfilments of 20-50 executors, pools whose tables outgrow
# their records, jobs whose last stage drains with nothing left to schedule
E50 = dict(num_executors=50, job_arrival_cap=60, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
# the wide instantiation (two executors per lane: a lane speaks for its earlier event, csrc/sss_sim.h lane_event): 65 (one lane
# with two executors), 100 (the reference's largest executor level), 128 (every lane with two; pool tables of 1024 slots)
E65 = dict(num_executors=65, job_arrival_cap=40, job_arrival_rate=8.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
E100 = dict(num_executors=100, job_arrival_cap=60, job_arrival_rate=8.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
E128 = dict(num_executors=128, job_arrival_cap=50, job_arrival_rate=1.5e-4, moving_delay=500.0, warmup_delay=100.0)


def oracle_summary(pack, cfg, policy_id, seed):
    env = OracleEnv(pack, cfg)
    r = C.c_double()
    n = env.lib.sso_run_episode(env.h, int(seed), policy_id, 10**9, C.byref(r))
    info = SsoObsInfo()
    env.lib.sso_obs_sizes(env.h, C.byref(info))
    out = (int(n), bits(r.value), bits(info.wall_time), int(env.lib.sso_event_count(env.h)))
    env.close()
    return out


# 31133 / 31158: seeds on which a completed job's cache slot was once handed on while an executor
# was still travelling to that job (found by the 4096-env GPU test, reproduced here)
@pytest.mark.parametrize("cfg,policy,policy_id,seeds,chunk", [
    (C2, "hash", 1, [31133, 31158, 7], 200),
    (C2, "fair", 0, [3, 4], 500),
    (E64, "hash", 1, [0, 1], 150),
    (E20, "fair", 0, [11], 300),
    (E50, "fair", 0, [0, 7], 250),
    (E50, "hash", 1, [2], 333),
    (E65, "fair", 0, [1], 250),
    (E100, "fair", 0, [0, 5], 250),
    (E100, "hash", 1, [4], 333),
    (E128, "fair", 0, [9], 200),
    (E128, "hash", 1, [3], 150),
])
def test_long_launches_match_oracle(cfg, policy, policy_id, seeds, chunk, pack):
    env = VecSparkSchedSimEnv(cfg, len(seeds), device="cpu", pack=pack, _lib=load_emu())
    env.reset(seed=seeds)
    for _ in range(40):
        env.rollout(policy, chunk)
        if all(env.header(k)["terminated"] or env.header(k)["err"] for k in range(len(seeds))):
            break
    for k, s in enumerate(seeds):
        h = env.header(k)
        assert h["err"] == 0 and h["terminated"] == 1, (s, h["err"], h["ep_steps"])
        exp = oracle_summary(pack, cfg, policy_id, s)
        assert (h["last_ep_steps"], bits(h["last_ep_return"]), bits(h["last_ep_wall"]), h["n_events"]) == exp, (s, h, exp)
    c = env.counters()
    assert c["n_batched_events"] > 0.5 * c["n_fast_events"] > 0  # the batch path did the bulk of the work
    assert c["n_batched_events"] >= 2 * c["n_rounds"] > 0 or cfg["num_executors"] >= 64
    env.close()


@pytest.mark.parametrize("cfg,policy,seeds,chunk,n_launches", [(C2, "hash", [5, 31133], 37, 12), (E64, "fair", [2], 61, 8), (E50, "fair", [0, 1], 41, 12),
                                                                (E50, "hash", [3], 29, 10), (E20, "hash", [1, 2], 23, 20),
                                                                (E65, "fair", [1, 6], 41, 10), (E100, "fair", [0, 2], 41, 12), (E100, "hash", [3], 29, 10),
                                                                (E128, "fair", [5], 37, 10), (E128, "hash", [1, 8], 23, 12)])
def test_batches_equal_one_event_at_a_time(cfg, policy, seeds, chunk, n_launches, pack):
    """the batch path is an optimisation of the one-at-a-time path, nothing else: same env bytes"""
    envs = [VecSparkSchedSimEnv(cfg, len(seeds), device="cpu", pack=pack, _lib=load_emu(v)) for v in ("", "_nobatch")]
    for e in envs:
        e.reset(seed=seeds)
    skip = slice(256, 272)  # SssHdr::n_batched, n_rounds: the only fields that may differ
    for it in range(n_launches):
        for e in envs:
            e.rollout(policy, chunk)
        a, b = (e._env_view.numpy().copy() for e in envs)
        a[:, skip] = 0
        b[:, skip] = 0
        assert np.array_equal(a, b), f"launch {it}: env state differs"
        for name in ("nodes", "edge_links", "dag_ptr", "exec_supplies", "obs_i32", "obs_f64"):
            assert np.array_equal(getattr(envs[0], name).numpy(), getattr(envs[1], name).numpy()), (it, name)
    c = envs[0].counters()
    assert c["n_batched_events"] > 0 and envs[1].counters()["n_batched_events"] == 0
    if cfg is E50:  # batches of released / arriving executors count as batched but not as "stage has more tasks" events
        assert c["n_batched_events"] > c["n_fast_events"]
    for e in envs:
        e.close()


# more jobs with pending events than LDS cache slots (10 at this sizing): the "deep" trace regime (jobs of up to 40 stages with
# thousands of tasks stay active for a long time) with the fair policy spreading 50 executors over ~40 active jobs
DEEP_E50 = dict(num_executors=50, job_arrival_cap=40, job_arrival_rate=1.0e-3, moving_delay=2000.0, warmup_delay=1000.0)


@pytest.mark.parametrize("policy,seeds,chunk,n_launches", [("fair", [0, 1], 41, 5), ("hash", [2], 29, 5)])
def test_runs_over_jobs_without_a_cache_slot_equal_one_event_at_a_time(policy, seeds, chunk, n_launches):
    """fast_run takes the task completions of jobs that have no LDS cache slot too (their records are read from and written back to
    HBM once per run): byte-identical env state to the one-at-a-time build, and nearly every such event goes through runs"""
    from spark_sched_sim_amd import workload

    pack = workload.profile_pack("deep")
    envs = [VecSparkSchedSimEnv(DEEP_E50, len(seeds), device="cpu", pack=pack, _lib=load_emu(v)) for v in ("", "_nobatch")]
    for e in envs:
        e.reset(seed=seeds)
    skip = slice(256, 272)  # SssHdr::n_batched, n_rounds: the only fields that may differ
    for it in range(n_launches):
        for e in envs:
            e.rollout(policy, chunk)
        a, b = (e._env_view.numpy().copy() for e in envs)
        a[:, skip] = 0
        b[:, skip] = 0
        assert np.array_equal(a, b), f"launch {it}: env state differs"
        for name in ("nodes", "edge_links", "dag_ptr", "exec_supplies", "obs_i32", "obs_f64"):
            assert np.array_equal(getattr(envs[0], name).numpy(), getattr(envs[1], name).numpy()), (it, name)
    c = envs[0].counters()
    assert c["n_batched_events"] > 0.9 * c["n_fast_events"] > 0, c   # (about 0.3 while runs stopped at the first event of a job without a slot)
    for e in envs:
        e.close()


def test_jump_table_is_numpys_advance():
    """sss_host.h sss_build_pcg_jump through its test export: state_{n+k} = A_k * state_n + C_k * inc"""
    lib = load_emu()
    tab = (C.c_uint64 * (129 * 4))()
    lib.sss_test_pcg_jump_table(tab)
    t = np.frombuffer(tab, dtype=np.uint64).reshape(129, 4)
    bg = np.random.PCG64(12345)
    st = bg.state["state"]
    s0, inc = int(st["state"]), int(st["inc"])
    mask = (1 << 128) - 1
    for k in (-64, -63, -17, -1, 0, 1, 2, 31, 63, 64):
        row = [int(x) for x in t[k + 64]]
        A, Cc = (row[0] << 64) | row[1], (row[2] << 64) | row[3]
        got = (A * s0 + Cc * inc) & mask
        ref = np.random.PCG64(12345)
        ref.advance(k % (1 << 128))
        assert got == int(ref.state["state"]["state"]), k


def test_level_threshold_table_is_the_reference_expression():
    """sss_host.h sss_build_lvl_thr through its test export: with n local executors strictly between two executor
    levels the reference draws `rand_pt = 1 + int(rng.random() * (right - left))` and keeps the lower level iff
    `rand_pt <= n - left` (data_samplers/tpch.py:222-229); rng.random() is (x >> 11) * 2**-53 for the raw output x.
    The table says: the upper level exactly when (x >> 11) >= thr[n]. Checked with numpy's float64 arithmetic on both
    sides of every threshold, on random mantissas, and against Generator.random() itself."""
    lib = load_emu()
    tab = (C.c_uint64 * 101)()
    lib.sss_test_lvl_thr_table(tab)
    thr = [int(x) for x in tab]
    levels = [5, 10, 20, 40, 50, 60, 80, 100]

    def upper(m, n, left, right):
        u = np.float64(m) * np.float64(2.0 ** -53)
        rand_pt = 1 + int(u * np.float64(right - left))
        return not (np.float64(rand_pt) <= np.float64(n) - np.float64(left))

    rng = np.random.default_rng(7)
    for n in range(1, 101):
        enclosing = [(a, b) for a, b in zip(levels, levels[1:]) if a < n < b]
        if n <= 5 or n in levels or not enclosing:
            assert thr[n] == 1 << 53, n  # closed interval: never the upper level (and no draw at all)
            continue
        left, right = enclosing[0]
        t = thr[n]
        assert 0 < t < (1 << 53)
        assert not upper(t - 1, n, left, right) and upper(t, n, left, right), n
        for m in [0, 1, (1 << 53) - 1] + [int(x) for x in rng.integers(0, 1 << 53, 200)]:
            assert upper(m, n, left, right) == (m >= t), (n, m)
    # the mantissa really is what Generator.random() uses
    g1, g2 = np.random.Generator(np.random.PCG64(99)), np.random.PCG64(99)
    raw = g2.random_raw(50)
    assert [float(np.float64(int(x) >> 11) * 2.0 ** -53) for x in raw] == [float(g1.random()) for _ in range(50)]
