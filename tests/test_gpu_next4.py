"""-m gpu: SURVEY 8(f) next-4 on the HIP path - the reference-exact RandomScheduler plugin and the
metrics percentiles through the single-env facade on cuda:0 against the recorded reference episode
(tests/golden/c1_random.npz), and a pack compiled from the reference's on-disk trace layout
(tpch.py:117-132) driving the kernels to the same trajectories as the in-memory pack."""
import numpy as np
import pytest
import torch

from golden_util import Golden, bits
from replay_util import replay_golden

pytestmark = pytest.mark.gpu


def test_random_scheduler_plugin_and_percentiles_on_the_gpu():
    from spark_sched_sim_amd import RandomScheduler, SparkSchedSimEnv, metrics

    g = Golden("c1_random")
    for seed in (7, 8):
        env = SparkSchedSimEnv(g.cfg, device="cuda:0")
        sched = RandomScheduler(seed=seed)
        obs, _ = env.reset(seed=seed, options=None)
        done, i = False, 0
        while not done:
            action, _ = sched.schedule(obs)
            i += 1
            assert (int(action["stage_idx"]), int(action["num_exec"])) == (int(g.ep(seed, "stage_idx")[i]), int(g.ep(seed, "num_exec")[i])), (seed, i)
            obs, reward, terminated, truncated, info = env.step(action)
            assert bits(reward) == int(g.ep(seed, "reward")[i]) and bits(info["wall_time"]) == int(g.ep(seed, "wall_time")[i]), (seed, i)
            done = terminated or truncated
        assert i == len(g.ep(seed, "reward")) - 1
        assert np.array_equal(np.asarray(metrics.job_durations(env)), g.ep(seed, "job_durations"))
        assert np.array_equal(metrics.job_duration_percentiles(env), np.percentile(g.ep(seed, "job_durations"), [25, 50, 75, 100]))  # metrics.py:21-23
        env.close()


def test_recorded_random_scheduler_trace_replays_bit_exact(pack):
    bad = replay_golden("c1_random", [7, 8], pack, device="cuda:0", full_obs_steps=40)
    assert not bad, "\n".join(bad[:10])


def test_pack_from_the_reference_trace_layout_drives_the_kernels(tmp_path, pack):
    """data/tpch/<size>/{adj_mat,task_duration}_<q>.npy -> pack -> HIP kernels: same bytes, same trajectories"""
    from spark_sched_sim_amd import VecSparkSchedSimEnv, workload

    workload.write_reference_layout(workload.make_raw_workload(), str(tmp_path))
    converted = workload.pack_from_reference_layout(str(tmp_path))
    assert converted == pack
    bad = replay_golden("c1_fair", [1234, 0], converted, device="cuda:0", full_obs_steps=10)
    assert not bad, "\n".join(bad[:10])
    cfg = dict(Golden("c1_fair").cfg)
    a = VecSparkSchedSimEnv(cfg, 64, device="cuda:0", pack=converted)
    b = VecSparkSchedSimEnv(cfg, 64, device="cuda:0", pack=pack)
    for e in (a, b):
        e.reset(seed=900)
        e.rollout("hash", 400)
    torch.cuda.synchronize()
    from spark_sched_sim_amd.vec_env import HDR_PROF
    sa, sb = a._env_view.clone(), b._env_view.clone()
    sa[:, HDR_PROF: HDR_PROF + 40] = 0  # shader-clock profiling counters: the only timing-dependent bytes of an env
    sb[:, HDR_PROF: HDR_PROF + 40] = 0
    assert torch.equal(sa, sb) and torch.equal(a.nodes, b.nodes) and torch.equal(a.obs_f64, b.obs_f64)
    a.close(), b.close()
