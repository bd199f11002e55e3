"""The single-env facade + host-side plugin surface under the CPU wave emulator: an episode loop
written exactly like the reference's `examples.run_episode` (examples.py:84-102) with the
RoundRobinScheduler plugin must reproduce the recorded reference episode, metrics included, and
invalid actions must raise what the reference raises."""
import numpy as np
import pytest

from emu_util import load_emu
from golden_util import Golden, bits
from spark_sched_sim_amd import RoundRobinScheduler, SparkSchedSimEnv, metrics


def run_episode(env_cfg, scheduler, seed, lib):
    env = SparkSchedSimEnv(env_cfg, device="cpu", _lib=lib)
    obs, _ = env.reset(seed=seed, options=None)
    terminated = truncated = False
    trace = []
    while not (terminated or truncated):
        action, _ = scheduler.schedule(obs)
        obs, reward, terminated, truncated, info = env.step(action)
        trace.append((int(action["stage_idx"]), int(action["num_exec"]), reward, info["wall_time"]))
    return env, trace


@pytest.mark.parametrize("name,seed,dyn", [("c1_fair", 1234, True), ("testyaml_fair", 3, True), ("c1_fifo", 5, False)])
def test_reference_style_episode_loop(name, seed, dyn):
    g = Golden(name)
    env, trace = run_episode(g.cfg, RoundRobinScheduler(g.cfg["num_executors"], dynamic_partition=dyn), seed, load_emu())
    assert len(trace) == len(g.ep(seed, "reward")) - 1
    for i, (si, ne, r, w) in enumerate(trace, start=1):
        assert (si, ne) == (int(g.ep(seed, "stage_idx")[i]), int(g.ep(seed, "num_exec")[i])), i
        assert bits(r) == int(g.ep(seed, "reward")[i]) and bits(w) == int(g.ep(seed, "wall_time")[i]), i
    # metrics: same values in the same order as the reference's list => identical mean
    assert np.array_equal(np.asarray(metrics.job_durations(env)), g.ep(seed, "job_durations"))
    assert bits(env.avg_job_duration) == bits(g.ep(seed, "avg_job_duration"))
    assert env.all_jobs_complete and env.num_completed_jobs == int(g.ep(seed, "num_jobs"))
    env.close()


def test_invalid_actions_raise_like_the_reference():
    g = Golden("tiny_hash")
    env = SparkSchedSimEnv(g.cfg, device="cpu", _lib=load_emu())
    obs, _ = env.reset(seed=0)
    n_nodes = obs["dag_batch"].nodes.shape[0]
    n_sched = int(obs["dag_batch"].nodes[:, 2].sum())
    with pytest.raises(ValueError, match="action space"):
        env.step({"stage_idx": n_nodes, "num_exec": 1})          # spark_sched_sim.py:276-277
    with pytest.raises(ValueError, match="action space"):
        env.step({"stage_idx": 0, "num_exec": 0})
    with pytest.raises(ValueError, match="action space"):
        env.step({"stage_idx": 0, "num_exec": 1, "job_idx": 0})  # Dict.contains rejects extra keys
    if n_sched < n_nodes:
        with pytest.raises(KeyError):
            env.step({"stage_idx": n_sched, "num_exec": 1})      # spark_sched_sim.py:284
    env2 = SparkSchedSimEnv(dict(g.cfg, num_executors=5), device="cpu", _lib=load_emu())
    obs2, _ = env2.reset(seed=1)
    # the env stays usable after a rejected action and the trajectory is unaffected
    obs_after, r, term, trunc, _ = env.step({"stage_idx": int(g.ep(0, "stage_idx")[1]), "num_exec": int(g.ep(0, "num_exec")[1])})
    assert bits(r) == int(g.ep(0, "reward")[1])
    with pytest.raises(ValueError, match="limit"):
        SparkSchedSimEnv(dict(g.cfg, job_arrival_cap=None, max_jobs=16), device="cpu", _lib=load_emu()).reset(seed=0)
    env.close()
    env2.close()


def test_action_space_follows_the_observation():
    """spark_sched_sim.py:85-94, 403-404: Dict(stage_idx: Discrete(n_nodes + 1, start=-1),
    num_exec: Discrete(E, start=1)); contains() is what step() validates against"""
    g = Golden("tiny_hash")
    env = SparkSchedSimEnv(g.cfg, device="cpu", _lib=load_emu())
    assert env.action_space["stage_idx"].n == 1 and env.action_space["num_exec"].n == g.cfg["num_executors"]
    obs, _ = env.reset(seed=0)
    n = len(obs["dag_batch"].nodes)
    sp = env.action_space
    assert sp["stage_idx"].n == n + 1 and sp["stage_idx"].start == -1
    assert sp.contains({"stage_idx": -1, "num_exec": 1}) and sp.contains({"stage_idx": n - 1, "num_exec": g.cfg["num_executors"]})
    assert not sp.contains({"stage_idx": n, "num_exec": 1}) and not sp.contains({"stage_idx": 0, "num_exec": 0})
    assert not sp.contains({"stage_idx": 0, "num_exec": 1, "job_idx": 0})
    env.close()


def test_reference_style_episode_loop_reproduces_the_reference_metric():
    """the loop every harness of the reference runs (examples.py:84-102, rollout_worker.py:135-157)
    against the facade: reset(seed) -> schedule(obs) -> step(action) until the episode ends, then
    metrics.avg_job_duration - fair scheduler, the reference's example config, seed 1234 -> the same
    average job duration the recorded reference episode has"""
    from spark_sched_sim_amd import metrics

    g = Golden("c1_fair")
    env = SparkSchedSimEnv(g.cfg, device="cpu", _lib=load_emu())
    sched = RoundRobinScheduler(g.cfg["num_executors"], dynamic_partition=True)
    obs, _ = env.reset(seed=1234, options=None)
    done = False
    while not done:
        action, _ = sched.schedule(obs)
        obs, _, terminated, truncated, _ = env.step(action)
        done = terminated or truncated
    got = metrics.avg_job_duration(env) * 1e-3
    assert bits(got) == bits(np.mean(g.ep(1234, "job_durations")) * 1e-3)
    env.close()


def test_observation_space_follows_the_episode():
    """spark_sched_sim.py:96-125: the observation space of the reference, with the two bounds that move
    (`dag_ptr` after every observation, :403)"""
    g = Golden("tiny_hash")
    env = SparkSchedSimEnv(g.cfg, device="cpu", _lib=load_emu())
    sp = env.observation_space
    assert set(sp.keys()) == {"dag_batch", "dag_ptr", "num_committable_execs", "source_job_idx", "exec_supplies"}
    assert sp["num_committable_execs"].n == g.cfg["num_executors"] + 1 and sp["exec_supplies"].feature_space.n == 2 * g.cfg["num_executors"]
    obs, _ = env.reset(seed=0)
    n = len(obs["dag_batch"].nodes)
    assert sp["dag_ptr"].feature_space.n == n + 1
    assert sp["dag_batch"].contains(obs["dag_batch"]) and sp["dag_ptr"].contains(obs["dag_ptr"])
    assert sp["num_committable_execs"].contains(obs["num_committable_execs"]) and sp["exec_supplies"].contains(obs["exec_supplies"])
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    vec = VecSparkSchedSimEnv(g.cfg, 2, device="cpu", _lib=load_emu())
    assert set(vec.single_observation_space.keys()) == set(sp.keys()) and vec.single_action_space["num_exec"].n == g.cfg["num_executors"]
    vec.close()
    env.close()


def test_step_returns_copies_of_the_small_vectors():
    """rewards / flags kept across steps must not change under the caller's feet (the observation tensors are views, by contract)"""
    import torch

    from spark_sched_sim_amd import VecSparkSchedSimEnv

    g = Golden("tiny_hash")
    vec = VecSparkSchedSimEnv(g.cfg, 3, device="cpu", _lib=load_emu())
    vec.reset(seed=0)
    kept = []
    for _ in range(12):
        _, rew, term, _, info = vec.step(vec.policy_actions("fair"))
        kept.append((rew, rew.clone(), info["wall_time"], info["wall_time"].clone()))
    for rew, rew0, wt, wt0 in kept:
        assert torch.equal(rew, rew0) and torch.equal(wt, wt0)
    assert any(float(k[2].max()) > 0 for k in kept)
    vec.close()


def test_edge_rows_are_rewritten_after_a_rebind(pack):
    """write_observation skips an env's edge rows while its active subgraph is what they were written from
    (csrc/sss_sim.h, INTEGRATION.md "Aliasing": the observation buffers are read-only for the caller). A caller that
    swaps or edits the buffers rebinds them (`rebind_buffers` -> sss_bind_buffers starts a new buffer generation) and
    gets the rows written again at the next step; without the rebind an in-place edit would persist - which is the
    documented contract, checked here as well so that a change of it is noticed."""
    import torch

    from emu_util import load_emu
    from spark_sched_sim_amd import VecSparkSchedSimEnv

    cfg = dict(num_executors=10, job_arrival_cap=12, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 2, device="cpu", pack=pack, _lib=load_emu())
    env.reset(seed=[5, 6])
    # a step that leaves the active subgraph as it is: no job arrives or completes, no stage completes
    for _ in range(50):
        before = env.header_field("n_active").clone(), env.obs_i32[:, 1].clone()
        snap = env.edge_links.clone()
        ver = [env.header(k)["n_events"] for k in range(2)]
        act = env.policy_actions("fair")
        env.step(act)
        same = torch.equal(env.edge_links, snap) and torch.equal(env.obs_i32[:, 1], before[1]) and int(env.obs_i32[:, 1].min()) > 0
        if same:
            break
    assert same, "no step with an unchanged subgraph found"
    good = env.edge_links.clone()
    n_edges = env.obs_i32[:, 1].clone()
    # scribble, step once more with the subgraph unchanged again if possible; either way a rebind restores the rows
    env.edge_links.fill_(-7)
    env.rebind_buffers()
    env.step(env.policy_actions("fair"))
    for k in range(2):
        ne = int(env.obs_i32[k, 1])
        assert ne > 0 and int(env.edge_links[k, :ne].min()) >= 0, "edge rows were not rewritten after the rebind"
    # and without a rebind the skipped rows are left alone (the read-only contract)
    snap2, ne2 = env.edge_links.clone(), env.obs_i32[:, 1].clone()
    env.edge_links[:, 0, 0] = -9
    env.step(env.policy_actions("fair"))
    unchanged = [k for k in range(2) if int(env.obs_i32[k, 1]) == int(ne2[k]) and torch.equal(env.edge_links[k, 1:], snap2[k, 1:])]
    for k in unchanged:
        assert int(env.edge_links[k, 0, 0]) in (-9, int(snap2[k, 0, 0]))  # kept (skipped) or rewritten (graph changed back to equal rows)
    env.close()
    assert good.shape == snap.shape and n_edges.numel() == 2


def test_late_hint_returns_the_latest_completed_copy_with_its_tag():
    """vec_env.LateHint on CPU tensors (copies are synchronous there): nothing posted -> the fill value and no tag; afterwards the latest
    post with the tag it was given - `decima_graph_on_device` tags a read-back with the set of list counters it belongs to"""
    import torch
    from spark_sched_sim_amd.vec_env import LateHint

    h = LateHint(4, torch.device("cpu"))
    v, tag = h.read_tagged()
    assert v.tolist() == [-1] * 4 and tag is None and h.read().tolist() == [-1] * 4
    h.post(torch.tensor([1, 2, 3, 4]), tag=1)
    v, tag = h.read_tagged()
    assert v.tolist() == [1, 2, 3, 4] and tag == 1
    src = torch.tensor([5, 6, 7, 8])
    h.post(src, tag=0)
    src[0] = 99  # (the hint holds a copy)
    v, tag = h.read_tagged()
    assert v.tolist() == [5, 6, 7, 8] and tag == 0
    v[1] = 0  # (and hands out copies)
    assert h.read().tolist() == [5, 6, 7, 8]
