#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz by running the REFERENCE env (imported, unmodified, from
/root/reference) on the build's frozen synthetic workload.

Runs only in the build container (where /root/reference is mounted); the GPU box never sees
the reference - it only sees the .npz outputs. Nothing of the reference is copied: this script
imports it, drives it through its public Gymnasium API and records what it returns.

    python tests/golden/make_golden.py            # all fixture sets
    python tests/golden/make_golden.py c1_fair    # one set

Per step a fixture stores the action taken, the returned reward / wall_time (f64 bit patterns),
`terminated`, the scalar observation fields and 64-bit digests (spark_sched_sim_amd/digest.py)
of `nodes` (f32), `edge_links`, `dag_ptr`, `exec_supplies` (all as int32); the first
FULL_OBS_STEPS observations of each episode are stored in full. Episode totals: final
`metrics.job_durations` and `env.avg_job_duration`.
"""
from __future__ import annotations

import os
import os.path as osp
import sys
import tempfile
import types

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
REF = os.environ.get("SSS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, osp.join(ROOT, "tests", "refharness"))  # gymnasium stand-in
sys.path.insert(1, REF)

from spark_sched_sim_amd import workload  # noqa: E402
from spark_sched_sim_amd.digest import digest_words, splitmix64  # noqa: E402

FULL_OBS_STEPS = 40

C1 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0,
          warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler")
C3 = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0,
          warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler")
TINY = dict(num_executors=5, job_arrival_cap=8, job_arrival_rate=1.0e-4, moving_delay=1500.0,
            warmup_delay=500.0, data_sampler_cls="TPCHDataSampler")
TESTYAML = dict(num_executors=50, job_arrival_cap=10, job_arrival_rate=4.0e-5, moving_delay=2000.0,
                warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler")  # reference test/test.yaml:36-41
BIGE = dict(num_executors=64, job_arrival_cap=30, job_arrival_rate=8.0e-5, moving_delay=2000.0,
            warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler")

# more than 64 executors (round 4: the wide instantiation of the kernels): the level table's top (tpch.py:238), and beyond it
# - 120 executors, few jobs: rows 101..119 of executor_intervals are (100, 100), row 120 stays (0, 0) (tpch.py:258-260)
E100 = dict(num_executors=100, job_arrival_cap=40, job_arrival_rate=1.0e-4, moving_delay=2000.0,
            warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler")
E120 = dict(num_executors=120, job_arrival_cap=6, job_arrival_rate=2.0e-5, moving_delay=2000.0,
            warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler")

DEEP = (list(workload.QUERY_SIZES), workload.NUM_QUERIES, workload.DEFAULT_SEED, "deep")

# name -> (env_cfg, policy, seeds, reset options[, (query sizes, number of queries, generator seed[, generator profile])])
SETS = {
    "c1_fair": (C1, "fair", [1234] + list(range(20)), None),
    "c1_hash": (C1, "hash", list(range(100, 112)), None),
    "c3_fair": (C3, "fair", [0, 1], None),
    "c3_hash": (C3, "hash", [7], None),
    "tiny_hash": (TINY, "hash", list(range(40)), None),
    "tiny_fair_tlimit": (dict(TINY, job_arrival_cap=None), "fair", list(range(8)), {"time_limit": 60000.0}),
    "testyaml_fair": (TESTYAML, "fair", [3, 4], None),
    "bige_hash": (BIGE, "hash", [0, 1, 2], None),
    "c1_fifo": (C1, "fifo", [5, 6], None),
    # discounted rewards (trainer config, config/decima_tpch.yaml:69 beta=5e-3): np.exp in the reward
    "c1_fair_beta": (dict(C1, beta=5.0e-3), "fair", [11, 12], None),
    # the reference's RandomScheduler plugin (legacy MT19937 RandomState(seed), heuristics/random_scheduler.py)
    "c1_random": (C1, "random", [7, 8], None),
    "e100_fair": (E100, "fair", [0, 1], None),
    "e100_hash": (E100, "hash", [2], None),
    "e120_hash": (E120, "hash", [0, 1, 2, 3], None),
    # a trace set of another shape - 5 queries x 2 sizes, its own generator seed - as a user would have it after editing
    # QUERY_SIZES / NUM_QUERIES (tpch.py:14-15); here those two module constants are set on the imported module
    "q5s2_fair": (dict(C1, job_arrival_cap=25), "fair", [0, 1, 2], None, (["2g", "10g"], 5, 77)),
    "q5s2_hash": (dict(C1, job_arrival_cap=25), "hash", [3, 4], None, (["2g", "10g"], 5, 77)),
    # the "deep" trace regime (workload.PROFILES: up to 40 stages, in-degree <= 6 over all predecessors, 4 .. 3000 tasks per stage,
    # no executor level 2, durations 50 ms .. 40 s; a 60 MB pack): long runs of task completions per decision, parents far upstream
    "deep_c1_fair": (C1, "fair", [0, 1], None, DEEP),
    "deep_c1_hash": (C1, "hash", [2], None, DEEP),
    "deep_c1_fifo": (dict(C1, job_arrival_cap=30), "fifo", [3], None, DEEP),
    "deep_e50_fair": (dict(C3, job_arrival_cap=100), "fair", [0], None, DEEP),
    "deep_e50_hash": (dict(C3, job_arrival_cap=60), "hash", [1], None, DEEP),
    "deep_e100_fair": (E100, "fair", [0], None, DEEP),
    "deep_e100_hash": (E100, "hash", [1], None, DEEP),
    # ... with discounted rewards, and bounded by a time limit instead of a job cap (episodes end in the middle of long stages)
    "deep_c1_fair_beta": (dict(C1, job_arrival_cap=20, beta=5.0e-3), "fair", [4], None, DEEP),
    "deep_tlimit_hash": (dict(C1, job_arrival_cap=None, job_arrival_rate=2.0e-6), "hash", [5, 6], {"time_limit": 2.0e7}, DEEP),
}


def import_reference():
    """import the reference env + its heuristic schedulers without running
    `schedulers/__init__.py` (which would pull in torch_geometric)."""
    import spark_sched_sim  # noqa: F401  (registers the env with the stand-in)
    pkg = types.ModuleType("schedulers")
    pkg.__path__ = [osp.join(REF, "schedulers")]
    sys.modules["schedulers"] = pkg
    from schedulers.heuristics.round_robin import RoundRobinScheduler
    import gymnasium as gym
    from spark_sched_sim import metrics
    return gym, RoundRobinScheduler, metrics


def hash_policy(obs, seed: int, step: int, p_none_permille: int = 30):
    """the build's counter-based pseudo-random policy (mirrored by the on-device policy):
    keyed on (seed, step) only, so it can be replayed against any implementation."""
    nodes = obs["dag_batch"].nodes
    n_sched = int(nodes[:, 2].sum()) if nodes.shape[0] else 0
    ncommit = int(obs["num_committable_execs"])
    h = splitmix64((seed << 32) ^ step)
    h2 = splitmix64(h)
    h3 = splitmix64(h2)
    if n_sched == 0 or (h3 % 1000) < p_none_permille:
        stage_idx = -1
    else:
        stage_idx = int(h % n_sched)
    num_exec = 1 + int(h2 % max(1, ncommit))
    return {"stage_idx": stage_idx, "num_exec": num_exec}


def run_episode(gym, metrics, env_cfg, policy, seed, options, sched_cls, sizes=tuple(workload.QUERY_SIZES)):
    sizes = list(sizes)
    env = gym.make("spark_sched_sim:SparkSchedSimEnv-v0", env_cfg=dict(env_cfg))
    if policy == "fair":
        sched = sched_cls(env_cfg["num_executors"], dynamic_partition=True)
    elif policy == "fifo":
        sched = sched_cls(env_cfg["num_executors"], dynamic_partition=False)
    elif policy == "random":
        from schedulers.heuristics.random_scheduler import RandomScheduler
        sched = RandomScheduler(seed=seed)
    else:
        sched = None
    obs, info = env.reset(seed=seed, options=dict(options) if options else None)
    time_limit = (options or {}).get("time_limit", np.inf)

    rec = {k: [] for k in ("stage_idx", "num_exec", "reward", "wall_time", "terminated", "ncommit",
                           "src_idx", "n_nodes", "n_edges", "n_jobs", "d_nodes", "d_edges", "d_ptr",
                           "d_sup")}
    full = []

    def record_obs(o, reward, wall, term):
        nodes = np.ascontiguousarray(o["dag_batch"].nodes, dtype=np.float32)
        el = np.ascontiguousarray(o["dag_batch"].edge_links, dtype=np.int32).reshape(-1, 2)
        ptr = np.asarray(o["dag_ptr"], dtype=np.int32)
        sup = np.asarray(o["exec_supplies"], dtype=np.int32)
        rec["reward"].append(np.float64(reward).view(np.uint64))
        rec["wall_time"].append(np.float64(wall).view(np.uint64))
        rec["terminated"].append(int(term))
        rec["ncommit"].append(int(o["num_committable_execs"]))
        rec["src_idx"].append(int(o["source_job_idx"]))
        rec["n_nodes"].append(nodes.shape[0])
        rec["n_edges"].append(el.shape[0])
        rec["n_jobs"].append(sup.size)
        rec["d_nodes"].append(digest_words(nodes))
        rec["d_edges"].append(digest_words(el))
        rec["d_ptr"].append(digest_words(ptr))
        rec["d_sup"].append(digest_words(sup))
        if len(full) < FULL_OBS_STEPS:
            full.append((nodes.copy(), el.copy(), ptr.copy(), sup.copy()))

    # entry 0 = the reset observation (action fields are placeholders)
    rec["stage_idx"].append(-2)
    rec["num_exec"].append(0)
    record_obs(obs, 0.0, info["wall_time"], False)

    step = 0
    error_step, error_msg = -1, ""
    terminated = truncated = False
    while not (terminated or truncated):
        if sched is not None:
            action, _ = sched.schedule(obs)
        else:
            action = hash_policy(obs, seed, step)
        action = {"stage_idx": int(action["stage_idx"]), "num_exec": int(action["num_exec"])}
        try:
            obs, reward, terminated, truncated, info = env.step(action)
        except AssertionError as e:
            # e.g. "[step]": the policy parked every executor (stage_idx=-1) with nothing left in
            # the event queue. The build reports this as a per-env error at the same step.
            error_step, error_msg = step, str(e)
            rec["stage_idx"].append(action["stage_idx"])
            rec["num_exec"].append(action["num_exec"])
            break
        # the StochasticTimeLimit wrapper's rule (wrappers/stochastic_time_limit.py:26-31)
        if info["wall_time"] >= time_limit:
            truncated = True
        rec["stage_idx"].append(action["stage_idx"])
        rec["num_exec"].append(action["num_exec"])
        record_obs(obs, reward, info["wall_time"], terminated)
        step += 1

    u64_keys = ("reward", "wall_time", "d_nodes", "d_edges", "d_ptr", "d_sup")
    out = {k: np.array([int(x) for x in v], dtype=np.uint64 if k in u64_keys else np.int64) for k, v in rec.items()}
    out["job_durations"] = np.asarray(metrics.job_durations(env), dtype=np.float64)
    out["avg_job_duration"] = np.float64(env.unwrapped.avg_job_duration) if len(env.unwrapped.job_duration_buff) else np.float64("nan")
    out["num_jobs"] = np.int64(len(env.unwrapped.jobs))
    out["num_completed"] = np.int64(env.unwrapped.num_completed_jobs)
    out["truncated"] = np.int64(truncated)
    out["error_step"] = np.int64(error_step)
    out["error_msg"] = np.asarray(error_msg)
    # per-job arrival times + template ids pin the reset-time sampling on their own
    jobs = env.unwrapped.jobs
    out["t_arrival"] = np.asarray([jobs[j].t_arrival for j in sorted(jobs)], dtype=np.float64)
    out["template"] = np.asarray(
        [workload.template_index(int(jobs[j].query_num), sizes.index(str(jobs[j].query_size)), len(sizes))
         for j in sorted(jobs)], dtype=np.int32)
    for i, (nodes, el, ptr, sup) in enumerate(full):
        out[f"full{i}_nodes"], out[f"full{i}_edges"] = nodes, el
        out[f"full{i}_ptr"], out[f"full{i}_sup"] = ptr, sup
    out["n_full"] = np.int64(len(full))
    return out


def random_regime(seed: int):
    """a trace-set regime, env configuration and policy drawn from `seed`: generator parameters far from the two committed profiles
    (tests/test_oracle_vs_live_reference.py: the oracle against the reference itself on regimes nobody looked at)"""
    rng = np.random.default_rng(1000 + seed)
    lo = int(rng.integers(2, 5))
    levels_all = [2, 5, 10, 20, 40, 50, 60, 80, 100]
    keep = sorted(rng.choice(len(levels_all), size=int(rng.integers(3, len(levels_all) + 1)), replace=False).tolist())
    prof = dict(stages=(lo, int(rng.integers(lo + 2, 31))), max_in=int(rng.integers(1, 6)), parent_window=[None, 2, 4, 8][int(rng.integers(0, 4))],
                tasks=(1, int(rng.integers(3, 60))), tasks_div=int(rng.integers(1, 4)), base=(int(rng.integers(20, 100)), int(rng.integers(500, 20000))),
                levels=[levels_all[i] for i in keep])
    cfg = dict(num_executors=int(rng.choice([3, 5, 8, 10, 16, 24, 37, 50, 64, 65, 100, 128])), job_arrival_cap=int(rng.integers(4, 11)), job_arrival_rate=float(10 ** rng.uniform(-4.3, -3.3)),
               moving_delay=float(rng.choice([0.0, 500.0, 2000.0])), warmup_delay=float(rng.choice([0.0, 300.0, 1000.0])), data_sampler_cls="TPCHDataSampler")
    policy = ["fair", "hash", "fifo"][int(rng.integers(0, 3))]
    shape = (["2g", "10g", "100g"], 4, 9000 + seed, prof)
    return cfg, policy, [int(rng.integers(0, 1000))], shape


def main(argv):
    if argv and argv[0] == "--random":  # python make_golden.py --random SEED OUT.npz
        seed, out_path = int(argv[1]), argv[2]
        SETS["random"] = (*random_regime(seed)[:3], None, random_regime(seed)[3])
        global HERE
        HERE, argv = osp.dirname(osp.abspath(out_path)), ["random"]
        main(argv)
        os.replace(osp.join(HERE, "random.npz"), out_path)
        return
    names = argv or list(SETS)
    cwd0 = os.getcwd()
    gym = sched_cls = metrics = None
    for name in names:
        env_cfg, policy, seeds, options = SETS[name][:4]
        shape = SETS[name][4] if len(SETS[name]) > 4 else None
        sizes, n_queries, raw_seed = shape[:3] if shape else (list(workload.QUERY_SIZES), workload.NUM_QUERIES, workload.DEFAULT_SEED)
        profile = shape[3] if shape and len(shape) > 3 else "default"
        raw = workload.make_raw_workload(raw_seed, sizes, n_queries, profile=profile)
        pack = workload.build_pack(raw)
        with tempfile.TemporaryDirectory() as tmp:
            workload.write_reference_layout(raw, tmp)
            os.chdir(tmp)  # the reference reads data/tpch relative to cwd (tpch.py:48,119)
            if gym is None:
                gym, sched_cls, metrics = import_reference()
            from spark_sched_sim.data_samplers import tpch
            keep = (tpch.QUERY_SIZES, tpch.NUM_QUERIES)
            tpch.QUERY_SIZES, tpch.NUM_QUERIES = list(sizes), n_queries  # (the reference's own values unless the set names a shape)
            try:
                blob = {}
                for seed in seeds:
                    ep = run_episode(gym, metrics, env_cfg, policy, seed, options, sched_cls, sizes)
                    for k, v in ep.items():
                        blob[f"s{seed}_{k}"] = v
                    print(f"{name} seed={seed}: {len(ep['stage_idx']) - 1} steps, "
                          f"{int(ep['num_completed'])}/{int(ep['num_jobs'])} jobs"
                          + (f"  ERROR at step {int(ep['error_step'])}: {ep['error_msg']}" if ep["error_step"] >= 0 else ""),
                          flush=True)
            finally:
                tpch.QUERY_SIZES, tpch.NUM_QUERIES = keep
                os.chdir(cwd0)
        blob["seeds"] = np.asarray(seeds, dtype=np.int64)
        blob["policy"] = np.asarray(policy)
        blob["pack_sha256"] = np.asarray(workload.pack_digest(pack))
        if shape:  # what tests/golden_util.py needs to rebuild the set's pack
            blob["trace_sizes"], blob["trace_queries"], blob["trace_seed"] = np.asarray(sizes), np.int64(n_queries), np.int64(raw_seed)
            if isinstance(profile, dict):
                import json
                blob["trace_profile_json"] = np.asarray(json.dumps(profile))
            elif profile != "default":
                blob["trace_profile"] = np.asarray(profile)
        blob["cfg_keys"] = np.asarray(sorted(k for k in env_cfg if k != "data_sampler_cls"))
        blob["cfg_vals"] = np.asarray(
            [np.nan if env_cfg[k] is None else float(env_cfg[k]) for k in sorted(env_cfg) if k != "data_sampler_cls"],
            dtype=np.float64)
        blob["time_limit"] = np.float64((options or {}).get("time_limit", np.inf))
        np.savez_compressed(osp.join(HERE, f"{name}.npz"), **blob)


if __name__ == "__main__":
    main(sys.argv[1:])
