#!/usr/bin/env python3
"""Regenerates tests/golden/decima_*.npz: the reference's Decima observation wrapper and GNN
(schedulers/decima/{env_wrapper,scheduler,utils}.py), imported unmodified from /root/reference
with functional stand-ins for torch_geometric / torch_sparse / torch_scatter
(tests/refharness/pygstubs), run on the build's frozen synthetic workload.

Build-container only (needs /root/reference). For a few seeds the env is driven by the
counter-based test policy (through DecimaActWrapper); at every step of the first STEPS steps the
fixture stores the wrapped observation (node features f32[N,5], stage_mask, exec_mask, DAG-layer
edge_masks) and the scores a DecimaScheduler with fixed random weights assigns: stage scores and,
for every active job, the executor-count scores. The weights themselves are stored too (they are
generated here from a torch seed; they are not the reference's model.pt).
"""
from __future__ import annotations

import os
import os.path as osp
import sys
import tempfile

import numpy as np

HERE = osp.dirname(osp.abspath(__file__))
ROOT = osp.dirname(osp.dirname(HERE))
REF = os.environ.get("SSS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, osp.join(ROOT, "tests", "refharness"))
sys.path.insert(0, osp.join(ROOT, "tests", "refharness", "pygstubs"))
sys.path.insert(2, REF)

import torch  # noqa: E402

from spark_sched_sim_amd import workload  # noqa: E402
from spark_sched_sim_amd.digest import splitmix64  # noqa: E402

STEPS = 90
CFGS = {
    "decima_c1": (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0,
                       warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler"), [3, 4]),
    "decima_e50": (dict(num_executors=50, job_arrival_cap=12, job_arrival_rate=8.0e-5, moving_delay=2000.0,
                        warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler"), [5]),
    # more than 64 executors: executor-count scores / exec_mask of 100 entries per job, the simulator's wide instantiation
    "decima_e100": (dict(num_executors=100, job_arrival_cap=12, job_arrival_rate=1.2e-4, moving_delay=2000.0,
                         warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler"), [6]),
    # the "deep" trace regime (workload.PROFILES: DAGs of up to 40 stages and 12 layers, in-degree <= 6, thousands of tasks per stage)
    "decima_deep": (dict(num_executors=10, job_arrival_cap=30, job_arrival_rate=4.0e-5, moving_delay=2000.0,
                         warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler"), [7, 8], "deep"),
}
AGENT = dict(embed_dim=16,
             gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(inplace=True, negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))  # config/decima_tpch.yaml:68-78


def main(argv):
    cwd0 = os.getcwd()
    if argv and argv[0] == "--random":  # python make_decima_golden.py --random SEED OUT.npz (tests/test_oracle_vs_live_reference.py)
        global STEPS, HERE
        seed, out_path = int(argv[1]), osp.abspath(argv[2])
        sys.path.insert(0, HERE)
        from make_golden import random_regime
        cfg, _, seeds, (sizes, n_q, raw_seed, prof) = random_regime(seed)
        CFGS.clear()
        CFGS["random"] = (cfg, seeds, ("random", sizes, n_q, raw_seed, prof))
        STEPS, HERE = 40, osp.dirname(out_path)
        with tempfile.TemporaryDirectory() as tmp:
            try:
                record(["random"], CFGS["random"][2], tmp)
            finally:
                os.chdir(cwd0)
        os.replace(osp.join(HERE, "random.npz"), out_path)
        return
    for profile in sorted({(c[2] if len(c) > 2 else "default") for c in CFGS.values()}):
        names = [n for n, c in CFGS.items() if (c[2] if len(c) > 2 else "default") == profile and (not argv or n in argv)]
        if names:
            with tempfile.TemporaryDirectory() as tmp:
                try:
                    record(names, profile, tmp)
                finally:
                    os.chdir(cwd0)


def record(names, profile, tmp):
    shape = None
    if isinstance(profile, tuple):  # ("random", sizes, number of queries, generator seed, generator parameters)
        _, sizes, n_q, raw_seed, prof = profile
        raw, shape = workload.make_raw_workload(raw_seed, sizes, n_q, profile=prof), (sizes, n_q, raw_seed, prof)
    else:
        raw = workload.make_raw_workload(profile=profile)
    if True:
        workload.write_reference_layout(raw, tmp)
        os.chdir(tmp)  # the reference reads data/tpch relative to cwd (tpch.py:48,119)
        import gymnasium as gym
        import spark_sched_sim  # noqa: F401
        if shape is not None:  # (a trace set of another shape: the sampler's two module constants, tpch.py:14-15)
            from spark_sched_sim.data_samplers import tpch
            tpch.QUERY_SIZES, tpch.NUM_QUERIES = list(shape[0]), shape[1]
        from schedulers.decima import utils as dutils
        from schedulers.decima.env_wrapper import DecimaEnvWrapper
        from schedulers.decima.scheduler import DecimaScheduler

        for name in names:
            env_cfg, seeds = CFGS[name][:2]
            torch.manual_seed(1234)
            sched = DecimaScheduler(num_executors=env_cfg["num_executors"], **AGENT)
            # give the biases (zeroed by the reference's constructor) some life
            with torch.no_grad():
                for n_, p in sched.named_parameters():
                    if "bias" in n_:
                        p.uniform_(-0.1, 0.1)
            sched.eval()
            blob = {f"w_{k}": v.numpy() for k, v in sched.state_dict().items()}
            blob["seeds"] = np.asarray(seeds)
            if shape is not None:
                import json
                blob["trace_sizes"], blob["trace_queries"], blob["trace_seed"] = np.asarray(shape[0]), np.int64(shape[1]), np.int64(shape[2])
                blob["trace_profile_json"] = np.asarray(json.dumps(shape[3]))
            elif profile != "default":
                blob["trace_profile"] = np.asarray(profile)
            blob["cfg_keys"] = np.asarray(sorted(k for k in env_cfg if k != "data_sampler_cls"))
            blob["cfg_vals"] = np.asarray([float(env_cfg[k]) for k in sorted(env_cfg) if k != "data_sampler_cls"])
            for seed in seeds:
                env = DecimaEnvWrapper(gym.make("spark_sched_sim:SparkSchedSimEnv-v0", env_cfg=dict(env_cfg)))
                obs, _ = env.reset(seed=seed)
                acts = []
                for t in range(STEPS):
                    raw_nodes = obs["dag_batch"].nodes
                    with torch.no_grad():
                        dag_batch = dutils.obs_to_pyg(obs)
                        h = sched.encoder(dag_batch)
                        stage_scores = sched.stage_policy_network(dag_batch, h).numpy()
                        n_jobs = len(obs["dag_ptr"]) - 1
                        exec_scores = [sched.exec_policy_network(dag_batch, h, j).numpy() for j in range(n_jobs)]
                    p = f"s{seed}_t{t}_"
                    blob[p + "nodes"] = np.asarray(raw_nodes, dtype=np.float32)
                    blob[p + "stage_mask"] = np.asarray(obs["stage_mask"], dtype=bool)
                    blob[p + "exec_mask"] = np.asarray(obs["exec_mask"], dtype=bool)
                    blob[p + "edge_masks"] = np.asarray(obs["edge_masks"], dtype=bool)
                    blob[p + "edge_links"] = np.asarray(obs["dag_batch"].edge_links, dtype=np.int32)
                    blob[p + "dag_ptr"] = np.asarray(obs["dag_ptr"], dtype=np.int32)
                    blob[p + "stage_scores"] = stage_scores
                    blob[p + "exec_scores"] = np.concatenate(exec_scores) if exec_scores else np.zeros(0, np.float32)
                    blob[p + "exec_counts"] = np.asarray([len(e) for e in exec_scores], dtype=np.int32)
                    # next action: counter-based test policy in Decima's action format
                    n_sched = int(np.asarray(obs["stage_mask"]).sum())
                    h1 = splitmix64((seed << 32) ^ t)
                    h2 = splitmix64(h1)
                    stage_idx = int(h1 % n_sched)
                    stage_node = int(np.flatnonzero(np.asarray(obs["stage_mask"]))[stage_idx])
                    job_idx = int(np.searchsorted(np.asarray(obs["dag_ptr"]), stage_node, side="right") - 1)
                    n_allowed = int(np.asarray(obs["exec_mask"])[job_idx].sum())
                    num_exec = int(h2 % max(1, n_allowed))  # Decima's num_exec is 0-based (env_wrapper.py:33-34)
                    acts.append((stage_idx, 1 + num_exec))
                    obs, r, term, trunc, _ = env.step({"stage_idx": stage_idx, "job_idx": job_idx, "num_exec": num_exec})
                    if term or trunc:
                        break
                blob[f"s{seed}_actions"] = np.asarray(acts, dtype=np.int32)
                print(name, "seed", seed, len(acts), "steps", flush=True)
            np.savez_compressed(osp.join(HERE, f"{name}.npz"), **blob)


if __name__ == "__main__":
    main(sys.argv[1:])
