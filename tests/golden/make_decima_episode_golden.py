#!/usr/bin/env python3
"""Regenerates tests/golden/decima_episode.npz: one full episode of the reference's harness
(examples.run_episode shape: DecimaEnvWrapper around the env, DecimaScheduler.schedule sampling
with Python's `random.choices`, `random.seed` fixed) on the build's frozen workload. Build-container
only; same import arrangement as make_decima_golden.py. Stores the weights (torch seed 1234) and the
per-step Decima actions, rewards and wall times."""
from __future__ import annotations

import os
import os.path as osp
import random
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

ENV_CFG = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0,
               data_sampler_cls="TPCHDataSampler")
AGENT = dict(embed_dim=16,
             gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(inplace=True, negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
SEED, PY_SEED = 21, 5


def main():
    raw = workload.make_raw_workload()
    with tempfile.TemporaryDirectory() as tmp:
        workload.write_reference_layout(raw, tmp)
        os.chdir(tmp)
        import gymnasium as gym
        import spark_sched_sim  # noqa: F401
        from schedulers.decima.scheduler import DecimaScheduler

        torch.manual_seed(1234)
        sched = DecimaScheduler(num_executors=ENV_CFG["num_executors"], **AGENT)
        with torch.no_grad():
            for n_, p in sched.named_parameters():
                if "bias" in n_:
                    p.uniform_(-0.1, 0.1)
        sched.eval()
        blob = {f"w_{k}": v.numpy().copy() for k, v in sched.state_dict().items()}
        env = gym.make("spark_sched_sim:SparkSchedSimEnv-v0", env_cfg=dict(ENV_CFG))
        env = sched.env_wrapper_cls(env)
        random.seed(PY_SEED)
        obs, _ = env.reset(seed=SEED, options=None)
        acts, rews, walls, lgs = [], [], [], []
        terminated = truncated = False
        while not (terminated or truncated):
            action, info = sched.schedule(obs)
            obs, reward, terminated, truncated, einfo = env.step(action)
            acts.append([action["stage_idx"], action["job_idx"], action["num_exec"]])
            rews.append(float(reward)), walls.append(float(einfo["wall_time"])), lgs.append(float(info["lgprob"]))
        blob.update(actions=np.asarray(acts, dtype=np.int64), rewards=np.asarray(rews), wall_times=np.asarray(walls),
                    lgprobs=np.asarray(lgs), seed=np.int64(SEED), py_seed=np.int64(PY_SEED))
        blob["cfg_keys"] = np.asarray(sorted(k for k in ENV_CFG if k != "data_sampler_cls"))
        blob["cfg_vals"] = np.asarray([float(ENV_CFG[k]) for k in sorted(ENV_CFG) if k != "data_sampler_cls"])
        np.savez_compressed(osp.join(HERE, "decima_episode.npz"), **blob)
        print("episode:", len(acts), "steps")


if __name__ == "__main__":
    main()
