#!/usr/bin/env python3
"""Times the REFERENCE's Decima policy in the loop and its PPO update (both imported, unmodified, from /root/reference with the
functional PyG stand-ins of tests/refharness/pygstubs) on the build's frozen synthetic trace set - the CPU baselines that stand next
to `decima_in_loop` and `ppo_config5_share` in bench.py's line (round-5 verdict, missing 2).

Runs ONLY in the build container (/root/reference does not exist on the GPU box). Adds two records to profiles/reference_python.json,
which is committed; bench.py quotes them with `measured_in_this_run: false` and the hardware stated.

  decima_c1    DecimaScheduler.schedule(obs) + env.step(action) through DecimaEnvWrapper (schedulers/decima/scheduler.py:71-99,
               env_wrapper.py:12-161; loop of examples.py:84-102), BASELINE config 1/2/4 sizing (10 executors, 50 jobs), random-init
               weights of the published architecture, one core (torch.set_num_threads(1), as rollout_worker.py:93 does)
  ppo_config5  BASELINE config 5's sizing (50 executors, 200 jobs, decima_tpch.yaml hyper-parameters) at 2 sequences x 2 rollouts:
               RolloutWorkerSync.collect_rollout (rollout_worker.py:133-160) per worker, then PPO.train_on_rollouts
               (trainers/ppo.py:51-138: 3 epochs x 10 minibatches) on them; samples / s of both

    python tools/time_reference_decima.py [decima] [ppo]
"""
from __future__ import annotations

import json
import os
import os.path as osp
import platform
import sys
import tempfile
import time
import types

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
REF = os.environ.get("SSS_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, osp.join(ROOT, "tests", "refharness"))
sys.path.insert(0, osp.join(ROOT, "tests", "refharness", "pygstubs"))
sys.path.insert(2, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

tb = types.ModuleType("torch.utils.tensorboard")
tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = tb

from spark_sched_sim_amd import workload  # noqa: E402

AGENT = dict(agent_cls="DecimaScheduler", embed_dim=16,
             gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(inplace=True, negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))  # config/decima_tpch.yaml:68-78
C1 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler")
C5 = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler",
          mean_time_limit=2.0e7)
TRAIN = dict(trainer_cls="PPO", device="cpu", num_iterations=1, num_sequences=2, num_rollouts=2, seed=42, artifacts_dir="artifacts", checkpointing_freq=50,
             use_tensorboard=False, num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04, beta_discount=5.0e-3, opt_cls="Adam",
             opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_decima(gym) -> dict:
    from schedulers.decima.env_wrapper import DecimaEnvWrapper
    from schedulers.decima.scheduler import DecimaScheduler

    torch.set_num_threads(1)  # rollout_worker.py:93
    torch.manual_seed(0)
    sched = DecimaScheduler(num_executors=C1["num_executors"], **{k: v for k, v in AGENT.items() if k != "agent_cls"})
    sched.eval()
    t_sched = t_step = 0.0
    steps = nodes = 0
    seeds = [0, 1, 2]
    for seed in seeds:
        env = DecimaEnvWrapper(gym.make("spark_sched_sim:SparkSchedSimEnv-v0", env_cfg=dict(C1)))
        obs, _ = env.reset(seed=seed)
        done = False
        while not done:
            t0 = time.perf_counter()
            action, _ = sched.schedule(obs)
            t1 = time.perf_counter()
            obs, _, term, trunc, _ = env.step(action)
            t2 = time.perf_counter()
            t_sched += t1 - t0
            t_step += t2 - t1
            steps += 1
            nodes += len(obs["dag_batch"].nodes)
            done = term or trunc
    return {"episodes": len(seeds), "steps": steps, "active_nodes_per_step": nodes / steps, "cores": 1,
            "env_steps_per_s": steps / (t_sched + t_step), "schedule_calls_per_s": steps / t_sched, "env_only_steps_per_s": steps / t_step,
            "schedule_share_of_time": t_sched / (t_sched + t_step),
            "what": "reference DecimaScheduler.schedule + DecimaEnvWrapper + env.step, 10 executors / 50 jobs, whole episodes, one torch thread"}


def time_ppo(gym) -> dict:
    from schedulers.decima.env_wrapper import DecimaEnvWrapper
    from spark_sched_sim.wrappers import StochasticTimeLimit
    from trainers import make_trainer
    from trainers.rollout_worker import RolloutWorkerSync

    trainer = make_trainer(dict(trainer=dict(TRAIN), agent=dict(AGENT), env=dict(C5)))
    env_cfg = trainer.env_cfg
    sched = trainer.scheduler
    base_seeds = np.repeat(TRAIN["seed"] + np.arange(TRAIN["num_sequences"]), TRAIN["num_rollouts"])
    workers = []
    for r, s in enumerate(base_seeds):
        w = RolloutWorkerSync()
        w.rank, w.base_seed, w.seed_step, w.reset_count = r, int(s), TRAIN["num_sequences"], 0
        env = StochasticTimeLimit(gym.make("spark_sched_sim:SparkSchedSimEnv-v0", env_cfg=env_cfg), env_cfg["mean_time_limit"])
        w.env = DecimaEnvWrapper(env)
        w.scheduler = sched  # (the reference gives every worker process a copy with the learner's state_dict, rollout_worker.py:97-102)
        workers.append(w)
    torch.set_num_threads(1)  # the workers' setting (rollout_worker.py:93)
    sched.eval()
    t0 = time.perf_counter()
    with torch.no_grad():
        buffers = [w.collect_rollout() for w in workers]
    t_collect = time.perf_counter() - t0
    samples = sum(len(b) for b in buffers)
    torch.set_num_threads(len(os.sched_getaffinity(0)))  # the learner process keeps torch's default thread count
    sched.train()
    t0 = time.perf_counter()
    info = trainer.train_on_rollouts(buffers)
    t_train = time.perf_counter() - t0
    return {"rollouts": len(buffers), "samples": samples, "collect_s": t_collect, "collect_env_steps_per_s_one_core": samples / t_collect,
            "train_s": t_train, "train_samples_per_s": samples * TRAIN["num_epochs"] / t_train, "train_threads": len(os.sched_getaffinity(0)),
            "learn_info": {k: float(v) for k, v in (info or {}).items() if isinstance(v, (int, float))},
            "what": "reference RolloutWorkerSync.collect_rollout x 4 (50 executors, 200 jobs, one thread, one after the other) then PPO.train_on_rollouts "
                    "(3 epochs x 10 minibatches; samples x epochs / s assumes no KL early stop - see learn_info) in the learner's default thread count"}


def main() -> None:
    if not osp.isdir(REF):
        raise SystemExit(f"{REF} not found: this script runs in the build container only")
    which = sys.argv[1:] or ["decima", "ppo"]
    path = osp.join(ROOT, "profiles", "reference_python.json")
    out = json.load(open(path))
    raw = workload.make_raw_workload()
    assert out["pack_sha256"] == workload.pack_digest(workload.build_pack(raw))
    cwd0 = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        workload.write_reference_layout(raw, tmp)
        os.chdir(tmp)
        try:
            import gymnasium as gym
            import spark_sched_sim  # noqa: F401
            hw = f"build container: {cpu_model()}, {len(os.sched_getaffinity(0))} vCPU (not the GPU box's host)"
            if "decima" in which:
                out["decima_c1"] = dict(time_decima(gym), hardware=hw)
                print(json.dumps(out["decima_c1"]), flush=True)
            if "ppo" in which:
                out["ppo_config5"] = dict(time_ppo(gym), hardware=hw)
                print(json.dumps(out["ppo_config5"]), flush=True)
        finally:
            os.chdir(cwd0)
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
