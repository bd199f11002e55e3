#!/usr/bin/env python3
"""A short PPO run on the batched env (GPU): prints one JSON line per iteration (average number of
concurrent jobs under the current policy - the quantity the reference's trainer tracks and
checkpoints on, trainer.py:133-142 - PPO statistics, samples, seconds). Evidence that the
collect -> returns -> baselines -> CLIP-loss -> Adam loop learns, not a tuned experiment."""
import argparse
import json
import os
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from spark_sched_sim_amd.training import Trainer  # noqa: E402

AGENT = dict(agent_cls="DecimaScheduler", embed_dim=16,
             gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=40)
    ap.add_argument("--sequences", type=int, default=32)
    ap.add_argument("--rollouts", type=int, default=4)
    ap.add_argument("--lr", type=float, default=3.0e-4)
    ap.add_argument("--on-env-error", default="truncate", choices=["raise", "truncate"])
    ap.add_argument("--executors", type=int, default=10, help="more than 64: the simulator's wide instantiation, executor-count draws over two counts per lane")
    ap.add_argument("--jobs", type=int, default=30)
    a = ap.parse_args()
    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=a.sequences, num_rollouts=a.rollouts, seed=42,
                 checkpointing_freq=10 ** 9, num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04,
                 beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=a.lr), max_grad_norm=0.5, artifacts_dir="/tmp/sss_demo", on_env_error=a.on_env_error)
    env = dict(num_executors=a.executors, job_arrival_cap=a.jobs, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0,
               mean_time_limit=2.0e7)
    tr = Trainer(AGENT, env, train, device="cuda:0")
    for it in range(a.iterations):
        t0 = time.perf_counter()
        tr.num_iterations = 1
        tr.history.clear()
        try:
            h = tr.train(verbose=False)[0]
        except RuntimeError as e:
            if hasattr(e, "case"):
                os.makedirs("gpurun_out", exist_ok=True)
                with open("gpurun_out/failed_case.json", "w") as fp:
                    json.dump(dict(e.case, env_cfg=tr.env_cfg), fp)
            raise
        torch.cuda.synchronize()
        print(json.dumps({"iteration": it, "avg_num_jobs": round(h["avg_num_jobs"], 4), "samples": h["samples"],
                          "policy_loss": round(h["policy loss"], 5), "entropy": round(h["entropy"], 4),
                          "kl": round(h["approx kl div"], 5), "env_errors": h["env_errors"], "seconds": round(time.perf_counter() - t0, 2)}), flush=True)


if __name__ == "__main__":
    main()
