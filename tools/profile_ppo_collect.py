#!/usr/bin/env python3
"""Where a collection iteration's time goes: `RolloutCollector` at one rank's share of BASELINE config 5
(1024 envs, 50 executors, 200 jobs), a bounded asynchronous collection; prints wall time per loop iteration, device-busy time per
iteration (torch profiler) and the top device kernels / host ops."""
import argparse
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from bench_ppo import AGENT  # noqa: E402
from spark_sched_sim_amd.training import Trainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=256)
    ap.add_argument("--rollouts", type=int, default=4)
    ap.add_argument("--duration", type=float, default=1.5e6, help="simulated ms per env and collection")
    ap.add_argument("--rows", type=int, default=25)
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--sync", action="store_true", help="synchronous rollouts (one episode per env: the reference's decima_tpch.yaml) instead")
    a = ap.parse_args()
    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=a.sequences, num_rollouts=a.rollouts, seed=42, checkpointing_freq=10 ** 9, num_epochs=3,
                 num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04, beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4),
                 max_grad_norm=0.5, artifacts_dir="/tmp/sss_ppo", collector_groups=a.groups, **({} if a.sync else {"rollout_duration": a.duration}))
    env = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=2.0e7)
    tr = Trainer(AGENT, env, train, device="cuda:0")
    tr.policy.eval()
    col = tr.collector
    run = (lambda: col.collect_sync(with_stats=False)) if a.sync else (lambda: col.collect_async(a.duration, with_stats=False))
    run()  # warm-up (async: also moves every env into its episode)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        t0 = time.perf_counter()
        ro = run()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
    iters, n = int(ro.active.shape[0]), int(ro.active.sum())
    ev = prof.key_averages()
    dev_us = sum(e.self_device_time_total for e in ev)
    print(f"iterations {iters}, samples {n}, wall {t1 - t0:.3f} s = {1e3 * (t1 - t0) / iters:.3f} ms / iteration, device busy {dev_us / iters:.1f} us / iteration, "
          f"{n / (t1 - t0) / 1e6:.3f} M env-steps/s")
    print(ev.table(sort_by="self_device_time_total", row_limit=a.rows, max_name_column_width=70))
    print(ev.table(sort_by="self_cpu_time_total", row_limit=a.rows, max_name_column_width=70))


if __name__ == "__main__":
    main()
