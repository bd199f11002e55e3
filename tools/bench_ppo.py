#!/usr/bin/env python3
"""One PPO iteration at scale (SURVEY 8f next-3 / BASELINE config 5 shape, one GPU's share):
`num_sequences x num_rollouts` envs collect synchronous rollouts with Decima sampling every action
on the device, then the PPO epochs run on the recorded compact graph. Prints one JSON line."""
import argparse
import json
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
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start when not already under torch.distributed.run")
    ap.add_argument("--sequences", type=int, default=256)
    ap.add_argument("--rollouts", type=int, default=4)
    ap.add_argument("--iterations", type=int, default=2)
    ap.add_argument("--executors", type=int, default=50)
    ap.add_argument("--jobs", type=int, default=200)
    ap.add_argument("--mean-time-limit", type=float, default=2.0e7)
    ap.add_argument("--rollout-duration", type=float, default=0.0)
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--device-index", type=int, default=None)
    ap.add_argument("--collector-groups", type=int, default=1)
    ap.add_argument("--no-train", action="store_true", help="collections only (for a rocprofv3 kernel trace of the collection alone: tools/profile_ppo_rocprof.sh)")
    ap.add_argument("--kernel-switch", action="append", default=[], metavar="NAME=0|1",
                    help="set a module-level switch of spark_sched_sim_amd.train_kernels (FUSED_HEAD_WGRAD, CONCAT_ONE_LAUNCH, SPLIT_INPUT, INDEXED_ROWS, ...): A/B timing")
    a = ap.parse_args()
    import os
    for sw in a.kernel_switch:
        from spark_sched_sim_amd import train_kernels
        name, val = sw.split("=")
        assert hasattr(train_kernels, name), name
        setattr(train_kernels, name, bool(int(val)))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:  # BASELINE config 5: `python tools/bench_ppo.py --gpus 8 --sequences 256 --rollouts 4`
        from spark_sched_sim_amd.distributed import launch_ranks
        raise SystemExit(launch_ranks(a.gpus, [osp.abspath(__file__)] + sys.argv[1:]))
    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=a.sequences, num_rollouts=a.rollouts, seed=42,
                 checkpointing_freq=10 ** 9, num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04,
                 beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5, artifacts_dir="/tmp/sss_ppo",
                 collector_groups=a.collector_groups)
    if a.rollout_duration:
        train["rollout_duration"] = a.rollout_duration
    env = dict(num_executors=a.executors, job_arrival_cap=a.jobs, job_arrival_rate=4.0e-5, moving_delay=2000.0,
               warmup_delay=1000.0, mean_time_limit=a.mean_time_limit)  # config/decima_tpch.yaml:80-86
    import os
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dev = f"cuda:{local if a.device_index is None else a.device_index}"
    torch.cuda.set_device(dev)
    if world > 1:  # BASELINE config 5: `torch.distributed.run --nproc-per-node 8 tools/bench_ppo.py --sequences 256 --rollouts 4`
        import torch.distributed as dist
        dist.init_process_group(a.dist_backend)
    tr = Trainer(AGENT, env, train, device=dev)
    out = []
    for it in range(a.iterations):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.policy.eval()
        ro = tr.collector.collect_async(a.rollout_duration, with_stats=False) if a.rollout_duration else tr.collector.collect_sync(with_stats=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr.policy.train()
        learn = {} if a.no_train else tr.ppo.train_on_rollouts(ro)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n = int(ro.active.sum())
        out.append({"iteration": it, "n_gpus": world, "envs_per_gpu": a.sequences * a.rollouts, "envs": a.sequences * a.rollouts, "samples": n, "longest_rollout": int(ro.active.shape[0]),
                    "collect_s": t1 - t0, "train_s": t2 - t1, "collect_env_steps_per_s": n / (t1 - t0),
                    "graph_nodes": int(ro.graph["x"].shape[0]), **learn})
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
