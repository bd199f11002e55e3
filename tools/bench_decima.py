#!/usr/bin/env python3
"""Decima-in-the-loop throughput (BASELINE config 4): env-steps/s with the GNN policy
sampling every action on the device. Not the headline metric (bench.py is); prints one JSON line.

One GPU: `python tools/bench_decima.py --envs 4096`. Several GPUs (config 4 = 8192 envs on 8 GPUs):
`python tools/bench_decima.py --gpus 8 --envs 1024` starts its own ranks (or run it under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1`) - `--envs` is per
rank, envs are sharded by global id, there is no per-step traffic; the timed region is bracketed by
barriers and the slowest rank's time is used."""
import argparse
import json
import os
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from spark_sched_sim_amd import VecSparkSchedSimEnv  # noqa: E402
from spark_sched_sim_amd.decima import DecimaPolicy  # noqa: E402

AGENT = dict(embed_dim=16,
             gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1, help="ranks to start when not already under torch.distributed.run")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--shards", type=int, default=1, help="split the envs into sub-batches on separate HIP streams (one shard's "
                    "step-kernel tail overlaps the other shards' GNN kernels)")
    ap.add_argument("--one-launch", action="store_true", help="per-env policy kernel (sss_decima_policy) instead of the row-parallel pipeline")
    ap.add_argument("--dist-backend", default="nccl")
    ap.add_argument("--device-index", type=int, default=None)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:  # start the ranks ourselves (nothing has touched the GPU yet)
        from spark_sched_sim_amd.distributed import launch_ranks
        raise SystemExit(launch_ranks(a.gpus, [osp.abspath(__file__)] + sys.argv[1:]))
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dev = f"cuda:{local if a.device_index is None else a.device_index}"
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group(a.dist_backend)
    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    S = max(1, a.shards)
    assert a.envs % S == 0
    envs = [VecSparkSchedSimEnv(cfg, a.envs // S, device=dev, auto_reset=True, seed_stride=a.envs * world) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    torch.manual_seed(0)
    policy = DecimaPolicy(num_executors=10, **AGENT).to(dev).eval()
    gens = [torch.Generator(device=dev).manual_seed(1 + k + 1000 * rank) for k in range(S)]
    for k, e in enumerate(envs):
        e.reset(seed=rank * a.envs + k * (a.envs // S))
    torch.cuda.synchronize()
    for i in range(a.warmup + a.steps):
        if i == a.warmup:
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
        for e, st, gen in zip(envs, streams, gens):
            with torch.cuda.stream(st):
                act, _ = policy.schedule_env(e, generator=gen, one_launch=a.one_launch)
                e.step(act)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    err = sum(int((e.obs_i32[:, 7] != 0).sum()) for e in envs)
    if world > 1:
        e_t = torch.tensor([err], device=dev)
        dist.all_reduce(e_t)
        err = int(e_t)
    if rank == 0:
        print(json.dumps({"metric": "env-steps/s with Decima in the loop", "value": world * a.envs * a.steps / dt, "n_gpus": world, "envs_per_gpu": a.envs,
                          "ms_per_step": 1e3 * dt / a.steps, "shards": S, "one_launch": a.one_launch, "err_envs": err}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
