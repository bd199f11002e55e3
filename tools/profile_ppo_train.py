#!/usr/bin/env python3
"""Where a PPO update's device time goes: collects one batch of synchronous rollouts at a reduced
share of BASELINE config 5, then runs `PPO.train_on_rollouts` under the torch profiler and prints the kernels by
total device time, plus the wall time of the pieces of one minibatch (select / forward / backward / optimiser)."""
import argparse
import json
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from spark_sched_sim_amd.training import Trainer, discounted_returns, ppo_loss, select_observations, sequence_baselines  # noqa: E402
from bench_ppo import AGENT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=32)
    ap.add_argument("--rollouts", type=int, default=4)
    ap.add_argument("--executors", type=int, default=50)
    ap.add_argument("--jobs", type=int, default=200)
    ap.add_argument("--top", type=int, default=25)
    a = ap.parse_args()
    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=a.sequences, num_rollouts=a.rollouts, seed=42,
                 checkpointing_freq=10 ** 9, num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04,
                 beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5, artifacts_dir="/tmp/sss_ppo")
    env = dict(num_executors=a.executors, job_arrival_cap=a.jobs, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=2.0e7)
    dev = "cuda:0"
    tr = Trainer(AGENT, env, train, device=dev)
    tr.policy.eval()
    t0 = time.perf_counter()
    ro = tr.collector.collect_sync(with_stats=False)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    n = int(ro.active.sum())
    print(json.dumps({"samples": n, "graph_nodes": int(ro.graph["x"].shape[0]), "collect_s": t1 - t0}))
    tr.policy.train()
    # pieces of one minibatch
    ppo = tr.ppo
    torch.cuda.synchronize(); p0 = time.perf_counter()
    returns = ppo.diff(ro) if ppo.diff is not None else discounted_returns(ro, ppo.beta)
    torch.cuda.synchronize(); p1 = time.perf_counter()
    baselines = sequence_baselines(ro, returns, ppo.num_sequences, ppo.num_rollouts)
    torch.cuda.synchronize(); p2 = time.perf_counter()
    print(json.dumps({"returns_s": p1 - p0, "baselines_s": p2 - p1, "rows": int(ro.active.shape[0])}))
    ids = ro.sample_ids()
    advgs = ro.flat(returns - baselines)
    acts = [ro.flat(ro.stage_sel), ro.flat(ro.job_idx), ro.flat(ro.exec_sel)]
    old_lg = ro.flat(ro.lgprobs)
    mb = torch.randperm(ids.numel(), device=ids.device)[: ids.numel() // 10 + 1]
    for rep in range(2):
        torch.cuda.synchronize(); a0 = time.perf_counter()
        g = select_observations(ro.graph, ids[mb])
        torch.cuda.synchronize(); a1 = time.perf_counter()
        loss, info = ppo_loss(tr.policy, g, acts[0][mb], acts[1][mb], acts[2][mb], advgs[mb], old_lg[mb], 0.2, 0.04)
        torch.cuda.synchronize(); a2 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); a3 = time.perf_counter()
        tr.policy.update_parameters(None)
        torch.cuda.synchronize(); a4 = time.perf_counter()
    jg = torch.cumsum(g["obs_jobs"], 0) - g["obs_jobs"] + acts[1][mb]
    print(json.dumps({"minibatch_edges": int(g["src"].numel()), "minibatch_jobs": int(g["job_obs"].numel()), "schedulable": int(g["stage_mask"].sum()),
                      "mean_allowed_executor_counts": float(g["job_cap"][jg].float().mean()), "layers": int(g["obs_depth"].max()),
                      "receivers": int(sum(int(r.numel()) for _, r in g.get("layers", [])))}))
    print(json.dumps({"minibatch_obs": int(mb.numel()), "minibatch_nodes": int(g["x"].shape[0]), "select_s": a1 - a0, "forward_loss_s": a2 - a1, "backward_s": a3 - a2, "optim_s": a4 - a3}))
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        t2 = time.perf_counter()
        ppo.train_on_rollouts(ro)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
    print(json.dumps({"train_s_profiled": t3 - t2}))
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=a.top, max_name_column_width=70))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=a.top, max_name_column_width=70))


if __name__ == "__main__":
    main()
