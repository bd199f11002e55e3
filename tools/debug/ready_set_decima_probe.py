#!/usr/bin/env python3
"""debug / experiment (round-5 verdict item 4a): what ready-set stepping would buy a Decima collection at one rank's share of
BASELINE config 5 (1024 envs, 50 executors, 200 jobs) BEFORE building a recorder for it. Lock-step: every iteration = Decima pass
over all envs + sss_step (ends with its slowest env). Ready-set: sss_step_bounded with an event budget; the Decima pass samples
only for the envs whose step completed (`active` mask), the others continue their step in the next launch. No record is kept in
either mode - this measures the loop's ceiling: completed env-steps per second.

usage: python tools/debug/ready_set_decima_probe.py [--envs 1024] [--budgets 0,16,24,32,48,64] [--iters 3000]"""
import argparse
import json
import os.path as osp
import sys
import time

import torch

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path.insert(0, ROOT)
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload  # noqa: E402
from spark_sched_sim_amd.decima import DecimaPolicy  # noqa: E402

AGENT = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
C5 = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1024)
    ap.add_argument("--budgets", default="0,16,24,32,48,64")
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--warm", type=int, default=1500)
    a = ap.parse_args()
    dev = "cuda:0"
    out = []
    for budget in [int(x) for x in a.budgets.split(",")]:
        env = VecSparkSchedSimEnv(C5, a.envs, device=dev, pack=workload.default_pack(), auto_reset=True)
        torch.manual_seed(0)
        policy = DecimaPolicy(num_executors=50, **AGENT).to(dev).eval()
        gen = torch.Generator(device=dev).manual_seed(1)
        env.reset(seed=0)
        ready = torch.ones(a.envs, dtype=torch.uint8, device=dev)
        keep_si = torch.full((a.envs,), -1, dtype=torch.int32, device=dev)
        keep_ne = torch.ones(a.envs, dtype=torch.int32, device=dev)

        def run(n):
            nonlocal ready
            for _ in range(n):
                if budget:
                    rb = ready.view(torch.bool)
                    act, _ = policy.schedule_env(env, generator=gen, active=rb)
                    # an env in the middle of its step keeps its (ignored) action entries; a ready one gets its new sample
                    torch.where(rb, act["stage_idx"], keep_si, out=keep_si)
                    torch.where(rb, act["num_exec"], keep_ne, out=keep_ne)
                    ready = env.step_bounded_async(keep_si, keep_ne, budget)
                else:
                    act, _ = policy.schedule_env(env, generator=gen)
                    env.step_async(act["stage_idx"], act["num_exec"])
        run(a.warm)
        torch.cuda.synchronize()
        s0 = int(env.header_field("n_steps").sum())
        t0 = time.perf_counter()
        run(a.iters)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s1 = int(env.header_field("n_steps").sum())
        err = int((env.obs_i32[:, 7] != 0).sum())
        rec = {"budget": budget, "envs": a.envs, "iterations": a.iters, "completed_env_steps_per_s": (s1 - s0) / dt, "ms_per_iteration": 1e3 * dt / a.iters,
               "completed_steps_per_iteration_and_env": (s1 - s0) / a.iters / a.envs, "envs_in_error_state": err}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        env.close()
    base = out[0]["completed_env_steps_per_s"] if out and out[0]["budget"] == 0 else None
    if base:
        best = max(out, key=lambda r: r["completed_env_steps_per_s"])
        print(json.dumps({"lock_step": base, "best_budget": best["budget"], "best": best["completed_env_steps_per_s"], "gain": best["completed_env_steps_per_s"] / base}))


if __name__ == "__main__":
    main()
