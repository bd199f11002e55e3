#!/usr/bin/env python3
"""The reference's one published result, re-run with this stack: "Decima attains a lower average job completion time than the fair
scheduler" at 50 jobs / 10 executors (reference README.md:5-7; the harness is examples.py:84-102 with metrics.avg_job_duration,
spark_sched_sim/metrics.py:12-14).

1. train a Decima policy with THIS repository's PPO (spark_sched_sim_amd/training.py; hyper-parameters of config/decima_tpch.yaml)
   on the batched env at the README's sizing, on the GPU, from random initial weights;
2. every `--eval-every` iterations and at the end: whole episodes (no time limit) of `--eval-envs` HELD-OUT seeds under
   (a) the policy with sampled actions (what the reference's DecimaScheduler.schedule does), (b) the same policy with arg-max actions,
   (c) the on-device fair policy (RoundRobinScheduler, dynamic partitioning) and (d) FIFO - the same seeds for all four, hence the
   same job sequences; per env the mean over its jobs of (t_completed - t_arrival), then mean and 95 % confidence interval over envs
   plus the paired difference to fair.

Writes one JSON record (--out) with the training curve, the evaluations and the wall-clock; synthetic trace set (the real TPC-H
traces cannot be fetched here), so the absolute numbers are this workload's, the comparison is the claim."""
import argparse
import json
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.abspath(__file__))))
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload  # noqa: E402
from spark_sched_sim_amd.training import SKIP_ENV, Trainer  # noqa: E402

AGENT = dict(agent_cls="DecimaScheduler", embed_dim=16,
             gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))


def episodes_under_decima(env, policy, seed0: int, greedy: bool, gen) -> dict:
    """every env plays one whole episode under the policy; finished (or failed) envs sit the remaining launches out"""
    B, dev = env.num_envs, env.device
    env.reset(seed=seed0)
    done = torch.zeros(B, dtype=torch.bool, device=dev)
    skip = torch.full((B,), SKIP_ENV, dtype=torch.int32, device=dev)
    steps = 0
    while True:
        for _ in range(64):
            act, _ = policy.schedule_env(env, generator=gen, active=~done, greedy=greedy)
            env.step_async(torch.where(done, skip, act["stage_idx"]).contiguous(), act["num_exec"])
            done = done | (env.obs_i32[:, 6] != 0) | (env.obs_i32[:, 7] != 0)
            steps += 1
        if bool(done.all()) or steps > 200_000:
            break
    return summarize(env)


def episodes_under_heuristic(env, name: str, seed0: int) -> dict:
    env.reset(seed=seed0)
    for _ in range(2000):
        env.rollout(name, 200)
        if bool(((env.header_field("terminated") != 0) | (env.obs_i32[:, 7] != 0)).all()):
            break
    return summarize(env)


def summarize(env) -> dict:
    st = env.rollout_stats()
    ok = (env.header_field("terminated") != 0) & (env.obs_i32[:, 7] == 0)
    return {"avg_job_duration_s": st["avg_job_duration"].clone(), "ok": ok.clone(), "avg_num_jobs": st["avg_num_jobs"].clone(),
            "steps": env.header_field("ep_steps").clone()}


def compare(results: dict) -> dict:
    """mean / CI per policy over the envs where EVERY policy finished its episode without error (a sampled action sequence can run
    into the reference's own "[step]" stall), and the paired difference to fair"""
    ok = None
    for r in results.values():
        ok = r["ok"] if ok is None else ok & r["ok"]
    n = int(ok.sum())
    out = {"envs_compared": n, "envs_excluded": int((~ok).sum())}
    fair = results["fair"]["avg_job_duration_s"][ok]
    for name, r in results.items():
        v = r["avg_job_duration_s"][ok]
        d = v - fair
        out[name] = {"avg_job_duration_s": float(v.mean()), "ci95": float(1.96 * v.std() / n ** 0.5), "avg_num_jobs": float(r["avg_num_jobs"][ok].mean()),
                     "steps_per_episode": float(r["steps"][ok].double().mean()),
                     "minus_fair_s": float(d.mean()), "minus_fair_ci95": float(1.96 * d.std() / n ** 0.5),
                     "envs_better_than_fair": float((d < 0).double().mean())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iterations", type=int, default=300)
    ap.add_argument("--sequences", type=int, default=64)
    ap.add_argument("--rollouts", type=int, default=4)
    ap.add_argument("--lr", type=float, default=3.0e-4)
    ap.add_argument("--executors", type=int, default=10)
    ap.add_argument("--jobs", type=int, default=50)
    ap.add_argument("--eval-envs", type=int, default=4096)
    ap.add_argument("--eval-every", type=int, default=50)
    ap.add_argument("--eval-seed", type=int, default=10_000_000)
    ap.add_argument("--entropy", type=float, default=0.04)
    ap.add_argument("--pack", default="default", choices=list(workload.PROFILES))
    ap.add_argument("--out", default="gpurun_out/decima_vs_fair.json")
    ap.add_argument("--emu", action="store_true", help="plumbing check without a GPU: the CPU wave-emulator build of the kernels (tests/emu), tiny sizes")
    a = ap.parse_args()
    dev, lib = "cuda:0", None
    if a.emu:
        sys.path.insert(0, osp.join(osp.dirname(osp.dirname(osp.abspath(__file__))), "tests"))
        from emu_util import load_emu
        dev, lib = "cpu", load_emu()
    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=a.sequences, num_rollouts=a.rollouts, seed=42, checkpointing_freq=10 ** 9, num_epochs=3,
                 num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=a.entropy, beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=a.lr),
                 max_grad_norm=0.5, artifacts_dir="/tmp/sss_dvf", on_env_error="truncate")
    env_cfg = dict(num_executors=a.executors, job_arrival_cap=a.jobs, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=2.0e7)
    pack = workload.profile_pack(a.pack)
    tr = Trainer(AGENT, env_cfg, train, device=dev, pack=pack, _lib=lib)
    eval_cfg = {k: v for k, v in env_cfg.items() if k != "mean_time_limit"}
    eval_env = VecSparkSchedSimEnv(eval_cfg, a.eval_envs, device=dev, pack=pack, _lib=lib)
    gen = torch.Generator(device=dev).manual_seed(7)

    heur = {name: episodes_under_heuristic(eval_env, name, a.eval_seed) for name in ("fair", "fifo")}
    rec = {"what": "Decima trained by this repository's PPO vs the fair scheduler: average job completion time over held-out job sequences (reference README.md:5-7)",
           "env": eval_cfg, "pack": a.pack, "train": {k: v for k, v in train.items() if k != "artifacts_dir"}, "eval_envs": a.eval_envs, "eval_seed": a.eval_seed,
           "curve": [], "evals": []}

    def evaluate(tag):
        tr.policy.eval()
        t0 = time.perf_counter()
        res = dict(heur)
        res["decima_sampled"] = episodes_under_decima(eval_env, tr.policy, a.eval_seed, False, gen)
        res["decima_greedy"] = episodes_under_decima(eval_env, tr.policy, a.eval_seed, True, gen)
        c = compare(res)
        c["after_iterations"], c["eval_seconds"] = tag, round(time.perf_counter() - t0, 1)
        rec["evals"].append(c)
        print(json.dumps(c), flush=True)
        tr.policy.train()

    evaluate(0)
    t_train = 0.0
    for it in range(a.iterations):
        t0 = time.perf_counter()
        tr.num_iterations = 1
        tr.history.clear()
        h = tr.train(verbose=False)[0]
        if dev != "cpu":
            torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t_train += dt
        row = {"iteration": it, "avg_num_jobs": round(h["avg_num_jobs"], 4), "samples": h["samples"], "entropy": round(h["entropy"], 4),
               "kl": round(h["approx kl div"], 5), "env_errors": h["env_errors"], "seconds": round(dt, 2)}
        rec["curve"].append(row)
        if it % 10 == 0:
            print(json.dumps(row), flush=True)
        if (it + 1) % a.eval_every == 0 or it + 1 == a.iterations:
            evaluate(it + 1)
            rec["train_seconds"] = round(t_train, 1)
            with open(a.out, "w") as fp:
                json.dump(rec, fp)
    last = rec["evals"][-1]
    best = min(rec["evals"], key=lambda c: c["decima_sampled"]["avg_job_duration_s"])
    rec["verdict"] = {"decima_sampled_beats_fair_at_end": last["decima_sampled"]["minus_fair_s"] + last["decima_sampled"]["minus_fair_ci95"] < 0,
                      "decima_greedy_beats_fair_at_end": last["decima_greedy"]["minus_fair_s"] + last["decima_greedy"]["minus_fair_ci95"] < 0,
                      "best_eval_after_iterations": best["after_iterations"]}
    with open(a.out, "w") as fp:
        json.dump(rec, fp)
    print(json.dumps(rec["verdict"]), flush=True)
    tr.close()
    eval_env.close()


if __name__ == "__main__":
    main()
