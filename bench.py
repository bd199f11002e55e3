#!/usr/bin/env python3
"""Throughput benchmark of the hot path: env-steps/s of the batched Spark-scheduling simulator.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs B] [--config c2|c3] [--policy hash|fair]
                    [--mode step|fused] [--no-cpu-baseline] [--no-c3] [--no-decima] [--sustained-s S]

A "step" is ONE batched step of all envs of a rank: on-device policy kernel + step kernel
(`--mode step`, the drop-in boundary: sss_policy + sss_step per step), or one iteration of the
fused rollout kernel (`--mode fused`, sss_rollout: identical per-step work, policy -> step ->
observe, without leaving the kernel). Envs auto-reset (next-step mode); only real step() calls are
counted (the device counts them), so `value` = real env steps of all ranks / max-over-ranks time.

Before anything is timed every env is rolled to its steady state with fused launches (episodes of
different seeds have different lengths, so after a few episodes the envs of a batch sit at all
phases of their episodes); `--warmup W` untimed steps in the measured mode follow.

Multi-GPU (`--gpus N`): with no torch.distributed environment bench.py starts its own N ranks
(fresh processes, one per GPU, 127.0.0.1 rendezvous); under `torch.distributed.run` it is one of
them. Envs are sharded, B per rank, no data-path collective; one RCCL all-gather of per-env episode
returns after the timed region (the stand-in for the reference's Pipe gather,
trainers/trainer.py:113-121). Weak scaling.

Prints ONE JSON line on rank 0 (see the driver contract) including `roofline` (SURVEY 8(d) model
bytes of the dominant kernel / its HIP-event-measured duration / 8 TB/s), `cpu_baseline` (the C
oracle, oracle/sss_oracle.c, timed on this box's host on a bounded sample), `step_tail` (how much
of a step launch is the wait for its slowest env), `sustained` (the same mode over about a second
of stepping when the K timed steps are shorter than that) and, at N = 1, a `c3` record with the same
measurements on BASELINE config 3 (4096 envs, 50 executors, 200 jobs, fair policy).
"""
from __future__ import annotations

import argparse
import json
import os
import os.path as osp
import sys
import time

ROOT = osp.dirname(osp.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # C2 sizing = reference examples.py:15-23 (10 executors, 50 jobs); C3 = config/decima_tpch.yaml:81-85
    "c2": dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0),
    "c3": dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0),
    # not a BASELINE config: config 3's sizing at the top of the reference's executor-level table (tpch.py:238) - more than
    # 64 executors run on the wide instantiation of the kernels (csrc/sss_hip_wide.hip: two executors per lane)
    "e100": dict(num_executors=100, job_arrival_cap=200, job_arrival_rate=8.0e-5, moving_delay=2000.0, warmup_delay=1000.0),
}
DEFAULT_POLICY = {"c2": "hash", "c3": "fair", "e100": "fair"}
PREROLL_STEPS = {"c2": 1500, "c3": 6000, "e100": 6000}  # a few episodes each (about 600 / 4000 steps long)
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(cfg: dict, policy: str, budget_s: float, pack_profile: str = "default") -> dict:
    """the C oracle (a port of the reference env, bit-identical trajectories) on ONE host core,
    same workload, whole episodes until ~budget_s of CPU time is spent"""
    import ctypes as C

    sys.path.insert(0, osp.join(ROOT, "tests"))
    from oracle_binding import OracleEnv
    from spark_sched_sim_amd import workload

    env = OracleEnv(workload.profile_pack(pack_profile), cfg)
    pol = {"fair": 0, "hash": 1}[policy]
    steps, eps = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        r = C.c_double()
        n = env.lib.sso_run_episode(env.h, 10_000 + eps, pol, 10**9, C.byref(r))
        assert n > 0, n
        steps += n
        eps += 1
    dt = time.perf_counter() - t0
    env.close()
    return {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{eps} episodes / {steps} steps of the same config and policy in {dt:.1f} s (C oracle, observation built every step)"}


_CPU_WORKER = """
import ctypes as C, json, sys, time
sys.path[:0] = [{root!r}, {tests!r}]
from oracle_binding import OracleEnv
from spark_sched_sim_amd import workload
cfg, pol, budget, wid = json.loads(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
env = OracleEnv(workload.profile_pack(sys.argv[5]), cfg)
steps = eps = 0
t0 = time.perf_counter()   # (the pack is built before the clock starts)
while time.perf_counter() - t0 < budget:
    r = C.c_double()
    steps += max(0, env.lib.sso_run_episode(env.h, 20000 + 1000 * wid + eps, pol, 10**9, C.byref(r)))
    eps += 1
print(json.dumps([steps, time.perf_counter() - t0]))
"""


def reference_python_baseline(config: str, profile: str = "default"):
    """the reference Python env itself, timed in the BUILD CONTAINER by tools/time_reference.py (it cannot travel to the GPU box:
    /root/reference does not exist there) - the committed record profiles/reference_python.json, quoted with its hardware"""
    path = osp.join(ROOT, "profiles", "reference_python.json")
    try:
        rec = json.load(open(path))
        if profile != "default":
            rec = rec[profile]  # (the same measurement on another synthetic trace regime: tools/time_reference.py --deep)
        c = rec["configs"][{"c2": "c1", "c3": "c3"}[config]]  # (KeyError for other configs: no record)
        return {"value": c["one_core"]["env_only"], "unit": "env-steps/s", "cores": 1, "kind": "reference",
                "all_cores": {"value": c["all_cores"]["env_only"], "processes": c["all_cores"]["processes"]},
                "env_plus_scheduler_incl_reset": c["one_core"]["including_reset"],
                "sample": f"{c['episodes']} episodes / {c['steps']} steps, fair scheduler, time.perf_counter around env.step only (examples.py:84-102 loop)",
                "hardware": rec["hardware"], "measured_in_this_run": False,
                "source": "profiles/reference_python.json (tools/time_reference.py; same executors / jobs / arrival rate as this record's config, 1 env)"}
    except Exception:
        return None


def usable_cores() -> int:
    """host cores this process may actually use: affinity mask and cgroup CPU quota included"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline_all_cores(cfg: dict, policy: str, budget_s: float, pack_profile: str = "default") -> dict:
    """the reference's own parallelism model (one process per env, trainers/trainer.py:264-293)
    with the C oracle: one child process per host core (plain subprocesses that never touch the GPU)"""
    import subprocess

    n = usable_cores()
    code = _CPU_WORKER.format(root=ROOT, tests=osp.join(ROOT, "tests"))
    pol = {"fair": 0, "hash": 1}[policy]
    procs = [subprocess.Popen([sys.executable, "-c", code, json.dumps(cfg), str(pol), str(budget_s), str(w), pack_profile],
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for w in range(n)]
    res = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=budget_s + 120)
            res.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    steps = sum(r[0] for r in res)
    dt = max(r[1] for r in res)
    return {"value": steps / dt, "unit": "env-steps/s", "cores": len(res), "kind": "port",
            "sample": f"{len(res)} processes x ~{budget_s:.0f} s of whole episodes, same config and policy (C oracle)"}


def measured_traffic(kernel: str, config: str, envs: int, events_per_step: float, key: str = "hi"):
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes), recorded under profiles/ by tools/collect_traffic.py for this
    exact kernel/config - and this regime: the record carries the events per step it was measured
    at, and it is only reported next to a run within 15 % of that. None otherwise."""
    path = osp.join(ROOT, "profiles", "traffic.json")
    if not osp.exists(path):
        return None
    try:
        for rec in json.load(open(path)):
            if rec["kernel"] == kernel and rec["config"] == config and rec["envs"] == envs:
                ref = rec.get("events_per_step")
                if ref and abs(events_per_step - ref) <= 0.15 * ref:
                    return rec["hbm_bytes_per_launch"] if key == "hi" else rec.get("hbm_bytes_per_launch_lo")
    except Exception:
        return None
    return None


FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense (the pass computes in exact fp32)


def binding_roofline(flops: float, hbm_bytes: float, seconds: float) -> dict:
    """the roofline record of a pass that is priced both ways: algorithmic flops against the dense fp32 MFMA peak and algorithmic
    HBM bytes against 8 TB/s - `bound` / `achieved` / `peak` / `frac` are those of the LARGER fraction (the roofline that binds),
    the other one rides along"""
    tf, gbs = flops / seconds / 1e12, hbm_bytes / seconds / 1e9
    m = {"bound": "mfma", "achieved": tf, "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / FP32_MFMA_PEAK_TFLOPS}
    h = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
    first, second = (m, h) if m["frac"] >= h["frac"] else (h, m)
    return dict(first, other_roofline=second)


def reference_decima_baseline(key: str):
    """the reference's own Decima code timed in the BUILD CONTAINER (tools/time_reference_decima.py; it cannot travel to the GPU box):
    the committed record profiles/reference_python.json, quoted with its hardware - never something timed in this run"""
    try:
        rec = json.load(open(osp.join(ROOT, "profiles", "reference_python.json")))[key]
    except Exception:
        return None
    return rec


def decima_in_loop(cfg: dict, B: int, dev, pack, steps: int = 100, warmup: int = 20, steady_after: int = 600, steady_steps: int = 1200) -> dict:
    """extra, not the headline: the same B envs with a sampled Decima action (GNN policy, random-init
    weights of the published architecture) for every env on every step - graph kernel, GNN kernels,
    sampling kernels, sss_step (spark_sched_sim_amd/decima.py). Two windows: steps 20..120 after the reset (the window
    the earlier rounds quote: every env early in its first episode) and `steady`: 1200 steps after 600 - two episode
    lengths under this policy; the envs stay roughly in phase, so the cost of a step swings with the phase of the episode
    (0.63 .. 0.89 ms, tools/debug/decima_windows.py) and only an average over episodes is a stable figure.

    `roofline`: the Decima pass (everything between two sss_step launches) priced against the dense fp32 MFMA peak - its HBM
    bytes are two orders of magnitude below what 8 TB/s moves in the same time, so the matrix cores are the binding roofline
    (DESIGN.md section 6): algorithmic flops of a pass (decima.algorithmic_cost: MLP rows x 2 * weights) / the pass's duration,
    HIP events around every 8th pass of the steady window. `cpu_baseline`: the reference's DecimaScheduler.schedule + env.step."""
    import torch

    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy, algorithmic_cost

    agent = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)),
                 policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
    env = VecSparkSchedSimEnv(cfg, B, device=dev, pack=pack, auto_reset=True)
    torch.manual_seed(0)
    policy = DecimaPolicy(num_executors=cfg["num_executors"], **agent).to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(1)
    env.reset(seed=0)

    def run(n: int, events=None) -> float:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            if events is not None and i % 8 == 0:
                e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
                e0.record()
                act, _ = policy.schedule_env(env, generator=gen)
                e1.record()
                env.step_async(act["stage_idx"], act["num_exec"])
                e2.record()
                events.append((e0, e1, e2))
            else:
                act, _ = policy.schedule_env(env, generator=gen)
                env.step_async(act["stage_idx"], act["num_exec"])  # (observations, rewards, flags: the env's buffers, read in place by the next pass)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    run(warmup)
    dt = run(steps)
    run(max(0, steady_after - warmup - steps))
    nodes = int(env.obs_i32[:, 0].sum())
    events: list = []
    dts = run(steady_steps, events)
    pass_ms = sum(a.elapsed_time(b) for a, b, _ in events) / len(events)
    step_ms = sum(b.elapsed_time(c) for _, b, c in events) / len(events)
    # the work of a pass on the observations the window ended on: rows of each MLP from the graph kernel's own output
    act, aux = policy.schedule_env(env, generator=gen, fresh_outputs=True)
    g = env.decima_graph()
    gid = (g["obs_job_off"] + aux["job_idx"]).clamp(0, max(0, g["job_cap"].numel() - 1))
    exec_rows = int((g["job_cap"][gid].clamp(0, cfg["num_executors"]) * aux["any_stage"]).sum())
    cost = algorithmic_cost(g, exec_rows)
    achieved = cost["flops"] / (pass_ms * 1e-3) / 1e12
    err = int((env.obs_i32[:, 7] != 0).sum())
    env.close()
    ref = reference_decima_baseline("decima_c1")
    out = {"value": B * steps / dt, "unit": "env-steps/s", "envs": B, "ms_per_step": 1e3 * dt / steps, "steps": steps, "envs_in_error_state": err,
           "steady": {"value": B * steady_steps / dts, "ms_per_step": 1e3 * dts / steady_steps, "steps": steady_steps, "after_steps": steady_after,
                      "active_nodes_per_env": nodes / B, "decima_pass_ms": pass_ms, "sss_step_ms": step_ms},
           "roofline": dict(binding_roofline(cost["flops"], cost["bytes_inference"], pass_ms * 1e-3),
                            kernel="the Decima pass: sss_decima_graph_kernel + sss_gnn_*_mfma kernels + sss_decima_sample kernels (dominant: sss_gnn_layer_mfma_kernel, one launch per DAG layer)",
                            dtype="f32", flops_per_pass=cost["flops"], mlp_rows_per_pass=cost["rows"], avg_pass_ms=pass_ms, passes_with_events=len(events),
                            hbm_bytes_per_pass_algorithmic=cost["bytes_inference"], traffic=None),
           "what": "every env gets a sampled Decima action every step (graph + GNN + sampling kernels, then sss_step)"}
    if ref is not None:
        out["cpu_baseline"] = {"value": ref["env_steps_per_s"], "unit": "env-steps/s", "cores": 1, "kind": "reference", "sample": f"{ref['episodes']} episodes / {ref['steps']} steps: {ref['what']}",
                               "schedule_share_of_time": ref["schedule_share_of_time"], "hardware": ref["hardware"], "measured_in_this_run": False,
                               "source": "profiles/reference_python.json decima_c1 (tools/time_reference_decima.py)"}
    return out


def ppo_config5_share(dev, sequences: int = 256, rollouts: int = 4) -> dict:
    """extra, not the headline: one GPU's share of BASELINE config 5 (reference trainers/trainer.py:85-162, trainers/ppo.py:51-138 with
    config/decima_tpch.yaml's hyper-parameters) - 256 job sequences x 4 rollouts = 1024 envs, 50 executors, 200 jobs: every env
    runs a whole episode under sampled Decima actions (synchronous collection, the record built on the device), then the PPO
    epochs run on the recorded compact graph. One warm-up iteration, one timed (the shape of tools/bench_ppo.py)."""
    import torch

    from spark_sched_sim_amd.training import Trainer

    agent = dict(agent_cls="DecimaScheduler", embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)),
                 policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=sequences, num_rollouts=rollouts, seed=42, checkpointing_freq=10 ** 9, num_epochs=3,
                 num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04, beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4),
                 max_grad_norm=0.5, artifacts_dir="/tmp/sss_ppo_bench")
    env = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=2.0e7)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.cuda.reset_peak_memory_stats(dev)
    base_alloc = torch.cuda.memory_allocated(dev)  # (what the earlier legs of this bench run still hold: not this record's)
    tr = Trainer(agent, env, train, device=str(dev))
    rec = None
    for it in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tr.policy.eval()
        ro = tr.collector.collect_sync(with_stats=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tr.policy.train()
        learn = tr.ppo.train_on_rollouts(ro)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n = int(ro.active.sum())
        rec = {"envs": sequences * rollouts, "samples": n, "longest_rollout": int(ro.active.shape[0]), "collect_s": t1 - t0, "train_s": t2 - t1,
               "iteration_s": t2 - t0, "collect_env_steps_per_s": n / (t1 - t0), "graph_nodes": int(ro.graph["x"].shape[0]),
               **{k: (float(v) if isinstance(v, (int, float)) else v) for k, v in learn.items()}}
        if it == 1:
            # the algorithmic work of the iteration from its own record (decima.algorithmic_cost): one forward pass over every recorded
            # observation = what the collection's policy passes computed; an optimiser step over a minibatch is forward + backward
            # (input and weight gradients) = 3 x its forward flops, `minibatches` of `num_batches` per epoch were taken
            from spark_sched_sim_amd.decima import algorithmic_cost
            g = ro.graph
            job_off = torch.cumsum(g["obs_jobs"], 0) - g["obs_jobs"]
            gid = (job_off[ro.sample_ids()] + ro.flat(ro.job_idx)).clamp(0, max(0, g["job_cap"].numel() - 1))
            cost = algorithmic_cost(g, int(g["job_cap"][gid].clamp(0, env["num_executors"]).sum()))
            mbs = int(learn.get("minibatches", train["num_epochs"] * train["num_batches"]))
            train_flops = 3.0 * cost["flops"] * mbs / train["num_batches"]
            train_bytes = cost["bytes_training"] * mbs / train["num_batches"]
            rec["train_roofline"] = dict(binding_roofline(train_flops, train_bytes, t2 - t1),
                                         kernel="the PPO update: sss_mlp_mfma_{fwd,bwdw} / sss_mlp_head_mfma_* kernels + row gather / segment-sum kernels + autograd glue",
                                         dtype="f32", flops=train_flops, flops_forward_whole_record=cost["flops"], minibatches=mbs, mlp_rows_whole_record=cost["rows"],
                                         hbm_bytes_algorithmic=train_bytes, hbm_bytes_model="every MLP's activations stored once and read once, gradients of the same size: 12 B x (in + h1 + h2 + out) per row "
                                         "(the GNN-shaped MLPs recompute theirs since round 6 and move less)", traffic=None,
                                         samples_x_epochs_per_s=n * mbs / train["num_batches"] / (t2 - t1))
            rec["collect_roofline"] = {"bound": "mfma", "kernel": "the collection's Decima passes (one per step over the active envs)", "achieved": cost["flops"] / (t1 - t0) / 1e12,
                                       "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": cost["flops"] / (t1 - t0) / 1e12 / FP32_MFMA_PEAK_TFLOPS, "flops": cost["flops"],
                                       "note": "the collection is latency-bound: ~7500 dependent steps of graph + GNN + sampling + sss_step launches over <= 1024 envs (DESIGN.md section 9.2)"}
            ref = reference_decima_baseline("ppo_config5")
            if ref is not None:
                rec["cpu_baseline"] = {"collect": {"value": ref["collect_env_steps_per_s_one_core"], "unit": "env-steps/s", "cores": 1},
                                       "train": {"value": ref["train_samples_per_s"], "unit": "samples x epochs / s", "cores": ref["train_threads"]},
                                       "kind": "reference", "sample": f"{ref['rollouts']} rollouts / {ref['samples']} samples: {ref['what']}", "hardware": ref["hardware"],
                                       "measured_in_this_run": False, "source": "profiles/reference_python.json ppo_config5 (tools/time_reference_decima.py)"}
        del ro
    rec["peak_memory_allocated_gb"] = torch.cuda.max_memory_allocated(dev) / 1e9
    rec["peak_memory_reserved_gb"] = torch.cuda.max_memory_reserved(dev) / 1e9
    rec["allocated_before_this_record_gb"] = base_alloc / 1e9
    rec["peak_memory_allocated_by_this_record_gb"] = (torch.cuda.max_memory_allocated(dev) - base_alloc) / 1e9
    rec["memory_note"] = ("peaks over both iterations incl. this record's own accounting pass over the whole record (algorithmic_cost); per phase "
                          "(tools/debug/ppo_memory_phases.py, a process of its own): ~31 GB while collecting, ~50 GB while updating")
    rec["what"] = ("one rank's share of BASELINE config 5 (PPO, decima_tpch.yaml): whole episodes of 1024 envs under sampled Decima actions, then 3 epochs x 10 "
                   "minibatches; second of two iterations")
    tr.close()
    torch.cuda.empty_cache()
    return rec


class Bench:
    """one rank's share of one configuration: B envs (optionally in sub-batches on their own streams)"""

    def __init__(self, args, config: str, policy: str, B: int, dev, rank: int, world: int, pack_profile: str | None = None):
        import torch

        from spark_sched_sim_amd import VecSparkSchedSimEnv, workload

        self.torch, self.args, self.config, self.policy, self.B, self.dev, self.rank, self.world = torch, args, config, policy, B, dev, rank, world
        self.cfg = CONFIGS[config]
        S = max(1, args.shards)
        assert B % S == 0, "--envs must be divisible by --shards"
        Bs = B // S
        self.pack_profile = pack_profile or getattr(args, "pack", "default")
        self.pack = workload.profile_pack(self.pack_profile)
        # env i of shard s of rank r: seed (r*S + s)*Bs + i = its global env id (placement invariant)
        lib = None
        if getattr(args, "lib", None):  # an A/B test build of the library (tests/gpu_variant.py), never the default
            from spark_sched_sim_amd.binding import load_library
            lib = load_library(osp.abspath(args.lib))
        self.shards = [VecSparkSchedSimEnv(self.cfg, Bs, device=dev, pack=self.pack, auto_reset=True, seed_stride=B * world, _lib=lib) for _ in range(S)]
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
        for k, e in enumerate(self.shards):
            e.reset(seed=(rank * S + k) * Bs)
        # steady state: every env a few episodes in, all phases of an episode present in the batch
        left = (PREROLL_STEPS[config] * (2 if self.pack_profile == "deep" else 1)) if args.preroll is None else args.preroll
        while left > 0:
            n = min(500, left)
            for e in self.shards:
                e.rollout(policy, n)
            left -= n
        torch.cuda.synchronize()

    def counters(self) -> dict:
        tot: dict = {}
        for e in self.shards:
            for key, v in e.counters().items():
                tot[key] = tot.get(key, 0) + v
        return tot

    def header_field(self, name):
        return self.torch.cat([e.header_field(name) for e in self.shards])

    def close(self):
        for e in self.shards:
            e.close()

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        self.torch.cuda.synchronize()

    def run(self, mode: str, n_steps: int, events=None):
        # every shard issues its own chain of launches on its own stream; chains are independent
        torch, args = self.torch, self.args
        every = max(1, int(getattr(args, "event_every", 1)))  # HIP events around every `every`-th launch of the timed region
        if mode == "step":
            for i in range(n_steps):
                for e, st in zip(self.shards, self.streams):
                    with torch.cuda.stream(st):
                        act = e.policy_actions(self.policy)
                        if events is not None and i % every == 0:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(st)
                            e.step_async(act["stage_idx"], act["num_exec"])
                            e1.record(st)
                            events.append((e0, e1))
                        else:
                            e.step_async(act["stage_idx"], act["num_exec"])
        elif mode == "bounded":  # launches with an event budget: an env whose step is cut goes on in the next launch (sss_step_bounded)
            for i in range(n_steps):
                for e, st in zip(self.shards, self.streams):
                    with torch.cuda.stream(st):
                        act = e.policy_actions(self.policy)
                        if events is not None and i % every == 0:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(st)
                            e.step_bounded_async(act["stage_idx"], act["num_exec"], args.bounded_events)
                            e1.record(st)
                            events.append((e0, e1))
                        else:
                            e.step_bounded_async(act["stage_idx"], act["num_exec"], args.bounded_events)
        else:
            done = 0
            while done < n_steps:
                n = min(args.fused_chunk, n_steps - done)
                for e, st in zip(self.shards, self.streams):
                    with torch.cuda.stream(st):
                        if events is not None:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(st)
                            e.rollout(self.policy, n)
                            e1.record(st)
                            events.append((e0, e1))
                        else:
                            e.rollout(self.policy, n)
                done += n

    def measure(self, mode: str, steps: int, warmup: int) -> dict:
        """W untimed + exactly K timed batched steps in `mode`; all-rank totals on every rank"""
        torch, args = self.torch, self.args
        self.run(mode, warmup)
        self.barrier()
        c0 = self.counters()
        events: list = []
        self.barrier()
        t0 = time.perf_counter()
        self.run(mode, steps, events)
        self.barrier()
        dt = time.perf_counter() - t0
        c1 = self.counters()
        kern_ms = sum(a.elapsed_time(b) for a, b in events)
        tot = torch.tensor([float(c1["n_steps"] - c0["n_steps"]), float(c1["n_events"] - c0["n_events"]),
                            float(c1["model_bytes"] - c0["model_bytes"]), kern_ms, float(len(events)),
                            float(c1["n_fast_events"] - c0["n_fast_events"]), float(c1["n_batched_events"] - c0["n_batched_events"]),
                            float(c1["n_rounds"] - c0["n_rounds"])], dtype=torch.float64, device=self.dev)
        tmax = torch.tensor([dt], dtype=torch.float64, device=self.dev)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        steps_all, evs_all, bytes_all, kern_ms_all, timed_launches_all, fast_all, batched_all, rounds_all = tot.cpu().tolist()
        dt_max = tmax.item()
        avg_launch_s = (kern_ms_all / timed_launches_all) * 1e-3  # (over the launches that carry events: every --event-every-th step launch)
        # all launches of the timed region: one per shard and batched step in the step / bounded modes; every fused launch carries events
        launches_all = float(steps * len(self.shards) * self.world) if mode in ("step", "bounded") else timed_launches_all
        bytes_per_launch = bytes_all / launches_all
        achieved = bytes_per_launch / avg_launch_s / 1e9
        kernel = {"step": "sss_step_kernel", "bounded": "sss_step_bounded_kernel"}.get(mode, "sss_rollout_kernel")
        evps = evs_all / max(1.0, steps_all)
        return {
            "value": steps_all / dt_max,
            "ms_per_step": dt_max / steps * 1e3,
            "launches_per_step": launches_all / self.world / steps if mode in ("step", "bounded") else launches_all / self.world / max(1, (steps + args.fused_chunk - 1) // args.fused_chunk),
            "events_per_s": evs_all / dt_max,
            "events_per_step": evps,
            "fast_path_event_frac": fast_all / max(1.0, evs_all),
            # share of all events handled by lane-parallel batches, and events per batch round
            "batched_event_frac": batched_all / max(1.0, evs_all),
            "events_per_batch": batched_all / max(1.0, rounds_all),
            # rank-0 shader-clock ticks per real env step spent in each phase (device counters)
            "phase_ticks_per_step": {k[6:]: (c1[k] - c0[k]) / max(1, c1["n_steps"] - c0["n_steps"]) for k in c1 if k.startswith("ticks_")},
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(kernel, self.config, self.B, evps),
                # the same passes with FETCH_SIZE at face value: the streaming calibration (x2) over-corrects narrow scattered
                # reads, which the counter reports as whole 128-byte lines (profiles/r03_traffic_calibration.json)
                "traffic_lo": measured_traffic(kernel, self.config, self.B, evps, "lo"),
                # `traffic` is NOT measured in this run: rocprofv3 cannot wrap itself around a region of a running
                # process, so the PMC passes are separate runs of this kernel / config (tools/collect_traffic.py)
                "traffic_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this kernel and config, same events per step within 15 %; null if none)",
                "bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_launch_s * 1e3,
                "kernel_time_frac_of_wall": avg_launch_s * (launches_all / self.world / len(self.shards)) / dt_max,
                "launches_with_events": timed_launches_all / self.world,
            },
        }

    def step_tail(self, launches: int = 24) -> dict:
        """why a step launch lasts as long as it does: per-env shader ticks of single launches (device
        counters read before and after each; outside the timed region). A launch ends with its
        slowest env: `slowest / mean` env ticks is the share of the launch spent waiting for it."""
        import numpy as np

        from spark_sched_sim_amd.vec_env import HDR_OFF, HDR_PROF

        torch, e = self.torch, self.shards[0]

        def snap():
            h = e._env_view[:, : e.dims.hdr_bytes].cpu().numpy()
            prof = np.ascontiguousarray(h[:, HDR_PROF: HDR_PROF + 40]).view(np.uint64).astype(np.int64)
            ev = np.ascontiguousarray(h[:, HDR_OFF["n_events"]: HDR_OFF["n_events"] + 8]).view(np.uint64).ravel().astype(np.int64)
            return prof[:, 1:].sum(1), ev  # action + events + reward + observe ticks (slow-path ticks are inside events)

        slow, mean, p99, evs, ms = [], [], [], [], []
        for _ in range(launches):
            t_a, ev_a = snap()
            act = e.policy_actions(self.policy)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            e.step_async(act["stage_idx"], act["num_exec"])
            e1.record()
            torch.cuda.synchronize()
            t_b, ev_b = snap()
            d = t_b - t_a
            k = int(np.argmax(d))
            slow.append(float(d[k])), mean.append(float(d.mean())), p99.append(float(np.sort(d)[-max(1, len(d) // 100)]))
            evs.append(int(ev_b[k] - ev_a[k])), ms.append(e0.elapsed_time(e1))
        return {"launches": launches, "slowest_env_ticks": float(np.mean(slow)), "mean_env_ticks": float(np.mean(mean)), "p99_env_ticks": float(np.mean(p99)),
                "slowest_over_mean": float(np.mean(slow) / max(1.0, np.mean(mean))), "slowest_env_events": float(np.mean(evs)),
                "launch_ms": float(np.mean(ms)), "what": "per-env shader ticks of single step launches; a launch ends with its slowest env"}


BOUNDED_EVENTS = {"c2": 48, "c3": 32, "c1": 48, "e100": 32}
# the "deep" trace regime: ~530-630 events per step; swept 768 .. 4096 (profiles/r06_bench.md section 2): a budget only cuts between
# event-loop rounds and a round is a run of hundreds of events there - at best equal to lock-step
BOUNDED_EVENTS_DEEP = {"c2": 2048, "c3": 2048}


def bounded_record(bench: "Bench", args, config: str, steps: int, warmup: int):
    """the same envs stepped by launches with an event budget (include/sss.h sss_step_bounded): a launch of sss_step ends with
    its slowest env; here an env whose step needs more than the budget continues in the next launch while the others take their
    next steps - the envs run at their own pace, as the reference's do in their worker processes. Env-steps per second of
    completed steps (per-env trajectories are the same bit for bit: tests/test_{emu,gpu}_bounded.py). NOT the headline `value`,
    which stays the lock-step one."""
    table = BOUNDED_EVENTS_DEEP if getattr(bench, "pack_profile", "default") == "deep" else BOUNDED_EVENTS
    budget = table.get(config, 24) if args.bounded_events is None else args.bounded_events
    if budget <= 0:
        return None
    saved = args.bounded_events
    args.bounded_events = budget
    try:
        k = max(steps, 200)
        r = bench.measure("bounded", k, max(warmup, 50))
    finally:
        args.bounded_events = saved
    return {"value": r["value"], "unit": "env-steps/s", "max_events_per_launch": budget, "launches": k, "ms_per_launch": r["ms_per_step"],
            "completed_steps_per_launch_and_env": r["value"] * r["ms_per_step"] * 1e-3 / bench.B, "events_per_step": r["events_per_step"],
            "roofline": r["roofline"],
            "what": "policy launch + sss_step_bounded per iteration; an env's step that exceeds the budget continues in the next launch"}


def deep_record(args, B: int, dev, rank: int, world: int) -> dict:
    """extra, not the headline: the same entry points on the second synthetic trace regime (workload.PROFILES["deep"]: DAGs of up to 40
    stages with parents anywhere upstream, 4 .. 3000 tasks per stage, durations 50 ms .. 40 s - a 60 MB pack, past the aggregate L2, with
    hundreds of task completions per scheduling decision), which the reference-recorded `deep_*` goldens pin (tests/golden/). C2 and
    C3 sizing: step mode (sss_policy + sss_step per batched step) and the fused rollout, each with its roofline; the C oracle on all host
    cores beside them on the same pack."""
    out: dict = {"pack": "deep", "what": "BASELINE config 2 / 3 sizing on the 'deep' trace regime (<= 40 stages, in-degree <= 6 over all predecessors, <= 3000 tasks "
                                          "per stage, 60 MB pack); on-device policies, auto-reset, every env past its first episodes"}
    for config in ("c2", "c3"):
        try:
            b = Bench(args, config, DEFAULT_POLICY[config], B, dev, rank, world, pack_profile="deep")
            k, w = max(10, min(args.steps, 200)), max(5, min(args.warmup, 30))
            r = b.measure("step", k, w)
            f = b.measure("fused", k, w)
            rec = {"value": r["value"], "unit": "env-steps/s", "ms_per_step": r["ms_per_step"], "steps": k, "events_per_step": r["events_per_step"], "events_per_s": r["events_per_s"],
                   "fast_path_event_frac": r["fast_path_event_frac"], "batched_event_frac": r["batched_event_frac"], "roofline": r["roofline"],
                   "step_tail": b.step_tail(8), "fused": {"value": f["value"], "ms_per_step": f["ms_per_step"], "events_per_s": f["events_per_s"], "roofline": f["roofline"]},
                   "mean_last_episode_return": b.header_field("last_ep_return").mean().item(), "envs_in_error_state": int((b.shards[0].obs_i32[:, 7] != 0).sum())}
            bd = bounded_record(b, args, config, k, w)
            if bd is not None:
                rec["bounded_launches"] = bd
            b.close()
            if not args.no_cpu_baseline:
                rec["cpu_baseline_all_cores"] = cpu_baseline_all_cores(CONFIGS[config], DEFAULT_POLICY[config], min(4.0, args.cpu_budget / 2), "deep")
                ref_py = reference_python_baseline(config, "deep")
                if ref_py is not None:
                    rec["cpu_baseline_reference_python"] = ref_py
            out[config] = rec
        except Exception as e:  # never let the extra record take the bench line down
            out[config] = {"error": repr(e)}
    return out


def run_ranks_myself(args) -> int:
    """--gpus N without a torch.distributed environment: N fresh rank processes (nothing here has touched the GPU)"""
    from spark_sched_sim_amd.distributed import launch_ranks

    return launch_ranks(args.gpus, [osp.abspath(__file__)] + sys.argv[1:])


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--config", default="c2", choices=list(CONFIGS))
    ap.add_argument("--policy", default=None, choices=["hash", "fair"])
    ap.add_argument("--mode", default="step", choices=["step", "fused"])
    ap.add_argument("--fused-chunk", type=int, default=50)
    ap.add_argument("--preroll", type=int, default=None, help="fused steps every env runs before anything is timed (default: a few episodes)")
    ap.add_argument("--pack", default="default", choices=["default", "deep"],
                    help="synthetic trace regime (spark_sched_sim_amd/workload.py PROFILES): 'default' = the frozen 2.7 MB set of SURVEY 8(d); 'deep' = "
                         "<= 40 stages, in-degree <= 6, <= 3000 tasks per stage, a 60 MB pack")
    ap.add_argument("--no-deep", action="store_true", help="skip the extra record on the 'deep' trace regime (N=1, --config c2 --pack default only)")
    ap.add_argument("--no-decima", action="store_true", help="skip the extra Decima-in-the-loop measurement (N=1, c2 only)")
    ap.add_argument("--no-ppo", action="store_true", help="skip the extra PPO-iteration record (one rank's share of BASELINE config 5; N=1, c2 only)")
    ap.add_argument("--event-every", type=int, default=None, help="HIP events (the roofline's launch durations) around every N-th step launch of the timed region: "
                    "an event pair per launch costs the stream ~5 us per step (profiles/r05_bench.md section 5), 3 %% of a config-2 step. Default: every 8th, "
                    "more often for short regions so that at least ~8 launches of the K timed steps carry events (K = 20: every 2nd)")
    ap.add_argument("--lib", default=None, help="path of a test build of the library to measure instead of the product (A/B timing; tests/gpu_variant.py builds them)")
    ap.add_argument("--no-e100", action="store_true", help="skip the 100-executor record (the wide instantiation; N=1, --config c2 only)")
    ap.add_argument("--no-c3", action="store_true", help="skip the BASELINE config 3 record (N=1, --config c2 only)")
    ap.add_argument("--shards", type=int, default=1,
                    help="split the rank's envs into this many independently stepped sub-batches, one HIP stream each "
                         "(a launch lasts as long as its slowest env; with several streams the tails overlap)")
    ap.add_argument("--single-mode", action="store_true", help="skip the second measurement in the other mode")
    ap.add_argument("--bounded-events", type=int, default=None,
                    help="event budget per launch of the extra `bounded_launches` record (sss_step_bounded; default: 48 at c2, 32 at c3; 0: skip the record)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=8.0, help="seconds of the single-core CPU leg (the all-core and config-3 legs take at most half of it each)")
    ap.add_argument("--sustained-s", type=float, default=1.2, help="a second, longer timed region in the same mode when the K timed steps last less than this (0: off)")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL on ROCm); 'gloo' only for plumbing tests")
    ap.add_argument("--device-index", type=int, default=None, help="override LOCAL_RANK -> device mapping (plumbing tests on one GPU)")
    args = ap.parse_args()
    if args.event_every is None:
        args.event_every = max(1, min(8, args.steps // 8))

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(run_ranks_myself(args))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP path has no CPU fallback")
    dev_index = local_rank if args.device_index is None else args.device_index
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(args.dist_backend)  # "nccl" is RCCL on ROCm

    policy = args.policy or DEFAULT_POLICY[args.config]
    B = args.envs
    bench = Bench(args, args.config, policy, B, dev, rank, world)
    cfg = bench.cfg

    primary = bench.measure(args.mode, args.steps, args.warmup)
    # K timed steps can be a few milliseconds (the driver runs --steps 20): the same measurement over a region of
    # about a second next to it, so that the line also carries a figure that does not depend on which 20 steps it met
    sustained = None
    if args.sustained_s > 0 and primary["ms_per_step"] * args.steps < 1e3 * args.sustained_s:
        k_long = int(min(200_000, max(args.steps + 1, 1e3 * args.sustained_s / max(1e-3, primary["ms_per_step"]))))
        long_run = bench.measure(args.mode, k_long, 0)
        sustained = {"value": long_run["value"], "unit": "env-steps/s", "steps": k_long, "ms_per_step": long_run["ms_per_step"], "mode": args.mode,
                     "events_per_step": long_run["events_per_step"], "roofline": long_run["roofline"],
                     "what": f"same mode, same envs, {k_long} timed batched steps right after the K = {args.steps} of `value`"}
    other_mode = "fused" if args.mode == "step" else "step"
    secondary = None if args.single_mode else bench.measure(other_mode, args.steps, args.warmup)
    tail = bench.step_tail() if (world == 1 and args.shards == 1) else None
    bounded = bounded_record(bench, args, args.config, args.steps, args.warmup) if world == 1 else None

    if world > 1:
        # the one exchange of the path: all-gather of per-env episode summaries (RCCL)
        from spark_sched_sim_amd.distributed import all_gather_episode_summaries

        table = torch.cat([all_gather_episode_summaries(e) for e in bench.shards])
        mean_return = table[:, 0].mean().item()
        ranks_seen = int(table.shape[0] // B)
    else:
        mean_return = bench.header_field("last_ep_return").mean().item()
        ranks_seen = 1
    # every env has finished at least one episode before the timed region: the measured regime is the steady state
    if (args.preroll is None or args.preroll >= PREROLL_STEPS[args.config]) and mean_return == 0.0:
        raise SystemExit("bench.py: the envs are not in their steady state (no finished episode)")

    out = None
    if rank == 0:
        workload = (f"{B} envs/GPU x ({cfg['num_executors']} executors, {cfg['job_arrival_cap']} TPC-H-format jobs, job_arrival_rate {cfg['job_arrival_rate']}/ms, "
                    f"synthetic frozen trace set{'' if args.pack == 'default' else ' of the ' + repr(args.pack) + ' regime'}), on-device '{policy}' policy, auto-reset, steady state (every env several episodes in), "
                    f"mode={args.mode} ({'sss_policy + sss_step per batched step' if args.mode == 'step' else 'sss_rollout, ' + str(args.fused_chunk) + ' steps per launch'})")
        out = {
            "metric": "env-steps/sec at 4096 batched envs (TPC-H, 10 exec)" if (args.config == "c2" and B == 4096) else f"env-steps/sec at {B} batched envs ({args.config})",
            "value": primary["value"],
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": primary["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64+f64",
            "data": "synthetic",
            "config": {"workload": workload, "envs_per_gpu": B, "policy": policy, "mode": args.mode,
                       "parallelism": f"env-shard x{world}", "streams_per_gpu": max(1, args.shards), "ranks_seen_by_all_gather": ranks_seen},
            "events_per_s": primary["events_per_s"],
            "events_per_step": primary["events_per_step"],
            "fast_path_event_frac": primary["fast_path_event_frac"],
            "batched_event_frac": primary["batched_event_frac"],
            "events_per_batch": primary["events_per_batch"],
            "phase_ticks_per_step": primary["phase_ticks_per_step"],
            "mean_last_episode_return": mean_return,
            # the roofline figures of the line are the SUSTAINED window's when there is one (>= --sustained-s of stepping in the
            # same mode): K = 20 timed steps are 3-4 ms and carry whatever bytes per launch those 20 steps happened to have
            "roofline": dict(sustained["roofline"], window=f"{sustained['steps']} batched steps (the `sustained` record; HIP events around every {args.event_every}th launch)") if sustained is not None
                        else dict(primary["roofline"], window=f"the K = {args.steps} timed steps (HIP events around every {args.event_every}th launch)"),
        }
        if sustained is not None:
            out["roofline_k_steps"] = dict(primary["roofline"], window=f"the K = {args.steps} timed steps of `value` (HIP events around every {args.event_every}th launch)")
            out["sustained"] = sustained
        if tail is not None:
            out["step_tail"] = tail
        if secondary is not None:
            # the same K batched steps through the other entry point (same per-step work, same trajectories)
            out["other_mode"] = dict(secondary, mode=other_mode)
        if bounded is not None:
            out["bounded_launches"] = bounded
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, policy, args.cpu_budget, args.pack)
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cfg, policy, min(4.0, args.cpu_budget / 2), args.pack)
            except Exception as e:  # never let the extra baseline take the bench line down
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
            ref_py = reference_python_baseline(args.config, args.pack)
            if ref_py is not None:
                out["cpu_baseline_reference_python"] = ref_py
    pack = bench.pack
    bench.close()
    del bench  # (the env's arenas are torch tensors: dropped with the last reference, not by close())

    # BASELINE config 3 next to the headline (N = 1): same measurements, its own roofline and CPU baseline
    if world == 1 and args.config == "c2" and not args.no_deep and args.shards == 1 and args.pack == "default":
        out["deep"] = deep_record(args, B, dev, rank, world)
    if world == 1 and args.config == "c2" and not args.no_c3 and args.shards == 1:
        try:
            b3 = Bench(args, "c3", DEFAULT_POLICY["c3"], B, dev, rank, world)
            k3, w3 = max(10, min(args.steps, 300)), max(5, min(args.warmup, 50))
            r3 = b3.measure("step", k3, w3)
            f3 = b3.measure("fused", k3, w3)
            s3 = None
            if args.sustained_s > 0 and r3["ms_per_step"] * k3 < 1e3 * args.sustained_s:
                k3_long = int(min(200_000, max(k3 + 1, 1e3 * args.sustained_s / max(1e-3, r3["ms_per_step"]))))
                l3 = b3.measure("step", k3_long, 0)
                s3 = {"value": l3["value"], "unit": "env-steps/s", "steps": k3_long, "ms_per_step": l3["ms_per_step"], "mode": "step", "roofline": l3["roofline"]}
            rec = {"what": f"BASELINE config 3: {B} envs x (50 executors, 200 jobs), fair policy, steady state; {k3} timed batched steps after {w3} warm-up steps",
                   "value": r3["value"], "unit": "env-steps/s", "ms_per_step": r3["ms_per_step"], "events_per_step": r3["events_per_step"],
                   "fast_path_event_frac": r3["fast_path_event_frac"], "batched_event_frac": r3["batched_event_frac"], "events_per_batch": r3["events_per_batch"],
                   "phase_ticks_per_step": r3["phase_ticks_per_step"], "roofline": s3["roofline"] if s3 is not None else r3["roofline"],
                   "roofline_k_steps": r3["roofline"], "step_tail": b3.step_tail(12),
                   "other_mode": dict(f3, mode="fused"), "mean_last_episode_return": b3.header_field("last_ep_return").mean().item()}
            bd3 = bounded_record(b3, args, "c3", k3, w3)
            if bd3 is not None:
                rec["bounded_launches"] = bd3
            b3.close()
            del b3
            if s3 is not None:
                rec["sustained"] = s3
            if not args.no_cpu_baseline:
                rec["cpu_baseline"] = cpu_baseline(CONFIGS["c3"], "fair", min(4.0, args.cpu_budget / 2), args.pack)
                rec["cpu_baseline_all_cores"] = cpu_baseline_all_cores(CONFIGS["c3"], "fair", min(4.0, args.cpu_budget / 2), args.pack)
                ref_py = reference_python_baseline("c3", args.pack)
                if ref_py is not None:
                    rec["cpu_baseline_reference_python"] = ref_py
            out["c3"] = rec
        except Exception as e:
            out["c3"] = {"error": repr(e)}
    # more than 64 executors (the wide instantiation of the kernels, csrc/sss_hip_wide.hip) next to the headline (N = 1): 100 executors,
    # 200 jobs, step mode at 1024 and at 4096 envs, with the C oracle on all host cores beside it
    if world == 1 and args.config == "c2" and not args.no_e100 and args.shards == 1:
        try:
            rec = {"what": "100 executors, 200 TPC-H-format jobs (job_arrival_rate 8e-05/ms), on-device 'fair' policy, auto-reset, steady state; step mode "
                           "(sss_policy + sss_step per batched step); the wide instantiation: two executors per lane"}
            ke, we = max(10, min(args.steps, 200)), max(5, min(args.warmup, 50))
            for Be in (1024, 4096):
                be = Bench(args, "e100", DEFAULT_POLICY["e100"], Be, dev, rank, world)
                re_ = be.measure("step", ke, we)
                fe = be.measure("fused", ke, we)
                rec[f"envs_{Be}"] = {"value": re_["value"], "unit": "env-steps/s", "ms_per_step": re_["ms_per_step"], "events_per_step": re_["events_per_step"],
                                     "batched_event_frac": re_["batched_event_frac"], "roofline": re_["roofline"], "step_tail": be.step_tail(8),
                                     "fused": {"value": fe["value"], "ms_per_step": fe["ms_per_step"]}}
                be.close()
                del be
            if not args.no_cpu_baseline:
                rec["cpu_baseline_all_cores"] = cpu_baseline_all_cores(CONFIGS["e100"], "fair", min(4.0, args.cpu_budget / 2))
            out["e100"] = rec
        except Exception as e:
            out["e100"] = {"error": repr(e)}
    if rank == 0:
        if world == 1 and not args.no_decima and args.config == "c2":
            try:  # SURVEY 8(f) next-1 on the same env sizing; never let it take the bench line down
                out["decima_in_loop"] = decima_in_loop(cfg, B, dev, pack)
            except Exception as e:
                out["decima_in_loop"] = {"error": repr(e)}
            try:  # BASELINE config 4 is 8192 envs over 8 GPUs: one rank's share
                out["decima_in_loop_1024"] = decima_in_loop(cfg, 1024, dev, pack)
            except Exception as e:
                out["decima_in_loop_1024"] = {"error": repr(e)}
        if world == 1 and not args.no_ppo and args.config == "c2":
            try:  # SURVEY 8(f) next-3 / BASELINE config 5 at one rank's share; never let it take the bench line down
                out["ppo_config5_share"] = ppo_config5_share(dev)
            except Exception as e:
                out["ppo_config5_share"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
