#!/usr/bin/env python3
"""Throughput benchmark of the hot path: env-steps/s of the batched Spark-scheduling simulator.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs B] [--config c2|c3] [--policy hash|fair]
                    [--mode step|fused] [--no-cpu-baseline]

A "step" is ONE batched step of all envs of a rank: on-device policy kernel + step kernel
(`--mode step`, the drop-in boundary: sss_policy + sss_step per step), or one iteration of the
fused rollout kernel (`--mode fused`, sss_rollout: identical per-step work, policy -> step ->
observe, without leaving the kernel). Envs auto-reset (next-step mode); only real step() calls are
counted (the device counts them), so `value` = real env steps of all ranks / max-over-ranks time.

Multi-GPU (`--gpus N`, launched by torch.distributed.run): envs are sharded, B per rank, no
data-path collective; one RCCL all-gather of per-env episode returns after the timed region
(the stand-in for the reference's Pipe gather, trainers/trainer.py:113-121). Weak scaling.

Prints ONE JSON line on rank 0 (see the driver contract) including `roofline` (SURVEY 8(d) model
bytes of the dominant kernel / its HIP-event-measured duration / 8 TB/s) and `cpu_baseline`
(the C oracle, oracle/sss_oracle.c, timed on this box's host on a bounded sample).
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
}
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def cpu_baseline(cfg: dict, policy: str, budget_s: float) -> dict:
    """the C oracle (a port of the reference env, bit-identical trajectories) on ONE host core,
    same workload, whole episodes until ~budget_s of CPU time is spent"""
    import ctypes as C

    sys.path.insert(0, osp.join(ROOT, "tests"))
    from oracle_binding import OracleEnv
    from spark_sched_sim_amd import workload

    env = OracleEnv(workload.default_pack(), cfg)
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
env = OracleEnv(workload.default_pack(), cfg)
steps = eps = 0
t0 = time.perf_counter()
while time.perf_counter() - t0 < budget:
    r = C.c_double()
    steps += max(0, env.lib.sso_run_episode(env.h, 20000 + 1000 * wid + eps, pol, 10**9, C.byref(r)))
    eps += 1
print(json.dumps([steps, time.perf_counter() - t0]))
"""


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


def cpu_baseline_all_cores(cfg: dict, policy: str, budget_s: float) -> dict:
    """the reference's own parallelism model (one process per env, trainers/trainer.py:264-293)
    with the C oracle: one child process per host core (plain subprocesses that never touch the GPU)"""
    import subprocess

    n = usable_cores()
    code = _CPU_WORKER.format(root=ROOT, tests=osp.join(ROOT, "tests"))
    pol = {"fair": 0, "hash": 1}[policy]
    procs = [subprocess.Popen([sys.executable, "-c", code, json.dumps(cfg), str(pol), str(budget_s), str(w)],
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


def measured_traffic(kernel: str, config: str, envs: int):
    """HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, corrected as
    MI355X_MICROARCH.md prescribes), recorded under profiles/ by tools/collect_traffic.py for this
    exact kernel/config; None when no matching measurement is committed."""
    path = osp.join(ROOT, "profiles", "traffic.json")
    if not osp.exists(path):
        return None
    try:
        for rec in json.load(open(path)):
            if rec["kernel"] == kernel and rec["config"] == config and rec["envs"] == envs:
                return rec["hbm_bytes_per_launch"]
    except Exception:
        return None
    return None


def decima_in_loop(cfg: dict, B: int, dev, pack, steps: int = 100, warmup: int = 20) -> dict:
    """extra, not the headline: the same B envs with a sampled Decima action (GNN policy, random-init
    weights of the published architecture) for every env on every step - graph kernel, GNN kernels,
    sampling kernels, sss_step (spark_sched_sim_amd/decima.py)"""
    import torch

    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    agent = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)),
                 policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
    env = VecSparkSchedSimEnv(cfg, B, device=dev, pack=pack, auto_reset=True)
    torch.manual_seed(0)
    policy = DecimaPolicy(num_executors=cfg["num_executors"], **agent).to(dev).eval()
    gen = torch.Generator(device=dev).manual_seed(1)
    env.reset(seed=0)
    for i in range(warmup + steps):
        if i == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        act, _ = policy.schedule_env(env, generator=gen)
        env.step(act)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    err = int((env.obs_i32[:, 7] != 0).sum())
    env.close()
    return {"value": B * steps / dt, "unit": "env-steps/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "envs_in_error_state": err,
            "what": "every env gets a sampled Decima action every step (graph + GNN + sampling kernels, then sss_step)"}


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
    ap.add_argument("--no-decima", action="store_true", help="skip the extra Decima-in-the-loop measurement (N=1, c2 only)")
    ap.add_argument("--shards", type=int, default=1,
                    help="split the rank's envs into this many independently stepped sub-batches, one HIP stream each "
                         "(a launch lasts as long as its slowest env; with several streams the tails overlap)")
    ap.add_argument("--single-mode", action="store_true", help="skip the second measurement in the other mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--evprof", action="store_true", help="library built with -DSSS_EVPROF (tools/evprof.sh): report ticks per event-loop round segment")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL on ROCm); 'gloo' only for plumbing tests")
    ap.add_argument("--device-index", type=int, default=None, help="override LOCAL_RANK -> device mapping (plumbing tests on one GPU)")
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU: the HIP path has no CPU fallback")
    dev_index = local_rank if args.device_index is None else args.device_index
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(args.dist_backend)  # "nccl" is RCCL on ROCm

    from spark_sched_sim_amd import VecSparkSchedSimEnv, workload

    cfg = CONFIGS[args.config]
    policy = args.policy or ("hash" if args.config == "c2" else "fair")
    B = args.envs
    S = max(1, args.shards)
    assert B % S == 0, "--envs must be divisible by --shards"
    Bs = B // S
    pack = workload.default_pack()
    # env i of shard s of rank r: seed (r*S + s)*Bs + i = its global env id (placement invariant)
    shards = [VecSparkSchedSimEnv(cfg, Bs, device=dev, pack=pack, auto_reset=True, seed_stride=B * world) for _ in range(S)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(dev)]
    for k, e in enumerate(shards):
        e.reset(seed=(rank * S + k) * Bs)
    torch.cuda.synchronize()

    class _All:  # the rank's whole batch, as the sum of its shards
        @staticmethod
        def counters():
            tot: dict = {}
            for e in shards:
                for key, v in e.counters().items():
                    tot[key] = tot.get(key, 0) + v
            return tot

        @staticmethod
        def header_field(name):
            return torch.cat([e.header_field(name) for e in shards])

        @staticmethod
        def close():
            for e in shards:
                e.close()

    env = _All

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(mode: str, n_steps: int, events=None):
        # every shard issues its own chain of launches on its own stream; chains are independent
        if mode == "step":
            for _ in range(n_steps):
                for e, st in zip(shards, streams):
                    with torch.cuda.stream(st):
                        act = e.policy_actions(policy)
                        if events is not None:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(st)
                            e.step_async(act["stage_idx"], act["num_exec"])
                            e1.record(st)
                            events.append((e0, e1))
                        else:
                            e.step_async(act["stage_idx"], act["num_exec"])
        else:
            done = 0
            while done < n_steps:
                n = min(args.fused_chunk, n_steps - done)
                for e, st in zip(shards, streams):
                    with torch.cuda.stream(st):
                        if events is not None:
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record(st)
                            e.rollout(policy, n)
                            e1.record(st)
                            events.append((e0, e1))
                        else:
                            e.rollout(policy, n)
                done += n

    def measure(mode: str) -> dict:
        """W untimed + exactly K timed batched steps in `mode`; all-rank totals on every rank"""
        run(mode, args.warmup)
        barrier()
        c0 = env.counters()
        events: list = []
        barrier()
        t0 = time.perf_counter()
        run(mode, args.steps, events)
        barrier()
        dt = time.perf_counter() - t0
        c1 = env.counters()
        kern_ms = sum(a.elapsed_time(b) for a, b in events)
        tot = torch.tensor([float(c1["n_steps"] - c0["n_steps"]), float(c1["n_events"] - c0["n_events"]),
                            float(c1["model_bytes"] - c0["model_bytes"]), kern_ms, float(len(events)),
                            float(c1["n_fast_events"] - c0["n_fast_events"]), float(c1["n_batched_events"] - c0["n_batched_events"]),
                            float(c1["n_rounds"] - c0["n_rounds"])], dtype=torch.float64, device=dev)
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        steps_all, evs_all, bytes_all, kern_ms_all, launches_all, fast_all, batched_all, rounds_all = tot.cpu().tolist()
        dt_max = tmax.item()
        avg_launch_s = (kern_ms_all / launches_all) * 1e-3
        bytes_per_launch = bytes_all / launches_all
        achieved = bytes_per_launch / avg_launch_s / 1e9
        kernel = "sss_step_kernel" if mode == "step" else "sss_rollout_kernel"
        return {
            "value": steps_all / dt_max,
            "ms_per_step": dt_max / args.steps * 1e3,
            "launches_per_step": launches_all / world / args.steps if mode == "step" else launches_all / world / max(1, (args.steps + args.fused_chunk - 1) // args.fused_chunk),
            "events_per_s": evs_all / dt_max,
            "events_per_step": evs_all / max(1.0, steps_all),
            "fast_path_event_frac": fast_all / max(1.0, evs_all),
            # share of all events handled by lane-parallel batches, and events per batch round
            "batched_event_frac": batched_all / max(1.0, evs_all),
            "events_per_batch": batched_all / max(1.0, rounds_all),
            # rank-0 shader-clock ticks per real env step spent in each phase (device counters)
            "phase_ticks_per_step": {k[6:]: (c1[k] - c0[k]) / max(1, c1["n_steps"] - c0["n_steps"]) for k in c1 if k.startswith("ticks_")},
            "roofline": {
                "bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(kernel, args.config, B),
                "bytes_per_launch": bytes_per_launch, "avg_launch_ms": avg_launch_s * 1e3,
                "kernel_time_frac_of_wall": (kern_ms_all / world) * 1e-3 / dt_max,
            },
        }

    ev0 = env.counters() if args.evprof else None
    primary = measure(args.mode)
    evprof = None
    if args.evprof:
        ev1 = env.counters()
        dd = {k: ev1[k] - ev0[k] for k in ev1}
        rounds = max(1, dd["evprof_rounds"])
        evprof = {"rounds_per_step": rounds / max(1, dd["n_steps"]), "batch_rounds_per_step": dd["n_rounds"] / max(1, dd["n_steps"]),
                  "ticks_per_round": {"batch_classify_or_early_exit": dd["ticks_slow_events"] / rounds, "batch_member_loop": dd["ticks_action"] / rounds,
                                      "batch_draw_commit": dd["ticks_events"] / rounds, "single_pop": dd["ticks_reward"] / rounds,
                                      "single_handler": dd["ticks_observe"] / rounds, "rng_refill": dd["evprof_refill"] / rounds},
                  "ticks_per_step_total": sum(dd[k] for k in ("ticks_slow_events", "ticks_action", "ticks_events", "ticks_reward", "ticks_observe", "evprof_refill")) / max(1, dd["n_steps"])}
    other_mode = "fused" if args.mode == "step" else "step"
    secondary = None if args.single_mode else measure(other_mode)

    if world > 1:
        # the one exchange of the path: all-gather of per-env episode summaries (RCCL)
        from spark_sched_sim_amd.distributed import all_gather_episode_summaries

        table = all_gather_episode_summaries(env)
        mean_return = table[:, 0].mean().item()
    else:
        mean_return = env.header_field("last_ep_return").mean().item()

    if rank == 0:
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
            "config": {
                "workload": f"{B} envs/GPU x ({cfg['num_executors']} executors, {cfg['job_arrival_cap']} TPC-H-format jobs, "
                            f"job_arrival_rate {cfg['job_arrival_rate']}/ms, synthetic frozen trace set), on-device '{policy}' policy, "
                            f"auto-reset, mode={args.mode} ({'sss_policy + sss_step per batched step' if args.mode == 'step' else 'sss_rollout, ' + str(args.fused_chunk) + ' steps per launch'})",
                "envs_per_gpu": B, "policy": policy, "mode": args.mode, "parallelism": f"env-shard x{world}",
                "streams_per_gpu": S,
            },
            "events_per_s": primary["events_per_s"],
            "events_per_step": primary["events_per_step"],
            "fast_path_event_frac": primary["fast_path_event_frac"],
            "batched_event_frac": primary["batched_event_frac"],
            "events_per_batch": primary["events_per_batch"],
            "phase_ticks_per_step": primary["phase_ticks_per_step"],
            "mean_last_episode_return": mean_return,
            "roofline": primary["roofline"],
        }
        if evprof is not None:
            out["evprof"] = evprof
        if secondary is not None:
            # the same K batched steps through the other entry point (same per-step work, same trajectories)
            out["other_mode"] = dict(secondary, mode=other_mode)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, policy, args.cpu_budget)
            try:
                out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cfg, policy, min(6.0, args.cpu_budget))
            except Exception as e:  # never let the extra baseline take the bench line down
                out["cpu_baseline_all_cores"] = {"error": repr(e)}
        if world == 1 and not args.no_decima and args.config == "c2":
            try:  # SURVEY 8(f) next-1 on the same env sizing; never let it take the bench line down
                out["decima_in_loop"] = decima_in_loop(cfg, B, dev, pack)
            except Exception as e:
                out["decima_in_loop"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    env.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
