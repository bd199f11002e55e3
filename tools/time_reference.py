#!/usr/bin/env python3
"""Times the REFERENCE Python env (imported, unmodified, from /root/reference) on the build's frozen synthetic trace set:
SURVEY 8(d) "re-measure here with the frozen generator seed: 6 episodes C1, 2 episodes C3, perf_counter around env.step only,
plus 8-process aggregate"; the loop is the reference's own harness (examples.py:84-102: reset, then schedule -> step until
terminated) with its fair scheduler (RoundRobinScheduler, dynamic_partition=True, examples.py:51).

Runs ONLY in the build container: /root/reference and the gymnasium stand-in (tests/refharness) do not exist on the GPU box.
It writes profiles/reference_python.json, which is committed; bench.py copies that record into its JSON line as
`cpu_baseline_reference_python` with the hardware it was measured on stated - a reported baseline, not something timed in the
bench run.

    python tools/time_reference.py            # 1 core + N-process aggregate, both configs -> profiles/reference_python.json
"""
from __future__ import annotations

import json
import os
import os.path as osp
import platform
import subprocess
import sys
import tempfile
import time

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
REF = os.environ.get("SSS_REFERENCE", "/root/reference")
CONFIGS = {
    # C1: examples.py:15-23 minus render_mode; C3 sizing: config/decima_tpch.yaml:81-85
    "c1": (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0,
                data_sampler_cls="TPCHDataSampler"), [1234, 0, 1, 2, 3, 4]),
    "c3": (dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0,
                data_sampler_cls="TPCHDataSampler"), [0, 1]),
}


# the "deep" trace regime (workload.PROFILES): fewer, much longer episodes (hundreds of task completions per decision)
DEEP_CONFIGS = {"c1": (CONFIGS["c1"][0], [0, 1]), "c3": (CONFIGS["c3"][0], [0])}


def worker(config: str, seed_offset: int, data_dir: str) -> dict:
    """one process: the config's episodes, one after the other; times env.step alone, env + scheduler, and with reset"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, osp.join(ROOT, "tests", "refharness"))  # gymnasium stand-in
    sys.path.insert(1, REF)
    sys.path.insert(0, osp.join(ROOT, "tests", "golden"))
    os.chdir(data_dir)  # the reference reads data/tpch relative to cwd (tpch.py:48,119)
    from make_golden import import_reference

    gym, sched_cls, _ = import_reference()
    env_cfg, seeds = (DEEP_CONFIGS if os.environ.get("SSS_TRACE_PROFILE") == "deep" else CONFIGS)[config]
    t_step = t_sched = t_reset = 0.0
    steps = 0
    for seed in seeds:
        env = gym.make("spark_sched_sim:SparkSchedSimEnv-v0", env_cfg=dict(env_cfg))
        sched = sched_cls(env_cfg["num_executors"], dynamic_partition=True)
        t0 = time.perf_counter()
        obs, _ = env.reset(seed=seed + seed_offset, options=None)
        t_reset += time.perf_counter() - t0
        done = False
        while not done:
            t0 = time.perf_counter()
            action, _ = sched.schedule(obs)
            t1 = time.perf_counter()
            obs, _, terminated, truncated, _ = env.step(action)
            t2 = time.perf_counter()
            t_sched += t1 - t0
            t_step += t2 - t1
            steps += 1
            done = terminated or truncated
    return {"steps": steps, "episodes": len(seeds), "t_step": t_step, "t_sched": t_sched, "t_reset": t_reset}


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def main() -> None:
    if len(sys.argv) >= 2 and sys.argv[1] == "--worker":
        print(json.dumps(worker(sys.argv[2], int(sys.argv[3]), sys.argv[4])))
        return
    if not osp.isdir(REF):
        raise SystemExit(f"{REF} not found: this script runs in the build container only")
    sys.path.insert(0, ROOT)
    from spark_sched_sim_amd import workload

    profile = "deep" if "--deep" in sys.argv else "default"
    if profile == "deep":  # a second record next to the default one: profiles/reference_python.json "deep"
        os.environ["SSS_TRACE_PROFILE"] = "deep"
    n_proc = len(os.sched_getaffinity(0))
    out = {"what": "the reference Python env (spark_sched_sim/spark_sched_sim.py, imported unmodified) with its fair scheduler on the build's "
                   "frozen synthetic trace set; loop of examples.py:84-102; time.perf_counter around env.step / scheduler.schedule / env.reset",
           "hardware": f"build container: {cpu_model()}, {n_proc} vCPU (not the GPU box's host)", "python": platform.python_version(),
           "unit": "env-steps/s", "configs": {}}
    raw = workload.make_raw_workload(profile=profile)
    out["pack_sha256"] = workload.pack_digest(workload.build_pack(raw))
    with tempfile.TemporaryDirectory() as tmp:
        workload.write_reference_layout(raw, tmp)

        def launch(config: str, off: int):
            return subprocess.Popen([sys.executable, osp.abspath(__file__), "--worker", config, str(off), tmp], stdout=subprocess.PIPE, text=True)

        for config in CONFIGS:
            one = json.loads(launch(config, 0).communicate()[0].strip().splitlines()[-1])
            t0 = time.perf_counter()
            procs = [launch(config, 1000 * (w + 1)) for w in range(n_proc)]
            many = [json.loads(p.communicate()[0].strip().splitlines()[-1]) for p in procs]
            wall = time.perf_counter() - t0
            rec = {
                "episodes": one["episodes"], "steps": one["steps"],
                "one_core": {"env_only": one["steps"] / one["t_step"], "env_plus_scheduler": one["steps"] / (one["t_step"] + one["t_sched"]),
                             "including_reset": one["steps"] / (one["t_step"] + one["t_sched"] + one["t_reset"])},
                "all_cores": {"processes": n_proc, "steps": sum(m["steps"] for m in many),
                              # every process runs the same number of episodes: aggregate = total steps / the slowest process's time
                              "env_only": sum(m["steps"] for m in many) / max(m["t_step"] for m in many),
                              "including_scheduler_and_reset": sum(m["steps"] for m in many) / max(m["t_step"] + m["t_sched"] + m["t_reset"] for m in many),
                              "wall_s_including_imports": wall},
            }
            out["configs"][config] = rec
            print(config, json.dumps(rec), flush=True)
    path = osp.join(ROOT, "profiles", "reference_python.json")
    if profile == "deep":
        whole = json.load(open(path))
        whole["deep"] = dict(out, trace_profile="deep")
        out = whole
    else:
        try:  # keep the records other tools added (tools/time_reference_decima.py, --deep)
            old = json.load(open(path))
            out.update({k: v for k, v in old.items() if k in ("decima_c1", "ppo_config5", "deep")})
        except Exception:
            pass
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
