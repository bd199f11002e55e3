"""debug: scoped profile of the lane-0 procedures (the -DSSS_EVPROF3 test build: python tests/gpu_variant.py evprof3 ahead of the gpurun call)"""
import ctypes as C, sys, os, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.binding import load_library
sys.path.insert(0, osp.join(ROOT, "tests"))
from gpu_variant import build_variant
NAMES = {0: "batch_released_events", 1: "trk_add_commitment", 2: "trk_remove_commitment", 3: "trk_move_executor_to_pool", 4: "job_record_stage_completion", 5: "task_duration",
         6: "find_backup_stage", 7: "execute_next_task", 8: "send_executor", 9: "get_idle_source_executors", 10: "move_idle_executors_all",
         11: "move_executor_to_stage", 12: "fulfill_commitments_from_source", 13: "batch_released_events: commit + lane-0 tail", 14: "handle_executor_arrival",
         15: "process_job_completion", 16: "handle_task_completion", 17: "take_action", 18: "jobtime_build_set", 19: "cache_acquire", 20: "batch_arrival_events",
         21: "find_schedulable_all", 22: "write_observation", 23: "env_begin", 24: "env_end", 25: "jobtime_sum", 26: "resume_simulation", 27: "do_reset",
         28: "do_step", 29: "run_policy", 30: "fast_run (per EVENT)", 31: "handle_popped", 32: "lean_released", 33: "fulfil_run", 34: "select_stage_wave",
         35: "fulfil_chunk", 36: "fulfil_order_commitments", 37: "pop_event_wave", 38: "lean_arrival", 39: "fulfil_common_wave", 40: "preflush_completing_job"}
CFG = {"c2": (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash"),
       "c3": (dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair"),
       "e100": (dict(num_executors=100, job_arrival_cap=200, job_arrival_rate=8.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair")}
N_ENVS = int(os.environ.get("SSS_ENVS", "4096"))
if os.environ.get("SSS_SECTIONS"):
    NAMES.update({1: "rel: reads .. first exit", 2: "rel: commitment scan", 3: "rel: classify", 4: "rel: window + ranking", 5: "rel: draws + lane commit", 6: "rel: lane-0 commitments",
                  7: "rel: per-lane pool records", 8: "rel: pools_staged", 9: "rel: sched clear", 10: "rel: sat bits", 11: "rel: send cache_acquire"})
if os.environ.get("SSS_SECTIONS") == "fast":
    NAMES.update({1: "fast: reads + classification", 2: "fast: t_stop, window", 3: "fast: ranking", 4: "fast: the loop (incl. first draw)", 5: "fast: write-back"})
    for k in range(6, 13):
        NAMES.pop(k, None)
if os.environ.get("SSS_SECTIONS") == "arr":
    NAMES.update({1: "arr: reads .. first exit", 2: "arr: classify", 3: "arr: window M", 4: "arr: before / same masks", 5: "arr: over test + draws", 6: "arr: lane commit",
                  7: "arr: per-lane pool records", 8: "arr: pools_staged PASS (job pools)", 9: "arr: pools_staged ENTER (stage pools)"})
    NAMES.update({7: "arr: up to the pair fetch", 10: "arr: pair fetch + stage (data in LDS)", 11: "arr: pair set operations", 12: "arr: pair flush (stores issued)",
                  8: "arr: slot refs, ballots, sync after the stores"})
# the timing build: tests/_build/libsss_hip_evprof3{,b,c}.so (python tests/gpu_variant.py evprof3 - ahead of the gpurun call)
lib = load_library(build_variant("evprof3" + {"": "", "fast": "c", "arr": "d"}.get(os.environ.get("SSS_SECTIONS", ""), "b")))
buf = (C.c_ulonglong * 96)()
mode = sys.argv[2] if len(sys.argv) > 2 else "fused"
# third argument: only record step launches of envs whose do_step took at least that many ticks (the tail of step mode)
min_ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lib.sss_debug_prof_min.argtypes = [C.c_ulonglong]
lib.sss_debug_prof_min_wide.argtypes = [C.c_ulonglong]
for name in sys.argv[1].split(","):
    cfg, pol = CFG[name]
    prof, prof_min = (lib.sss_debug_prof_wide, lib.sss_debug_prof_min_wide) if cfg["num_executors"] > 64 else (lib.sss_debug_prof, lib.sss_debug_prof_min)
    env = VecSparkSchedSimEnv(cfg, N_ENVS, device="cuda:0", pack=workload.default_pack(), auto_reset=True, _lib=lib)
    env.reset(seed=0)
    env.rollout(pol, 600)
    torch.cuda.synchronize()
    prof(buf)
    prof_min(min_ticks)
    c0 = env.counters()
    if mode == "fused":
        for _ in range(6):
            env.rollout(pol, 50)
    else:
        for _ in range(300):
            env.step(env.policy_actions(pol))
    torch.cuda.synchronize()
    prof(buf)
    c1 = env.counters()
    prof_min(0)
    steps = c1["n_steps"] - c0["n_steps"]; evs = c1["n_events"] - c0["n_events"]
    if min_ticks:
        steps = buf[2 * 28 + 1]  # per recorded (slow) step
    print(f"== {name} {mode}: {steps} steps, {evs/steps:.1f} events/step; per STEP: ticks (calls, ticks/call)")
    rows = [(buf[2 * i] / steps, buf[2 * i + 1] / steps, NAMES[i]) for i in NAMES if buf[2 * i + 1]]
    for t, n, nm in sorted(rows, reverse=True):
        print(f"  {nm:34s} {t:10.0f}  ({n:7.3f} calls, {t / max(n, 1e-9):9.0f} each)")
    env.close()
