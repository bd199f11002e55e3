"""debug: why event-loop rounds end (CPU wave emulator, -DSSS_BATCH_STATS build): counts per exit point of the batch
paths for a config. usage: python tools/debug/batch_census.py [c2|c3|e50] [steps] [default|deep]"""
import ctypes as C, os.path as osp, subprocess, sys
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT, osp.join(ROOT, "tests")]
subprocess.run(["make", "-s", "-C", osp.join(ROOT, "tests", "emu"), "../_build/libsss_emu_stats.so"], check=True)
from emu_util import load_emu
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
CFG = {"c2": (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash"),
       "c3": (dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair"),
       "e100": (dict(num_executors=100, job_arrival_cap=200, job_arrival_rate=8.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair"),
       "e50": (dict(num_executors=50, job_arrival_cap=60, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair")}
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
profile = sys.argv[3] if len(sys.argv) > 3 else "default"
cfg, pol = CFG[name]
lib = load_emu("_stats")
env = VecSparkSchedSimEnv(cfg, 2, device="cpu", pack=workload.profile_pack(profile), _lib=lib, auto_reset=True)
env.reset(seed=[11, 12])
env.rollout(pol, 300)
stats = (C.c_longlong * 128).in_dll(lib, "sss_batch_stats")
for i in range(128):
    stats[i] = 0
c0 = env.counters()
env.rollout(pol, steps)
c1 = env.counters()
ev = c1["n_events"] - c0["n_events"]
print(f"{name}: {c1['n_steps'] - c0['n_steps']} steps, {ev} events, fast {c1['n_fast_events'] - c0['n_fast_events']}, batched {c1['n_batched_events'] - c0['n_batched_events']}, rounds {c1['n_rounds'] - c0['n_rounds']}")
for i in range(128):
    if stats[i]:
        print(f"  stat[{i:3d}] = {stats[i]}")
