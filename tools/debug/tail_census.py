"""debug: what the event-loop rounds of the HEAVIEST steps consist of (CPU wave emulator, -DSSS_BATCH_STATS build): one env
stepped one call at a time, the census counters' deltas per step; steps are binned by their number of events and the bins'
means printed. usage: python tools/debug/tail_census.py [c2|c3] [steps] [seed]"""
import ctypes as C, os.path as osp, subprocess, sys
import numpy as np
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT, osp.join(ROOT, "tests")]
subprocess.run(["make", "-s", "-C", osp.join(ROOT, "tests", "emu"), "../_build/libsss_emu_stats.so"], check=True)
from emu_util import load_emu
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
CFG = {"c2": (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash"),
       "c3": (dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair")}
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 11
cfg, pol = CFG[name]
lib = load_emu("_stats")
env = VecSparkSchedSimEnv(cfg, 1, device="cpu", pack=workload.default_pack(), _lib=lib, auto_reset=True)
env.reset(seed=[seed])
stats = (C.c_longlong * 128).in_dll(lib, "sss_batch_stats")
rows = []
prev = np.array(list(stats), dtype=np.int64)
ev_prev = env.counters()["n_events"]
for t in range(steps):
    env.step(env.policy_actions(pol))
    cur = np.array(list(stats), dtype=np.int64)
    ev = env.counters()["n_events"]
    rows.append(np.concatenate([[ev - ev_prev], cur - prev]))
    prev, ev_prev = cur, ev
rows = np.array(rows)
LABEL = {1 + 93: "fast-run events", 1 + 90: "fast_run calls", 1 + 31: "released batches committed", 1 + 33: "  their members", 1 + 46: "  freed members",
         1 + 32: "  pools through the staging area", 1 + 64: "released: exit none/single/not head", 1 + 65: "released: exit head no member (commitment)",
         1 + 66: "released: exit classify", 1 + 67: "released: exit enters source", 1 + 68: "released: exit window empty", 1 + 69: "released: exit cut to nothing",
         1 + 70: "released: exit lemire", 1 + 34: "arrival batches committed", 1 + 35: "  their members", 1 + 36: "  job pools staged", 1 + 37: "  stage pools staged",
         1 + 38: "  parked", 1 + 80: "arrival: exit not head", 1 + 81: "arrival: exit head not member", 1 + 43: "single TASK_FINISHED (slow path)",
         1 + 39: "single EXECUTOR_READY", 1 + 100: "trk_move calls", 1 + 102: "  old pool > 8 slots", 1 + 104: "  new pool > 8 slots", 1 + 53: "fulfil chunks", 1 + 55: "  items in chunks",
         1 + 56: "fulfil serial items", 1 + 110: "n_commits at single TF (sum)", 1 + 24: "single TF: no commitment", 1 + 25: "single TF: to common", 1 + 26: "single TF: send",
         1 + 27: "single TF: park", 1 + 28: "single TF: start", 1 + 29: "single TF: completes stage",
         1 + 120: "lean_released tried", 1 + 121: "  exit: not such an event / no commitment", 1 + 122: "  exit: classification", 1 + 111: "lean_released handled",
         1 + 112: "  start", 1 + 113: "  park", 1 + 114: "  send", 1 + 115: "  idle -> job pool", 1 + 116: "  idle -> common pool", 1 + 117: "  of those: to a backup stage / none found", 1 + 125: "lean_arrival handled", 1 + 126: "  start", 1 + 118: "job completion: pool flushed by the wave", 1 + 119: "  executors", 1 + 127: "released batches: uniform path", 1 + 123: "fulfil common suffix (wave)", 1 + 124: "  items"}
ev = rows[:, 0]
if len(sys.argv) > 4 and sys.argv[4] == "cost":
    # bins by an estimate of the step's ticks (kilo-ticks per item from the GPU's scoped profile) instead of its events
    c = lambda i: rows[:, 1 + i].astype(float)
    ev = (1.2 * c(93) + 1.5 * c(90) + 14.5 * (c(43) + c(39)) + 3.0 * (c(64) + c(65) + c(66) + c(67)) + 22.0 * c(31) + 0.3 * c(33) + 15.5 * c(34) + 40.0).astype(int)
print(f"{name}: {steps} steps, events/step mean {ev.mean():.1f}, p99 {np.percentile(ev, 99):.0f}, max {ev.max()}")
bins = [(0, 20), (20, 60), (60, 120), (120, 200), (200, 10**9)] if not (len(sys.argv) > 4 and sys.argv[4] == "cost") else [(0, 150), (150, 300), (300, 450), (450, 600), (600, 10**9)]
hdr = "".join(f"{f'[{a},{b if b < 10**8 else chr(8734)})':>14s}" for a, b in bins)
print(f"{'per step, by events in the step':46s}{hdr}")
print(f"{'steps in bin':46s}" + "".join(f"{int(((ev >= a) & (ev < b)).sum()):14d}" for a, b in bins))
for col, lab in LABEL.items():
    vals = []
    for a, b in bins:
        m = (ev >= a) & (ev < b)
        vals.append(rows[m, col].mean() if m.any() else float("nan"))
    if any(v == v and v > 0 for v in vals):
        print(f"{lab:46s}" + "".join(f"{v:14.2f}" for v in vals))
