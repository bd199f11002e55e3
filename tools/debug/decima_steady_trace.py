"""debug: per-kernel time of the Decima step in the STEADY window (after 600 steps), from a rocprofv3 --kernel-trace CSV of
tools/debug/decima_steady_run.py. usage: python tools/debug/decima_steady_trace.py <kernel_trace.csv> [steps_in_tail]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
tail_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sss_step_kernel")]
start = idx[-tail_steps - 1]
tail = rows[start:idx[-1] + 1]
t0, t1 = int(tail[0]["Start_Timestamp"]), int(tail[-1]["End_Timestamp"])
agg = collections.defaultdict(lambda: [0, 0, 0])
prev_end = None
for r in tail:
    a = agg[r["Kernel_Name"][:70]]
    a[0] += 1; a[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if prev_end is not None:
        a[2] += max(0, int(r["Start_Timestamp"]) - prev_end)  # device idle right before this kernel
    prev_end = max(prev_end or 0, int(r["End_Timestamp"]))
busy = sum(v[1] for v in agg.values())
print(f"{tail_steps} steps: {1e-3*(t1-t0)/tail_steps:.1f} us per step wall, {1e-3*busy/tail_steps:.1f} us of kernels per step")
for k, (n, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:70s} {n/tail_steps:6.2f} per step  {1e-3*t/n:8.1f} us each  {1e-3*t/tail_steps:8.1f} us per step  {1e-3*g/tail_steps:6.1f} us idle before")
