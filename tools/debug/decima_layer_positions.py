"""debug: the DAG-layer launches of a Decima step by POSITION (first launch = deepest layer), averaged over the steady window of a
rocprofv3 --kernel-trace CSV of tools/debug/decima_steady_run.py. usage: python tools/debug/decima_layer_positions.py <kernel_trace.csv> [steps]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tail_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("sss_step_kernel")]
tail = rows[idx[-tail_steps - 1]:idx[-1] + 1]
pos, sums, gaps, prev_end = 0, {}, {}, None
for r in tail:
    nm = r["Kernel_Name"]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if nm.startswith("sss_gnn_layer_mfma_kernel"):
        a = sums.setdefault(pos, [0, 0, 0])
        a[0] += 1; a[1] += e - s; a[2] += s - prev_end if prev_end else 0
        pos += 1
    else:
        pos = 0
    prev_end = e
for p, (n, t, g) in sorted(sums.items()):
    print(f"  layer launch {p}: {1e-3 * t / n:7.1f} us each, {1e-3 * g / n:5.1f} us idle before it ({n} launches)")
print(f"  sum {sum(1e-3 * t / n for n, t, g in sums.values()):.1f} us + idle {sum(1e-3 * g / n for n, t, g in sums.values()):.1f} us")
