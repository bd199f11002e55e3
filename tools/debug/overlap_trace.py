"""debug: from a rocprofv3 --kernel-trace CSV (kernel_trace.csv): how much of the sss_step_kernel launches' time other kernels ran beside them,
and the busy / idle time of the device over the traced window. usage: python tools/debug/overlap_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows]
ks.sort()
n = len(ks)
ks = ks[n // 3:]  # steady part
t0, t1 = ks[0][0], max(e for _, e, _, _ in ks)
steps = [(s, e) for s, e, nm, _ in ks if nm.startswith("sss_step_kernel")]
others = [(s, e) for s, e, nm, _ in ks if not nm.startswith("sss_step_kernel")]
def union(iv):
    out = []
    for s, e in sorted(iv):
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out
def overlap(a, b):
    i = j = 0; tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo: tot += hi - lo
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return tot
us, uo, ua = union(steps), union(others), union(steps + others)
ts, to, ta = (sum(e - s for s, e in u) for u in (us, uo, ua))
print(f"window {1e-6*(t1-t0):.2f} ms, {len(steps)} step launches; step kernels {1e-6*ts:.2f} ms, other kernels {1e-6*to:.2f} ms, any kernel {1e-6*ta:.2f} ms ({100*ta/(t1-t0):.1f}% of the window)")
print(f"other kernels running beside a step kernel: {1e-6*overlap(us, uo):.2f} ms = {100*overlap(us, uo)/max(ts,1):.1f}% of the step kernels' time")
print("queues:", sorted(set(q for _, _, _, q in ks)))
