"""debug helper: one-line summaries of bench.py JSON lines"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "ERR", e); continue
    def line(tag, m, top=None):
        r = m["roofline"]
        print(f"{tag:14s} {m['value']/1e6:7.2f}M steps/s  {m['ms_per_step']:.3f} ms/step  launch {r['avg_launch_ms']:.3f} ms  ev/step {m['events_per_step']:.1f}"
              f"  batched {m.get('batched_event_frac', 0):.2f} x{m.get('events_per_batch', 0):.2f}  ticks {({a: int(b) for a, b in m['phase_ticks_per_step'].items()})}")
    print(f)
    line(d["config"]["mode"], d)
    if "other_mode" in d: line(d["other_mode"]["mode"], d["other_mode"])
    if "c3" in d:
        line("c3 " + d["c3"].get("mode", "step"), d["c3"])
        if "other_mode" in d["c3"]: line("c3 " + d["c3"]["other_mode"]["mode"], d["c3"]["other_mode"])
        if "step_tail" in d["c3"]: print("   c3 tail", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d["c3"]["step_tail"].items() if k != "what"})
    if "cpu_baseline" in d: print("   cpu 1 core %.2fM, all cores %.2fM (%s)" % (d["cpu_baseline"]["value"]/1e6, d.get("cpu_baseline_all_cores", {}).get("value", 0)/1e6, d.get("cpu_baseline_all_cores", {}).get("cores")))
