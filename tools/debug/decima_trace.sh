#!/bin/bash
# debug: kernel time vs wall time of a Decima-in-the-loop step (are the launches / host syncs the bound?)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dtrace
for n in 4096 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dtrace/e$n -o t -- python3 tools/bench_decima.py --envs $n --steps 100 --warmup 20 > gpurun_out/dtrace/e$n.json 2> gpurun_out/dtrace/e$n.err
  python3 tools/bench_decima.py --envs $n --steps 100 --warmup 20 > gpurun_out/dtrace/e${n}_plain.json 2>/dev/null
done
python3 - <<'PY'
import csv, glob, json
for n in (4096, 1024):
    f = glob.glob(f"gpurun_out/dtrace/e{n}/**/*kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    calls = sum(int(r["Calls"]) for r in rows)
    d = json.load(open(f"gpurun_out/dtrace/e{n}_plain.json"))
    print(n, "envs: kernel time per step %.3f ms, launches per step %.1f, wall per step (unprofiled) %.3f ms, %.2f M env-steps/s" % (tot / 120 / 1e6, calls / 120, d["ms_per_step"], d["value"] / 1e6))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:26]:
        print("   %-46s calls/step %5.1f  avg %7.1f us  per step %7.1f us" % (r["Name"][:46], int(r["Calls"]) / 120, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 120 / 1e3))
PY
