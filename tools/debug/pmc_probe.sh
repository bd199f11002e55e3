#!/bin/bash
# debug: instruction-cache and issue counters of the step kernel (one PMC pass per counter group)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|SQ_WAIT_INST|SQ_INSTS_(VALU|SALU|LDS|SMEM|VMEM)|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_ACTIVE_INST|SQ_INST_CYCLES|SQ_WAIT_ANY|SQ_WAVES" | head -60 > gpurun_out/pmc/avail.txt
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_BUSY_CYCLES" "SQ_IFETCH SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc/$tag -o t -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-decima --no-c3 --no-ppo --no-e100 --single-mode --preroll 300 > /dev/null 2> gpurun_out/pmc/$tag.err
done
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open("gpurun_out/pmc/summary.txt", "w") as out:
    for k, d in acc.items():
        if "sss_step" not in k: continue
        for c, v in sorted(d.items()):
            tail = v[-40:]
            out.write(f"{k[:30]:30s} {c:24s} n={len(v):4d} mean(last 40)={sum(tail)/len(tail):.1f}\n")
PY
