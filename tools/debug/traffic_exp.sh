#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/texp
for flag in NONE TRAFFIC_NO_SLOT_WB TRAFFIC_NO_OBS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -D$flag -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/texp/${flag}_$c -o t -- python3 tools/debug/traffic_probe.py > /dev/null 2>&1
  done
done
python3 - <<'PY'
import csv, glob
for flag in ("NONE", "TRAFFIC_NO_SLOT_WB", "TRAFFIC_NO_OBS"):
    out = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/texp/{flag}_{c}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] == c and "sss_step_kernel" in row["Kernel_Name"]:
                    vals.append(float(row["Counter_Value"]))
        tail = vals[-100:]
        out[c] = sum(tail) / max(len(tail), 1)
    print(flag, "fetch KiB/launch %.0f (x2 = %.1f MB) write KiB/launch %.0f (%.1f MB)" % (out["FETCH_SIZE"], out["FETCH_SIZE"] * 2 * 1024 / 1e6, out["WRITE_SIZE"], out["WRITE_SIZE"] * 1024 / 1e6))
PY
