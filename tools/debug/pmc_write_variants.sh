cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_pmc
cp spark_sched_sim_amd/csrc/libsss_hip.so /tmp/prod.so
for v in prod nolean syncwg pair17; do
  if [ $v != prod ]; then cp tests/_build/libsss_hip_$v.so spark_sched_sim_amd/csrc/libsss_hip.so; else cp /tmp/prod.so spark_sched_sim_amd/csrc/libsss_hip.so; fi
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r04_pmc/$v -o t -- python3 tools/debug/traffic_probe.py > /dev/null 2>&1
  python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("gpurun_out/r04_pmc/$v/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="WRITE_SIZE" and r["Kernel_Name"].startswith("sss_step_kernel"): rows.append(float(r["Counter_Value"]))
print("$v", "launches", len(rows), "mean KiB", sum(rows[-100:])/max(1,len(rows[-100:])))
PY
done
cp /tmp/prod.so spark_sched_sim_amd/csrc/libsss_hip.so
