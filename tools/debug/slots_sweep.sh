#!/bin/bash
# debug: LDS job-cache slots vs occupancy at large job capacities (C3): builds with -DSSS_FALLBACK_SLOTS=n
set -e
cd "$(dirname "$0")/../.."
trap 'python -m spark_sched_sim_amd.build --force > /dev/null' EXIT
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -DSSS_FALLBACK_SLOTS=$n -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
  python bench.py --config c3 --steps 200 --warmup 50 --no-cpu-baseline --no-decima 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('slots $n: step', round(d['value']/1e6,2), 'M (launch', round(d['roofline']['avg_launch_ms'],3), 'ms) fused', round(d['other_mode']['value']/1e6,2), 'M')"
done
