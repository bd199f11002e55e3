#!/bin/bash
# debug: scoped profile of the lane-0 procedures. usage: evprof3.sh c2,c3 [fused|step]
set -e
cd "$(dirname "$0")/../.."
trap 'python -m spark_sched_sim_amd.build --force > /dev/null' EXIT  # put the product library back whatever happens
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -DSSS_EVPROF3 $SSS_EXTRA_FLAGS -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
python tools/debug/evprof3.py "$@" 2>&1 | grep -v amdgpu.ids
