#!/bin/bash
# debug: which events still go one at a time in the slowest env of a step launch (-DSSS_TAILSTAT build)
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -DSSS_TAILSTAT -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
python tools/debug/tail_env.py "$@" 2>&1 | grep -v amdgpu.ids
python -m spark_sched_sim_amd.build --force > /dev/null
