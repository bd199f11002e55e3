#!/bin/bash
# debug helper: rebuilds the HIP library with extra -D flags on the GPU box and runs find_bad_env.py
set -e
cd "$(dirname "$0")/../.."
trap 'python -m spark_sched_sim_amd.build --force > /dev/null' EXIT  # the variants overwrite the product library: put it back
mkdir -p gpurun_out/dbg
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -I spark_sched_sim_amd/csrc $v -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
  echo "== variant [$v]"
  python tools/debug/find_bad_env.py 4096 2>&1 | grep -v amdgpu.ids | cut -c1-150 | head -8
done
