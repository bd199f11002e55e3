#!/bin/bash
# rocprofv3 kernel trace + PMC traffic of the round-2 build (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof/step_c2 -o r02_step -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-decima --no-c3 --single-mode > gpurun_out/r02_prof/step_c2.json 2> gpurun_out/r02_prof/step_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof/fused_c2 -o r02_fused -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-decima --no-c3 --single-mode --mode fused > gpurun_out/r02_prof/fused_c2.json 2> gpurun_out/r02_prof/fused_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof/step_c3 -o r02_step_c3 -- python3 bench.py --config c3 --steps 200 --warmup 50 --no-cpu-baseline --no-decima --single-mode > gpurun_out/r02_prof/step_c3.json 2> gpurun_out/r02_prof/step_c3.err
python3 tools/collect_traffic.py > gpurun_out/r02_prof/traffic.log 2>&1
find gpurun_out/r02_prof -name "*kernel_stats.csv" | head
