#!/bin/bash
# rocprofv3 kernel trace (+ stats) of the simulator kernels for a round's profiles/ (runs on the GPU box):
#   tools/profile_round.sh r03
# step / fused launches at BASELINE config 2 sizing and step launches at config 3; the *_kernel_stats.csv files are what
# gets copied to profiles/<round>_*.csv. HBM traffic (PMC passes) is collected separately by tools/collect_traffic.py.
R=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/${R}_prof
mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/step_c2 -o step_c2 -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-decima --no-c3 --no-ppo --no-e100 --no-deep --single-mode --sustained-s 0 --bounded-events 0 > $D/step_c2.json 2> $D/step_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/fused_c2 -o fused_c2 -- python3 bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-decima --no-c3 --no-ppo --no-e100 --no-deep --single-mode --mode fused --sustained-s 0 --bounded-events 0 > $D/fused_c2.json 2> $D/fused_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/step_c3 -o step_c3 -- python3 bench.py --config c3 --steps 200 --warmup 50 --no-cpu-baseline --no-decima --no-ppo --no-e100 --no-deep --single-mode --sustained-s 0 --bounded-events 0 > $D/step_c3.json 2> $D/step_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/step_e100 -o step_e100 -- python3 bench.py --config e100 --envs 1024 --steps 200 --warmup 50 --no-cpu-baseline --single-mode --sustained-s 0 --bounded-events 0 > $D/step_e100.json 2> $D/step_e100.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/decima -o decima -- python3 tools/bench_decima.py --envs 4096 --steps 120 > $D/decima.json 2> $D/decima.err
find $D -name "*kernel_stats.csv" | while read f; do echo "== $f"; head -6 "$f" | cut -c1-160; done
