#!/bin/bash
# debug: -DSSS_EVPROF -DSSS_EVPROF2: share of pool moves / idle-set building in the one-event handler time
set -e
cd "$(dirname "$0")/../.."
for cfg in c2 c3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -DSSS_EVPROF -DSSS_EVPROF2 -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
  python bench.py --config $cfg --steps 400 --warmup 100 --no-cpu-baseline --no-decima --single-mode --mode fused --evprof 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['evprof']; r=p['rounds_per_step']; t=p['ticks_per_round']
print('$cfg per step: single_handler', round(t['single_handler']*r), 'pool moves', round(t['rng_refill']*r), 'idle sets (ticks; counted in evprof_rounds slot)', p['rounds_per_step'])"
done
python -m spark_sched_sim_amd.build --force > /dev/null
