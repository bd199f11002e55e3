#!/bin/bash
# GPU box: per-event cycle breakdown of the event loop (see profiles/r01_bench.md). Builds the library
# with -DSSS_EVPROF (phase counters re-purposed: action = pop, reward = handler, observe = whole loop);
# a second build adds -DSSS_EXPERIMENT_NO_GATHER (durations not read from the table: WRONG results,
# timing only) to size the L2 gather inside the fast-path handler.
set -e
cd $GRAFT_REPO_ROOT
for extra in "" "-DSSS_EXPERIMENT_NO_GATHER"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DSSS_EVPROF $extra -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
  python bench.py --steps 400 --warmup 50 --no-cpu-baseline --single-mode --mode fused 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fused [$extra]', d['value'], d['phase_ticks_per_step'], d['events_per_step'], d['fast_path_event_frac'])"
done
