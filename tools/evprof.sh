#!/bin/bash
# GPU box: where an event-loop round's time goes. Builds the library with -DSSS_EVPROF (the header's
# phase counters are re-purposed: slow_events = batch part 1 (loads, classification, window min, ballot; also
# the whole of a round the batch path left early), action = batch member loop, events = batch draws +
# commit, reward = one-event pop, observe = one-event handler; pad0 = generator refills; pad1 = rounds),
# runs the fused bench and prints ticks per round. The regular library is rebuilt afterwards.
set -e
cd "$(dirname "$0")/.."
for cfg in c2 c3; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-function -DSSS_EVPROF "$@" -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
  python bench.py --config $cfg --steps 400 --warmup 100 --no-cpu-baseline --no-decima --single-mode --mode fused --evprof 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['evprof']; print('$cfg fused', round(d['value']/1e6,2), 'M steps/s; events/step', round(d['events_per_step'],1), 'batched', round(d['batched_event_frac'],2), 'x', round(d['events_per_batch'],2)); print(json.dumps(p))"
done
python -m spark_sched_sim_amd.build --force > /dev/null
