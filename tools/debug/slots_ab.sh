#!/bin/bash
# A/B on the GPU box: product library vs slot-count variants at 1024 envs, config-3 sizing
cd $GRAFT_REPO_ROOT
cp spark_sched_sim_amd/csrc/libsss_hip.so /tmp/product.so
for v in product slots24 slots48; do
  if [ $v != product ]; then cp tests/_build/libsss_hip_$v.so spark_sched_sim_amd/csrc/libsss_hip.so; fi
  python bench.py --config c3 --envs 1024 --steps 400 --warmup 50 --no-cpu-baseline --no-decima --single-mode --sustained-s 0 --bounded-events 0 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v: 1024 envs step', round(d['value']/1e6,2), 'M, launch', round(d['roofline']['avg_launch_ms'],4), 'ms')"
  python tools/profile_ppo_collect.py --sync --rows 3 2>/dev/null | grep "^iterations"
done
cp /tmp/product.so spark_sched_sim_amd/csrc/libsss_hip.so
