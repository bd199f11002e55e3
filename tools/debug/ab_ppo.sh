#!/bin/bash
# A/B timing of the PPO update's kernel switches (spark_sched_sim_amd/train_kernels.py) on one box: tools/debug/ab_ppo.sh
cd $GRAFT_REPO_ROOT
for sw in "FUSED_WGRAD=1" "FUSED_HEAD_WGRAD=0" "CONCAT_ONE_LAUNCH=0" "SPLIT_INPUT=0" "FUSED_WGRAD=1"; do
echo "== $sw"
timeout 600 python tools/bench_ppo.py --iterations 3 --kernel-switch $sw 2>&1 | tail -1 | python -c "
import json,sys
for r in json.loads(sys.stdin.readline()): print(r['collect_s'], r['train_s'], r['samples'])"
done
