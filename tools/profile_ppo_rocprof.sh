#!/bin/bash
# rocprofv3 kernel stats of ONE collection and ONE update of the PPO iteration at one rank's share of BASELINE config 5
# (1024 envs, 50 executors, 200 jobs; runs on the GPU box):   tools/profile_ppo_rocprof.sh r06
# Two traced runs from the same seed: a collection alone, and a collection followed by the update; the update's kernels are the
# second run's minus the first's, kernel by kernel (tools/kernel_stats_diff.py). Outputs under gpurun_out/<round>_prof/:
#   <round>_ppo_collect_kernel_stats.csv   <round>_ppo_train_kernel_stats.csv   (copy them to profiles/)
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
D=gpurun_out/${R}_prof
mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D/ppo_collect -o ppo_collect -- python3 tools/bench_ppo.py --iterations 1 --no-train > $D/ppo_collect.json 2> $D/ppo_collect.err
rocprofv3 --kernel-trace --stats --output-format csv -d $D/ppo_iter -o ppo_iter -- python3 tools/bench_ppo.py --iterations 1 > $D/ppo_iter.json 2> $D/ppo_iter.err
C=$(find $D/ppo_collect -name "*kernel_stats.csv" | head -1)
I=$(find $D/ppo_iter -name "*kernel_stats.csv" | head -1)
cp "$C" $D/${R}_ppo_collect_kernel_stats.csv
python3 tools/kernel_stats_diff.py "$I" "$C" > $D/${R}_ppo_train_kernel_stats.csv
cat $D/ppo_collect.json $D/ppo_iter.json
head -12 $D/${R}_ppo_collect_kernel_stats.csv | cut -c1-200
head -16 $D/${R}_ppo_train_kernel_stats.csv | cut -c1-200
