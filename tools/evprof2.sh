set -e
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DSSS_EVPROF2 -I spark_sched_sim_amd/csrc -o spark_sched_sim_amd/csrc/libsss_hip.so spark_sched_sim_amd/csrc/sss_hip.hip
python bench.py --steps 400 --warmup 50 --no-cpu-baseline --single-mode --mode fused 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('fused', d['value'], d['phase_ticks_per_step'], d['events_per_step'], d['fast_path_event_frac'])"
