"""debug: a short steady-state step-mode run at C2 for PMC passes (no parity checks: used with builds that break the semantics on purpose)"""
import sys, os.path as osp
sys.path[:0] = [osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))]
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
env = VecSparkSchedSimEnv(cfg, 4096, device="cuda:0", pack=workload.default_pack(), auto_reset=True)
env.reset(seed=0)
env.rollout("hash", 1500)
for _ in range(150):
    a = env.policy_actions("hash")
    env.step_async(a["stage_idx"], a["num_exec"])
torch.cuda.synchronize()
