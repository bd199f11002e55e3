import sys, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
cfg = dict(num_executors=5, job_arrival_cap=8, job_arrival_rate=1.0e-4, moving_delay=1500.0, warmup_delay=500.0)
env = VecSparkSchedSimEnv(cfg, 2, device="cuda:0", pack=workload.default_pack())
print("created", flush=True)
env.reset(seed=1)
torch.cuda.synchronize()
print("reset ok", env.obs_i32.cpu(), flush=True)
env.rollout("fair", 5)
torch.cuda.synchronize()
print("rollout ok", env.obs_i32.cpu(), flush=True)
