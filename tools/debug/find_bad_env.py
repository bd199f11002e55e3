"""debug helper: runs the C2 / hash full-batch case on the GPU and prints the envs that failed or did not finish"""
import sys, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT, osp.join(ROOT, "tests")]
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
C2 = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecSparkSchedSimEnv(C2, B, device="cuda:0", pack=workload.default_pack())
env.reset(seed=31000)
for it in range(8):
    env.rollout("hash", 200)
torch.cuda.synchronize()
term = env.header_field("terminated").cpu().numpy()
err = env.obs_i32[:, 7].cpu().numpy()
herr = env.header_field("err").cpu().numpy()
steps = env.header_field("ep_steps").cpu().numpy()
bad = [i for i in range(B) if term[i] == 0]
print("not terminated:", len(bad))
for i in bad[:20]:
    print("env", i, "seed", 31000 + i, "err(obs)", err[i], "err(hdr)", herr[i], "ep_steps", steps[i], env.header(i))
