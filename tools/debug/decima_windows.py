import os.path as osp, sys, time
sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.decima import DecimaPolicy
agent = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)), policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
env = VecSparkSchedSimEnv(cfg, 4096, device="cuda:0", pack=workload.default_pack(), auto_reset=True)
torch.manual_seed(0)
policy = DecimaPolicy(num_executors=10, **agent).to("cuda:0").eval()
gen = torch.Generator(device="cuda:0").manual_seed(1)
env.reset(seed=0)
done = 0
for upto in (20, 120, 320, 520, 800, 1200, 2000, 2200, 4000, 4200, 8000, 8200):
    n = upto - done
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        act, _ = policy.schedule_env(env, generator=gen)
        env.step(act)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"steps {done}..{upto}: {1e3*dt/n:.3f} ms/step, nodes/env now {int(env.obs_i32[:,0].sum())/4096:.1f}, episodes {int(env.header_field('episodes').sum())}")
    done = upto
