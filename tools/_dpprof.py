import sys, torch
sys.path.insert(0,'.')
from spark_sched_sim_amd import VecSparkSchedSimEnv
from spark_sched_sim_amd.decima import DecimaPolicy
AGENT = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)), policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
for B in (64, 1024):
    env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", auto_reset=True)
    pol = DecimaPolicy(num_executors=10, **AGENT).to("cuda:0").eval()
    env.reset(seed=0)
    for _ in range(300): env.step(env.policy_actions("fair"))
    acts, a = pol.act_env(env, 1, want_prof=True)
    torch.cuda.synchronize()
    p = a["prof"].double()
    print(B, "mean cycles per phase [analysis, prep, layers, summaries, stage, exec], depth, nodes:", [round(x) for x in p.mean(0).tolist()], "max total", int(p[:, :6].sum(1).max()))
