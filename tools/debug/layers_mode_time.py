"""debug: the DAG layers as one launch (a wave per observation) against a launch per layer, by observation size: ms per Decima step in
windows of a run that starts with empty envs (graphs grow from ~10 to ~200 nodes per observation at config 2), for layers_mode 1 / 2 / 0.
usage: python tools/debug/layers_mode_time.py [envs] [c2|c3]"""
import sys, time, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.decima import DecimaPolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name = sys.argv[2] if len(sys.argv) > 2 else "c2"
E, J = (10, 50) if name == "c2" else (50, 200)
cfg = dict(num_executors=E, job_arrival_cap=J, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
agent = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)), policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
dev = torch.device("cuda:0")
for mode in [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "120")]:
    env = VecSparkSchedSimEnv(cfg, B, device=dev, pack=workload.default_pack(), auto_reset=True)
    torch.manual_seed(0)
    pol = DecimaPolicy(num_executors=E, **agent).to(dev).eval()
    pol._layers_mode = mode
    gen = torch.Generator(device=dev).manual_seed(1)
    env.reset(seed=0)
    row = []
    for w in range(8):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            act, _ = pol.schedule_env(env, generator=gen)
            env.step_async(act["stage_idx"], act["num_exec"])
        torch.cuda.synchronize()
        row.append(f"{(time.perf_counter() - t0) / 40 * 1e3:.3f} ms @ {int(env.obs_i32[:, 0].sum()) / B:.0f} nodes (max {int(env.obs_i32[:, 0].max())})")
    print(f"layers_mode {mode} ({name}, {B} envs): " + " | ".join(row), flush=True)
    env.close()
