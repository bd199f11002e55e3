"""debug: ms per Decima step in the early and the steady window (bench.py decima_in_loop's loop). usage: python tools/debug/decima_steady_time.py [envs]"""
import sys, time, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.decima import DecimaPolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
agent = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)), policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
dev = torch.device("cuda:0")
env = VecSparkSchedSimEnv(cfg, B, device=dev, pack=workload.default_pack(), auto_reset=True)
torch.manual_seed(0)
pol = DecimaPolicy(num_executors=10, **agent).to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(1)
env.reset(seed=0)
import os
SPIN = float(os.environ.get("SSS_SPIN_US", "0")) * 1e-6  # (extra host time per step: is the host on the critical path?)
def run(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        act, _ = pol.schedule_env(env, generator=gen)
        env.step_async(act["stage_idx"], act["num_exec"])
        if SPIN:
            t1 = time.perf_counter() + SPIN
            while time.perf_counter() < t1:
                pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
run(20)
early = run(100)
run(480)
steady = run(1200)
print(f"early {early:.4f} ms, steady {steady:.4f} ms per Decima step ({B} envs)")
