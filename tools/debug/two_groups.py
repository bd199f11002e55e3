"""debug: Decima in the loop with the envs in G groups, each on its own stream (group A's step kernel under group B's policy pass)"""
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
pack = workload.default_pack()
for G in (1, 2, 4):
    envs = [VecSparkSchedSimEnv(cfg, B // G, device=dev, pack=pack, auto_reset=True) for _ in range(G)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    torch.manual_seed(0)
    pols = [DecimaPolicy(num_executors=10, **agent).to(dev).eval() for _ in range(G)]
    gens = [torch.Generator(device=dev).manual_seed(1 + k) for k in range(G)]
    for k, e in enumerate(envs):
        with torch.cuda.stream(streams[k]):
            e.reset(seed=k * (B // G))
    def run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            for k in range(G):
                with torch.cuda.stream(streams[k]):
                    act, _ = pols[k].schedule_env(envs[k], generator=gens[k])
                    envs[k].step_async(act["stage_idx"], act["num_exec"])
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        return time.perf_counter() - t0, th
    run(20)
    dt, th = run(100)
    run(480)
    dts, ths = run(1200)
    print(f"G={G}: early {1e3*dt/100:.3f} ms/step (host enqueue {1e3*th/100:.3f}), steady {1e3*dts/1200:.3f} ms/step (host {1e3*ths/1200:.3f}) -> {B*1200/dts/1e6:.2f} M env-steps/s", flush=True)
    for e in envs: e.close()
