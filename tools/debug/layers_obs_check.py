"""debug: the DAG layers in one launch (a wave per observation, layers_mode 2) against the launch per layer (layers_mode 1): the
encoder's three outputs must be bit-identical on live observations; then the time of a Decima step either way.
usage: python tools/debug/layers_obs_check.py [envs] [c2|c3]"""
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
env = VecSparkSchedSimEnv(cfg, B, device=dev, pack=workload.default_pack(), auto_reset=True)
torch.manual_seed(0)
pol = DecimaPolicy(num_executors=E, **agent).to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(1)
env.reset(seed=0)
pol.bind_kernels(env._b)
bad = 0
for step in range(900):
    if step % 100 == 0:
        g = env.decima_graph()  # exact sizes: new buffers per call
        outs = []
        for mode in (1, 2):
            pol._layers_mode = mode
            h = pol._encode_kernels(g)
            outs.append({k: v.clone() for k, v in h.items()})
        torch.cuda.synchronize()
        same = all(torch.equal(outs[0][k].view(torch.int32), outs[1][k].view(torch.int32)) for k in outs[0])
        bad += not same
        lc = g["layer_cnt"][:int(g["max_depth"])].long()
        tiles = ((lc + 15) // 16).sum().item() / B
        ts = []
        for mode in (1, 2):
            pol._layers_mode = mode
            for _ in range(3):
                pol._encode_kernels(g)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                pol._encode_kernels(g)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"step {step}: {g['x'].shape[0]} nodes, depth {int(g['max_depth'])}: {'identical' if same else 'DIFFERENT'}; list entries per layer {lc.sum(1).tolist()}, "
              f"{lc.sum().item() / B:.1f} per observation in {tiles:.1f} tiles; largest observation {int(g['obs_nodes'].max())} nodes; encode {ts[0]:.0f} us per layer launches, {ts[1]:.0f} us one launch", flush=True)
    pol._layers_mode = 2 if step % 2 else 1
    act, _ = pol.schedule_env(env, generator=gen)
    env.step_async(act["stage_idx"], act["num_exec"])
torch.cuda.synchronize()
for mode in (1, 2, 1, 2):
    pol._layers_mode = mode
    for _ in range(20):
        act, _ = pol.schedule_env(env, generator=gen)
        env.step_async(act["stage_idx"], act["num_exec"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        act, _ = pol.schedule_env(env, generator=gen)
        env.step_async(act["stage_idx"], act["num_exec"])
    torch.cuda.synchronize()
    print(f"layers_mode {mode}: {(time.perf_counter() - t0) / 300 * 1e3:.4f} ms per Decima step ({B} envs, {name})", flush=True)
print("mismatching checks:", bad)
sys.exit(1 if bad else 0)
