"""debug: where a wave of the one-launch DAG layers (csrc/sss_gnn_mfma.h sss_gnn_layers_obs_kernel) spends its cycles - the
-DGNN_OBS_PROF timing build (tests/gpu_variant.py obsprof). usage: python tools/debug/layers_obs_prof.py [envs] [steps before]"""
import ctypes as C, sys, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT, osp.join(ROOT, "tests")]
import torch
from gpu_variant import load_variant
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.decima import DecimaPolicy
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
before = int(sys.argv[2]) if len(sys.argv) > 2 else 800
lib = load_variant("obsprof")
cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
agent = dict(embed_dim=16, gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(negative_slope=0.2)), policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
dev = torch.device("cuda:0")
env = VecSparkSchedSimEnv(cfg, B, device=dev, pack=workload.default_pack(), auto_reset=True, _lib=lib)
torch.manual_seed(0)
pol = DecimaPolicy(num_executors=10, **agent).to(dev).eval()
gen = torch.Generator(device=dev).manual_seed(1)
env.reset(seed=0)
pol.bind_kernels(env._b)
pol._layers_mode = 2
for _ in range(before):
    act, _ = pol.schedule_env(env, generator=gen)
    env.step_async(act["stage_idx"], act["num_exec"])
torch.cuda.synchronize()
g = env.decima_graph()
out = (C.c_ulonglong * 16)()
pol._encode_kernels(g)
torch.cuda.synchronize()
lib.sss_debug_obs_prof(out)
pol._encode_kernels(g)
torch.cuda.synchronize()
lib.sss_debug_obs_prof(out)
v = list(out)
waves, tiles, groups, slots = v[11], v[10], v[13], v[12]
print(f"{g['x'].shape[0]} nodes, {B} observations; {waves} waves with layers, {tiles} tiles ({tiles / max(waves, 1):.1f} per wave), {groups} slot groups, {slots} slots evaluated")
print(f"wave total: mean {v[0] / max(waves, 1):.0f} cycles, max {v[1]}")
names = {2: "set-up + weights", 3: "lists", 4: "tile: loads of the row's node", 5: "tile: edge slots' loads", 6: "tile: children's loads", 7: "tile: message MLPs", 8: "tile: aggregate + update + store", 9: "fence"}
for i, nm in names.items():
    per = "per tile" if i in (4, 8) else "per slot group" if i in (5, 6, 7) else "per wave"
    d = tiles if i in (4, 8) else groups if i in (5, 6, 7) else waves
    print(f"  {nm:36s} {v[i] / max(waves, 1):9.0f} cycles per wave   {v[i] / max(d, 1):8.0f} {per}")
