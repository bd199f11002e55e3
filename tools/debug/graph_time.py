"""debug: time of the Decima graph kernel alone (offsets scan + graph build on live observations), on the product library or on test
builds of it (names from tests/gpu_variant.py VARIANTS; profiles/r05_graph_kernel.txt was made with builds that ended the kernel after a
phase). usage: python tools/debug/graph_time.py [envs] [c2|c3] [product|variant ...]"""
import sys, os.path as osp
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT, osp.join(ROOT, "tests")]
import torch
from gpu_variant import load_variant
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
name = sys.argv[2] if len(sys.argv) > 2 else "c2"
E, J = (10, 50) if name == "c2" else (50, 200)
cfg = dict(num_executors=E, job_arrival_cap=J, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
dev = torch.device("cuda:0")
for var in (sys.argv[3:] or ["product"]):
    lib = None if var == "product" else load_variant(var)
    env = VecSparkSchedSimEnv(cfg, B, device=dev, pack=workload.default_pack(), auto_reset=True, _lib=lib)
    env.reset(seed=0)
    env.rollout("fair", 1500)
    torch.cuda.synchronize()
    nodes = int(env.obs_i32[:, 0].sum())
    for _ in range(5):
        env.decima_graph_on_device()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        env.decima_graph_on_device()
    e1.record()
    torch.cuda.synchronize()
    print(f"{var:10s} {name} {B} envs, {nodes / B:.1f} nodes per env: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (scan + clear + graph kernel)", flush=True)
    env.close()
