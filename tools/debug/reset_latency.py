"""debug: latency of ONE env's reset (masked sss_reset launch), C2 and C3 sizing"""
import sys, os.path as osp, time
ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT]
import torch
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
CFG = {"c2": dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0),
       "c3": dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)}
for name, cfg in CFG.items():
    env = VecSparkSchedSimEnv(cfg, 4096, device="cuda:0", pack=workload.default_pack())
    env.reset(seed=0)
    torch.cuda.synchronize()
    for nsel in (1, 4096):
        mask = torch.zeros(4096, dtype=torch.bool, device="cuda:0"); mask[:nsel] = True
        ts = []
        for it in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.reset(seed=100 * it, mask=mask); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(name, "reset of", nsel, "env(s): ms", [round(t, 3) for t in ts])
    env.close()
