#!/usr/bin/env python3
"""debug: what a launch of sss_step_bounded waits for - per-env shader ticks by phase (action + fulfil, event loop, reward,
observation) of single launches, the slowest env's split, against the launch's duration. usage: bounded_probe.py c2|c3 budget"""
import os.path as osp
import sys

import numpy as np
import torch

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload  # noqa: E402
from spark_sched_sim_amd.vec_env import HDR_OFF, HDR_PROF  # noqa: E402

config, budget = sys.argv[1], int(sys.argv[2])
cfg = bench.CONFIGS[config]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
e = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=workload.default_pack(), auto_reset=True, seed_stride=B)
e.reset(seed=0)
left = bench.PREROLL_STEPS[config]
while left > 0:
    e.rollout(bench.DEFAULT_POLICY[config], min(500, left))
    left -= 500
torch.cuda.synchronize()


def snap():
    h = e._env_view[:, : e.dims.hdr_bytes].cpu().numpy()
    prof = np.ascontiguousarray(h[:, HDR_PROF: HDR_PROF + 40]).view(np.uint64).astype(np.int64)
    ev = np.ascontiguousarray(h[:, HDR_OFF["n_events"]: HDR_OFF["n_events"] + 8]).view(np.uint64).ravel().astype(np.int64)
    return prof, ev


pol = bench.DEFAULT_POLICY[config]
for _ in range(200):
    a = e.policy_actions(pol)
    e.step_bounded_async(a["stage_idx"], a["num_exec"], budget) if budget else e.step_async(a["stage_idx"], a["num_exec"])
rows = []
for _ in range(24):
    p0, ev0 = snap()
    a = e.policy_actions(pol)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    ready = e.step_bounded_async(a["stage_idx"], a["num_exec"], budget) if budget else None
    if not budget:
        e.step_async(a["stage_idx"], a["num_exec"])
    t1.record()
    torch.cuda.synchronize()
    p1, ev1 = snap()
    d = p1 - p0
    tot = d[:, 1:].sum(1)
    k = int(tot.argmax())
    rows.append((t0.elapsed_time(t1), tot.mean(), tot.max(), d[k, 1], d[k, 2], d[k, 3], d[k, 4], (ev1 - ev0)[k], (ev1 - ev0).mean(), (ev1 - ev0).max(),
                 float(ready.float().mean()) if ready is not None else 1.0))
r = np.array(rows)
print(f"{config} budget {budget}: launch {r[:,0].mean():.3f} ms = {r[:,0].mean()*2.1e6/1e3:.0f} k ticks at 2.1 GHz; recorded env ticks mean {r[:,1].mean()/1e3:.0f} k, slowest {r[:,2].mean()/1e3:.0f} k "
      f"(action {r[:,3].mean()/1e3:.0f} k, events {r[:,4].mean()/1e3:.0f} k, reward {r[:,5].mean()/1e3:.0f} k, observe {r[:,6].mean()/1e3:.0f} k; its events {r[:,7].mean():.1f}); "
      f"events per env mean {r[:,8].mean():.1f} max {r[:,9].mean():.1f}; ready {r[:,10].mean():.3f}")
