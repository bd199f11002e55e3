#!/usr/bin/env python3
"""GPU soak with verification: 4096 envs auto-reset through many episodes in the fused rollout
kernel; afterwards the LAST finished episode of a sample of envs (episode k of env i has seed
base + i + k * num_envs) is re-run on the C oracle (tests/oracle_binding.py - test infrastructure)
and must match in steps, return and final wall time, bit for bit."""
import ctypes as C
import json
import os.path as osp
import sys

import numpy as np
import torch

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
sys.path[:0] = [ROOT, osp.join(ROOT, "tests")]
from oracle_binding import OracleEnv, SsoObsInfo  # noqa: E402
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload  # noqa: E402

CASES = [
    ("c2_hash0", dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash", 1, 60000),
    ("c3_fair", dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 40000),
    ("e64_fair", dict(num_executors=64, job_arrival_cap=100, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 30000),
    # round 4: the wave-uniform single-event paths and the pair staging under other regimes - random actions at 50 executors (backup
    # scheduling, sends), zero delays, bursts - and the wide instantiation
    ("e50_hash0", dict(num_executors=50, job_arrival_cap=60, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash", 1, 30000),
    ("e33_zero_delays_hash0", dict(num_executors=33, job_arrival_cap=80, job_arrival_rate=1.0e-4, moving_delay=0.0, warmup_delay=0.0), "hash", 1, 30000),
    ("burst_fair", dict(num_executors=20, job_arrival_cap=120, job_arrival_rate=4.0e-4, moving_delay=500.0, warmup_delay=100.0), "fair", 0, 30000),
    ("e100_fair_wide", dict(num_executors=100, job_arrival_cap=100, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 12000),
    # round 5: the wide instantiation's lane-parallel event machinery (two executors per lane: csrc/sss_sim.h lane_event) - random
    # actions (backup scheduling, sends, parks), every lane with two executors, zero delays, 65 executors (one lane with two)
    ("e100_hash0_wide", dict(num_executors=100, job_arrival_cap=100, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0), "hash", 1, 12000),
    ("e128_fair_wide", dict(num_executors=128, job_arrival_cap=80, job_arrival_rate=2.0e-4, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 10000),
    ("e128_hash0_wide", dict(num_executors=128, job_arrival_cap=60, job_arrival_rate=2.0e-4, moving_delay=500.0, warmup_delay=100.0), "hash", 1, 10000),
    ("e65_fair_wide", dict(num_executors=65, job_arrival_cap=100, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 12000),
    ("e90_zero_delays_hash0_wide", dict(num_executors=90, job_arrival_cap=80, job_arrival_rate=1.5e-4, moving_delay=0.0, warmup_delay=0.0), "hash", 1, 10000),
    # round 6: the "deep" trace set (a sixth element names the pack profile) - fast runs over jobs without a cache slot (more jobs with pending
    # events than LDS slots), 32-bit task counters in the thousands, both instantiations
    ("deep_c2_hash0", dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "hash", 1, 6000, "deep"),
    ("deep_c3_fair", dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 12000, "deep"),
    ("deep_e50_burst_hash0", dict(num_executors=50, job_arrival_cap=60, job_arrival_rate=1.0e-3, moving_delay=500.0, warmup_delay=100.0), "hash", 1, 6000, "deep"),
    ("deep_e100_fair_wide", dict(num_executors=100, job_arrival_cap=80, job_arrival_rate=2.0e-4, moving_delay=2000.0, warmup_delay=1000.0), "fair", 0, 5000, "deep"),
    ("deep_e128_burst_hash0_wide", dict(num_executors=128, job_arrival_cap=60, job_arrival_rate=1.0e-3, moving_delay=500.0, warmup_delay=100.0), "hash", 1, 4000, "deep"),
]


def bits(x):
    return np.float64(x).view(np.uint64)


def main():
    stride = int(sys.argv[1]) if len(sys.argv) > 1 else 64  # every stride-th env is replayed on the oracle (usage: soak_verify.py [stride] [case name part])
    only = sys.argv[2] if len(sys.argv) > 2 else ""
    B, base = 4096, 777
    for name, cfg, policy, pid, steps, *rest in CASES:
        if only not in name:
            continue
        pack = workload.profile_pack(rest[0] if rest else "default")
        env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=pack, auto_reset=True)
        env.reset(seed=base)
        for _ in range(steps // 1000):
            env.rollout(policy, 1000)
        torch.cuda.synchronize()
        ep = env.header_field("episodes").cpu().numpy()
        st = env.header_field("last_ep_steps").cpu().numpy()
        rt = env.header_field("last_ep_return").cpu().numpy()
        wl = env.header_field("last_ep_wall").cpu().numpy()
        err = env.obs_i32[:, 7].cpu().numpy()
        bad, checked = 0, 0
        for i in list(range(0, B, stride)):
            if ep[i] == 0 or err[i]:
                continue
            k = int(ep[i]) - 1
            o = OracleEnv(pack, cfg)
            r = C.c_double()
            n = o.lib.sso_run_episode(o.h, base + i + k * B, pid, 10 ** 9, C.byref(r))
            info = SsoObsInfo()
            o.lib.sso_obs_sizes(o.h, C.byref(info))
            ok = (int(st[i]), bits(rt[i]), bits(wl[i])) == (int(n), bits(r.value), bits(info.wall_time))
            bad += not ok
            checked += 1
            o.close()
        print(json.dumps({"case": name, "steps_per_env": steps, "episodes_total": int(ep.sum()), "min_episodes_per_env": int(ep.min()),
                          "envs_in_error": int((err != 0).sum()), "error_codes": sorted(set(err.tolist())), "checked": checked, "mismatches": bad}), flush=True)
        env.close()


if __name__ == "__main__":
    main()
