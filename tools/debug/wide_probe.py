"""GPU debugging aid: B envs of a config stepped one call at a time on cuda:0 next to the C oracle (same on-device policy's
actions fed to both); prints the first step at which an env's error code / reward / wall time / node rows differ, with the
source line of the kernel-side check that failed (header field err_line).
    python tools/debug/wide_probe.py E J policy [B] [max_steps] [rate]"""
import os.path as osp
import sys

import numpy as np
import torch

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path[:0] = [ROOT, osp.join(ROOT, "tests")]
from golden_util import bits  # noqa: E402
from oracle_binding import OracleEnv  # noqa: E402
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload  # noqa: E402

E, J, pol = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
B = int(sys.argv[4]) if len(sys.argv) > 4 else 8
T = int(sys.argv[5]) if len(sys.argv) > 5 else 3000
rate = float(sys.argv[6]) if len(sys.argv) > 6 else 1.0e-4
fused = len(sys.argv) > 7 and sys.argv[7] == "fused"
pack = workload.default_pack()
cfg = dict(num_executors=E, job_arrival_cap=J, job_arrival_rate=rate, moving_delay=2000.0, warmup_delay=1000.0)
env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=pack)
env.reset(seed=7000)
print("after reset: err", env.obs_i32[:, 7].tolist()[:8], "line", env.header_field("err_line").tolist()[:8])
if fused:
    for k in range(20):
        env.rollout(pol, 50)
        err = env.obs_i32[:, 7].cpu().numpy()
        if err.any():
            print("fused: after", 50 * (k + 1), "steps err", err.tolist()[:16], "lines", env.header_field("err_line").tolist()[:16], "ep_steps", env.header_field("ep_steps").tolist()[:16])
            break
    else:
        print("fused ok")
    sys.exit(0)
os_ = [OracleEnv(pack, cfg) for _ in range(B)]
for k, o in enumerate(os_):
    o.reset(7000 + k)
alive = [True] * B
for t in range(T):
    a = env.policy_actions(pol)
    si, ne = a["stage_idx"].cpu().numpy(), a["num_exec"].cpu().numpy()
    a["stage_idx"][torch.tensor([not x for x in alive], device="cuda:0")] = -(2 ** 31)
    env.step(a)
    oi, of = env.obs_i32.cpu().numpy(), env.obs_f64.cpu().numpy()
    nodes = env.nodes.cpu().numpy()
    for k, o in enumerate(os_):
        if not alive[k]:
            continue
        e, r, done = o.step(int(si[k]), int(ne[k]))
        bad = int(oi[k, 7]) != e or (e == 0 and (bits(of[k, 0]) != bits(r) or bits(of[k, 1]) != bits(o.info().wall_time)
                                                  or not np.array_equal(nodes[k, : oi[k, 0]], o.obs()[1])))
        if bad:
            print(f"step {t} env {k}: gpu err {int(oi[k, 7])} line {int(env.header_field('err_line')[k])} oracle err {e}; reward {of[k, 0]!r} vs {r!r}; "
                  f"wall {of[k, 1]!r} vs {o.info().wall_time!r}; action ({int(si[k])}, {int(ne[k])})")
            alive[k] = False
        elif e or done:
            alive[k] = False
    if not any(alive):
        break
print("done after", t + 1, "steps")
