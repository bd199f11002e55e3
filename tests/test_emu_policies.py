"""On-device policies (csrc/sss_sim.h policy_fair / policy_hash) and the fused rollout kernel,
under the CPU wave emulator: the action streams they produce must be the ones the reference's own
RoundRobinScheduler produced when the golden trajectories were recorded (c1_fair / c1_fifo), resp.
the recorded hash-policy stream (tiny_hash), and the fused kernel must land in the same state."""
import numpy as np
import pytest
import torch

from emu_util import load_emu
from golden_util import Golden, bits
from spark_sched_sim_amd import VecSparkSchedSimEnv


def run_policy_episode(name, policy, param, seeds, pack, device="cpu", lib=None, max_steps=None, fused=0):
    g = Golden(name)
    env = VecSparkSchedSimEnv(g.cfg, len(seeds), device=device, pack=pack, _lib=lib)
    env.reset(seed=seeds)
    n_rec = [len(g.ep(s, "reward")) for s in seeds]
    T = max(n_rec) - 1 if max_steps is None else max_steps
    bad = []
    if fused:
        env.rollout(policy, T, param)
        for k, s in enumerate(seeds):
            n = min(T, n_rec[k] - 1)
            hdr = env.header(k)
            if n == n_rec[k] - 1:
                ok = hdr["terminated"] == 1 and bits(hdr["wall_time"]) == int(g.ep(s, "wall_time")[n]) and hdr["ep_steps"] == n
                exp_ret = 0.0
                for rr in g.ep(s, "reward")[1:].view(np.float64).tolist():
                    exp_ret += rr  # same left-to-right f64 accumulation as the device's ep_return
                ok = ok and bits(hdr["ep_return"]) == bits(exp_ret)
            else:
                ok = bits(hdr["wall_time"]) == int(g.ep(s, "wall_time")[n])
            if not ok:
                bad.append(f"{name} seed {s}: fused rollout ended at wall={hdr['wall_time']} steps={hdr['ep_steps']} term={hdr['terminated']}")
        env.close()
        return bad
    alive = [True] * len(seeds)
    for i in range(1, T + 1):
        act = env.policy_actions(policy, param)
        si, ne = act["stage_idx"].cpu().numpy().copy(), act["num_exec"].cpu().numpy().copy()
        env.step(act)
        of = env.obs_f64.cpu().numpy()
        for k, s in enumerate(seeds):
            if not alive[k] or i >= n_rec[k]:
                alive[k] = False
                continue
            exp = (int(g.ep(s, "stage_idx")[i]), int(g.ep(s, "num_exec")[i]))
            if (int(si[k]), int(ne[k])) != exp or bits(of[k, 0]) != int(g.ep(s, "reward")[i]) or bits(of[k, 1]) != int(g.ep(s, "wall_time")[i]):
                bad.append(f"{name} seed {s} step {i}: action {(int(si[k]), int(ne[k]))} expected {exp}; reward {of[k, 0]}")
                alive[k] = False
        if not any(alive):
            break
    env.close()
    return bad


@pytest.mark.parametrize("name,policy,param,seeds,max_steps", [
    ("c1_fair", "fair", 0, [1234, 0, 1], None),
    ("c1_fifo", "fifo", 0, [5], 400),
    ("tiny_hash", "hash", 30, list(range(8)), None),
    ("testyaml_fair", "fair", 0, [3, 4], None),
])
def test_device_policy_reproduces_recorded_actions(name, policy, param, seeds, max_steps, pack):
    bad = run_policy_episode(name, policy, param, seeds, pack, lib=load_emu(), max_steps=max_steps)
    assert not bad, "\n".join(bad[:10])


def test_fused_rollout_equals_stepwise(pack):
    bad = run_policy_episode("c1_fair", "fair", 0, [1234, 2], pack, lib=load_emu(), fused=1)
    bad += run_policy_episode("tiny_hash", "hash", 30, [0, 1, 2, 3], pack, lib=load_emu(), fused=1)
    assert not bad, "\n".join(bad[:10])
