"""Replays recorded action traces through the batched env (HIP build on a GPU, or the CPU wave
emulator build of the same kernel source) and compares every step with the golden fixtures."""
from __future__ import annotations

import numpy as np
import torch

from golden_util import Golden, bits
from spark_sched_sim_amd import VecSparkSchedSimEnv
from spark_sched_sim_amd.digest import digest_words


SKIP = -2147483648  # stage_idx of an env that takes no part in a launch (include/sss.h)


def step_in_bounded_launches(env, stage_idx: torch.Tensor, num_exec: torch.Tensor, bounded, launch_no: list) -> None:
    """one step of every env through `sss_step_bounded` only (include/sss.h): launches with an event budget until every env's step
    is complete; an env that is done sits the remaining launches out. `bounded`: the budget, or "mixed" - another one every launch."""
    done = torch.zeros(env.num_envs, dtype=torch.bool, device=env.device)
    while not bool(done.all()):
        budget = (launch_no[0] * 7) % 23 + 1 if bounded == "mixed" else int(bounded)
        launch_no[0] += 1
        si = torch.where(done, torch.full_like(stage_idx, SKIP), stage_idx).contiguous()
        ready = env.step_bounded_async(si, num_exec, budget).bool()
        done = done | ready
        assert launch_no[0] < 10 ** 7


def replay_golden(name: str, seeds, pack: bytes, device: str, lib=None, full_obs_steps: int = 0, max_steps: int | None = None,
                  reward_rtol: float = 0.0, rewards_out: list | None = None, bounded=None):
    """one env per seed, all stepped together; returns a list of mismatch descriptions. `bounded`: every step is taken through
    the bounded entry point (`sss_step_bounded`, an event budget per launch) instead of `sss_step`"""
    g = Golden(name)
    pack = g.pack(pack)
    seeds = list(seeds)
    cfg = dict(g.cfg)
    if cfg.get("job_arrival_cap") is None:
        cfg["max_jobs"] = 64
    env = VecSparkSchedSimEnv(cfg, len(seeds), device=device, pack=pack, _lib=lib)
    env.reset(seed=seeds, options={"time_limit": g.time_limit})
    n_rec = [len(g.ep(s, "reward")) for s in seeds]
    if max_steps is not None:
        n_rec = [min(n, max_steps) for n in n_rec]
    err_step = [int(g.ep(s, "error_step")) for s in seeds]
    alive = [True] * len(seeds)
    bad: list[str] = []
    def check_totals(k, s):
        if err_step[k] >= 0 or (max_steps is not None and max_steps < len(g.ep(s, "reward"))):
            return
        ta, tc, order, tmpl = env.job_times(k)
        if not np.array_equal(ta.view(np.uint64), g.ep(s, "t_arrival").view(np.uint64)) or not np.array_equal(tmpl.astype(np.int32), g.ep(s, "template")):
            bad.append(f"{name} seed {s}: arrival times / templates differ")
        wall = env.obs_f64[k, 1].item()
        arrived = env.header(k)["next_arrival"]  # metrics.job_durations covers active + completed jobs only
        dur = np.sort(np.minimum(tc[:arrived], wall) - ta[:arrived])
        if not np.array_equal(dur, np.sort(g.ep(s, "job_durations"))):
            bad.append(f"{name} seed {s}: job durations differ")

    i = 0
    launch_no = [0]
    while True:
        oi = env.obs_i32.cpu().numpy()
        of = env.obs_f64.cpu().numpy()
        need_arrays = any(alive)
        if need_arrays:
            nodes, el = env.nodes.cpu().numpy(), env.edge_links.cpu().numpy()
            ptr, sup = env.dag_ptr.cpu().numpy(), env.exec_supplies.cpu().numpy()
        for k, s in enumerate(seeds):
            if not alive[k]:
                continue
            if err_step[k] >= 0 and i - 1 == err_step[k]:
                if int(oi[k, 7]) != 5:
                    bad.append(f"{name} seed {s}: expected the '[step]' stall error at step {i - 1}, got err={int(oi[k, 7])}")
                alive[k] = False
                continue
            if i >= n_rec[k]:
                alive[k] = False
                check_totals(k, s)  # now: shorter episodes keep being stepped (with no-op actions) below
                continue
            o = oi[k]
            got = (int(o[0]), int(o[1]), int(o[2]), int(o[4]), int(o[5]))
            exp = tuple(int(g.ep(s, kk)[i]) for kk in ("n_nodes", "n_edges", "n_jobs", "ncommit", "src_idx"))
            ok = got == exp and bits(of[k, 1]) == int(g.ep(s, "wall_time")[i])
            if i > 0:
                exp_r = float(g.ep(s, "reward")[i: i + 1].view(np.float64)[0])
                if reward_rtol:  # beta > 0: np.exp in the reference is not bit-reproducible
                    r_ok = abs(of[k, 0] - exp_r) <= reward_rtol * max(1.0, abs(exp_r))
                else:
                    r_ok = bits(of[k, 0]) == bits(exp_r)
                if rewards_out is not None:
                    rewards_out.append((s, i, float(of[k, 0])))
                ok = ok and r_ok and bool(o[6]) == bool(g.ep(s, "terminated")[i]) and int(o[7]) == 0
            n, ne, a = got[0], got[1], got[2]
            if ok:
                d = (digest_words(nodes[k, :n]), digest_words(el[k, :ne]), digest_words(ptr[k, : a + 1]), digest_words(sup[k, :a]))
                ed = tuple(int(g.ep(s, kk)[i]) for kk in ("d_nodes", "d_edges", "d_ptr", "d_sup"))
                ok = d == ed
            if ok and i < min(full_obs_steps, int(g.ep(s, "n_full"))):
                ok = (np.array_equal(nodes[k, :n].view(np.uint32), g.ep(s, f"full{i}_nodes").view(np.uint32))
                      and np.array_equal(el[k, :ne], g.ep(s, f"full{i}_edges"))
                      and np.array_equal(ptr[k, : a + 1], g.ep(s, f"full{i}_ptr"))
                      and np.array_equal(sup[k, :a], g.ep(s, f"full{i}_sup")))
            if not ok:
                bad.append(f"{name} seed {s} step {i}: got {got} err={int(o[7])} reward={of[k, 0]!r} wall={of[k, 1]!r}; expected {exp}")
                alive[k] = False
            elif i == n_rec[k] - 1 and not (err_step[k] >= 0):
                # last recorded step: totals are checked now, before the longer episodes of the
                # batch keep this env stepping with no-op actions
                alive[k] = False
                check_totals(k, s)
        if not any(alive):
            break
        si = torch.full((len(seeds),), -1, dtype=torch.int32)
        ne_ = torch.ones(len(seeds), dtype=torch.int32)
        for k, s in enumerate(seeds):
            st = g.ep(s, "stage_idx")
            if alive[k] and i + 1 < len(st):
                si[k], ne_[k] = int(st[i + 1]), int(g.ep(s, "num_exec")[i + 1])
        if bounded is None:
            env.step({"stage_idx": si.to(env.device), "num_exec": ne_.to(env.device)})
        else:
            step_in_bounded_launches(env, si.to(env.device), ne_.to(env.device), bounded, launch_no)
        i += 1
    env.close()
    return bad
