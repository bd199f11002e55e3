"""helpers of the workload-boundary tests: small trace sets to mutate, and a lock-step comparison of the batched env (HIP build on
a GPU, or the CPU wave-emulator build of the same kernel source) with the C oracle under the counter-based test policy, every
step with the full observation."""
from __future__ import annotations

import numpy as np
import torch

from oracle_binding import OracleEnv
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.digest import splitmix64

SMALL_SIZES, SMALL_QUERIES, SMALL_SEED = ["2g", "10g"], 4, 4242


def small_raw(profile: str = "default") -> dict:
    """a 4-query x 2-size trace set in the reference's schema (fresh lists every call: tests mutate it)"""
    return workload.make_raw_workload(SMALL_SEED, SMALL_SIZES, SMALL_QUERIES, profile=profile)


def lockstep_vs_oracle(pack: bytes, cfg: dict, seeds, n_steps: int, device: str, lib=None, p_none_permille: int = 30) -> list[str]:
    """one env per seed next to one oracle per seed; returns the mismatches (empty: none). Compared bit for bit at every step:
    the scalar observation fields, wall time, reward, termination, and the node / edge / dag_ptr / exec_supplies arrays."""
    seeds = list(seeds)
    B = len(seeds)
    env = VecSparkSchedSimEnv(cfg, B, device=device, pack=pack, _lib=lib)
    env.reset(seed=seeds)
    oracles = [OracleEnv(pack, cfg) for _ in seeds]
    for o, s in zip(oracles, seeds):
        assert o.reset(s) == 0
    done = [False] * B
    bad: list[str] = []
    for t in range(n_steps):
        oi, of = env.obs_i32.cpu().numpy(), env.obs_f64.cpu().numpy()
        nodes, el = env.nodes.cpu().numpy(), env.edge_links.cpu().numpy()
        ptr, sup = env.dag_ptr.cpu().numpy(), env.exec_supplies.cpu().numpy()
        si, ne = np.full(B, -1, np.int32), np.ones(B, np.int32)
        for k, o in enumerate(oracles):
            if done[k]:
                continue
            info, onodes, oel, optr, osup = o.obs()
            got = tuple(int(x) for x in oi[k, :6])
            exp = (info.n_nodes, info.n_edges, info.n_jobs, info.n_schedulable, info.num_committable_execs, info.source_job_idx)
            ok = got == exp and int(oi[k, 7]) == 0 and np.float64(of[k, 1]).view(np.uint64) == np.float64(info.wall_time).view(np.uint64)
            ok = ok and np.array_equal(nodes[k, : info.n_nodes].view(np.uint32), onodes.view(np.uint32)) and np.array_equal(el[k, : info.n_edges], oel)
            ok = ok and np.array_equal(ptr[k, : info.n_jobs + 1], optr) and np.array_equal(sup[k, : info.n_jobs], osup)
            if not ok:
                bad.append(f"seed {seeds[k]} step {t}: env {got} err={int(oi[k, 7])} wall={of[k, 1]!r}; oracle {exp} wall={info.wall_time!r}")
                done[k] = True
                continue
            h = splitmix64((seeds[k] << 32) ^ t)
            h2 = splitmix64(h)
            none = info.n_schedulable == 0 or splitmix64(h2) % 1000 < p_none_permille
            si[k] = -1 if none else h % info.n_schedulable
            ne[k] = 1 + h2 % max(1, info.num_committable_execs)
        if all(done):
            break
        env.step({"stage_idx": torch.from_numpy(si).to(env.device), "num_exec": torch.from_numpy(ne).to(env.device)})
        rew, term = env.obs_f64[:, 0].cpu().numpy(), env.obs_i32[:, 6].cpu().numpy()
        for k, o in enumerate(oracles):
            if done[k]:
                continue
            e, r, tm = o.step(int(si[k]), int(ne[k]))
            if e == 5 and int(env.obs_i32[k, 7]) == 5:   # the reference's own "[step]" stall, reached by both
                done[k] = True
            elif e != 0 or np.float64(r).view(np.uint64) != np.float64(rew[k]).view(np.uint64) or bool(term[k]) != tm:
                bad.append(f"seed {seeds[k]} step {t}: reward env {rew[k]!r} oracle {r!r} (oracle err {e}, env err {int(env.obs_i32[k, 7])})")
                done[k] = True
            elif tm:
                done[k] = True
    env.close()
    for o in oracles:
        o.close()
    return bad
