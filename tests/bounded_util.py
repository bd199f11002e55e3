"""`sss_step_bounded` against `sss_step` (shared by the emulator and the GPU tests)."""
from __future__ import annotations

import numpy as np
import torch

from spark_sched_sim_amd import VecSparkSchedSimEnv
from spark_sched_sim_amd.vec_env import HDR_OFF, HDR_PROF

SKIP = -2147483648
OUT = ("nodes", "edge_links", "dag_ptr", "exec_supplies", "obs_i32", "obs_f64")


def _rows(env, name):
    t = getattr(env, name)
    return t.reshape(env.num_envs, -1).cpu().numpy().copy()


def check_bounded_steps(device, lib, cfg, seeds, policy, n_steps, budgets, pack=None):
    """every env's k-th completed step under `step_bounded_async` leaves the outputs the k-th `step` leaves (observation arrays,
    reward, wall time, flags - bit for bit), whatever the budget; after `n_steps` completed steps per env the env states are the
    same bytes (timers, batch counters and the scratch a cut step parks in the state aside). Returns launches per budget."""
    B = len(seeds)
    kw = dict(device=device, auto_reset=True, _lib=lib)  # (an env whose episode ends starts its next one: a launch like any other)
    if pack is not None:
        kw["pack"] = pack
    ref = VecSparkSchedSimEnv(cfg, B, **kw)
    ref.reset(seed=list(seeds))
    want = []
    for _ in range(n_steps):
        a = ref.policy_actions(policy)
        ref.step_async(a["stage_idx"], a["num_exec"])
        want.append({k: _rows(ref, k) for k in OUT})
    assert not any((w["obs_i32"][:, 7] != 0).any() for w in want), "the reference run reported an error"
    d = ref.dims
    keep = np.ones(d.env_stride, dtype=bool)
    keep[HDR_PROF: HDR_PROF + 40] = False            # shader-clock timers
    keep[256:272] = False                            # n_batched, n_rounds: how events were grouped
    keep[280:288] = False                            # mid_step (0 on both sides), step_events (left over from the last cut step)
    keep[304:320] = False                            # wall_old, n_old_active: the same
    state_ref = ref._env_view.cpu().numpy().copy()
    launches = {}
    for budget in budgets:
        env = VecSparkSchedSimEnv(cfg, B, **kw)
        env.reset(seed=list(seeds))
        count = np.zeros(B, dtype=np.int64)
        n_launch = n_cut = 0
        while (count < n_steps).any():
            a = env.policy_actions(policy)
            frozen = torch.from_numpy(count >= n_steps).to(env.device)
            stage_idx = torch.where(frozen, torch.full_like(a["stage_idx"], SKIP), a["stage_idx"]).contiguous()
            if budget == "mixed":  # a different budget every launch, and every fifth launch a plain step() (it finishes cut steps too)
                if n_launch % 5 == 4:
                    env.step_async(stage_idx, a["num_exec"])
                    ready = np.ones(B, dtype=bool)
                else:
                    ready = env.step_bounded_async(stage_idx, a["num_exec"], (n_launch * 7) % 23 + 1).cpu().numpy().astype(bool)
            else:
                ready = env.step_bounded_async(stage_idx, a["num_exec"], budget).cpu().numpy().astype(bool)
            n_launch += 1
            got = {k: _rows(env, k) for k in OUT}
            for b in np.nonzero(~frozen.cpu().numpy())[0]:
                if not ready[b]:
                    n_cut += 1
                    continue
                w = want[count[b]]
                for k in OUT:
                    assert np.array_equal(got[k][b], w[k][b]), (budget, int(b), int(count[b]), k)
                count[b] += 1
            assert n_launch < 200 * n_steps, "no progress"
        state = env._env_view.cpu().numpy().copy()
        cmp = keep.copy()
        # the old-active list of a cut step sits behind the active list (2 bytes per job slot each)
        n_slots = ref.dims.job_cap
        cmp[d.off_active + 2 * n_slots: d.off_active + 4 * n_slots] = False
        diff = (state != state_ref) & cmp[None, :]
        # entries of the active list behind n_active are leftovers of longer lists (a cut step parks its list as it is then)
        n_act = np.ascontiguousarray(state[:, HDR_OFF["n_active"]: HDR_OFF["n_active"] + 4]).view(np.int32).ravel()
        assert np.array_equal(n_act, np.ascontiguousarray(state_ref[:, HDR_OFF["n_active"]: HDR_OFF["n_active"] + 4]).view(np.int32).ravel())
        # entries of the commitment list behind the live ones are dead (never read again, written back only while live: which bytes
        # they hold follows from where launches ended); hot block: header, then ev[ME] (16 B), ex_loc (4), ex_job (2),
        # ex_task_stage (1), ex_executing (1), c_src (4), c_dst (4), c_seq (4), c_n (2), ME = 64 or 128 executor entries
        ME = 128 if cfg["num_executors"] > 64 else 64
        n_com = np.ascontiguousarray(state[:, HDR_OFF["n_commits"]: HDR_OFF["n_commits"] + 4]).view(np.int32).ravel()
        assert np.array_equal(n_com, np.ascontiguousarray(state_ref[:, HDR_OFF["n_commits"]: HDR_OFF["n_commits"] + 4]).view(np.int32).ravel())
        c0 = d.hdr_bytes + ME * (16 + 4 + 2 + 1 + 1)
        for b in range(B):
            diff[b, d.off_active + 2 * int(n_act[b]): d.off_active + 2 * n_slots] = False
            for base, w in ((c0, 4), (c0 + 4 * ME, 4), (c0 + 8 * ME, 4), (c0 + 12 * ME, 2)):
                diff[b, base + w * int(n_com[b]): base + w * ME] = False
        assert not diff.any(), (budget, np.nonzero(diff.any(0))[0][:10])
        launches[budget] = (n_launch, n_cut)
        env.close()
    ref.close()
    return launches
