"""shared by the emulator and GPU training tests: replays tests/golden/ppo_c1.npz (recorded from the
reference's rollout workers, returns / baseline calculators, evaluate_actions, CLIP loss and
optimiser step; see tests/golden/make_ppo_golden.py) on the batched env."""
import os.path as osp

import numpy as np
import torch

from decima_util import AGENT
from spark_sched_sim_amd import VecSparkSchedSimEnv
from spark_sched_sim_amd.decima import DecimaPolicy, select_observations
from spark_sched_sim_amd.digest import splitmix64
from spark_sched_sim_amd.training import DifferentialReturns, PPO, RolloutCollector, discounted_returns, ppo_loss, sequence_baselines

HERE = osp.dirname(osp.abspath(__file__))
REWARD_RTOL = 1e-12   # beta > 0: rewards go through exp(); everything else in the env is exact
RETURN_RTOL = 1e-10   # recurrences of up to ~300 exp()-weighted terms over those rewards
SCORE_ATOL = 2e-5     # float32 GNN, different summation order


def counter_act_fn(g, step_counts):
    """the fixture's stand-in for sampling (make_ppo_golden.CounterPolicy), batched"""
    dev = g["x"].device
    B = g["n_obs"]
    n_sched = torch.zeros(B, dtype=torch.long, device=dev).index_add_(0, g["node_obs"], g["stage_mask"].long()).cpu().numpy()
    sel, h2s = np.zeros(B, dtype=np.int64), np.zeros(B, dtype=np.uint64)
    for b in range(B):
        h1 = splitmix64(((1000 + b) << 32) ^ int(step_counts[b]))
        h2s[b] = splitmix64(h1)
        sel[b] = h1 % max(1, int(n_sched[b]))
    stage_sel = torch.from_numpy(sel).to(dev)
    node = (g["sched_rank"] == stage_sel[g["node_obs"]]).nonzero(as_tuple=True)[0]  # one per env that has a stage
    has = torch.from_numpy(n_sched > 0).to(dev)
    job_gid = torch.zeros(B, dtype=torch.long, device=dev)
    job_gid[has] = g["node_job"][node]
    n_allowed = torch.zeros(B, dtype=torch.long, device=dev)
    n_allowed[has] = g["job_cap"][job_gid[has]]
    n_allowed = n_allowed.cpu().numpy()
    job_off = torch.cumsum(g["obs_jobs"], 0) - g["obs_jobs"]
    ex = np.asarray([int(h2s[b]) % max(1, int(n_allowed[b])) for b in range(B)], dtype=np.int64)
    lg = torch.tensor([-1.0 - 0.001 * ((int(step_counts[b]) + 1) % 7) for b in range(B)], dtype=torch.float32, device=dev)
    return {"stage_sel": stage_sel, "job_idx": torch.where(has, job_gid - job_off, torch.zeros_like(job_gid)),
            "exec_sel": torch.from_numpy(ex).to(dev), "lgprob": lg, "any_stage": has}


def load_fixture():
    g = np.load(osp.join(HERE, "golden", "ppo_c1.npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], [float(v) for v in g["cfg_vals"]]))
    mean_tl = cfg.pop("mean_time_limit")
    cfg["num_executors"] = int(cfg["num_executors"])
    cfg["job_arrival_cap"] = int(cfg["job_arrival_cap"])
    cfg["beta"] = float(g["beta"])
    return g, cfg, mean_tl


def make_policy(g, prefix, dev, **kw):
    policy = DecimaPolicy(num_executors=10, **AGENT, **kw)
    policy.load_state_dict({k[len(prefix):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)})
    return policy.to(dev)


def check_rollout(ro, g, prefix, b):
    r = ro.rollout(b)
    assert np.array_equal(r["actions"], g[prefix + "actions"]), prefix
    assert np.array_equal(r["wall_times"].view(np.uint64), g[prefix + "wall_times"].view(np.uint64)), prefix
    np.testing.assert_allclose(r["rewards"], g[prefix + "rewards"], rtol=REWARD_RTOL, atol=0)
    return r


def check_sync_pipeline(device, lib):
    g, cfg, mean_tl = load_fixture()
    base_seeds = [int(s) for s in g["base_seeds"]]
    env = VecSparkSchedSimEnv(cfg, len(base_seeds), device=device, _lib=lib)
    dev = env.device
    col = RolloutCollector(env, mean_tl, base_seeds, seed_step=2, num_executors=10, act_fn=counter_act_fn)
    for it in range(2):
        ro = col.collect_sync()
        for b in range(len(base_seeds)):
            p = f"sync{it}_r{b}_"
            r = check_rollout(ro, g, p, b)
            np.testing.assert_allclose(r["lgprobs"], g[p + "lgprobs"], rtol=1e-6)
            assert float(col.tl_env.time_limit[b]) == float(g[p + "time_limit"])
            st = np.asarray([ro.stats[k][b] for k in ("avg_job_duration", "avg_num_jobs", "num_completed_jobs", "num_job_arrivals")])
            np.testing.assert_allclose(st, g[p + "stats"], rtol=1e-12, equal_nan=True)
    # ---- returns and baselines of the second iteration's rollouts ----------------------------
    B = len(base_seeds)
    ret = discounted_returns(ro, cfg["beta"])
    base = sequence_baselines(ro, ret, num_sequences=2, num_rollouts=2)
    for b in range(B):
        n = int(ro.lengths[b])
        np.testing.assert_allclose(ret[:n, b].cpu().numpy(), g[f"returns_r{b}"], rtol=RETURN_RTOL)
        np.testing.assert_allclose(base[:n, b].cpu().numpy(), g[f"baselines_r{b}"], rtol=RETURN_RTOL)
    diff = DifferentialReturns(700)
    for call in range(2):
        d = diff(ro)
        np.testing.assert_allclose(diff.avg_num_jobs, float(g[f"diff_avg_num_jobs{call}"]), rtol=1e-12)
        for b in range(B):
            n = int(ro.lengths[b])
            np.testing.assert_allclose(d[:n, b].cpu().numpy(), g[f"diffret{call}_r{b}"], rtol=RETURN_RTOL, atol=1e-6)
    # ---- evaluate_actions, CLIP loss, one optimiser step on the recorded minibatch -------------
    policy = make_policy(g, "w_", dev, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5).train()
    ids = ro.sample_ids()
    mb = torch.from_numpy(g["mb_idx"]).to(dev)
    sub = select_observations(ro.graph, ids[mb])
    acts = [ro.flat(ro.stage_sel)[mb], ro.flat(ro.job_idx)[mb], ro.flat(ro.exec_sel)[mb]]
    res = policy.evaluate_actions(sub, *acts)
    np.testing.assert_allclose(res["lgprobs"].detach().cpu().numpy(), g["mb_lgprobs"], atol=SCORE_ATOL)
    np.testing.assert_allclose(res["entropies"].detach().cpu().numpy(), g["mb_entropies"], atol=SCORE_ATOL)
    adv = ro.flat(ret - base)[mb]
    loss, info = ppo_loss(policy, sub, *acts, adv, torch.from_numpy(g["mb_old_lgprobs"]).to(dev), 0.2, 0.04)
    np.testing.assert_allclose(float(loss.detach()), float(g["mb_loss"]), atol=2e-5)
    np.testing.assert_allclose([float(info[k]) for k in ("policy_loss", "entropy_loss", "approx_kl_div")], g["mb_info"], atol=2e-5)
    loss.backward()
    grads = {k: p.grad.clone().cpu() for k, p in policy.named_parameters()}
    policy.update_parameters(None)
    # Adam's first step moves every element by ~lr * g / (|g| + 1e-8): a wrong gradient sign shows
    # up as 2 * lr = 6e-4. Elements whose gradient is round-off noise around an exact zero (the last
    # bias of the stage scorer: softmax is shift-invariant) are amplified by that rule and skipped.
    worst, checked, total = 0.0, 0, 0
    for k, v in policy.state_dict().items():
        ref0, ref1 = torch.from_numpy(g["w_" + k]), torch.from_numpy(g["w1_" + k])
        assert float((ref1 - ref0).abs().max()) > 0  # the step really changed this tensor in the reference
        solid = grads[k].abs() > 1e-6
        total += solid.numel()
        checked += int(solid.sum())
        if solid.any():
            worst = max(worst, float((v.cpu() - ref1).abs()[solid].max()))
    assert checked >= 0.97 * total, (checked, total)
    assert worst <= 2e-5, worst
    env.close()
    return ro


def check_async_pipeline(device, lib):
    g, cfg, mean_tl = load_fixture()
    base_seeds = [int(s) for s in g["base_seeds"]][:2]
    env = VecSparkSchedSimEnv(cfg, 2, device=device, _lib=lib)
    col = RolloutCollector(env, mean_tl, base_seeds, seed_step=2, num_executors=10, act_fn=counter_act_fn)
    for it in range(2):
        ro = col.collect_async(float(g["async_duration"]))
        for b in range(2):
            p = f"async{it}_r{b}_"
            r = check_rollout(ro, g, p, b)
            assert np.array_equal(r["resets"], g[p + "resets"]), p
    env.close()


def reference_smoke_test_config(artifacts_dir: str) -> dict:
    """the hyper-parameters of the reference's own smoke test, in its config schema (reference test/test.yaml:1-41: one PPO
    iteration, one job sequence x two rollouts, 50 executors, 10 jobs) - built here instead of kept as a YAML file"""
    return {
        "trainer": dict(trainer_cls="PPO", num_iterations=1, num_sequences=1, num_rollouts=2, seed=42, artifacts_dir=artifacts_dir,
                        checkpointing_freq=50, use_tensorboard=False, num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01,
                        entropy_coeff=0.04, beta_discount=5.0e-3, opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5),
        "agent": dict(agent_cls="DecimaScheduler", embed_dim=16,
                      gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(inplace=True, negative_slope=0.2)),
                      policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh")),
        "env": dict(num_executors=50, job_arrival_cap=10, moving_delay=2000.0, mean_time_limit=2.0e7, job_arrival_rate=4.0e-5,
                    warmup_delay=1000.0, data_sampler_cls="TPCHDataSampler"),
    }


def check_rows_ops(binding, device, n):
    """`train_kernels.rows_op` (include/sss.h sss_rows_op) against torch's index_select / index_add_ / index_copy on the same
    data; the gather is exact, the sums agree up to the order of the additions"""
    import pytest

    from spark_sched_sim_amd.train_kernels import ROWS_GATHER, ROWS_SCATTER, ROWS_SCATTER_ADD, ROWS_SEGMENT_SUM, ROWS_TAKE, ROWS_UPDATE, rows_op, segment_offsets

    gen = torch.Generator().manual_seed(11)
    rnd = lambda *shape: torch.randn(shape, generator=gen).to(device)  # noqa: E731
    for width in (1, 2, 5, 6, 16, 21, 35, 36, 64):  # (2, 6: the 8-bytes-per-lane form of GATHER / SCATTER where the pointers allow it)
        rows = max(3, n // 3)
        table = rnd(rows, width)
        idx = torch.randint(0, rows, (n,), generator=gen).to(device)
        # gather into a whole matrix and into a column slice of a wider one
        out = torch.empty((n, width), device=device)
        rows_op(ROWS_GATHER, idx, out, table, binding=binding)
        assert torch.equal(out, table[idx])
        wide = torch.full((n, width + 7), -1.0, device=device)
        rows_op(ROWS_GATHER, idx, wide[:, 3:3 + width], table, binding=binding)
        assert torch.equal(wide[:, 3:3 + width], table[idx]) and bool((wide[:, :3] == -1).all()) and bool((wide[:, 3 + width:] == -1).all())
        # scatter-add, from a whole matrix and from a column slice
        src = rnd(n, width + 4)
        for a in (src[:, :width].contiguous(), src[:, 2:2 + width]):
            acc = table.clone()
            rows_op(ROWS_SCATTER_ADD, idx, a, acc, binding=binding)
            assert torch.allclose(acc, table.clone().index_add_(0, idx, a), rtol=1e-5, atol=1e-5)
        # the two receiver operations (ids without repeats)
        uniq = torch.randperm(rows, generator=gen)[: max(1, rows // 2)].to(device)
        a, h, c = rnd(uniq.numel(), width), rnd(rows, width), rnd(rows, width)
        want = h.clone().index_copy_(0, uniq, a + c[uniq])
        rows_op(ROWS_UPDATE, uniq, a, h, c, binding=binding)
        assert torch.equal(h, want)
        gh, gi = rnd(rows, width), rnd(rows, width)
        want_a, want_gh, want_gi = gh[uniq].clone(), gh.clone().index_fill_(0, uniq, 0.0), gi.clone().index_add_(0, uniq, gh[uniq])
        got_a = torch.empty_like(a)
        rows_op(ROWS_TAKE, uniq, got_a, gh, gi, binding=binding)
        assert torch.equal(got_a, want_a) and torch.equal(gh, want_gh) and torch.equal(gi, want_gi)
        # stores to rows without repeats; sums over ranges of rows (empty segments included)
        tab = rnd(rows, width)
        want = tab.clone().index_copy_(0, uniq, a)
        rows_op(ROWS_SCATTER, uniq, a, tab, binding=binding)
        assert torch.equal(tab, want)
        owner = torch.sort(torch.randint(0, rows, (n,), generator=gen))[0].to(device)
        ptr = segment_offsets(owner, rows)
        assert int(ptr[0]) == 0 and int(ptr[-1]) == n
        for a2 in (src[:, :width].contiguous(), src[:, 2:2 + width]):
            sums = torch.full((rows, width), 7.0, device=device)
            rows_op(ROWS_SEGMENT_SUM, ptr, a2, sums, binding=binding)
            assert torch.allclose(sums, torch.zeros((rows, width), device=device).index_add_(0, owner, a2), rtol=1e-5, atol=1e-5)
    # several tables side by side in one launch (sss_rows_concat): the two score networks' input rows and others, row counts on both
    # sides of a wave's 64 rows, a part without indices, a part without a gradient
    from spark_sched_sim_amd.train_kernels import rows_concat
    for widths in ((5, 16, 16, 16), (35, 1), (3, 16, 16, 1), (64,), (1, 1, 1, 1), (21, 43)):
        for m in sorted({1, 63, 64, 65, n}):
            tabs = [rnd(max(3, m // 2) if k else m, w) for k, w in enumerate(widths)]
            idxs = [None] + [torch.randint(0, t.shape[0], (m,), generator=gen).to(device) for t in tabs[1:]]
            out = torch.full((m, sum(widths)), -3.0, device=device)
            rows_concat(0, out, tabs, idxs, binding=binding)
            assert torch.equal(out, torch.cat([t if ix is None else t[ix] for t, ix in zip(tabs, idxs)], -1)), (widths, m)
            g = rnd(m, sum(widths))
            accs = [t.clone() for t in tabs]
            skip = len(widths) - 1 if len(widths) > 1 else -1
            rows_concat(1, g, [a if k != skip else a.shape[1] for k, a in enumerate(accs)], idxs, binding=binding)
            off = 0
            for k, (t, a, ix) in enumerate(zip(tabs, accs, idxs)):
                want = t if k == skip else (t + g[:, off:off + t.shape[1]] if ix is None else t.clone().index_add_(0, ix, g[:, off:off + t.shape[1]]))
                assert torch.allclose(a, want, rtol=1e-5, atol=1e-5), (widths, m, k)
                off += t.shape[1]
    with pytest.raises(ValueError):
        rows_concat(0, torch.empty((2, 70), device=device), [rnd(2, 35), rnd(2, 35)], [None, None], binding=binding)
    empty = torch.zeros((0,), dtype=torch.long, device=device)
    rows_op(ROWS_GATHER, empty, torch.empty((0, 16), device=device), rnd(4, 16), binding=binding)  # nothing to do: no launch
    with pytest.raises(ValueError):
        rows_op(ROWS_GATHER, torch.zeros(2, dtype=torch.long, device=device), torch.empty((2, 65), device=device), rnd(4, 65), binding=binding)
    with pytest.raises(ValueError):
        rows_op(7, torch.zeros(2, dtype=torch.long, device=device), torch.empty((2, 16), device=device), rnd(4, 16), binding=binding)


def check_record_on_device(device, lib, num_envs, tiny_arena=False, num_executors=10):
    """`RolloutCollector(record_on_device=True)` (the step's graph stays on the device and is appended to a `GraphArena` there,
    flags read a few steps late) against `record_on_device=False` (sizes and flags read every step, per-step graphs concatenated
    at the end): same policy parameters, same generator seed -> the same record, array by array, twice in a row (the second
    collection starts from the first one's arena sizes)"""
    import spark_sched_sim_amd.training as T

    cfg = dict(num_executors=num_executors, job_arrival_cap=8, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    dev = torch.device(device)
    out = {}
    old_init = T.GraphArena.__init__
    if tiny_arena:
        def small_init(self, env, capacity=None):
            old_init(self, env, capacity)
            for name, dt, per, kind, _ in self.ARRAYS:  # one step's maximum and a bit: every few steps the headroom rule fires
                self.capacity[kind] = self.step_max[kind] + 3
                self.buf[name] = self.buf[name][: self.capacity[kind]].clone()
        T.GraphArena.__init__ = small_init
    try:
        for on_dev in (True, False):
            env = VecSparkSchedSimEnv(cfg, num_envs, device=device, auto_reset=False, _lib=lib)
            torch.manual_seed(1)
            pol = DecimaPolicy(num_executors=num_executors, **AGENT).to(dev)
            gen = torch.Generator(device=dev if dev.type == "cuda" else "cpu")
            gen.manual_seed(99)
            col = RolloutCollector(env, 5.0e5, list(range(21, 21 + num_envs)), seed_step=num_envs, num_executors=num_executors, policy=pol, generator=gen,
                                   record_on_device=on_dev)
            out[on_dev] = [col.collect_sync(with_stats=False) for _ in range(2)]
            assert (col._arena_sizes is not None) == on_dev  # (the loop without waits was the one that ran)
            env.close()
    finally:
        T.GraphArena.__init__ = old_init
    for ra, rb in zip(out[True], out[False]):
        assert ra.active.shape == rb.active.shape and int(ra.active.sum()) > 20 * num_envs
        for name in ("active", "t_before", "t_after", "rewards", "stage_sel", "job_idx", "exec_sel", "lgprobs", "resets"):
            assert torch.equal(getattr(ra, name), getattr(rb, name)), name
        assert set(ra.graph) >= set(rb.graph) - {"_ranges"}
        for k, v in rb.graph.items():
            if torch.is_tensor(v):
                assert ra.graph[k].dtype == v.dtype and torch.equal(ra.graph[k], v), k
            elif k in ("n_obs", "n_pad"):
                assert ra.graph[k] == v, k
        assert torch.equal(ra.sample_ids(), rb.sample_ids())


def check_record_kernels(binding, device):
    """`sss_discounted_returns` / `sss_sequence_baselines` against the tensor-op forms of training.discounted_returns /
    sequence_baselines on a random record: envs with different lengths (one of them empty -> the mean skips it), repeated step
    times (zero-length steps: duplicate knots of the value curves), with and without empty rollouts. Bit for bit on the GPU
    (same exp, same operation order); the emulator's host loops use libm's exp (1e-12 relative)."""
    from spark_sched_sim_amd.training import Rollouts

    gen = torch.Generator().manual_seed(0)
    for T, B, R in ((37, 12, 4), (200, 64, 4), (5, 3, 1), (23, 16, 8), (19, 27, 9), (11, 140, 140)):  # (R >= 8: numpy's pairwise order)
        n = torch.randint(0, T + 1, (B,), generator=gen)
        n[min(3, B - 1)] = 0
        dt = torch.rand((T, B), generator=gen, dtype=torch.float64) * 5e4
        dt[torch.rand((T, B), generator=gen) < 0.2] = 0.0
        ta = torch.cumsum(dt, 0)
        tb = torch.cat([torch.zeros((1, B), dtype=torch.float64), ta[:-1]])  # non-decreasing, repeats where dt = 0
        rw = -torch.rand((T, B), generator=gen, dtype=torch.float64) * 1e4
        z = torch.zeros((T, B), dtype=torch.long)
        for drop_empty in (False, True):
            nn = torch.where(n == 0, torch.ones_like(n), n) if drop_empty else n
            a = (torch.arange(T)[:, None] < nn[None, :]).to(device)
            ro = Rollouts(graph={}, active=a, t_before=tb.to(device) * a, t_after=ta.to(device) * a, rewards=rw.to(device) * a, stage_sel=z, job_idx=z, exec_sel=z,
                          lgprobs=z.float(), resets=z.bool())
            ref = Rollouts(**{**ro.__dict__, "active": a.cpu(), "t_before": ro.t_before.cpu(), "t_after": ro.t_after.cpu(), "rewards": ro.rewards.cpu()})
            r_ref = T_discounted(ref)
            r_got = discounted_returns(ro, 5e-3, binding=binding)
            assert torch.allclose(r_got.cpu(), r_ref, rtol=1e-12, atol=1e-9), float((r_got.cpu() - r_ref).abs().max())
            b_ref = sequence_baselines(ref, r_ref, B // R, R)
            b_got = sequence_baselines(ro, r_ref.to(device), B // R, R, binding=binding)
            assert torch.equal(b_got.cpu(), b_ref), float((b_got.cpu() - b_ref).abs().max())
            if device != "cpu":
                r_dev = T_discounted(ro)   # the same loop as tensor operations on the GPU: the same exp
                assert torch.equal(r_got, r_dev)


def T_discounted(ro):
    """training.discounted_returns' tensor-op loop, whatever device the record is on (returns_calculator.py:67-76)"""
    T, B = ro.active.shape
    dt = ro.t_after - ro.t_before
    out = torch.zeros_like(ro.rewards)
    R = torch.zeros(B, dtype=torch.float64, device=ro.rewards.device)
    for k in range(T - 1, -1, -1):
        R = torch.where(ro.active[k], ro.rewards[k] + torch.exp(-5e-3 * 1e-3 * dt[k]) * R, R)
        out[k] = R
    return out * ro.active


def check_bit_lists(binding, device):
    """`decima.bit_lists` (include/sss.h sss_bit_lists + sss_prefix_rows) against `((bits >> l) & 1).nonzero()` layer by layer:
    sizes around the chunk and wave boundaries, all 32 bits, sparse and empty masks"""
    from spark_sched_sim_amd.decima import bit_lists

    gen = torch.Generator().manual_seed(0)
    for n, L, density in ((1, 1, 1.0), (63, 3, 0.5), (2048, 9, 0.5), (2049, 9, 0.1), (70_001, 32, 0.5), (300_000, 9, 0.02), (5_000, 4, 0.0)):
        bits = torch.randint(0, 2 ** 32, (n,), generator=gen, dtype=torch.int64) & ((1 << L) - 1)
        bits[torch.rand(n, generator=gen) >= density] = 0
        bits = torch.where(bits >= 2 ** 31, bits - 2 ** 32, bits).to(torch.int32).to(device)
        got = bit_lists(bits, L, binding=binding)
        assert len(got) == L
        for l in range(L):
            want = ((bits >> l) & 1).nonzero(as_tuple=True)[0]
            assert got[l].dtype == torch.int64 and torch.equal(got[l], want), (n, L, l)


def check_record_on_device_with_a_failing_env(device, lib, num_envs=6):
    """an env that reports an error in the middle of a collection (its sampled action is corrupted: a stage index outside the
    schedulable stages) - `on_env_error="truncate"`: the env leaves the collection, the record is the same with and without
    `record_on_device` (flags read late or not); `"raise"`: both raise, naming the env and its step"""
    import pytest

    cfg = dict(num_executors=10, job_arrival_cap=8, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    dev = torch.device(device)
    bad_env, bad_step = 2, 9

    def collector(on_dev, mode):
        env = VecSparkSchedSimEnv(cfg, num_envs, device=device, auto_reset=False, _lib=lib)
        torch.manual_seed(1)
        pol = DecimaPolicy(num_executors=10, **AGENT).to(dev)
        gen = torch.Generator(device=dev if dev.type == "cuda" else "cpu")
        gen.manual_seed(5)
        col = RolloutCollector(env, 5.0e5, list(range(31, 31 + num_envs)), seed_step=num_envs, num_executors=10, policy=pol, generator=gen,
                               record_on_device=on_dev, on_env_error=mode)
        act, calls = pol.act, [0]

        def corrupted(g, generator=None):
            a = act(g, generator)
            if calls[0] == bad_step:
                a["stage_sel"][bad_env] = 10_000
                if "env_stage_idx" in a:
                    a["env_stage_idx"][bad_env] = 10_000
            calls[0] += 1
            return a
        pol.act = corrupted
        return env, col

    out = {}
    for on_dev in (True, False):
        env, col = collector(on_dev, "truncate")
        out[on_dev] = col.collect_sync(with_stats=False)
        assert col.env_errors == 1 and (col._arena_sizes is not None) == on_dev
        env.close()
    ra, rb = out[True], out[False]
    assert int(ra.active[:, bad_env].sum()) == bad_step  # the failing step is not recorded, the env sits out the rest
    for name in ("active", "t_before", "t_after", "rewards", "stage_sel", "job_idx", "exec_sel", "lgprobs", "resets"):
        assert torch.equal(getattr(ra, name), getattr(rb, name)), name
    for k, v in rb.graph.items():
        if torch.is_tensor(v):
            assert torch.equal(ra.graph[k], v), k
    for on_dev in (True, False):
        env, col = collector(on_dev, "raise")
        with pytest.raises(RuntimeError) as ei:
            col.collect_sync(with_stats=False)
        assert f"env {bad_env} " in str(ei.value) and f"rollout step {bad_step}" in str(ei.value)
        env.close()
    # the LAST step of a collection fails for every env that is left (one env, its action corrupted): the step is dropped from the
    # record, but it took a counter of the policy's sampling stream - with the flags read late as well, so that the collection
    # AFTER it draws the same samples either way
    num_envs_saved, num_envs = num_envs, 1
    bad_env = 0
    two = {}
    for on_dev in (True, False):
        env, col = collector(on_dev, "truncate")
        first = col.collect_sync(with_stats=False)
        calls_after = col.policy._calls
        second = col.collect_sync(with_stats=False)
        two[on_dev] = (first, calls_after, second)
        assert int(first.active.sum()) == bad_step and col.env_errors == 1
        env.close()
    assert two[True][1] == two[False][1] == bad_step + 1
    for ra, rb in ((two[True][0], two[False][0]), (two[True][2], two[False][2])):
        for name in ("active", "t_before", "t_after", "rewards", "stage_sel", "job_idx", "exec_sel", "lgprobs"):
            assert torch.equal(getattr(ra, name), getattr(rb, name)), name
    assert int(two[True][2].active.sum()) > 0
    num_envs = num_envs_saved


def check_segment_categorical(binding, device, n_seg):
    """`train_kernels.segment_categorical` (include/sss.h sss_segment_categorical) against the tensor-op forms of
    `DecimaPolicy.evaluate_actions` - decima._segment_log_softmax + index_add_ for the stage softmax (den_eps 1e-16), torch.softmax +
    clamp + log for the executor softmax (den_eps 0) - values and, through a random linear functional of (lg, ent), the gradients
    w.r.t. the scores; empty segments, one-row segments, a probability small enough for the clamp to bite"""
    from spark_sched_sim_amd.decima import _excl_cumsum, _segment_log_softmax
    from spark_sched_sim_amd.train_kernels import segment_categorical

    gen = torch.Generator().manual_seed(23)
    sizes = torch.randint(0, 12, (n_seg,), generator=gen)
    sizes[0], sizes[1], sizes[-1] = 1, 0, 0
    ptr = torch.zeros(n_seg + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(sizes, 0)
    rows = int(ptr[-1])
    scores = (torch.randn(rows, generator=gen) * 3.0)
    big = int(torch.argmax(sizes))
    scores[int(ptr[big])] = -40.0  # (its probability falls below eps: the clamp bites, no gradient through it)
    chosen = (torch.rand(n_seg, generator=gen) * sizes.clamp(min=1)).long().clamp(max=(sizes - 1).clamp(min=0))
    owner = torch.repeat_interleave(torch.arange(n_seg), sizes)
    w_lg, w_ent = torch.randn(n_seg, generator=gen), torch.randn(n_seg, generator=gen)
    full = sizes > 0
    ptr_d, chosen_d = ptr.to(device), chosen.to(device)
    for den_eps in (1e-16, 0.0):
        s_ref = scores.clone().to(device).requires_grad_(True)
        if den_eps:
            p, lp = _segment_log_softmax(s_ref, owner.to(device), n_seg)
        else:  # torch.softmax over the padded [n_seg, max] matrix, as evaluate_actions does for the executor counts
            width = int(sizes.max())
            col = torch.arange(rows) - _excl_cumsum(sizes)[owner]
            mat = torch.full((n_seg, width), float("-inf"), device=device).index_put((owner.to(device), col.to(device)), s_ref)
            pe = torch.softmax(mat, 1)
            eps = torch.finfo(pe.dtype).eps
            pe = torch.where(torch.isfinite(mat), pe.clamp(min=eps, max=1 - eps), torch.ones_like(pe))
            p, lp = pe[owner.to(device), col.to(device)], pe.log()[owner.to(device), col.to(device)]
        lg_ref = torch.zeros(n_seg, device=device)
        lg_ref[full.to(device)] = lp[(ptr[:-1] + chosen)[full].to(device)]
        ent_ref = -torch.zeros(n_seg, device=device).index_add_(0, owner.to(device), lp * p)
        (lg_ref * w_lg.to(device) + ent_ref * w_ent.to(device)).sum().backward()
        s_k = scores.clone().to(device).requires_grad_(True)
        lg, ent = segment_categorical(s_k, ptr_d, chosen_d, den_eps, binding=binding)
        (lg * w_lg.to(device) + ent * w_ent.to(device)).sum().backward()
        assert torch.allclose(lg, lg_ref.detach(), rtol=1e-5, atol=1e-5) and torch.allclose(ent, ent_ref.detach(), rtol=1e-5, atol=1e-5), den_eps
        assert torch.allclose(s_k.grad, s_ref.grad, rtol=1e-4, atol=1e-5), (den_eps, float((s_k.grad - s_ref.grad).abs().max()))
        assert float(s_k.grad[int(ptr[big])]) == 0.0 or abs(float(s_k.grad[int(ptr[big])])) < 1e-12
