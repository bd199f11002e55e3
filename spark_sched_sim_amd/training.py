"""Rollout collection and PPO over the batched env.

The reference trains with one OS process per env: a `RolloutWorker` loop per process
(reference trainers/rollout_worker.py:118-206), a Pipe gather, then returns / baselines / the CLIP
loss on the learner (trainers/trainer.py:171-216, trainers/utils/returns_calculator.py:45-76,
trainers/utils/baselines.py:12-37, trainers/ppo.py:51-138). Here every "worker" is one env of a
`VecSparkSchedSimEnv`, all envs advance together on the GPU, the recorded observations stay on the
device as one compact graph (`decima.compact_graph`), and the per-rollout post-processing is a set
of padded tensor ops. The arithmetic per element follows the reference's; tolerances are in the tests.

Layout of a `Rollouts` record (T = longest rollout of the batch, B = envs): every per-step quantity
is a [T, B] tensor and `active[t, b]` says whether env b recorded a step at position t (a rollout's
steps are always a prefix). Observation (t, b) is observation id t*B + b of `graph`.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Sequence

import numpy as np
import torch

from .decima import concat_graphs, select_observations
from .wrappers import VecStochasticTimeLimit

SKIP_ENV = -(2 ** 31)  # include/sss.h SSS_SKIP_ENV
EPS = 1e-8             # trainers/ppo.py:12


@dataclass
class Rollouts:
    graph: dict[str, Any]
    active: torch.Tensor        # bool[T,B]
    t_before: torch.Tensor      # f64[T,B]  wall time (sync) / elapsed time (async) when the action was taken
    t_after: torch.Tensor       # f64[T,B]  the same clock after the step
    rewards: torch.Tensor       # f64[T,B]
    stage_sel: torch.Tensor     # i64[T,B]
    job_idx: torch.Tensor       # i64[T,B]
    exec_sel: torch.Tensor      # i64[T,B]
    lgprobs: torch.Tensor       # f32[T,B]
    resets: torch.Tensor        # bool[T,B] (async: the env was reset right after this step)
    stats: dict[str, np.ndarray] = field(default_factory=dict)
    obs_index: torch.Tensor | None = None  # i64[T,B]: which observation of `graph` belongs to (t, b); None: t * B + b

    @property
    def lengths(self) -> torch.Tensor:
        return self.active.sum(0)

    def rollout(self, b: int) -> dict[str, np.ndarray]:
        """env b's record in the shape of the reference's `RolloutBuffer` (rollout_worker.py:18-45)"""
        n = int(self.active[:, b].sum())
        tb, ta = self.t_before[:n, b].cpu().numpy(), self.t_after[:n, b].cpu().numpy()
        return {"wall_times": np.concatenate([tb, ta[-1:]]) if n else np.zeros(1),
                "rewards": self.rewards[:n, b].cpu().numpy(),
                "actions": torch.stack([self.stage_sel[:n, b], self.job_idx[:n, b], self.exec_sel[:n, b]], 1).cpu().numpy(),
                "lgprobs": self.lgprobs[:n, b].cpu().numpy(),
                "resets": self.resets[:n, b].nonzero(as_tuple=True)[0].cpu().numpy()}

    def sample_ids(self) -> torch.Tensor:
        """observation ids of all recorded steps in the reference's order (rollout-major:
        `chain(*obsns_list)`, ppo.py:58-62)"""
        T, B = self.active.shape
        ids = self.obs_index if self.obs_index is not None else (
            torch.arange(T, device=self.active.device)[:, None] * B + torch.arange(B, device=self.active.device)[None, :])
        return ids.t()[self.active.t()]

    def flat(self, x: torch.Tensor) -> torch.Tensor:
        """a [T,B] quantity as the flat per-sample vector matching `sample_ids`"""
        return x.t()[self.active.t()]


ActFn = Callable[[dict[str, Any], torch.Tensor], dict[str, torch.Tensor]]


class GraphArena:
    """The record of a collection's observations as ONE compact graph (what `concat_graphs` of the per-step graphs gives), built
    on the device while collecting: `append` copies the step's graph out of the env's capacity buffers (`decima_graph_on_device`)
    behind what is there, at cursors that stay on the device, shifting its ids to arena ids (include/sss.h sss_arena_append).
    The host never learns a step's sizes; it keeps `headroom` steps' worth of the env's capacities free beyond the last cursor
    it has seen (`ensure`), growing the arrays geometrically when that is not the case."""

    NODES, EDGES, JOBS, OBS = 0, 1, 2, 3
    # (name, dtype, elements per row, what it has one row per, what its ids name: 0 nothing, 1 nodes, 2 jobs, 3 observations)
    ARRAYS = (("x", torch.float32, 5, 0, 0), ("node_obs", torch.int64, 1, 0, 3), ("node_loc", torch.int64, 1, 0, 0), ("node_job", torch.int64, 1, 0, 2),
              ("gen", torch.int32, 1, 0, 0), ("stage_mask", torch.bool, 1, 0, 0), ("sched_rank", torch.int64, 1, 0, 0), ("node_recv", torch.int32, 1, 0, 0),
              ("src", torch.int64, 1, 1, 1), ("dst", torch.int64, 1, 1, 1), ("edge_obs", torch.int64, 1, 1, 3), ("edge_layers", torch.int32, 1, 1, 0),
              ("job_obs", torch.int64, 1, 2, 3), ("job_cap", torch.int64, 1, 2, 0), ("job_first", torch.int64, 1, 2, 1),
              ("obs_nodes", torch.int64, 1, 3, 0), ("obs_jobs", torch.int64, 1, 3, 0), ("obs_depth", torch.int32, 1, 3, 0))

    def __init__(self, env, capacity: Sequence[int] | None = None):
        self.env, self.dev, self.B = env, env.device, env.num_envs
        d = env.dims
        self.step_max = (self.B * d.node_cap, self.B * d.edge_cap, self.B * d.job_cap, self.B)  # what one step can add at most
        self.capacity = [max(int(c), 16 * m) for c, m in zip(capacity or (0, 0, 0, 0), self.step_max)]
        self.capacity[3] = max(self.capacity[3], 1024 * self.B)
        self.cursor = torch.zeros(8, dtype=torch.int64, device=self.dev)
        self.seen = [0, 0, 0, 0]   # the last cursors the host has seen
        self.seen_steps = 0        # ... and how many steps they cover
        self.steps = 0             # steps appended (enqueued)
        self.buf = {name: torch.empty((self.capacity[kind], per) if per > 1 else (self.capacity[kind],), dtype=dt, device=self.dev)
                    for name, dt, per, kind, _ in self.ARRAYS}
        self._args = None

    def _build_args(self, g):
        from .binding import SssArenaArgs, SssArenaArray
        a = SssArenaArgs()
        a.n_arrays, a.n_obs = len(self.ARRAYS), self.B
        a.totals_dev, a.cursor_dev = g["totals_dev"].data_ptr(), self.cursor.data_ptr()
        for k in range(4):
            a.capacity[k] = self.capacity[k]
        a.rows_hint = 0
        for i, (name, dt, per, kind, shift) in enumerate(self.ARRAYS):
            src = g[name]
            assert src.dtype == dt and src.is_contiguous(), name
            a.arrays[i] = SssArenaArray(src.data_ptr(), self.buf[name].data_ptr(), src.element_size(), per, kind, shift)
        self._args = (a, tuple(g[name].data_ptr() for name, *_ in self.ARRAYS) + (g["totals_dev"].data_ptr(),))

    def note(self, cursors: Sequence[int], steps: int) -> None:
        """the host has seen the cursors as they were after `steps` appended steps"""
        self.seen, self.seen_steps = [int(c) for c in cursors[:4]], int(steps)

    def ensure(self, sync) -> None:
        """room for the next step whatever its size: the steps the host has not seen the cursors of may each have added a
        step's maximum. Grows (after `sync()`, which must leave `note` up to date) when that bound does not fit."""
        def short():
            ahead = self.steps - self.seen_steps + 1
            return [k for k in range(4) if self.seen[k] + ahead * self.step_max[k] > self.capacity[k]]
        if not short():
            return
        sync()
        for k in short():
            # (x 1.3, not x 2: at BASELINE config 5 the record is ~21 GB per rank and a doubled arena can hold twice that during a first
            # collection; a growth step is one device copy of the arrays of that kind, ~10 ms)
            new_cap = max(int(1.3 * self.capacity[k]), self.seen[k] + 8 * self.step_max[k])
            for name, dt, per, kind, _ in self.ARRAYS:
                if kind == k:
                    old = self.buf[name]
                    self.buf[name] = torch.empty((new_cap, *old.shape[1:]), dtype=dt, device=self.dev)
                    self.buf[name][: self.seen[k]] = old[: self.seen[k]]
            self.capacity[k] = new_cap
        self._args = None

    def append(self, g: dict[str, Any], rows_hint: int = 0) -> None:
        import ctypes

        from .binding import device_of
        key = tuple(g[name].data_ptr() for name, *_ in self.ARRAYS) + (g["totals_dev"].data_ptr(),)
        if self._args is None or self._args[1] != key:
            self._build_args(g)
        a = self._args[0]
        a.rows_hint = int(rows_hint)
        with device_of(self.dev):
            self.env._b.check(self.env._b.lib.sss_arena_append(ctypes.byref(a), self.env._stream()))
        self.steps += 1

    def finish(self, n_steps: int, sizes: Sequence[int] | None = None) -> dict[str, Any]:
        """the graph of the first `n_steps` steps: `sizes` = the cursors (nodes, edges, jobs) as they were after that many steps
        (`note`), default: everything appended - the steps behind them are dropped"""
        cur = [int(c) for c in self.cursor[:6].tolist()]
        if cur[5]:
            raise RuntimeError("the graph arena overflowed (the headroom rule of GraphArena.ensure was violated)")
        if sizes is None:
            sizes = cur[:3]
        assert all(0 <= int(s) <= c for s, c in zip(sizes[:3], cur[:3])) and n_steps <= cur[4]
        sizes = (int(sizes[0]), int(sizes[1]), int(sizes[2]), n_steps * self.B)
        out: dict[str, Any] = {name: self.buf[name][: sizes[kind]] for name, _, _, kind, _ in self.ARRAYS}
        out["gen"] = out["gen"].long()
        out["n_obs"], out["n_pad"] = n_steps * self.B, self.env.dims.node_cap
        out["_by_observation"] = True  # (appended step by step, env by env: decima.select_observations cuts such a record by ranges)
        return out


class RolloutCollector:
    """the reference's `RolloutWorkerSync` / `RolloutWorkerAsync` loops for all envs at once.

    Env b plays worker rank b: seed of its k-th episode = base_seeds[b] + seed_step * k
    (rollout_worker.py:118-120, trainer.py:264-269), time limit from `StochasticTimeLimit`'s rule.
    `act_fn(g, step_counts)` (g = compact graph of the active envs' observations) returns
    `DecimaPolicy.act`'s dict; the default samples from `policy`. `step_counts` is an int64 tensor ON THE ENV'S DEVICE (the
    collection kernels update it in place): an `act_fn` that mixes it with CPU tensors has to move it (`.cpu()` is one sync
    per step). Envs that are done (sync) or have filled their duration (async) are frozen with
    SSS_SKIP_ENV until the others catch up."""

    def __init__(self, env, mean_time_limit: float, base_seeds: Sequence[int], seed_step: int, num_executors: int,
                 policy=None, act_fn: ActFn | None = None, generator: torch.Generator | None = None, on_env_error: str = "raise", groups: int = 1,
                 record_on_device: bool = True):
        """groups: the envs are split into that many groups that take their steps alternately, each on its own HIP stream (the
        idea: while one group's step launch waits for its slowest env the other group's policy kernels have the device). The
        record and the results per env are the same as with one group; on one MI355X it is SLOWER (every launch of the policy
        pass is latency-bound and costs a half-size group as much as the whole batch: 0.88 -> 1.84 ms per row of the record
        with two groups, profiles/r03_ppo.md), so 1 is the default.
        record_on_device: synchronous collection with the default policy sampling keeps the host out of the loop - the step's
        graph stays in the env's capacity buffers with its sizes on the device, `GraphArena` appends it to the record there, and
        the flags of a step are read a few steps late (the steps enqueued in between find every env frozen when the collection
        turns out to be over). The record is the same; False reads the sizes and the flags every step (two waits per step).
        on_env_error: what to do when an env reports an error from `step` - in practice the
        reference's `[step]` assertion (spark_sched_sim.py:212-215), which valid Decima actions can
        trigger (tests/golden/stall_case.json). "raise" = the reference's behaviour (the worker
        aborts, the trainer stops, rollout_worker.py:110-112 / trainer.py:117-124); "truncate" = the
        rollout ends before the failing step and training goes on (`env_errors` counts them)."""
        assert on_env_error in ("raise", "truncate")
        self.groups = max(1, min(int(groups), env.num_envs))
        self.record_on_device = bool(record_on_device)
        self._arena_sizes = None  # the previous collection's graph sizes: the next arena's starting capacity
        self.on_env_error = on_env_error
        self.env_errors = 0
        self.env = env
        self.tl_env = VecStochasticTimeLimit(env, mean_time_limit)
        self.base_seeds = np.asarray(base_seeds, dtype=np.int64)
        assert self.base_seeds.shape == (env.num_envs,)
        self.seed_step = int(seed_step)
        self.reset_count = np.zeros(env.num_envs, dtype=np.int64)
        self.step_counts = torch.zeros(env.num_envs, dtype=torch.long, device=env.device)
        self.E = num_executors
        self.policy = policy
        self.generator = generator
        if policy is not None:
            policy.bind_kernels(env._b)  # inference on the fused GNN kernels; training stays on autograd
        self.act_fn = act_fn
        self._obs = None
        self._wall = None
        self._pending_reset = None

    @property
    def seeds(self) -> np.ndarray:
        return self.base_seeds + self.seed_step * self.reset_count

    def _reset(self, mask: torch.Tensor | None = None):
        obs, _ = self.tl_env.reset(seed=[int(s) for s in self.seeds], mask=mask)
        self.reset_count += 1 if mask is None else mask.cpu().numpy().astype(np.int64)
        return obs

    def _stats(self) -> dict[str, np.ndarray]:
        """rollout_worker.py:122-130 for every env"""
        return {k: v.cpu().numpy() for k, v in self.env.rollout_stats().items()}

    def _loop(self, asynchronous: bool, duration: float, with_stats: bool) -> Rollouts:
        """Both worker loops. Per step of a group of envs: the compact graph of its active envs' observations (recorded),
        the policy's sample, `sss_collect_step` phase 0 (actions; the other groups' envs skip), `sss_step`,
        `sss_collect_step` phase 1 (time limit, who failed / finished / goes on, the step's row of the record, the envs'
        clocks). The flags the control flow needs are read when the group's NEXT step is about to be enqueued - by then the
        other groups' work has been enqueued behind it (one device->host read per group and step, plus the graph totals).
        The record lives in [T_cap, B] device arrays that grow geometrically."""
        import contextlib
        import ctypes

        from .binding import SssCollectArgs, device_of
        env, dev, B, G = self.env, self.env.device, self.env.num_envs, self.groups
        if not asynchronous or self._obs is None:
            self._obs = self._reset()
            self._wall = torch.zeros(B, dtype=torch.float64, device=dev)
        wall = self._wall
        if asynchronous and self._pending_reset is not None:
            self._reset(mask=self._pending_reset)
            wall = torch.where(self._pending_reset, torch.zeros_like(wall), wall)
        self._pending_reset = None
        wall = wall.clone()
        elapsed = torch.zeros(B, dtype=torch.float64, device=dev)
        active = torch.ones(B, dtype=torch.uint8, device=dev)  # (bool view for the graph kernel and the caller's policy)
        pending = torch.zeros(B, dtype=torch.uint8, device=dev)
        spec = (("active", torch.uint8), ("t_before", torch.float64), ("t_after", torch.float64), ("rewards", torch.float64), ("stage_sel", torch.int64),
                ("job_idx", torch.int64), ("exec_sel", torch.int64), ("lgprobs", torch.float32), ("resets", torch.uint8))
        cap = 1024
        rec = {k: torch.zeros((cap, B), dtype=dt, device=dev) for k, dt in spec}  # (zeros: rows a group never reaches read as inactive)
        flags = torch.zeros((cap, G, 8), dtype=torch.int32, device=dev)
        group_of = (torch.arange(B, device=dev) * G) // max(B, 1)
        member = [(group_of == k).to(torch.uint8) for k in range(G)] if G > 1 else [None]
        member_b = [m.view(torch.bool) if m is not None else None for m in member]
        use_streams = G > 1 and dev.type == "cuda"
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        # the groups' streams are created once per collector: the per-stream work buffers of the graph / encoder calls
        # (vec_env._layer_scratch, decima._enc_scratch) are keyed by stream and would pile up with fresh streams per call
        if use_streams and len(getattr(self, "_streams", ())) != G:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
        streams = list(self._streams) if use_streams else [None] * G
        for st in streams:
            if st is not None:
                st.wait_stream(main)
        acts = [(torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(G)]
        lib = env._b.lib
        # the loop without waits (see __init__, record_on_device)
        fast = (self.record_on_device and not asynchronous and G == 1 and self.act_fn is None and self.policy is not None and B > 0
                and self.policy._use_kernels() and env.graph_kernel_fits)
        LAG, R = 4, 8  # flags are read at most LAG steps late; rings of R slots for what a step leaves for the host
        arena = None
        if fast:
            cuda = dev.type == "cuda"
            pin = (lambda t: t.pin_memory()) if cuda else (lambda t: t)
            cap0 = None
            if self._arena_sizes:
                d = env.dims
                step_max = (B * d.node_cap, B * d.edge_cap, B * d.job_cap, B)
                cap0 = [int(1.1 * c) + (LAG + 4) * m for c, m in zip(self._arena_sizes, step_max)]
            arena = GraphArena(env, cap0)
            ring_flags, ring_cur = pin(torch.zeros((R, 8), dtype=torch.int32)), pin(torch.zeros((R, 8), dtype=torch.int64))
            ring_acts = [(torch.empty(B, dtype=torch.int32, device=dev), torch.empty(B, dtype=torch.int32, device=dev)) for _ in range(R)]
            ring_ev = [torch.cuda.Event() for _ in range(R)] if cuda else None
        n_kept, kept_sizes = 0, [0, 0, 0, 0]  # fast: steps whose observations are part of the record, the arena's cursors after them
        n_lockstep = 0  # steps a loop that reads every step's flags before issuing the next would have issued (a dropped last step included)
        calls0 = getattr(self.policy, "_calls", 0) if fast else 0
        graphs: dict[tuple[int, int], dict[str, Any]] = {}
        issued = [0] * G             # steps enqueued per group
        unread: list[Any] = [None] * G  # (t, graph, stage_idx, num_exec) of the group's step whose flags have not been read yet
        alive = [B > 0] * G
        n_failed = 0

        def sync_all():
            if use_streams:
                for st in streams:
                    st.synchronize()

        def read_flags(k: int) -> None:
            """what became of group k's last step: the ONE device->host read per group and step beside the graph totals"""
            t, g, stage_idx, num_exec = unread[k]
            unread[k] = None
            handle_flags(k, t, g, stage_idx, num_exec, flags[t, k, :5].tolist())

        def handle_flags(k: int, t: int, g, stage_idx, num_exec, vals) -> None:
            nonlocal n_failed, n_kept, kept_sizes, n_lockstep
            any_bad, any_done, any_left, bad_env, any_recorded = vals
            n_lockstep = t + 1
            if any_bad:
                n_bad = int(pending.sum()) - n_failed
                n_failed += n_bad
                self.env_errors += n_bad
                if self.on_env_error == "raise":
                    from .binding import ERROR_NAMES
                    b = bad_env - 1
                    code = int(env.obs_i32[b, 7])
                    err = RuntimeError(f"env {b} (seed {int(self.seeds[b] - self.seed_step)}), rollout step {t}: "
                                       f"{ERROR_NAMES.get(code, code)}; action stage_idx={int(stage_idx[b])} num_exec={int(num_exec[b])}")
                    # what a bug report needs: the env's seed, time limit and action history
                    err.case = {"seed": int(self.seeds[b] - self.seed_step), "time_limit": float(self.tl_env.time_limit[b]), "code": code,
                                "stage_idx": [int(x) for x in rec["stage_sel"][:t, b]] + [int(stage_idx[b])],
                                "num_exec": [int(x) + 1 for x in rec["exec_sel"][:t, b]] + [int(num_exec[b])]}
                    raise err
                # truncate: the failing step is not recorded and the env sits out the rest of this
                # collection (async: it starts its next episode at the next collection)
                if not any_recorded:
                    alive[k] = False
                    return
            if g is not None:
                graphs[(t, k)] = g
            n_kept = t + 1
            if arena is not None:
                kept_sizes = list(arena.seen)  # (take_late has just noted the cursors after this step)
            if asynchronous and any_done:
                done = rec["resets"][t].view(torch.bool)
                self._reset(mask=done if member_b[k] is None else done & member_b[k])
            alive[k] = bool(any_left)

        def enqueue(k: int) -> None:
            nonlocal cap, flags
            t = issued[k]
            if t == cap:  # grow the record (every stream has to be done with the old arrays)
                sync_all()
                for name in rec:
                    rec[name] = torch.cat([rec[name], torch.zeros_like(rec[name])])
                flags = torch.cat([flags, torch.zeros_like(flags)])
                cap *= 2
            act_b = active.view(torch.bool) if member_b[k] is None else active.view(torch.bool) & member_b[k]
            if fast:
                arena.ensure(drain_all)
                g = self.env.decima_graph_on_device(act_b)  # (sizes stay on the device; appended to the arena below)
            else:
                g = self.env.decima_graph(act_b)  # recorded for training
            a = self.act_fn(g, self.step_counts) if self.act_fn is not None else self.policy.act(g, self.generator)
            if fast:
                hint = g["totals_hint"].tolist()
                arena.append(g, rows_hint=max(5 * hint[0], hint[1], hint[2], B) if hint[0] >= 0 else 0)
            sel = [a[n] if a[n].dtype == torch.int64 and a[n].is_contiguous() else a[n].to(torch.int64).contiguous() for n in ("stage_sel", "job_idx", "exec_sel")]
            lg = a["lgprob"] if a["lgprob"].dtype == torch.float32 and a["lgprob"].is_contiguous() else a["lgprob"].float().contiguous()
            stage_idx, num_exec = ring_acts[t % R] if fast else acts[k]
            c = SssCollectArgs(B, int(asynchronous), t, float(duration), env.obs_f64.data_ptr(), env.obs_i32.data_ptr(), env.obs_i32.stride(0),
                               self.tl_env.time_limit.data_ptr(), active.data_ptr(), wall.data_ptr(), elapsed.data_ptr(), self.step_counts.data_ptr(),
                               pending.data_ptr(), sel[0].data_ptr(), sel[1].data_ptr(), sel[2].data_ptr(), lg.data_ptr(), stage_idx.data_ptr(), num_exec.data_ptr(),
                               *(rec[name].data_ptr() for name, _ in spec), flags[t, k].data_ptr(), member[k].data_ptr() if member[k] is not None else None)
            stream = env._stream()
            with device_of(dev):
                env._b.check(lib.sss_collect_step(ctypes.byref(c), 0, stream))
            env.step_async(stage_idx, num_exec)
            with device_of(dev):
                env._b.check(lib.sss_collect_step(ctypes.byref(c), 1, stream))
            if fast:  # what the host wants of this step, without waiting for it: its flags and the arena's cursors after it
                ring_flags[t % R].copy_(flags[t, k], non_blocking=True)
                ring_cur[t % R].copy_(arena.cursor, non_blocking=True)  # (ring_cur[.., 4] = steps appended: which step the cursors are of)
                if ring_ev is not None:
                    ring_ev[t % R].record(torch.cuda.current_stream(dev))
                unread[k] = None
                late.append((t, stage_idx, num_exec))
            else:
                unread[k] = (t, g, stage_idx, num_exec)
            issued[k] = t + 1

        late: list[Any] = []  # fast: steps enqueued whose flags have not been looked at, oldest first

        def take_late() -> None:
            """the oldest enqueued step's flags (waits for that step if it has not finished)"""
            t, stage_idx, num_exec = late.pop(0)
            if ring_ev is not None:
                ring_ev[t % R].synchronize()
            arena.note(ring_cur[t % R].tolist(), t + 1)
            handle_flags(0, t, None, stage_idx, num_exec, ring_flags[t % R, :5].tolist())

        def drain_all() -> None:
            while late and alive[0]:
                take_late()

        try:
            while fast and alive[0]:
                # look at the steps that have finished, and never run more than LAG steps ahead of the flags
                while late and alive[0] and (len(late) >= LAG or ring_ev is None or ring_ev[late[0][0] % R].query()):
                    take_late()
                if alive[0]:
                    enqueue(0)
                    if ring_ev is None:
                        take_late()
            while not fast and (any(alive) or any(u is not None for u in unread)):
                for k in range(G):
                    with (torch.cuda.stream(streams[k]) if streams[k] is not None else contextlib.nullcontext()):
                        if unread[k] is not None:
                            read_flags(k)
                        if alive[k]:
                            enqueue(k)
        finally:
            sync_all()
            if use_streams:
                for st in streams:
                    main.wait_stream(st)
        if n_failed:
            self._pending_reset = pending.view(torch.bool)
        self._obs, self._wall = True, wall  # (_obs: the envs are inside their episodes)
        keys = sorted(graphs)
        T = n_kept if fast else max((t for t, _ in keys), default=-1) + 1
        out = {name: rec[name][:T] for name, _ in spec}
        obs_index = None
        if G > 1:  # observation (t, b) sits in the graph its group recorded at step t
            pos = np.full((max(T, 1), G), 0, dtype=np.int64)
            for i, (t, k) in enumerate(keys):
                pos[t, k] = i
            obs_index = torch.from_numpy(pos[:T]).to(dev)[:, group_of] * B + torch.arange(B, device=dev)[None, :]
        if fast:
            if dev.type == "cuda":
                torch.cuda.current_stream(dev).synchronize()
            graph = arena.finish(T, kept_sizes)  # (a last step that failed for every env left is dropped, like the steps enqueued behind the end)
            # the steps enqueued behind the last one found every env frozen, but each took a draw counter of the policy's sampling
            # stream (decima._sample_kernels): hand them back, so that the next collection draws what it would have drawn - the loop
            # that waits for every step's flags has issued n_lockstep steps (a last step that failed for every env left took its
            # counter too, although it is not part of the record)
            if hasattr(self.policy, "_calls"):
                self.policy._calls = calls0 + n_lockstep
            self._arena_sizes = [int(graph["x"].shape[0]), int(graph["src"].numel()), int(graph["job_obs"].numel()), T * B]
        else:
            graph = concat_graphs([graphs[key] for key in keys])
        return Rollouts(graph=graph, active=out["active"].view(torch.bool), t_before=out["t_before"], t_after=out["t_after"],
                        rewards=out["rewards"], stage_sel=out["stage_sel"], job_idx=out["job_idx"], exec_sel=out["exec_sel"],
                        lgprobs=out["lgprobs"], resets=out["resets"].view(torch.bool), stats=self._stats() if with_stats else {}, obs_index=obs_index)

    def collect_sync(self, with_stats: bool = True) -> Rollouts:
        """one full episode per env (rollout_worker.py:133-159)"""
        return self._loop(False, 0.0, with_stats)

    def collect_async(self, rollout_duration: float, with_stats: bool = True) -> Rollouts:
        """`rollout_duration` ms of simulated time per env, episodes restarting in place
        (rollout_worker.py:162-206)"""
        return self._loop(True, float(rollout_duration), with_stats)


# ---- returns (trainers/utils/returns_calculator.py) -----------------------------------------------

def _record_kernels(ro: Rollouts, binding=None):
    """the binding to run the [T, B] record kernels with (include/sss.h sss_discounted_returns / sss_sequence_baselines): the
    library when the record is on a GPU - no fallback there - or the one a caller passes (tests: the emulator's host backend)"""
    if binding is not None:
        return binding
    if ro.active.is_cuda:
        from .train_kernels import _binding
        return _binding()
    return None


def _c64(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if t.dtype == torch.float64 else t.double().contiguous()


def discounted_returns(ro: Rollouts, beta: float, binding=None) -> torch.Tensor:
    """R_k = r_k + exp(-beta * 1e-3 * dt_k) * R_{k+1} per rollout (returns_calculator.py:67-76); f64[T,B]"""
    T, B = ro.active.shape
    b = _record_kernels(ro, binding)
    if b is not None and T > 0 and B > 0:  # one kernel, a thread per env (the loop below: T x 5 launches, 0.30 s at BASELINE config 5)
        import ctypes

        from .binding import SssReturnsArgs, device_of
        dev = ro.active.device
        act, tb, ta, rw = ro.active.contiguous().view(torch.uint8), _c64(ro.t_before), _c64(ro.t_after), _c64(ro.rewards)
        out = torch.empty((T, B), dtype=torch.float64, device=dev)
        a = SssReturnsArgs(T, B, act.data_ptr(), tb.data_ptr(), ta.data_ptr(), rw.data_ptr(), float(beta), out.data_ptr())
        with device_of(dev):
            b.check(b.lib.sss_discounted_returns(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))
        return out
    dt = ro.t_after - ro.t_before
    out = torch.zeros_like(ro.rewards)
    R = torch.zeros(B, dtype=torch.float64, device=ro.rewards.device)
    for k in range(T - 1, -1, -1):
        R = torch.where(ro.active[k], ro.rewards[k] + torch.exp(-beta * 1e-3 * dt[k]) * R, R)
        out[k] = R
    return out * ro.active


class DifferentialReturns:
    """average-reward ("differential") returns with the moving estimate of the mean number of
    concurrent jobs kept in a circular buffer of (dt, reward) rows (returns_calculator.py:6-65, 78-89)"""

    def __init__(self, buff_cap: int):
        self.cap = int(buff_cap)
        self.data = np.zeros((self.cap, 2))
        self.avg_num_jobs: float | None = None

    def _extend(self, new: np.ndarray) -> None:
        if new.shape[0] > self.cap:
            new = new[-self.cap:]
        keep = self.cap - new.shape[0]
        if keep > 0:
            self.data[:keep] = self.data[-keep:]
        self.data[keep:] = new

    def __call__(self, ro: Rollouts) -> torch.Tensor:
        dt = ro.t_after - ro.t_before
        rows = np.stack([ro.flat(dt).cpu().numpy(), ro.flat(ro.rewards).cpu().numpy()], 1)
        self._extend(rows[rows[:, 0] > 0])
        total_time, rew_sum = self.data.sum(0)
        self.avg_num_jobs = -rew_sum / total_time
        T, B = ro.active.shape
        out = torch.zeros_like(ro.rewards)
        R = torch.zeros(B, dtype=torch.float64, device=ro.rewards.device)
        for k in range(T - 1, -1, -1):
            # R = -(job_time - expected_job_time) + R with job_time = -r (returns_calculator.py:57-60)
            R = torch.where(ro.active[k], -(-ro.rewards[k] - dt[k] * self.avg_num_jobs) + R, R)
            out[k] = R
        return out * ro.active


# ---- baselines (trainers/utils/baselines.py) --------------------------------------------------------

def _interp(x: torch.Tensor, xp: torch.Tensor, fp: torch.Tensor, n: torch.Tensor) -> torch.Tensor:
    """numpy.interp along the last axis for batches: xp rows are non-decreasing with `n` valid
    entries (the rest padding). Same case analysis as numpy's compiled_interp: clamp outside the
    range, the LAST knot j with xp[j] <= x, exact hit -> fp[j], else slope * (x - xp[j]) + fp[j]."""
    Tm = xp.shape[-1]
    ar = torch.arange(Tm, device=xp.device)
    xp_s = torch.where(ar < n[..., None], xp, torch.full_like(xp, float("inf")))
    j = (torch.searchsorted(xp_s, x.contiguous(), right=True) - 1).clamp(min=0)
    last = (n[..., None] - 1).clamp(min=0).expand_as(j)
    j = torch.minimum(j, last)
    j1 = torch.minimum(j + 1, last)
    x0, x1, y0, y1 = xp.gather(-1, j), xp.gather(-1, j1), fp.gather(-1, j), fp.gather(-1, j1)
    slope = (y1 - y0) / (x1 - x0)
    v = slope * (x - x0) + y0
    exact = (x == x0) | (j == last) | (x < x0)
    return torch.where(exact, y0, v)


def numpy_sum_order(terms: list) -> Any:
    """the sum of `terms` (tensors, or floats) in the order numpy's `add.reduce` takes over a float64 axis - what `y_hat.mean()`
    does in baselines.py:33 (numpy's DOUBLE_pairwise_sum): fewer than 8 terms one after the other from 0.0; up to 128 terms eight
    strided partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the tail; above that the range is halved (the
    first half rounded down to a multiple of 8). tests/test_emu_training.py checks it against numpy itself."""
    n = len(terms)
    if n < 8:
        res = torch.zeros_like(terms[0]) if n and torch.is_tensor(terms[0]) else 0.0
        for t in terms:
            res = res + t
        return res
    if n <= 128:
        r = list(terms[:8])
        i = 8
        while i < n - (n % 8):
            for k in range(8):
                r[k] = r[k] + terms[i + k]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for t in terms[i:]:
            res = res + t
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return numpy_sum_order(terms[:n2]) + numpy_sum_order(terms[n2:])


def sequence_baselines(ro: Rollouts, values: torch.Tensor, num_sequences: int, num_rollouts: int, binding=None) -> torch.Tensor:
    """for each group of `num_rollouts` consecutive envs (the rollouts of one job sequence) the mean
    over the group's rollouts of their piecewise-linear value curves, evaluated at each rollout's own
    step times (baselines.py:12-37; times = `wall_times[:-1]`, trainer.py:206-207). f64[T,B]"""
    T, B = ro.active.shape
    assert B == num_sequences * num_rollouts
    G, R = num_sequences, num_rollouts
    b = _record_kernels(ro, binding)
    if b is not None and T > 0 and B > 0:  # one kernel, a thread per (step, env) query (below: ~20 operations on [G, R, R, T] tensors)
        import ctypes

        from .binding import SssBaselineArgs, device_of
        dev = ro.active.device
        act, ts, ys = ro.active.contiguous().view(torch.uint8), _c64(ro.t_before), _c64(values)
        n = ro.active.sum(0).to(torch.int64).contiguous()
        out = torch.empty((T, B), dtype=torch.float64, device=dev)
        a = SssBaselineArgs(T, B, R, int(not bool((n > 0).all())), act.data_ptr(), ts.data_ptr(), ys.data_ptr(), n.data_ptr(), out.data_ptr())
        with device_of(dev):
            b.check(b.lib.sss_sequence_baselines(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))
        return out
    ts = ro.t_before.t().reshape(G, R, T)
    ys = values.t().reshape(G, R, T)
    n = ro.active.sum(0).reshape(G, R)
    x = ts[:, :, None, :].expand(G, R, R, T)          # query: rollout i's times ...
    xp = ts[:, None, :, :].expand(G, R, R, T)         # ... on rollout j's curve
    fp = ys[:, None, :, :].expand(G, R, R, T)
    nn = n[:, None, :].expand(G, R, R)
    y_hat = _interp(x.contiguous(), xp.contiguous(), fp.contiguous(), nn.contiguous())
    if bool((n > 0).all()):  # (the mean over the sequence's rollouts in numpy's summation order)
        return (numpy_sum_order([y_hat[:, :, j, :] for j in range(R)]) / R).reshape(B, T).t() * ro.active
    # some rollout recorded nothing (its env failed on its first step, on_env_error="truncate"):
    # it has no curve and is left out of its group's mean
    has = (n > 0).to(torch.float64)
    acc = numpy_sum_order([y_hat[:, :, j, :] * has[:, None, j, None] for j in range(R)])
    return (acc / has.sum(1).clamp(min=1)[:, None, None]).reshape(B, T).t() * ro.active


# ---- PPO (trainers/ppo.py) ----------------------------------------------------------------------------

def ppo_loss(policy, g: dict[str, Any], stage_sel, job_idx, exec_sel, advantages: torch.Tensor, old_lgprobs: torch.Tensor,
             clip_range: float, entropy_coeff: float, adv_stats: tuple[float, float] | None = None, weight: float = 1.0):
    """the CLIP loss of ppo.py:104-138 on one minibatch (a compact graph + per-observation vectors).
    Several ranks, each holding a part of the minibatch: `adv_stats` = (mean, std) of the WHOLE minibatch's advantages (ppo.py:113-116
    normalises over the minibatch - the reference has one learner and every sample in it) and `weight` = this rank's share of the
    minibatch x number of ranks, so that the ranks' AVERAGED gradients are the gradient of the whole minibatch's mean loss."""
    res = policy.evaluate_actions(g, stage_sel, job_idx, exec_sel)
    advgs = advantages.float()
    mean, std = (advgs.mean(), advgs.std()) if adv_stats is None else adv_stats
    advgs = (advgs - mean) / (std + EPS)
    log_ratio = res["lgprobs"] - old_lgprobs
    ratio = log_ratio.exp()
    policy_loss = -torch.min(advgs * ratio, advgs * torch.clamp(ratio, 1 - clip_range, 1 + clip_range)).mean()
    entropy_loss = -res["entropies"].mean()
    loss = policy_loss + entropy_coeff * entropy_loss
    if weight != 1.0:
        loss = loss * weight
    with torch.no_grad():
        approx_kl = ((ratio - 1) - log_ratio).mean()
    return loss, {"policy_loss": policy_loss.detach(), "entropy_loss": entropy_loss.detach(), "approx_kl_div": approx_kl}


class PPO:
    """the reference's PPO trainer (trainers/ppo.py + the parts of trainers/trainer.py it uses) on
    the batched env. With `torch.distributed` initialised every rank trains on its own env shard and
    gradients are averaged across ranks before each optimiser step; the KL early-stop test uses the
    all-rank mean so that every rank takes the same number of steps."""

    def __init__(self, policy, train_cfg: dict[str, Any], generator: torch.Generator | None = None):
        self.policy = policy
        self.entropy_coeff = train_cfg.get("entropy_coeff", 0.0)
        self.clip_range = train_cfg.get("clip_range", 0.2)
        self.target_kl = train_cfg.get("target_kl", 0.01)
        self.num_epochs = train_cfg.get("num_epochs", 10)
        self.num_batches = train_cfg.get("num_batches", 3)
        self.num_sequences = int(train_cfg["num_sequences"])
        self.num_rollouts = int(train_cfg["num_rollouts"])
        assert ("reward_buff_cap" in train_cfg) ^ ("beta_discount" in train_cfg), \
            "must provide exactly one of `reward_buff_cap` and `beta_discount` in config"  # trainer.py:63-65
        self.beta = train_cfg.get("beta_discount")
        self.diff = DifferentialReturns(train_cfg["reward_buff_cap"]) if "reward_buff_cap" in train_cfg else None
        self.generator = generator

    def preprocess(self, ro: Rollouts):
        returns = self.diff(ro) if self.diff is not None else discounted_returns(ro, self.beta)
        baselines = sequence_baselines(ro, returns, self.num_sequences, self.num_rollouts)
        return returns, baselines

    def _dist(self):
        import torch.distributed as dist
        return dist if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 else None

    def train_on_rollouts(self, ro: Rollouts) -> dict[str, float]:
        returns, baselines = self.preprocess(ro)
        ids = ro.sample_ids()
        advgs = ro.flat(returns - baselines)
        acts = [ro.flat(ro.stage_sel), ro.flat(ro.job_idx), ro.flat(ro.exec_sel)]
        old_lg = ro.flat(ro.lgprobs)
        n = ids.numel()
        bs = n // self.num_batches + 1
        dist = self._dist()
        pol, ent, kls = [], [], []
        go = True
        for _ in range(self.num_epochs):
            if not go:
                break
            perm = torch.randperm(n, generator=self.generator, device=ids.device if self.generator is None or self.generator.device.type != "cpu" else "cpu").to(ids.device)
            # single process: the reference's DataLoader batches (size n // num_batches + 1, ppo.py:66-71).
            # several ranks: exactly num_batches chunks everywhere, so that the collectives line up
            chunks = torch.tensor_split(perm, self.num_batches) if dist else [perm[s: s + bs] for s in range(0, n, bs)]
            whole = None
            if dist:
                # A minibatch is the ranks' chunks TOGETHER, as if one learner held every sample (the reference's does: the workers'
                # rollouts are gathered into one process, trainer.py:113-121, ppo.py:51-138): its advantages are normalised with the
                # mean / std over all ranks' chunks and every rank's loss counts by its share of the samples. One all-reduce per
                # epoch carries the sums of all its minibatches: (count, sum, sum of squares) each, f64.
                a64 = advgs.double()
                whole = torch.stack([torch.stack([torch.tensor(float(mb.numel()), dtype=torch.float64, device=ids.device), a64[mb].sum(), (a64[mb] ** 2).sum()])
                                     for mb in chunks])
                dist.all_reduce(whole)
                whole = whole.cpu()
            for ci, mb in enumerate(chunks):
                stats, weight = None, 1.0
                if dist:
                    N, S1, S2 = (float(v) for v in whole[ci])
                    usable = mb.numel() >= 1 and N >= 2  # (a single sample in the whole minibatch has no advantage std)
                    if usable:
                        mean = S1 / N
                        stats = (mean, max(0.0, (S2 - N * mean * mean) / (N - 1)) ** 0.5)  # (torch.std: the unbiased estimate)
                        weight = mb.numel() * dist.get_world_size() / N
                else:
                    usable = mb.numel() >= 2  # a single sample has no advantage std (the reference would fail on it)
                if usable:
                    g = select_observations(ro.graph, ids[mb])
                    loss, info = ppo_loss(self.policy, g, acts[0][mb], acts[1][mb], acts[2][mb], advgs[mb], old_lg[mb],
                                          self.clip_range, self.entropy_coeff, stats, weight)
                    kl = info["approx_kl_div"]
                else:
                    loss, info, kl = None, None, torch.zeros((), device=ids.device)
                if dist:  # the early-stop test looks at the whole minibatch's mean KL (ppo.py:89-93)
                    stat = torch.stack([kl.float() * (mb.numel() if usable else 0), torch.ones_like(kl.float()) * (mb.numel() if usable else 0)])
                    dist.all_reduce(stat)
                    kl = stat[0] / stat[1].clamp(min=1)
                elif not usable:
                    continue
                kl = float(kl)
                if usable:
                    pol.append(float(info["policy_loss"]))
                    ent.append(float(info["entropy_loss"]))
                kls.append(kl)
                if self.target_kl is not None and kl > 1.5 * self.target_kl:
                    go = False
                    break
                if usable:
                    loss.backward()
                if dist:
                    # ONE all-reduce per optimiser step: the whole gradient (20.8 k floats = 83 KB for the
                    # published architecture) as a single bucket - latency-bound on xGMI, so fewer, larger
                    # messages is all there is to tune
                    params = list(self.policy.parameters())
                    for p in params:
                        if p.grad is None:
                            p.grad = torch.zeros_like(p)
                    flat = torch.cat([p.grad.reshape(-1) for p in params])
                    dist.all_reduce(flat)
                    flat /= dist.get_world_size()
                    off = 0
                    for p in params:
                        k = p.numel()  # not `n`: that is the number of samples the epochs permute
                        p.grad.copy_(flat[off: off + k].view_as(p))
                        off += k
                self.policy.update_parameters(None)
        # (`minibatches`: optimiser steps taken - fewer than epochs x batches when the KL test stopped the epochs, ppo.py:89-93)
        return {"policy loss": abs(float(np.mean(pol))), "entropy": abs(float(np.mean(ent))), "approx kl div": abs(float(np.mean(kls))), "minibatches": len(pol)}


class Trainer:
    """the reference's training loop (trainers/trainer.py:27-168) with the rollout workers replaced
    by the batched env: per iteration collect `num_sequences x num_rollouts` rollouts (per rank),
    update the policy, track / checkpoint the best parameters.

    With torch.distributed initialised (one process per GPU) every rank owns `num_sequences` job
    sequences of its own (base seeds are a function of the global sequence id, so the union over
    ranks does not depend on the placement), rollouts of one sequence stay on one rank (their
    baseline needs no traffic), gradients are averaged across ranks and ONE all-gather per
    iteration moves the per-rollout statistics (trainer.py:113-121's Pipe gather)."""

    def __init__(self, agent_cfg: dict[str, Any], env_cfg: dict[str, Any], train_cfg: dict[str, Any],
                 device: str | torch.device | None = None, _lib=None, pack: bytes | None = None):
        """`pack`: the workload pack the envs run on (None: the frozen default trace set)"""
        import torch.distributed as dist

        from .decima import DecimaPolicy
        from .vec_env import VecSparkSchedSimEnv

        self.rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.seed = int(train_cfg["seed"])
        torch.manual_seed(self.seed)
        self.num_iterations = int(train_cfg["num_iterations"])
        self.num_sequences = int(train_cfg["num_sequences"])
        self.num_rollouts = int(train_cfg["num_rollouts"])
        self.rollout_duration = train_cfg.get("rollout_duration")
        self.checkpointing_freq = int(train_cfg.get("checkpointing_freq", 50))
        self.artifacts_dir = train_cfg.get("artifacts_dir", "artifacts")
        self.env_cfg = dict(env_cfg)
        if "beta_discount" in train_cfg:
            self.env_cfg["beta"] = train_cfg["beta_discount"]  # trainer.py:70-72
        assert agent_cfg.get("agent_cls", "DecimaScheduler") == "DecimaScheduler"
        dev = torch.device(device if device is not None else train_cfg.get("device", "cuda:0"))
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", 0)
        self.device = dev
        if dev.type == "cuda" and train_cfg.get("allocator_rounding", True):
            # The sizes of a minibatch's tensors (GBs at BASELINE config 5) differ by a percent from one minibatch to the next; the
            # caching allocator then keeps asking the driver for slightly larger blocks (0.1 s per hipMalloc, 0.9 s of a 3.4 s
            # update: profiles/r04_ppo.md). Rounding request sizes up to eighths of a power of two makes the blocks reusable.
            try:
                (getattr(torch._C, "_accelerator_setAllocatorSettings", None) or torch.cuda.memory._set_allocator_settings)("roundup_power2_divisions:8")
            except Exception:  # (an allocator backend without the option)
                pass
        E = int(self.env_cfg["num_executors"])
        kw = {k: v for k, v in agent_cfg.items() if k != "agent_cls"}
        self.policy = DecimaPolicy(num_executors=E, opt_cls=train_cfg["opt_cls"], opt_kwargs=train_cfg.get("opt_kwargs"),
                                   max_grad_norm=train_cfg.get("max_grad_norm"), **kw).to(dev)
        B = self.num_sequences * self.num_rollouts
        sim_cfg = {k: v for k, v in self.env_cfg.items() if k not in ("mean_time_limit", "dataset")}
        self.env = VecSparkSchedSimEnv(sim_cfg, B, device=dev, _lib=_lib, pack=pack)
        total_sequences = self.num_sequences * self.world
        seq_ids = self.rank * self.num_sequences + np.arange(self.num_sequences)
        base_seeds = np.repeat(self.seed + seq_ids, self.num_rollouts)  # trainer.py:264-266
        gen = torch.Generator(device=dev if dev.type == "cuda" else "cpu")
        gen.manual_seed(self.seed * 1000003 + self.rank)
        self.collector = RolloutCollector(self.env, self.env_cfg["mean_time_limit"], base_seeds, total_sequences, E,
                                          policy=self.policy, generator=gen, on_env_error=train_cfg.get("on_env_error", "raise"),
                                          # (`collector_groups` > 1: alternating groups of envs on their own streams - measured
                                          # slower on one MI355X, profiles/r03_ppo.md; one group is the default)
                                          groups=int(train_cfg.get("collector_groups", 1)))
        self.ppo = PPO(self.policy, train_cfg, generator=gen)
        self.history: list[dict[str, float]] = []

    def _gather_stats(self, stats: dict[str, np.ndarray]) -> dict[str, np.ndarray]:
        import torch.distributed as dist
        keys = sorted(stats)
        local = torch.from_numpy(np.stack([stats[k] for k in keys], 1)).to(self.device)
        if self.world > 1:
            out = [torch.empty_like(local) for _ in range(self.world)]
            dist.all_gather(out, local)
            local = torch.cat(out, 0)
        arr = local.cpu().numpy()
        return {k: arr[:, i] for i, k in enumerate(keys)}

    def train(self, verbose: bool = True) -> list[dict[str, float]]:
        import json
        import os
        import os.path as osp
        from copy import deepcopy

        ckpt_dir = osp.join(self.artifacts_dir, "checkpoints")
        if self.rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
        best = None
        for i in range(self.num_iterations):
            state_dict = deepcopy(self.policy.state_dict())
            self.policy.eval()
            ro = (self.collector.collect_async(self.rollout_duration) if self.rollout_duration else self.collector.collect_sync())
            self.policy.train()
            learn = self.ppo.train_on_rollouts(ro)
            stats = self._gather_stats(ro.stats)
            with np.errstate(all="ignore"):
                avg_num_jobs = (self.ppo.diff.avg_num_jobs if self.ppo.diff is not None else None) or float(np.mean(stats["avg_num_jobs"]))
            if not best or avg_num_jobs < best["avg_num_jobs"]:  # trainer.py:138-142
                best = {"iteration": i, "avg_num_jobs": float(np.round(avg_num_jobs, 3)), "state_dict": state_dict,
                        "completed_job_count": int(np.mean(stats["num_completed_jobs"]))}
            if (i + 1) % self.checkpointing_freq == 0 and self.rank == 0:
                d = osp.join(ckpt_dir, f"{i + 1}")
                os.makedirs(d, exist_ok=True)
                torch.save(best.pop("state_dict"), osp.join(d, "model.pt"))
                with open(osp.join(d, "state.json"), "w") as fp:
                    json.dump(best, fp)
                best = None
            elif (i + 1) % self.checkpointing_freq == 0:
                best = None
            rec = dict(learn, iteration=i, avg_num_jobs=float(avg_num_jobs), samples=int(ro.active.sum()), env_errors=self.collector.env_errors,
                       episode_length=float(ro.lengths.float().mean()))
            self.history.append(rec)
            if verbose and self.rank == 0:
                print(f"Iteration {i + 1} complete. Avg. # jobs: {avg_num_jobs:.3f}", flush=True)
        return self.history

    def close(self) -> None:
        self.env.close()


def make_trainer(cfg: dict[str, Any], device: str | torch.device | None = None, _lib=None) -> Trainer:
    """by-name factory like the reference's (trainers/__init__.py:7-13): `cfg` is the parsed YAML
    with its `trainer` / `agent` / `env` sections"""
    trainer_cls = cfg["trainer"]["trainer_cls"]
    assert trainer_cls == "PPO", f"'{trainer_cls}' is not a valid trainer."  # PPO is the trainer of BASELINE config 5
    return Trainer(agent_cfg=cfg["agent"], env_cfg=cfg["env"], train_cfg=cfg["trainer"], device=device, _lib=_lib)
