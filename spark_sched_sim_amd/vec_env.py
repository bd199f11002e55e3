"""Batched Spark-scheduling environment: the host-side mirror of the reference's
`SparkSchedSimEnv` (reference spark_sched_sim/spark_sched_sim.py:29-245) over `num_envs`
independent simulations that live in HBM and are stepped by the HIP kernels behind include/sss.h.

PyTorch only provides device memory and streams here: every buffer is a torch tensor whose raw
pointer is handed to the C ABI, so observations come back as device tensors without a copy.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Sequence

import numpy as np
import torch

from . import workload
from .binding import ERROR_NAMES, POLICY_IDS, Binding, SssBuffers, SssCfg, SssDecimaGraph, SssDecimaLists, device_of

OBS_FIELDS = ("n_nodes", "n_edges", "n_jobs", "n_schedulable", "num_committable_execs", "source_job_idx",
              "terminated", "err")
NUM_NODE_FEATURES = 3  # reference spark_sched_sim.py:25


class BatchedObs(dict):
    """dict of device tensors; rows beyond n_nodes[i] / n_edges[i] / n_jobs[i] of env i are padding.

    keys: nodes f32[B, node_cap, 3], edge_links i32[B, edge_cap, 2], dag_ptr i32[B, job_cap+1],
    exec_supplies i32[B, job_cap], plus one i32[B] tensor per OBS_FIELDS entry.
    """


def _mask_u8(mask: torch.Tensor | None) -> torch.Tensor | None:
    """an env mask as the u8 array the kernels read: a view when it already is one byte per env (bool / uint8, contiguous)"""
    if mask is None:
        return None
    if mask.is_contiguous() and mask.dtype in (torch.bool, torch.uint8):
        return mask.view(torch.uint8)
    return mask.to(torch.uint8).contiguous()


class LateHint:
    """A few device integers read back WITHOUT waiting for them: `post(t)` enqueues a copy of `t` to pinned host memory on the current
    stream, `read()` returns the most recent copy that has COMPLETED (all `fill` until one has) - two host buffers taken in turn,
    each with an event recorded behind its copy, so that a buffer is never read while a copy into it may be in flight (a plain
    pinned tensor read while the previous call's non_blocking copy was still running gave torn / half-old values). The values only
    size grids and work buffers: late ones are fine, undefined ones are not. CPU tensors: the copy is synchronous."""

    def __init__(self, n: int, device: torch.device, fill: int = -1):
        self.cuda = device.type == "cuda"
        self.fill = torch.full((n,), fill, dtype=torch.int64)
        bufs = [torch.full((n,), fill, dtype=torch.int64) for _ in range(2)]
        self.bufs = [b.pin_memory() for b in bufs] if self.cuda else bufs
        self.events = [torch.cuda.Event() for _ in range(2)] if self.cuda else None
        self.posted = [False, False]
        self.tags = [None, None]
        self.last = 1

    def read(self) -> torch.Tensor:
        return self.read_tagged()[0]

    def read_tagged(self):
        """(values, the `tag` their `post` was given); (fill, None) until a copy has completed. With the host several steps ahead of
        the device BOTH buffers can be waiting for their copies: the values of the last completed one are kept and returned again
        (returning the fill value then sized every grid by its capacity for as long as the host stayed ahead)"""
        for k in (self.last, 1 - self.last):
            if self.posted[k] and (not self.cuda or self.events[k].query()):
                self.kept = (self.bufs[k].clone(), self.tags[k])
                break
        kept = getattr(self, "kept", None)
        return (kept[0].clone(), kept[1]) if kept is not None else (self.fill.clone(), None)

    def post(self, t: torch.Tensor, tag=None) -> None:
        k = 1 - self.last
        if self.cuda and self.posted[k] and not self.events[k].query():
            return  # (the copy before last has not even finished: the host is far ahead - skip this one rather than wait)
        self.bufs[k].copy_(t, non_blocking=True)
        if self.cuda:
            self.events[k].record(torch.cuda.current_stream(t.device))
        self.posted[k], self.last, self.tags[k] = True, k, tag


class VecSparkSchedSimEnv:
    """`num_envs` reference environments in one object.

    env_cfg keys are the reference's (`num_executors`, `job_arrival_cap`, `job_arrival_rate`,
    `moving_delay`, `warmup_delay`, optional `beta`; `data_sampler_cls` must be "TPCHDataSampler"
    if present) plus optional `max_jobs` (arena capacity when `job_arrival_cap` is None).
    """

    def __init__(self, env_cfg: dict[str, Any], num_envs: int, device: str | torch.device = "cuda:0",
                 pack: bytes | None = None, auto_reset: bool = False, seed_stride: int | None = None,
                 _lib: C.CDLL | None = None) -> None:
        self.device = torch.device(device)
        if self.device.type != "cuda" and _lib is None:
            raise RuntimeError("VecSparkSchedSimEnv runs on an AMD GPU (device='cuda:N'); there is no CPU path")
        sampler = env_cfg.get("data_sampler_cls", "TPCHDataSampler")
        if sampler != "TPCHDataSampler":
            raise ValueError(f"'{sampler}' is not a valid data sampler.")  # data_samplers/__init__.py:12-14
        self.num_envs = int(num_envs)
        self.num_executors = int(env_cfg["num_executors"])
        self.env_cfg = dict(env_cfg)
        self.auto_reset = bool(auto_reset)
        self.seed_stride = int(seed_stride if seed_stride is not None else num_envs)
        self._b = Binding(_lib)
        self._pack = pack if pack is not None else workload.default_pack()
        cap = env_cfg.get("job_arrival_cap")
        self._cfg = SssCfg(self.num_executors, int(cap) if cap else 0, int(env_cfg.get("max_jobs") or 0), 0,
                           float(env_cfg["job_arrival_rate"]), float(env_cfg["moving_delay"]),
                           float(env_cfg["warmup_delay"]), float(env_cfg.get("beta", 0.0)))
        self.dims = self._b.query_dims(self._cfg, self._pack, self.num_envs)
        dev_index = self.device.index or 0
        self._h = self._b.create(self._cfg, self._pack, self.num_envs, dev_index if self.device.type == "cuda" else -1)
        d, B, dev = self.dims, self.num_envs, self.device
        # the arena must start zeroed (lifetime counters) and 256-byte aligned
        self._state_raw = torch.zeros(d.state_bytes + 256, dtype=torch.uint8, device=dev)
        off = (-self._state_raw.data_ptr()) % 256
        self.state = self._state_raw[off: off + d.state_bytes]
        self.nodes = torch.zeros((B, d.node_cap, NUM_NODE_FEATURES), dtype=torch.float32, device=dev)
        self.edge_links = torch.zeros((B, d.edge_cap, 2), dtype=torch.int32, device=dev)
        self.dag_ptr = torch.zeros((B, d.job_cap + 1), dtype=torch.int32, device=dev)
        self.exec_supplies = torch.zeros((B, d.job_cap), dtype=torch.int32, device=dev)
        self.obs_i32 = torch.zeros((B, d.obs_i32), dtype=torch.int32, device=dev)
        self.obs_f64 = torch.zeros((B, d.obs_f64), dtype=torch.float64, device=dev)
        self.rebind_buffers()
        self._seeds = torch.zeros(B, dtype=torch.int64, device=dev)
        self._tl = torch.full((B,), float("inf"), dtype=torch.float64, device=dev)
        self._mask = torch.ones(B, dtype=torch.uint8, device=dev)
        self._env_view = self.state.view(B, d.env_stride)
        self._act_stage = torch.zeros(B, dtype=torch.int32, device=dev)
        self._act_nexec = torch.ones(B, dtype=torch.int32, device=dev)
        self._closed = False
        self._dg_pool: dict[int, torch.Tensor] = {}
        self._layer_scratch: dict = {}
        self._never = None
        from .workload import pack_max_depth
        self.max_dag_depth = pack_max_depth(self._pack)  # bound of an observation's DAG layers: sizes sss_gnn_encode's layer launches
        # gymnasium.vector.VectorEnv attributes; the per-env action space is the reference's at
        # construction (its stage_idx bound follows each observation: valid range is [-1, n_nodes[i]))
        from .spaces import make_action_space, make_observation_space
        self.single_action_space = make_action_space(self.num_executors)
        self.action_space = self.single_action_space
        # per-env observation space of the reference (its two episode-dependent bounds are those of the env at
        # construction here; the per-env views of the facade keep them current, env.py)
        self.single_observation_space = make_observation_space(self.num_executors, NUM_NODE_FEATURES)
        self.observation_space = self.single_observation_space

    # ---- plumbing ---------------------------------------------------------------------

    def rebind_buffers(self) -> None:
        """(re)registers the observation / state tensors with the library (include/sss.h sss_bind_buffers). Every
        call starts a new buffer generation: rows the env skips while nothing changed (the edge rows of an
        unchanged active subgraph) are written again at the next step. The supported way to hand over new
        buffers - or buffers whose contents were changed behind the env's back: assign the tensors, then call this."""
        bufs = SssBuffers(self.state.data_ptr(), self.nodes.data_ptr(), self.edge_links.data_ptr(),
                          self.dag_ptr.data_ptr(), self.exec_supplies.data_ptr(), self.obs_i32.data_ptr(),
                          self.obs_f64.data_ptr())
        self._b.check(self._b.lib.sss_bind_buffers(self._h, C.byref(bufs)))

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def _obs(self) -> BatchedObs:
        o = BatchedObs(nodes=self.nodes, edge_links=self.edge_links, dag_ptr=self.dag_ptr,
                       exec_supplies=self.exec_supplies)
        for k, name in enumerate(OBS_FIELDS):
            o[name] = self.obs_i32[:, k]
        return o

    def close(self) -> None:
        if not self._closed:
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)
            self._b.lib.sss_destroy(self._h)
            self._closed = True

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- Gymnasium-style API ----------------------------------------------------------

    def reset(self, *, seed: int | Sequence[int] | None = None, options: dict[str, Any] | None = None,
              mask: torch.Tensor | None = None):
        """reference `reset(seed, options)` (spark_sched_sim.py:127-186) for every env (or the
        envs selected by `mask`). `seed`: int -> env i gets seed + i; sequence -> per env;
        None -> fresh entropy, as `gymnasium.Env.reset(seed=None)` does."""
        B = self.num_envs
        if seed is None:
            seeds = np.random.SeedSequence().generate_state(B, np.uint64) >> np.uint64(1)
        elif isinstance(seed, (int, np.integer)):
            seeds = np.arange(B, dtype=np.uint64) + np.uint64(int(seed))
        else:
            seeds = np.asarray(list(seed), dtype=np.uint64)
            if seeds.shape != (B,):
                raise ValueError("need one seed per env")
        self._seeds.copy_(torch.from_numpy(seeds.view(np.int64)))
        tl = (options or {}).get("time_limit", np.inf)
        tl_arr = np.broadcast_to(np.asarray(tl, dtype=np.float64), (B,)).copy()
        self._tl.copy_(torch.from_numpy(tl_arr))
        mask_ptr = None
        if mask is not None:
            self._mask.copy_(mask.to(torch.uint8))
            mask_ptr = self._mask.data_ptr()
        self._b.check(self._b.lib.sss_reset(self._h, self._seeds.data_ptr(), self._tl.data_ptr(), mask_ptr, self._stream()))
        return self._obs(), {"wall_time": self.obs_f64[:, 1]}

    def step_async(self, stage_idx: torch.Tensor, num_exec: torch.Tensor) -> None:
        """launches one `step(action)` per env on the current stream and returns immediately"""
        assert stage_idx.dtype == torch.int32 and num_exec.dtype == torch.int32
        assert stage_idx.device == self.device and stage_idx.is_contiguous() and num_exec.is_contiguous()
        self._b.check(self._b.lib.sss_step(self._h, stage_idx.data_ptr(), num_exec.data_ptr(), int(self.auto_reset),
                                           self.seed_stride, self._stream()))

    def step_bounded_async(self, stage_idx: torch.Tensor, num_exec: torch.Tensor, max_events: int, ready: torch.Tensor | None = None) -> torch.Tensor:
        """`step_async` with an event budget per launch (include/sss.h sss_step_bounded): an env whose step needs more than
        `max_events` events is cut between two events and goes on in the next call (its action entries are then ignored).
        Returns `ready` u8[B] (the env's buffer unless one is given): 1 = the env's step is complete and its outputs are
        current, 0 = it continues - its observation / reward / flags are still the previous ones. Envs run at their own
        pace, as the reference's envs do in their worker processes; each env's trajectory is the one `step` gives."""
        assert stage_idx.dtype == torch.int32 and num_exec.dtype == torch.int32
        assert stage_idx.device == self.device and stage_idx.is_contiguous() and num_exec.is_contiguous()
        if ready is None:
            if getattr(self, "_ready", None) is None:
                self._ready = torch.ones(self.num_envs, dtype=torch.uint8, device=self.device)
            ready = self._ready
        assert ready.dtype == torch.uint8 and ready.is_contiguous() and ready.numel() == self.num_envs
        self._b.check(self._b.lib.sss_step_bounded(self._h, stage_idx.data_ptr(), num_exec.data_ptr(), int(self.auto_reset), self.seed_stride, int(max_events),
                                                   ready.data_ptr(), self._stream()))
        return ready

    def step(self, actions):
        """reference `step(action)` (spark_sched_sim.py:188-221), batched.
        `actions`: {"stage_idx": i32[B], "num_exec": i32[B]} (tensors on the env's device).
        Returns (obs, reward f64[B], terminated bool[B], truncated bool[B], info).

        Aliasing contract: `obs` holds VIEWS of the env's observation buffers - the next step / reset
        overwrites them in place (zero-copy hand-off to a policy on the device). The small per-env
        vectors a loop typically keeps across steps - reward, terminated, info["wall_time"],
        info["err"] - are copies. `step_async` + the buffers themselves is the copy-free path."""
        if set(actions.keys()) != {"stage_idx", "num_exec"}:
            raise ValueError("invalid action: does not belong to the action space")  # Dict.contains
        self.step_async(actions["stage_idx"], actions["num_exec"])
        obs = self._obs()
        # the small per-env vectors are COPIES (two copy launches + one compare): reward / wall_time, terminated / err
        f64 = self.obs_f64.clone()
        i32 = self.obs_i32[:, 6:8].clone()
        terminated = i32[:, 0] != 0
        if self._never is None or self._never.shape != terminated.shape:
            self._never = torch.zeros_like(terminated)  # (truncation is the time-limit wrapper's business: always False here; shared, read-only)
        return obs, f64[:, 0], terminated, self._never, {"wall_time": f64[:, 1], "err": i32[:, 1]}

    # ---- on-device policies and fused rollouts ---------------------------------------------

    def policy_actions(self, policy: str = "fair", param: int = 0):
        """one action per env from an on-device heuristic ("fair" / "fifo" = the reference's
        RoundRobinScheduler with / without dynamic partitioning, "hash" = the build's counter-based
        uniform-random policy); returns {"stage_idx", "num_exec"} device tensors (re-used buffers)."""
        self._b.check(self._b.lib.sss_policy(self._h, POLICY_IDS[policy], int(param), self._act_stage.data_ptr(),
                                             self._act_nexec.data_ptr(), self._stream()))
        return {"stage_idx": self._act_stage, "num_exec": self._act_nexec}

    def rollout(self, policy: str, n_steps: int, param: int = 0) -> None:
        """n_steps x (policy -> step) per env inside one kernel launch (asynchronous)"""
        self._b.check(self._b.lib.sss_rollout(self._h, POLICY_IDS[policy], int(param), int(n_steps), int(self.auto_reset),
                                              self.seed_stride, self._stream()))

    def decima_graph_on_device(self, active: torch.Tensor | None = None, num_tasks_scale: float = 200.0, work_scale: float = 1e5) -> dict[str, Any]:
        """`decima_graph` without the device->host round trip: the graph's totals (nodes, edges, jobs, schedulable nodes) stay on
        the device (`g["totals_dev"]`, i64[4]); every array is a buffer of the env at its CAPACITY (num_envs x node_cap nodes, ...),
        only its leading `totals_dev[k]` rows are meaningful, and the kernels downstream read the counts themselves
        (sss_gnn_args::n_rows_dev, sss_gnn_encode_args::n_nodes_dev). `g["totals_hint"]`: the previous call's totals, copied back
        without waiting (pinned memory) - grid sizes only. For act-and-forget inference (`DecimaPolicy.schedule_env`): the buffers are
        overwritten by the next call; graphs that are kept (rollout recording, training) use `decima_graph`."""
        B, dev, d = self.num_envs, self.device, self.dims
        assert self.graph_kernel_fits, "the graph kernel's LDS working set does not fit this node capacity"
        D = self.max_dag_depth
        ws = getattr(self, "_dg_dev", None)
        if ws is None:
            Mc, Ec, Jc = B * d.node_cap, B * d.edge_cap, B * d.job_cap
            e = lambda n, dt, *tail: torch.empty((max(n, 1), *tail), dtype=dt, device=dev)  # noqa: E731
            ws = self._dg_dev = {
                "x": e(Mc, torch.float32, 5), "node_obs": e(Mc, torch.int64), "node_loc": e(Mc, torch.int64), "node_job": e(Mc, torch.int64),
                "sched_rank": e(Mc, torch.int64), "gen": e(Mc, torch.int32), "node_recv": e(Mc, torch.int32), "stage_mask": e(Mc, torch.bool),
                "src": e(Ec, torch.int64), "dst": e(Ec, torch.int64), "edge_obs": e(Ec, torch.int64), "edge_layers": e(Ec, torch.int32),
                "job_obs": e(Jc, torch.int64), "job_cap": e(Jc, torch.int64), "job_first": e(Jc, torch.int64), "obs_depth": e(B, torch.int32),
                "job_nodes": e(Jc, torch.int64), "out_start": e(Mc, torch.int64), "out_deg": e(Mc, torch.int32),
                "layer_cnt": e(32, torch.int32, B), "sched_list": e(Mc, torch.int64),
                "scan": e(2, torch.int64, 4, B),
                # two sets of list counters (i64[33][32] each: a counter per layer and block of envs + the blocks' largest observation) used in turn - a launch reserves on
                # one and clears the other - with the graph's four totals between them: [set 0 | totals | set 1], so that the totals and
                # the current set are one contiguous piece for the read-back of the grid-size hints
                "pp": torch.zeros(1056 + 4 + 1056, dtype=torch.int64, device=dev),
                "recv": e(Mc * max(D, 1), torch.int64), "hint": LateHint(1060, dev), "epoch": 0}
            ws["tot"] = ws["pp"][1056:1060]
        act8 = _mask_u8(active)
        scan, tot = ws["scan"], ws["tot"]
        stream = self._stream()
        mask_ptr = act8.data_ptr() if act8 is not None else None
        kept = ws.get("args")
        if kept is None or kept[0] != (num_tasks_scale, work_scale, self.obs_i32.data_ptr()):
            # the buffers are the env's own, the same at every call: the argument structure and the dict of views are made once
            # (building them cost more host time per step than the two launches), the mask pointer is written per call
            off, cnt_t = scan[0], scan[1]
            stride = ws["recv"].numel() // max(D, 1)
            a = SssDecimaGraph(mask_ptr, off[0].data_ptr(), off[2].data_ptr(), off[1].data_ptr(),
                               float(num_tasks_scale), float(work_scale), *(ws[k].data_ptr() for k in (
                                   "x", "node_obs", "node_loc", "node_job", "sched_rank", "gen", "node_recv", "stage_mask", "src", "dst",
                                   "edge_obs", "edge_layers", "job_obs", "job_cap", "job_first", "obs_depth", "job_nodes", "out_start", "out_deg", "layer_cnt")),
                               off[3].data_ptr(), ws["sched_list"].data_ptr(), None, ws["recv"].data_ptr(), stride, None, 33 * 32)
            g0 = {k: ws[k] for k in ("x", "node_obs", "node_loc", "node_job", "sched_rank", "gen", "node_recv", "stage_mask", "src", "dst", "edge_obs",
                                     "edge_layers", "job_obs", "job_cap", "job_first", "obs_depth", "job_nodes", "out_start", "out_deg", "layer_cnt", "sched_list")}
            g0["n_obs"], g0["n_pad"], g0["max_depth"] = B, d.node_cap, D
            g0["obs_nodes"], g0["obs_jobs"], g0["obs_node_off"], g0["obs_job_off"] = cnt_t[0], cnt_t[2], off[0], off[2]
            g0["totals_dev"], g0["_binding"] = tot, self._b
            ptrs = (self.obs_i32.data_ptr(), self.obs_i32.stride(0), scan[0].data_ptr(), scan[1].data_ptr(), tot.data_ptr())
            kept = ws["args"] = ((num_tasks_scale, work_scale, self.obs_i32.data_ptr()), a, g0, stride, off, ptrs)
        _, a, g0, stride, off, ptrs = kept
        a.active_dev = mask_ptr
        with device_of(dev):
            self._b.check(self._b.lib.sss_prefix_rows(ptrs[0], 1, ptrs[1], mask_ptr, 4, B, ptrs[2], ptrs[3], ptrs[4], stream))
            ws["epoch"] += 1
            par, pp = ws["epoch"] & 1, ws["pp"]
            sets = (pp[0:1056], pp[1060:2116])
            a.layer_totals_dev, a.layer_totals_clear_dev = sets[par].data_ptr(), sets[1 - par].data_ptr()
            self._b.check(self._b.lib.sss_decima_graph_build(self._h, C.byref(a), stream))
            g = dict(g0)
            g["layer_totals"] = sets[par]
            g["_layer_lists"] = ({"recv": ws["recv"], "stride": stride, "epoch": ws["epoch"]}, ws["epoch"])
            # the latest earlier call's numbers that have arrived: totals (M, Ed, J, S) and the lists' piece lengths (-1: none yet)
            hv, tag = ws["hint"].read_tagged()
            g["totals_hint"], g["layer_hint"] = (hv[:4], hv[:1056]) if tag is None else (hv[1056:1060], hv[:1056]) if tag == 0 else (hv[:4], hv[4:1060])
            ws["hint"].post(pp[0:1060] if par == 0 else pp[1056:2116], tag=par)
        g["_keepalive"] = (off, act8)
        return g

    @property
    def graph_kernel_fits(self) -> bool:
        """whether include/sss.h sss_decima_graph_build takes this env's capacities (its per-env LDS working set: 8 bytes per node
        slot + 8 per job slot; 16-bit node / edge slots; job templates of at most 24 DAG layers = longest path in edges) - else the
        graph comes from tensor ops"""
        d = self.dims
        return 8 * d.node_cap + 8 * (d.job_cap + 1) <= 65536 and d.node_cap <= 65535 and d.edge_cap <= 65535 and self.max_dag_depth <= 24

    def decima_graph(self, active: torch.Tensor | None = None, num_tasks_scale: float = 200.0, work_scale: float = 1e5,
                     reuse_buffers: bool = False) -> dict[str, Any]:
        """the current observations of all envs (or of those with `active[b]` True) as Decima's
        compact graph - `decima.compact_graph(decima.decima_observation(obs))` produced by ONE kernel
        (include/sss.h sss_decima_graph_build) instead of ~150 tensor ops, plus per-edge / per-node
        DAG-layer bits and each schedulable node's `stage_idx`. One device->host sync (totals).
        `reuse_buffers`: write into buffers kept by the env (views into them are returned and are
        overwritten by the next such call) - for act-and-forget inference loops; graphs that are
        kept (rollout recording) must use the default."""
        B, dev = self.num_envs, self.device
        if not self.graph_kernel_fits:
            # the kernel's per-node LDS working set does not fit: same graph from tensor ops on the device
            from .decima import compact_graph, decima_observation
            if not getattr(self, "_warned_graph_fallback", False):
                import warnings
                self._warned_graph_fallback = True
                warnings.warn(f"Decima graph kernel not usable for this env (node capacity {self.dims.node_cap}, edge capacity {self.dims.edge_cap}, "
                              f"deepest job template {self.max_dag_depth} DAG layers; limits: 8 B x nodes + 8 B x jobs <= 64 KB of LDS, 65535 slots, 24 layers): "
                              "observations are transformed by tensor ops instead (same results, several times slower per step)", RuntimeWarning, stacklevel=2)
            f = decima_observation(self._obs(), self.num_executors, self.dims.stage_stride, int(num_tasks_scale), work_scale)
            if active is not None:
                for k in ("node_valid", "job_valid", "edge_valid", "stage_mask"):
                    f[k] = f[k] & active[:, None]
                f["n_nodes"], f["depth"] = f["n_nodes"] * active, f["depth"] * active
            return compact_graph(f)
        # per-env counts (n_nodes, n_edges, n_jobs; 0 for inactive envs), their exclusive prefix sums and totals:
        # one small kernel over the obs_i32 rows in place (include/sss.h sss_prefix_rows) instead of a handful of
        # tensor ops; the totals are the one device->host sync
        act8 = _mask_u8(active)
        # (rows of the scan: n_nodes, n_edges, n_jobs, n_sched = columns 0..3 of obs_i32)
        scan = torch.empty((2, 4, B), dtype=torch.int64, device=dev)
        tot = torch.empty(4, dtype=torch.int64, device=dev)
        with device_of(dev):  # (sss_prefix_rows takes no handle: it launches on the CURRENT device)
            self._b.check(self._b.lib.sss_prefix_rows(self.obs_i32.data_ptr(), 1, self.obs_i32.stride(0), act8.data_ptr() if act8 is not None else None, 4, B,
                                                     scan[0].data_ptr(), scan[1].data_ptr(), tot.data_ptr(), self._stream()))
        off, cnt_t = scan[0], scan[1]  # [4, B] each
        M, Ed, J, S = (int(v) for v in tot.tolist())
        # buffers hold at least one element so that their pointers are never NULL; `g` gets exact views
        pool = self._dg_pool if reuse_buffers else None

        def mk(n, dt, *tail, _c=[0]):
            if pool is None:
                return torch.empty((max(n, 1), *tail), dtype=dt, device=dev)
            k = _c[0]
            _c[0] += 1
            t = pool.get(k)
            if t is None or t.shape[0] < max(n, 1) or t.dtype != dt:
                t = pool[k] = torch.empty((max(2 * n, 1024), *tail), dtype=dt, device=dev)  # grows geometrically
            return t[: max(n, 1)]
        buf = {"x": mk(M, torch.float32, 5), "node_obs": mk(M, torch.int64), "node_loc": mk(M, torch.int64), "node_job": mk(M, torch.int64),
               "sched_rank": mk(M, torch.int64), "gen": mk(M, torch.int32), "node_recv": mk(M, torch.int32), "stage_mask": mk(M, torch.bool),
               "src": mk(Ed, torch.int64), "dst": mk(Ed, torch.int64), "edge_obs": mk(Ed, torch.int64), "edge_layers": mk(Ed, torch.int32),
               "job_obs": mk(J, torch.int64), "job_cap": mk(J, torch.int64), "job_first": mk(J, torch.int64), "obs_depth": mk(B, torch.int32),
               "job_nodes": mk(J, torch.int64), "out_start": mk(M, torch.int64), "out_deg": mk(M, torch.int32),
               "layer_cnt": torch.empty((32, B), dtype=torch.int32, device=dev), "sched_list": mk(S, torch.int64)}
        size = {"x": M, "node_obs": M, "node_loc": M, "node_job": M, "sched_rank": M, "gen": M, "node_recv": M, "stage_mask": M,
                "src": Ed, "dst": Ed, "edge_obs": Ed, "edge_layers": Ed, "job_obs": J, "job_cap": J, "job_first": J, "obs_depth": B,
                "job_nodes": J, "out_start": M, "out_deg": M, "layer_cnt": 32, "sched_list": S}
        g = {k: v[: size[k]] for k, v in buf.items()}
        # the layers' lists of receiving nodes, written by the graph kernel itself into work space the env keeps per stream
        # (max_dag_depth lists of up to M ids; lengths in `layer_totals`): valid until the next decima_graph on this stream
        D = self.max_dag_depth
        skey = (dev, self._stream())
        ls = self._layer_scratch.get(skey)
        if ls is None or ls["recv"].numel() < max(M, 1) * max(D, 1):
            ls = self._layer_scratch[skey] = {"recv": torch.empty(max(2 * M, 1 << 14) * max(D, 1), dtype=torch.int64, device=dev), "epoch": 0}
        ls["epoch"] += 1
        ls["stride"] = ls["recv"].numel() // max(D, 1)
        layer_totals = torch.zeros(33 * 32, dtype=torch.int64, device=dev)  # (i64[33][32]: a counter per layer and block of envs, the blocks' largest observation)
        a = SssDecimaGraph(act8.data_ptr() if act8 is not None else None, off[0].data_ptr(), off[2].data_ptr(), off[1].data_ptr(),
                           float(num_tasks_scale), float(work_scale), *(buf[k].data_ptr() for k in (
                               "x", "node_obs", "node_loc", "node_job", "sched_rank", "gen", "node_recv", "stage_mask", "src", "dst",
                               "edge_obs", "edge_layers", "job_obs", "job_cap", "job_first", "obs_depth", "job_nodes", "out_start", "out_deg", "layer_cnt")),
                           off[3].data_ptr(), buf["sched_list"].data_ptr(), layer_totals.data_ptr(), ls["recv"].data_ptr(), ls["stride"], None, 33 * 32)
        self._b.check(self._b.lib.sss_decima_graph_build(self._h, C.byref(a), self._stream()))
        g["n_obs"], g["n_pad"] = B, self.dims.node_cap
        g["max_depth"] = self.max_dag_depth
        g["layer_totals"], g["_layer_lists"] = layer_totals, (ls, ls["epoch"])
        g["obs_nodes"], g["obs_jobs"] = cnt_t[0], cnt_t[2]
        g["obs_node_off"], g["obs_job_off"] = off[0], off[2]
        g["_keepalive"] = (off, act8)
        g["_binding"] = self._b
        return g

    @staticmethod
    def decima_layer_lists(g: dict[str, Any]) -> list[torch.Tensor]:
        """for every DAG layer of a graph from `decima_graph` the ids of the nodes it updates (one
        small kernel for all layers; one device->host sync for the list sizes)"""
        if "recv_lists" not in g:
            lc = g["layer_cnt"]  # i32[32, B]
            dev_l = lc.device
            scan = torch.empty((32, lc.shape[1]), dtype=torch.int64, device=dev_l)
            tot = torch.empty(32, dtype=torch.int64, device=dev_l)
            b = g["_binding"]
            stream = torch.cuda.current_stream(dev_l).cuda_stream if dev_l.type == "cuda" else 0
            with device_of(dev_l):
                b.check(b.lib.sss_prefix_rows(lc.data_ptr(), lc.stride(0), 1, None, 32, lc.shape[1], scan.data_ptr(), None, tot.data_ptr(), stream))
            totals = tot.tolist()
            n_layers = max((lvl + 1 for lvl, c in enumerate(totals) if c), default=0)
            base = [0] * 32
            for lvl in range(1, 32):
                base[lvl] = base[lvl - 1] + totals[lvl - 1]
            recv = torch.empty(max(sum(totals), 1), dtype=torch.int64, device=dev_l)
            if n_layers:
                env_off = scan  # exclusive prefix of the per-env receiver counts, per layer
                a = SssDecimaLists(g["obs_node_off"].data_ptr(), g["obs_nodes"].data_ptr(), g["node_recv"].data_ptr(), env_off.data_ptr(),
                                   (C.c_int64 * 32)(*base), recv.data_ptr(), n_layers)
                with device_of(dev_l):
                    b.check(b.lib.sss_decima_layer_lists(g["n_obs"], C.byref(a), stream))
                g["_keepalive_lists"] = env_off
            g["recv_lists"] = [recv[base[lvl]: base[lvl] + totals[lvl]] for lvl in range(n_layers)]
        return g["recv_lists"]

    def raise_on_error(self) -> None:
        """the reference raises from inside step(); the batched env records a per-env code. This
        turns recorded codes into the reference's exception types (one device->host sync)."""
        err = self.obs_i32[:, 7].cpu().numpy()
        bad = np.nonzero(err)[0]
        if bad.size:
            code = int(err[bad[0]])
            msg = f"env(s) {bad.tolist()[:8]}: {ERROR_NAMES.get(code, code)}"
            if code in (5, 7):
                raise AssertionError(msg)
            if code == 2:
                raise KeyError(msg)
            raise ValueError(msg)

    # ---- per-env views (host copies) for unmodified reference-style plugins ---------------

    def obs_view(self, i: int) -> dict[str, Any]:
        """the reference observation dict of env i (spark_sched_sim.py:393-399) as numpy/python
        objects - what `Scheduler.schedule(obs)` plugins written for the reference consume."""
        from .spaces import GraphInstance

        s = self.obs_i32[i].cpu().numpy()
        n, ne, a = int(s[0]), int(s[1]), int(s[2])
        nodes = self.nodes[i, :n].cpu().numpy()
        el = self.edge_links[i, :ne].cpu().numpy().astype(np.int64)
        return {
            "dag_batch": GraphInstance(nodes, np.zeros(ne, dtype=int), el),
            "dag_ptr": self.dag_ptr[i, : a + 1].cpu().tolist(),
            "num_committable_execs": int(s[4]),
            "source_job_idx": int(s[5]),
            "exec_supplies": self.exec_supplies[i, :a].cpu().tolist(),
        }

    def _hdr_i32(self, i: int, byte_off: int) -> int:
        return int(self._env_view[i, byte_off: byte_off + 4].cpu().numpy().view(np.int32)[0])

    def job_times(self, i: int):
        """(t_arrival f64[J], t_completed f64[J], completion_order i16[J], template id i16[J]) of env
        i's current episode (between launches the HBM copy of every record is current)"""
        d = self.dims
        row = self._env_view[i].cpu().numpy()
        hdr = row[: d.hdr_bytes]
        J = int(hdr[HDR_OFF["J"]: HDR_OFF["J"] + 4].view(np.int32)[0])
        ta = row[d.off_t_arrival: d.off_t_arrival + 8 * J].view(np.float64).copy()
        tc = row[d.off_t_completed: d.off_t_completed + 8 * J].view(np.float64).copy()
        jobs = row[d.off_jobs: d.off_jobs + d.job_rec_bytes * J].reshape(J, d.job_rec_bytes)
        order = jobs[:, 56:58].copy().view(np.int16).ravel()
        gs_base = jobs[:, 60:64].copy().view(np.int32).ravel()
        tmpl = np.searchsorted(workload.pack_section(self._pack, "tmpl_stage_off"), gs_base).astype(np.int16)
        return ta, tc, order, tmpl

    def job_duration_buff(self, i: int) -> list[float]:
        """env i's deque of the last 200 completed-job durations, oldest first (spark_sched_sim.py:121, 693-697)"""
        d = self.dims
        h = self.header(i)
        ring = self._env_view[i, d.off_dur_ring: d.off_dur_ring + 8 * 200].cpu().numpy().view(np.float64)
        return [float(ring[(h["dur_head"] + k) % 200]) for k in range(h["dur_n"])]

    def header(self, i: int) -> dict[str, Any]:
        hdr = self._env_view[i, : self.dims.hdr_bytes].cpu().numpy()
        out = {}
        for name, off in HDR_OFF.items():
            w = HDR_W[name]
            out[name] = hdr[off: off + np.dtype(w).itemsize].view(w)[0].item()
        return out

    def header_field(self, name: str) -> torch.Tensor:
        """device tensor [B] of one header scalar (zero-copy view of the arena)"""
        off, w = HDR_OFF[name], HDR_W[name]
        tw = {np.float64: torch.float64, np.uint64: torch.int64, np.int32: torch.int32, np.uint32: torch.int32}[w]
        n = np.dtype(w).itemsize
        return self._env_view[:, off: off + n].view(tw).squeeze(1)

    def rollout_stats(self) -> dict[str, torch.Tensor]:
        """what the reference's rollout workers report per env (rollout_worker.py:122-130:
        `avg_job_duration` [s, over the last <=200 completed jobs, nan if none], `avg_num_jobs`
        (metrics.py:16-18), `num_completed_jobs`, `num_job_arrivals`) for ALL envs as f64[B] device
        tensors - no per-env host round trips"""
        d, B = self.dims, self.num_envs
        J = d.job_cap
        f64 = lambda off, n: self._env_view[:, off: off + 8 * n].view(torch.float64)  # noqa: E731
        ta, tc = f64(d.off_t_arrival, J), f64(d.off_t_completed, J)
        wall = self.header_field("wall_time")
        arrived = torch.arange(J, device=self.device)[None, :] < self.header_field("next_arrival")[:, None]
        dur = torch.where(arrived, torch.minimum(tc, wall[:, None]) - ta, torch.zeros_like(ta))
        n_ring = self.header_field("dur_n").to(torch.float64)
        ring = f64(d.off_dur_ring, 200)
        head = self.header_field("dur_head").long()
        k = (torch.arange(200, device=self.device)[None, :] - head[:, None]) % 200  # position of slot in deque order
        ring_sum = torch.where(k < n_ring[:, None], ring, torch.zeros_like(ring)).sum(1)
        done, act = self.header_field("n_completed").to(torch.float64), self.header_field("n_active").to(torch.float64)
        return {"avg_job_duration": ring_sum / n_ring * 1e-3, "avg_num_jobs": dur.sum(1) / wall, "num_completed_jobs": done,
                "num_job_arrivals": done + act}

    def counters(self) -> dict[str, int]:
        """lifetime totals over all envs: real step() calls, events popped, SURVEY 8(d) model bytes"""
        hdr = self._env_view[:, : self.dims.hdr_bytes].cpu().numpy()
        tot = {}
        for name in ("n_steps", "n_events", "model_bytes"):
            off = HDR_OFF[name]
            tot[name] = int(np.ascontiguousarray(hdr[:, off: off + 8]).view(np.uint64).sum())
        prof = np.ascontiguousarray(hdr[:, HDR_PROF: HDR_PROF + 40]).view(np.uint64).sum(axis=0)
        for k, name in enumerate(("ticks_slow_events", "ticks_action", "ticks_events", "ticks_reward", "ticks_observe")):
            tot[name] = int(prof[k])
        for name in ("n_fast", "n_batched", "n_rounds"):
            off = HDR_OFF[name]
            tot[name + "_events" if name != "n_rounds" else name] = int(np.ascontiguousarray(hdr[:, off: off + 8]).view(np.uint64).sum())
        return tot


# byte offsets inside SssHdr (csrc/sss_layout.h); pinned by tests/test_abi.py against the header compiled with gcc
HDR_PROF = 208  # uint64 prof[5]: shader-clock ticks in slow-path handlers, action + fulfil, event loop, reward, observe
HDR_OFF = {"wall_time": 40, "time_limit": 48, "seed": 56, "n_steps": 64, "n_events": 72, "model_bytes": 80,
           "counter": 88, "next_arrival": 92, "J": 96, "n_active": 100, "n_completed": 104, "curr_source": 108,
           "n_sched": 112, "n_commits": 124, "terminated": 136, "err": 140, "need_reset": 144, "dur_head": 148, "dur_n": 152,
           "episodes": 156, "last_reward": 160, "ep_return": 168, "last_ep_return": 176, "last_ep_wall": 184,
           "ep_steps": 192, "last_ep_steps": 196, "next_arrival_t": 200, "n_fast": 248, "n_batched": 256, "n_rounds": 264,
           "err_line": 272}  # source line of the kernel-side check that failed last (diagnostics for bug reports)
HDR_W = {"wall_time": np.float64, "time_limit": np.float64, "seed": np.uint64, "n_steps": np.uint64,
         "n_events": np.uint64, "model_bytes": np.uint64, "counter": np.uint32, "next_arrival": np.int32,
         "J": np.int32, "n_active": np.int32, "n_completed": np.int32, "curr_source": np.uint32,
         "n_sched": np.int32, "n_commits": np.int32, "terminated": np.int32, "err": np.int32, "need_reset": np.int32,
         "dur_head": np.int32, "dur_n": np.int32, "episodes": np.int32, "last_reward": np.float64,
         "ep_return": np.float64, "last_ep_return": np.float64, "last_ep_wall": np.float64,
         "ep_steps": np.int32, "last_ep_steps": np.int32, "next_arrival_t": np.float64, "n_fast": np.uint64,
         "n_batched": np.uint64, "n_rounds": np.uint64, "err_line": np.uint64}
