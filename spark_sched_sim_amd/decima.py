"""Decima on the batched env: the observation transform of the reference's `DecimaObsWrapper`
(reference schedulers/decima/env_wrapper.py:37-143, DAG-layer edge masks of decima/utils.py:238-267)
and the Decima GNN policy (decima/scheduler.py:16-385) as plain PyTorch over ALL envs of a
`VecSparkSchedSimEnv` at once - no torch_geometric / torch_sparse / torch_scatter, no host round
trip: the env's observation tensors are consumed where they are (device, padded per env).

This is the first "next" row of SURVEY 8(f); the simulator itself does not depend on it.

Layout: `decima_observation` works on the env's own tensors, padded per env to the env's capacities
(N = node_cap, A = job_cap, Ed = edge_cap; validity from `n_nodes / n_jobs / n_edges`).
`compact_graph` then drops the padding: a batch of observations becomes ONE flat graph (nodes, jobs
and edges of all observations back to back, like the reference's `collate_obsns`), which is what
the GNN runs on and what `GraphArena` stores for training. `DecimaPolicy`'s parameter names match
the reference's `DecimaScheduler.state_dict()` so its checkpoints load unchanged.
"""
from __future__ import annotations

from typing import Any

import torch
import torch.nn as nn

NUM_NODE_FEATURES = 5  # env_wrapper.py:9
NUM_DAG_FEATURES = 3   # scheduler.py:33


_NO_HINT = [-1] * 32  # sss_gnn_encode_args.layer_rows_hint: no idea (the launch is sized by the node count)


def decima_observation(obs, num_executors: int, max_depth: int, num_tasks_scale: int = 200, work_scale: float = 1e5,
                       edge_masks: bool = False) -> dict[str, torch.Tensor]:
    """`DecimaObsWrapper.observation` for every env of a `BatchedObs`, padded per env.

    Returns (B = envs): x f32[B,N,5], node_valid / stage_mask bool[B,N], node_job i64[B,N]
    (job slot of each node, A for padding), job_valid bool[B,A], exec_mask bool[B,A,E],
    commit_caps i64[B,A], edge_src / edge_dst i64[B,Ed] (N for padding), edge_valid bool[B,Ed],
    gen i64[B,N] (topological generation of each node in the active subgraph), depth i64[B]
    (= number of DAG-layer masks the reference builds), has_mp bool[B] (the reference skips message
    passing when an observation has a single DAG layer, scheduler.py:196-198).
    `max_depth` bounds the longest path (e.g. the pack's stages-per-job bound).
    With `edge_masks=True` also the reference's dense edge_masks bool[L,B,Ed], L = deepest DAG in
    the batch (one device->host sync); the policy itself works from `gen` and does not need them.
    """
    nodes = obs["nodes"]
    B, N, _ = nodes.shape
    dev = nodes.device
    E = num_executors
    n_nodes, n_jobs, n_edges = obs["n_nodes"].long(), obs["n_jobs"].long(), obs["n_edges"].long()
    A = obs["exec_supplies"].shape[1]
    Ed = obs["edge_links"].shape[1]
    ar_n = torch.arange(N, device=dev)
    ar_a = torch.arange(A, device=dev)
    node_valid = ar_n[None, :] < n_nodes[:, None]
    job_valid = ar_a[None, :] < n_jobs[:, None]

    # node -> job slot from dag_ptr (rows are only maintained up to n_jobs + 1 entries)
    ptr_hi = obs["dag_ptr"][:, 1:].long()
    ptr_hi = torch.where(job_valid, ptr_hi, torch.full_like(ptr_hi, N + 1))
    node_job = torch.searchsorted(ptr_hi.contiguous(), ar_n[None, :].expand(B, N).contiguous(), right=True)
    node_job = torch.where(node_valid, node_job, torch.full_like(node_job, A))

    # cap on executors that can be committed to each job (env_wrapper.py:72-82)
    supplies = torch.where(job_valid, obs["exec_supplies"].long(), torch.zeros_like(obs["exec_supplies"].long()))
    ncommit = obs["num_committable_execs"].long()
    gap = (E - supplies).clamp(min=0)
    commit_caps = torch.minimum(gap, ncommit[:, None])
    src = obs["source_job_idx"].long()
    is_src_job = ar_a[None, :] == src[:, None]
    commit_caps = torch.where(is_src_job & job_valid, ncommit[:, None].expand(B, A), commit_caps)
    commit_caps = torch.where(job_valid, commit_caps, torch.zeros_like(commit_caps))

    # node features (env_wrapper.py:110-143); the int -> f64 -> f32 roundings follow numpy's
    pad = torch.zeros((B, 1), dtype=torch.long, device=dev)
    caps_n = torch.cat([commit_caps, pad], 1).gather(1, node_job)
    sup_n = torch.cat([supplies, pad], 1).gather(1, node_job)
    x = torch.zeros((B, N, NUM_NODE_FEATURES), dtype=torch.float32, device=dev)
    # divisors are device tensors: torch's GPU kernels turn "/ python_scalar" into a multiply by the
    # reciprocal, which is not the reference's (numpy) division
    e_f64 = torch.full((1,), float(E), dtype=torch.float64, device=dev)
    x[..., 0] = (caps_n.double() / e_f64).float()
    x[..., 1] = torch.where(node_job == src[:, None], 1.0, -1.0)
    x[..., 2] = (sup_n.double() / e_f64).float()
    rem, dur = nodes[..., 0], nodes[..., 1]
    x[..., 3] = rem / torch.full((1,), float(num_tasks_scale), dtype=torch.float32, device=dev)
    x[..., 4] = rem * dur / torch.full((1,), float(work_scale), dtype=torch.float32, device=dev)
    x = torch.where(node_valid[..., None], x, torch.zeros_like(x))
    stage_mask = (nodes[..., 2] != 0) & node_valid

    exec_mask = torch.arange(E, device=dev)[None, None, :] < commit_caps[..., None]

    # topological generations of the active subgraph (decima/utils.py:246-247) by relaxation
    edge_valid = torch.arange(Ed, device=dev)[None, :] < n_edges[:, None]
    el = obs["edge_links"].long()
    e_src = torch.where(edge_valid, el[..., 0], torch.full_like(el[..., 0], N))
    e_dst = torch.where(edge_valid, el[..., 1], torch.full_like(el[..., 1], N))
    gen = torch.zeros((B, N + 1), dtype=torch.long, device=dev)
    for _ in range(max_depth):
        cand = torch.where(edge_valid, gen.gather(1, e_src) + 1, torch.zeros_like(e_src))
        gen = gen.scatter_reduce(1, e_dst, cand, "amax", include_self=True)
        gen[:, N] = 0
    gen_n = gen[:, :N]
    max_gen = torch.where(node_valid, gen_n, torch.zeros_like(gen_n)).amax(1)
    out = {"x": x, "node_valid": node_valid, "stage_mask": stage_mask, "node_job": node_job, "job_valid": job_valid,
           "exec_mask": exec_mask, "edge_src": e_src, "edge_dst": e_dst, "edge_valid": edge_valid,
           "gen": gen_n, "has_mp": max_gen > 0, "depth": max_gen, "commit_caps": commit_caps,
           "dag_start": obs["dag_ptr"][:, :-1].long(), "n_nodes": n_nodes, "n_edges": n_edges}
    if edge_masks:
        # DAG-layer masks (decima/utils.py:249-267): mask l = edges with both ends in
        # (generation l) U (its successors)
        masks = []
        for lvl in range(int(max_gen.max())):
            in_lvl = torch.cat([(gen_n == lvl) & node_valid, torch.zeros((B, 1), dtype=torch.bool, device=dev)], 1)
            succ = torch.zeros((B, N + 1), dtype=torch.long, device=dev)
            succ = succ.scatter_reduce(1, e_dst, (in_lvl.gather(1, e_src) & edge_valid).long(), "amax", include_self=True)
            in_m = in_lvl | (succ > 0)
            masks.append(in_m.gather(1, e_src) & in_m.gather(1, e_dst) & edge_valid)
        out["edge_masks"] = torch.stack(masks) if masks else torch.zeros((0, B, Ed), dtype=torch.bool, device=dev)
    return out


def _excl_cumsum(v: torch.Tensor) -> torch.Tensor:
    return torch.cumsum(v, 0) - v


def compact_graph(f: dict[str, torch.Tensor]) -> dict[str, Any]:
    """the batch as ONE graph over the valid nodes only (what `collate_obsns` / PyG batching does in
    the reference, decima/utils.py:117-204). Flat tensors, M nodes / J jobs / Ed edges in total:
    x f32[M,5], node_obs / node_loc / node_job / gen i64[M] (observation, position inside it, global
    job id, topological generation), stage_mask bool[M], src / dst i64[Ed] (global node ids),
    sched_rank i64[M] (index among the observation's schedulable stages, -1 if not schedulable),
    edge_obs i64[Ed], job_obs i64[J], job_cap i64[J] (allowed executor counts = 1..cap),
    job_first i64[J] (global id of the job's first node), n_obs, obs_nodes i64[n_obs],
    obs_depth i64[n_obs]. Padding never reaches the MLPs. Costs a few device->host syncs (sizes)."""
    x = f["x"]
    B = x.shape[0]
    env_n, loc_n = f["node_valid"].nonzero(as_tuple=True)
    env_j, loc_j = f["job_valid"].nonzero(as_tuple=True)
    env_e, loc_e = f["edge_valid"].nonzero(as_tuple=True)
    n_nodes = f["n_nodes"]
    node_off = _excl_cumsum(n_nodes)
    job_off = _excl_cumsum(f["job_valid"].sum(1))
    rank = f["stage_mask"].long().cumsum(1) - 1
    return {"x": x[env_n, loc_n], "node_obs": env_n, "node_loc": loc_n, "n_pad": x.shape[1],
            "sched_rank": torch.where(f["stage_mask"], rank, torch.full_like(rank, -1))[env_n, loc_n],
            "node_job": job_off[env_n] + f["node_job"][env_n, loc_n], "gen": f["gen"][env_n, loc_n],
            "stage_mask": f["stage_mask"][env_n, loc_n],
            "src": node_off[env_e] + f["edge_src"][env_e, loc_e], "dst": node_off[env_e] + f["edge_dst"][env_e, loc_e],
            "edge_obs": env_e, "job_obs": env_j, "job_cap": f["commit_caps"][env_j, loc_j],
            "job_first": node_off[env_j] + f["dag_start"][env_j, loc_j], "n_obs": B,
            "obs_nodes": n_nodes, "obs_jobs": f["job_valid"].sum(1), "obs_depth": f["depth"]}


def bit_lists(bits: torch.Tensor, n_layers: int, binding=None) -> list[torch.Tensor]:
    """[positions e with bit l of bits[e] set, ascending, for l < n_layers] for an int32 mask array - `((bits >> l) & 1).nonzero()`
    for all layers in three launches and ONE device->host read (include/sss.h sss_bit_lists + sss_prefix_rows) instead of a
    `nonzero` with its read per layer"""
    import ctypes

    from .binding import SssBitListArgs, device_of
    if binding is None:
        from .train_kernels import _binding
        binding = _binding()
    dev, n = bits.device, int(bits.numel())
    assert bits.dtype == torch.int32 and bits.is_contiguous() and 1 <= n_layers <= 32
    if n == 0:
        return [torch.zeros(0, dtype=torch.int64, device=dev) for _ in range(n_layers)]
    chunk = 2048
    n_chunks = (n + chunk - 1) // chunk
    cnt = torch.empty((n_chunks, n_layers), dtype=torch.int32, device=dev)
    off = torch.empty((n_layers, n_chunks), dtype=torch.int64, device=dev)
    tot = torch.empty(n_layers, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    a = SssBitListArgs(bits.data_ptr(), n, n_layers, chunk, n_chunks, 0, cnt.data_ptr(), None, (ctypes.c_int64 * 32)(), None)
    with device_of(dev):
        binding.check(binding.lib.sss_bit_lists(ctypes.byref(a), stream))
        binding.check(binding.lib.sss_prefix_rows(cnt.data_ptr(), 1, n_layers, None, n_layers, n_chunks, off.data_ptr(), None, tot.data_ptr(), stream))
    sizes = [int(v) for v in tot.tolist()]  # (the one read)
    out = torch.empty(max(sum(sizes), 1), dtype=torch.int64, device=dev)
    base, run = (ctypes.c_int64 * 32)(), 0
    for l, k in enumerate(sizes):
        base[l], run = run, run + k
    a.phase, a.off_dev, a.out_dev, a.base = 1, off.data_ptr(), out.data_ptr(), base
    with device_of(dev):
        binding.check(binding.lib.sss_bit_lists(ctypes.byref(a), stream))
    return [out[base[l]: base[l] + sizes[l]] for l in range(n_layers)]


def edges_grouped_by_source(g: dict[str, Any]) -> bool:
    """whether `src` is non-decreasing, i.e. every node's out-edges are consecutive - what the segment sums of the kernel message
    passing rely on. True by construction for graphs written by the graph kernel (they carry each node's out-edge range,
    `out_start`); checked once (one device->host read) and remembered for any other graph - a reference-format observation
    may list its `edge_links` in any order."""
    if "_src_sorted" not in g:
        src = g["src"]
        g["_src_sorted"] = "out_start" in g or src.numel() < 2 or bool((src[1:] >= src[:-1]).all())
    return g["_src_sorted"]


def graph_layers(g: dict[str, Any]) -> list[torch.Tensor]:
    """for every DAG layer l (decima/utils.py:249-267): (ids of the edges whose two ends lie in
    (generation l) U (its successors) - the reference's `edge_masks[l]` as an index list, ids of the
    nodes that are the source end of one of them - the nodes the layer updates)"""
    if "layers" not in g and "edge_layers" in g:
        # the graph kernel (include/sss.h sss_decima_graph_build) already marked every edge / node
        # with the layers it belongs to: one bit test + nonzero per layer
        depth = int(g["obs_depth"].max()) if g["x"].shape[0] else 0
        el, nr = g["edge_layers"], g["node_recv"]
        if el.is_cuda and 1 <= depth <= 32 and el.numel() >= 8192 and el.dtype == nr.dtype == torch.int32:
            g["layers"] = list(zip(bit_lists(el.contiguous(), depth), bit_lists(nr.contiguous(), depth)))  # two reads instead of 2 x depth
        else:
            g["layers"] = [(((el >> lvl) & 1).nonzero(as_tuple=True)[0], ((nr >> lvl) & 1).nonzero(as_tuple=True)[0]) for lvl in range(depth)]
        return g["layers"]
    if "layers" not in g:
        gen, src, dst = g["gen"], g["src"], g["dst"]
        M = gen.numel()
        depth = int(gen.max()) if M else 0
        layers = []
        for lvl in range(depth):
            in_m = gen == lvl
            hit = torch.zeros(M, dtype=torch.int32, device=gen.device).index_add_(0, dst, in_m[src].to(torch.int32))
            in_m = in_m | (hit > 0)
            e = (in_m[src] & in_m[dst]).nonzero(as_tuple=True)[0]
            recv = torch.zeros(M, dtype=torch.bool, device=gen.device).index_fill_(0, src[e], True).nonzero(as_tuple=True)[0]
            layers.append((e, recv))
        g["layers"] = layers
    return g["layers"]


def _take(t: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """`t[idx]` along the first axis; large gathers of 4-byte-multiple rows on the GPU go through the row kernel (include/sss.h
    sss_rows_op GATHER moves bits, whatever the dtype: the library's index kernel ran at a tenth of its rate, profiles/r04_ppo.md)"""
    row_bytes = t.element_size() * (t[0].numel() if t.dim() > 1 else 1) if t.numel() else 0
    if not (t.is_cuda and idx.numel() >= 8192 and t.is_contiguous() and row_bytes % 4 == 0 and 4 <= row_bytes <= 256 and t.dtype != torch.bool):
        return t[idx]
    from .train_kernels import ROWS_GATHER, rows_op
    w = row_bytes // 4
    src = t.view(-1).view(torch.float32).view(t.shape[0], w)
    out = torch.empty((idx.numel(), *t.shape[1:]), dtype=t.dtype, device=t.device)
    rows_op(ROWS_GATHER, idx, out.view(-1).view(torch.float32).view(idx.numel(), w), src)
    return out


def select_observations(g: dict[str, Any], obs_idx: torch.Tensor) -> dict[str, Any]:
    """the sub-batch made of observations `obs_idx` (i64[k], in that order) of a compact graph,
    re-labelled - a PPO minibatch out of a `GraphArena`"""
    dev = g["x"].device
    n_obs = g["n_obs"]
    new_of_old = None  # (the general form's look-up table, built below when it is needed)

    def pick(owner):  # members of the selected observations, grouped by new observation id (stable)
        new_owner = new_of_old[owner]
        keep = (new_owner >= 0).nonzero(as_tuple=True)[0]
        order = torch.sort(new_owner[keep], stable=True)[1]
        return keep[order]

    # An arena whose members are stored observation by observation (what `concat_graphs` of per-step graphs gives: owners
    # non-decreasing) is cut by RANGES: the work is proportional to the minibatch, not to the arena - at BASELINE config 5 the
    # arena is 211 M nodes and a minibatch 2.5 M, and `pick` reads the whole arena three times per minibatch.
    cache = g.get("_ranges")
    if cache is None:
        # (`_by_observation`: the graph says of itself that its members are stored observation by observation - a GraphArena's
        # record is, by construction - and its per-observation node / job counts are its own `obs_nodes` / `obs_jobs`: no pass over
        # the 211 M node ids of a config-5 record to find that out)
        known = bool(g.get("_by_observation"))

        def ranges(owner, cnt=None):
            if not known and owner.numel() > 1 and not bool((owner[1:] >= owner[:-1]).all()):
                return None
            if cnt is None:
                cnt = torch.bincount(owner, minlength=n_obs)
            return cnt, torch.cumsum(cnt, 0) - cnt
        r = [ranges(g["node_obs"], g["obs_nodes"] if known else None), ranges(g["job_obs"], g["obs_jobs"] if known else None), ranges(g["edge_obs"])]
        cache = g["_ranges"] = r if all(x is not None for x in r) else False
    if cache:
        # Members of an observation are a RANGE of the arena's rows, so ids can be re-labelled by arithmetic on minibatch-sized
        # arrays: member m of observation o (new id o') that sat at arena row r sits at new row r - first_old[o] + first_new[o'].
        # (The general form below builds arena-sized look-up tables per minibatch - 211 M + 32 M entries at BASELINE config 5, 2 GB
        # written and then read through at random for every minibatch.)
        k = obs_idx.numel()

        def cut(cnt_off):
            cnt, off = cnt_off[0][obs_idx], cnt_off[1][obs_idx]
            total = int(cnt.sum())
            first_new = torch.cumsum(cnt, 0) - cnt
            which = torch.repeat_interleave(torch.arange(k, device=dev), cnt, output_size=total)  # new observation id of every member
            shift = off - first_new                                                              # old row - new row, per observation
            return shift[which] + torch.arange(total, device=dev), which, shift
        (kn, wn, sn), (kj, wj, sj), (ke, we, _) = cut(cache[0]), cut(cache[1]), cut(cache[2])
        T = _take
        return {"x": T(g["x"], kn), "node_obs": wn, "node_loc": T(g["node_loc"], kn), "node_job": T(g["node_job"], kn) - sj[wn],
                "gen": T(g["gen"], kn), "stage_mask": g["stage_mask"][kn], "sched_rank": T(g["sched_rank"], kn), "n_pad": g["n_pad"],
                "src": T(g["src"], ke) - sn[we], "dst": T(g["dst"], ke) - sn[we], "edge_obs": we,
                "job_obs": wj, "job_cap": T(g["job_cap"], kj), "job_first": T(g["job_first"], kj) - sn[wj],
                "n_obs": int(k), "obs_nodes": g["obs_nodes"][obs_idx], "obs_jobs": g["obs_jobs"][obs_idx], "obs_depth": g["obs_depth"][obs_idx],
                **({"edge_layers": T(g["edge_layers"], ke), "node_recv": T(g["node_recv"], kn)} if "edge_layers" in g else {})}
    else:
        new_of_old = torch.full((n_obs,), -1, dtype=torch.long, device=dev)
        new_of_old[obs_idx] = torch.arange(obs_idx.numel(), device=dev)
        kn, kj, ke = pick(g["node_obs"]), pick(g["job_obs"]), pick(g["edge_obs"])
    node_new = torch.full((g["x"].shape[0],), -1, dtype=torch.long, device=dev)
    node_new[kn] = torch.arange(kn.numel(), device=dev)
    job_new = torch.full((g["job_obs"].numel(),), -1, dtype=torch.long, device=dev)
    job_new[kj] = torch.arange(kj.numel(), device=dev)
    T = _take
    return {"x": T(g["x"], kn), "node_obs": T(new_of_old, T(g["node_obs"], kn)), "node_loc": T(g["node_loc"], kn),
            "node_job": T(job_new, T(g["node_job"], kn)), "gen": T(g["gen"], kn), "stage_mask": g["stage_mask"][kn],
            "sched_rank": T(g["sched_rank"], kn), "n_pad": g["n_pad"],
            "src": T(node_new, T(g["src"], ke)), "dst": T(node_new, T(g["dst"], ke)), "edge_obs": T(new_of_old, T(g["edge_obs"], ke)),
            "job_obs": T(new_of_old, T(g["job_obs"], kj)), "job_cap": T(g["job_cap"], kj), "job_first": T(node_new, T(g["job_first"], kj)),
            "n_obs": int(obs_idx.numel()), "obs_nodes": g["obs_nodes"][obs_idx], "obs_jobs": g["obs_jobs"][obs_idx],
            "obs_depth": g["obs_depth"][obs_idx],
            **({"edge_layers": T(g["edge_layers"], ke), "node_recv": T(g["node_recv"], kn)} if "edge_layers" in g else {})}


def concat_graphs(gs: list[dict[str, Any]]) -> dict[str, Any]:
    """observations of several compact graphs back to back (ids shifted)"""
    out: dict[str, Any] = {}
    n_off = j_off = o_off = 0
    parts: dict[str, list] = {k: [] for k in ("x", "node_obs", "node_loc", "node_job", "gen", "stage_mask", "sched_rank", "src", "dst",
                                             "edge_obs", "job_obs", "job_cap", "job_first", "obs_nodes", "obs_jobs", "obs_depth")}
    if gs and all("edge_layers" in g for g in gs):
        parts["edge_layers"], parts["node_recv"] = [], []
    for g in gs:
        shift = {"node_obs": o_off, "edge_obs": o_off, "job_obs": o_off, "node_job": j_off, "src": n_off, "dst": n_off,
                 "job_first": n_off}
        for k in parts:
            parts[k].append(g[k] + shift[k] if k in shift else g[k])
        n_off += g["x"].shape[0]
        j_off += g["job_obs"].numel()
        o_off += g["n_obs"]
    for k, v in parts.items():
        out[k] = torch.cat(v)
    out["n_obs"] = o_off
    out["n_pad"] = max(g["n_pad"] for g in gs) if gs else 0
    out["gen"] = out["gen"].long()
    return out


# (in, hidden 1, hidden 2, out) of the seven MLPs of the published architecture (config/decima_tpch.yaml:68-78: embed 16, GNN MLPs
# [32, 16], policy MLPs [64, 64]); a row of an MLP costs 2 * (in * h1 + h1 * h2 + h2 * out) flops
MLP_DIMS = {"prep": (NUM_NODE_FEATURES, 32, 16, 16), "msg": (16, 32, 16, 16), "update": (16, 32, 16, 16), "dag": (NUM_NODE_FEATURES + 16, 32, 16, 16),
            "glob": (16, 32, 16, 16), "stage": (NUM_NODE_FEATURES + 3 * 16, 64, 64, 1), "exec": (NUM_DAG_FEATURES + 2 * 16 + 1, 64, 64, 1)}


def _popcount_sum(t: torch.Tensor, bits: int = 32) -> int:
    return int(sum(int(((t >> b) & 1).sum()) for b in range(bits))) if t.numel() else 0


def algorithmic_cost(g: dict[str, Any], exec_rows: int) -> dict[str, Any]:
    """Algorithmic work of ONE forward pass of the published Decima architecture over the observations of compact graph `g`
    (scheduler.py:142-385) - what bench.py prices `decima_in_loop` and `ppo_config5_share` against:

      rows    MLP evaluations the pass needs: prep and dag once per active node; msg once per (edge, DAG layer whose mask holds the
              edge) and update once per (receiving node, layer) (scheduler.py:196-241); glob once per active job; stage once per
              schedulable node; exec once per allowed (job, executor count) pair of the chosen jobs (`exec_rows`, given by the caller)
      flops   sum over rows of 2 * (in * h1 + h1 * h2 + h2 * out): the matrix products only (activations, sums, softmax not counted)
      bytes_inference   HBM bytes when nothing but inputs and outputs of each MLP touch memory: 4 B * (in + out) per row
      bytes_training    forward + backward with every MLP's activations stored once and read once, gradients of the same size
                        written and read: 12 B * (in + h1 + h2 + out) per row
    A training step (forward + backward: input and weight gradients) is 3 x the forward flops."""
    rows = {"prep": int(g["x"].shape[0]), "dag": int(g["x"].shape[0]), "glob": int(g["job_obs"].numel()), "stage": int(g["stage_mask"].sum()),
            "exec": int(exec_rows), "msg": _popcount_sum(g["edge_layers"]) if "edge_layers" in g else 0,
            "update": _popcount_sum(g["node_recv"]) if "node_recv" in g else 0}
    flops = sum(n * 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[3]) for k, n in rows.items() for d in (MLP_DIMS[k],))
    return {"rows": rows, "flops": float(flops), "bytes_inference": float(sum(n * 4 * (MLP_DIMS[k][0] + MLP_DIMS[k][3]) for k, n in rows.items())),
            "bytes_training": float(sum(n * 12 * sum(MLP_DIMS[k]) for k, n in rows.items()))}


def make_mlp(input_dim: int, hid_dims: list[int], output_dim: int, act_cls: str, act_kwargs: dict[str, Any] | None = None) -> nn.Sequential:
    """Linear / activation stack with the reference's layer numbering (decima/utils.py:44-64)"""
    act = getattr(torch.nn.modules.activation, act_cls)
    kwargs = dict(act_kwargs or {})
    kwargs.pop("inplace", None)
    layers: list[nn.Module] = []
    prev = input_dim
    dims = list(hid_dims) + [output_dim]
    # nn.Linear whose weight gradient comes from the hand-written kernel, inside an nn.Sequential that evaluates the
    # architecture's three-layer MLPs with one forward and one backward kernel in the training path
    from .train_kernels import KernelLinear, KernelMLP
    for i, d in enumerate(dims):
        layers.append(KernelLinear(prev, d))
        if i < len(dims) - 1:
            layers.append(act(**kwargs))
        prev = d
    return KernelMLP(*layers)


class _NodeEncoder(nn.Module):
    def __init__(self, nf: int, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp_prep = make_mlp(nf, output_dim=emb, **mlp_kwargs)
        self.mlp_msg = make_mlp(emb, output_dim=emb, **mlp_kwargs)
        self.mlp_update = make_mlp(emb, output_dim=emb, **mlp_kwargs)

    KERNEL_MESSAGE_PASSING = True

    def _kernel_message_passing(self, h_init: torch.Tensor) -> bool:
        """the training path on a GPU, with the published 16 -> 32 -> 16 -> 16 LeakyReLU MLPs"""
        from .train_kernels import MIN_ROWS, KernelMLP
        return (self.KERNEL_MESSAGE_PASSING and h_init.is_cuda and h_init.dtype == torch.float32 and torch.is_grad_enabled() and h_init.requires_grad
                and h_init.shape[0] >= MIN_ROWS and h_init.shape[1] == 16 and all(isinstance(m, KernelMLP) and m._fused_spec() for m in (self.mlp_msg, self.mlp_update)))

    def forward(self, g: dict[str, Any], per_obs_skip: bool) -> torch.Tensor:
        """child -> parent ("reverse flow") message passing one DAG layer at a time, deepest layer
        first (scheduler.py:192-236). Per layer only that layer's edges and receiving nodes are
        touched: messages are evaluated per edge (a child with two parents in the layer is evaluated
        twice - in-degree is small), the update once per receiving node.

        A graph with a single DAG layer gets `mlp_prep` only (scheduler.py:196-198, 238-243). The
        reference applies that test to whatever it is given: one observation when scheduling, the
        whole collated batch when training. `per_obs_skip` selects the former for every observation
        of the batch independently (batched inference == the reference's one-at-a-time inference)."""
        x, src, dst = g["x"], g["src"], g["dst"]
        M = x.shape[0]
        h_init = self.mlp_prep(x)
        layers = graph_layers(g)
        if not layers:
            return h_init
        # nodes that are never the source end of an edge start from update(h_init), the rest from 0 (the reference evaluates
        # the update network on every node and masks; only the rows that are kept are evaluated here)
        from .train_kernels import gather_rows, segment_sum
        is_parent = torch.zeros(M, dtype=torch.bool, device=x.device).index_fill_(0, src, True)
        leaf = (~is_parent).nonzero(as_tuple=True)[0]
        h = segment_sum(self.mlp_update(gather_rows(h_init, leaf, unique=True)), leaf, M, "unique")
        if self._kernel_message_passing(h_init) and edges_grouped_by_source(g):
            # one autograd node for the whole loop, both MLPs on the MLP kernels (train_kernels._MessagePassFn): a receiver's
            # messages are summed as a RANGE of the layer's edge rows - which they are when the edges are stored source node by
            # source node (every graph built here; checked once per graph for graphs from elsewhere, else the form below)
            from .train_kernels import message_passing
            plan = [(_take(dst, e), torch.searchsorted(recv, _take(src, e)), recv) for e, recv in reversed(layers)]  # (_take: the row kernel's gather)
            h = message_passing(h_init, h, plan, self.mlp_msg, self.mlp_update)
        else:
            # (row gathers as index_select: its backward is one index_add_, the advanced-indexing form sorts its indices first)
            for e, recv in reversed(layers):
                msg = self.mlp_msg(h.index_select(0, dst[e]))
                agg = torch.zeros_like(h_init).index_add_(0, src[e], msg)
                h = h.index_copy(0, recv, h_init.index_select(0, recv) + self.mlp_update(agg.index_select(0, recv)))
        if per_obs_skip:
            h = torch.where((g["obs_depth"] > 0)[g["node_obs"]][:, None], h, h_init)
        return h


class _DagEncoder(nn.Module):
    def __init__(self, nf: int, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp = make_mlp(nf + emb, output_dim=emb, **mlp_kwargs)

    def forward(self, h_node: torch.Tensor, g: dict[str, Any]) -> torch.Tensor:
        """per-job sums f32[J,emb] (scheduler.py:246-262)"""
        from .train_kernels import segment_sum
        from .train_kernels import KernelMLP
        # (the update's kernels read a row's two pieces where they are: the [M, 21] concatenation and the slices of its gradient are never built)
        y = self.mlp.forward_cat(g["x"], h_node) if isinstance(self.mlp, KernelMLP) else self.mlp(torch.cat([g["x"], h_node], -1))
        return segment_sum(y, g["node_job"], g["job_obs"].numel(), "sorted")  # (an observation's nodes are stored job by job)


class _GlobalEncoder(nn.Module):
    def __init__(self, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp = make_mlp(emb, output_dim=emb, **mlp_kwargs)

    def forward(self, h_dag: torch.Tensor, g: dict[str, Any]) -> torch.Tensor:
        """per-observation sums f32[n_obs,emb] (scheduler.py:265-283)"""
        from .train_kernels import segment_sum
        y = self.mlp(h_dag)
        return segment_sum(y, g["job_obs"], g["n_obs"], "sorted")


class _Encoder(nn.Module):
    def __init__(self, nf: int, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.node_encoder = _NodeEncoder(nf, emb, mlp_kwargs)
        self.dag_encoder = _DagEncoder(nf, emb, mlp_kwargs)
        self.global_encoder = _GlobalEncoder(emb, mlp_kwargs)


class _ScoreNet(nn.Module):
    def __init__(self, input_dim: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp_score = make_mlp(input_dim, output_dim=1, **mlp_kwargs)


def _segment_log_softmax(scores: torch.Tensor, owner: torch.Tensor, n_seg: int):
    """(probs clamped like torch.distributions' clamp_probs, their logs) of a softmax taken inside
    each segment - the reference's `utils.evaluate` (decima/utils.py:26-41)"""
    mx = torch.full((n_seg,), float("-inf"), dtype=scores.dtype, device=scores.device).scatter_reduce(0, owner, scores.detach(), "amax")
    ex = (scores - mx[owner]).exp()
    den = torch.zeros(n_seg, dtype=scores.dtype, device=scores.device).index_add_(0, owner, ex)
    probs = ex / (den[owner] + 1e-16)
    eps = torch.finfo(probs.dtype).eps
    probs = probs.clamp(min=eps, max=1 - eps)
    return probs, probs.log()


class DecimaPolicy(nn.Module):
    """the reference's Decima architecture and trainable-scheduler surface (scheduler.py:16-139),
    evaluated over a batch of observations at once"""

    def __init__(self, num_executors: int, embed_dim: int, gnn_mlp_kwargs: dict[str, Any], policy_mlp_kwargs: dict[str, Any],
                 state_dict_path: str | None = None, opt_cls: str | None = None, opt_kwargs: dict[str, Any] | None = None,
                 max_grad_norm: float | None = None, **_unused):
        super().__init__()
        self.name = "Decima"
        self.num_executors = num_executors
        self.max_grad_norm = max_grad_norm
        self.encoder = _Encoder(NUM_NODE_FEATURES, embed_dim, gnn_mlp_kwargs)
        self.stage_policy_network = _ScoreNet(NUM_NODE_FEATURES + 3 * embed_dim, policy_mlp_kwargs)
        self.exec_policy_network = _ScoreNet(NUM_DAG_FEATURES + 2 * embed_dim + 1, policy_mlp_kwargs)
        for n_, p in self.named_parameters():  # scheduler.py:66-69
            if "bias" in n_:
                p.data.zero_()
        if state_dict_path:
            self.load_state_dict(torch.load(state_dict_path, map_location="cpu"))
        self.optim = getattr(torch.optim, opt_cls)(self.parameters(), **(opt_kwargs or {})) if opt_cls else None

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    # ---- fused inference kernels (include/sss.h sss_gnn_launch) -------------------------------

    def bind_kernels(self, binding) -> "DecimaPolicy":
        """use the hand-written GNN kernels for inference (`act`); `binding` is the env's
        `spark_sched_sim_amd.binding.Binding` (env._b). Training (`evaluate_actions`) always runs
        on autograd tensor ops."""
        self._kb = binding
        self._packed = None
        return self

    def _kernel_arch_ok(self) -> bool:
        def dims(mlp):
            return [(m.in_features, m.out_features) for m in mlp if isinstance(m, nn.Linear)]

        def acts(mlp):
            return [type(m) for m in mlp if not isinstance(m, nn.Linear)]
        enc = self.encoder
        gnn = [enc.node_encoder.mlp_prep, enc.node_encoder.mlp_msg, enc.node_encoder.mlp_update, enc.dag_encoder.mlp, enc.global_encoder.mlp]
        want = [[(5, 32), (32, 16), (16, 16)]] + [[(16, 32), (32, 16), (16, 16)]] * 2 + [[(21, 32), (32, 16), (16, 16)], [(16, 32), (32, 16), (16, 16)]]
        if [dims(m) for m in gnn] != want or any(acts(m) != [nn.LeakyReLU, nn.LeakyReLU] for m in gnn):
            return False
        pol = [self.stage_policy_network.mlp_score, self.exec_policy_network.mlp_score]
        return [dims(m) for m in pol] == [[(53, 64), (64, 64), (64, 1)], [(36, 64), (64, 64), (64, 1)]] and all(acts(m) == [nn.Tanh, nn.Tanh] for m in pol)

    def _packed_weights(self) -> dict[str, torch.Tensor]:
        # re-pack when the parameters moved: optimiser steps bump every parameter's version together, so
        # the first and the last one stand for all (update_parameters / load_state_dict also invalidate)
        ps = getattr(self, "_plist", None)
        if ps is None:
            ps = self._plist = list(self.parameters())
        ver = (ps[0]._version, ps[-1]._version, ps[0].device)
        if getattr(self, "_packed", None) is None or self._packed[0] != ver:
            def pack(mlp):  # [W1, b1, W2^T, b2, W3, b3] (include/sss.h sss_gnn_launch)
                lin = [m for m in mlp if isinstance(m, nn.Linear)]
                parts = [lin[0].weight, lin[0].bias, lin[1].weight.t(), lin[1].bias, lin[2].weight, lin[2].bias]
                return torch.cat([t.detach().float().contiguous().reshape(-1) for t in parts]).contiguous()
            def pack16(mlp):  # the same MLP (IN -> H1 -> 16 -> 16) as the 16-lanes-per-row image of csrc/sss_gnn16.h
                lin = [m for m in mlp if isinstance(m, nn.Linear)]
                W1, b1, W2, b2, W3, b3 = (t.detach().float() for t in (lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias, lin[2].weight, lin[2].bias))
                h1, n_in = W1.shape
                h2 = W2.shape[0]
                q1, q2 = h1 // 16, h2 // 16
                parts = [W1.reshape(q1, 16, n_in).permute(2, 1, 0),            # w1[i][g][q] = W1[g + 16 q][i]
                         b1.reshape(q1, 16).t(),                                 # b1[g][q]
                         W2.reshape(q2, 16, q1, 16).permute(3, 2, 1, 0),       # w2[jj][q][g][r] = W2[g + 16 r][jj + 16 q]
                         b2.reshape(q2, 16).t(),                                 # b2[g][r]
                         W3.t() if W3.shape[0] == 16 else W3.reshape(q2, 16).t(),   # w3[k][g] = W3[g][k]  /  one output: w3[g][r] = W3[0][g + 16 r]
                         b3]
                flat = torch.cat([t.contiguous().reshape(-1) for t in parts])
                return torch.nn.functional.pad(flat, (0, (-flat.numel()) % 4)).contiguous()
            def head_image(mlp, cols):  # a policy head (IN -> 64 -> 64 -> 1) as the A-operand images of csrc/sss_gnn_mfma.h MfmaHead
                lin = [m for m in mlp if isinstance(m, nn.Linear)]
                W1, b1, W2, b2 = (t.detach().float() for t in (lin[0].weight, lin[0].bias, lin[1].weight, lin[1].bias))
                dev = W1.device
                lane = torch.arange(64, device=dev)
                i, q = lane & 15, lane >> 4
                col = torch.as_tensor(cols, dtype=torch.long, device=dev)  # [U, 16]: column of W1 feature f of segment u multiplies, -1 = padding
                U = col.shape[0]
                step = torch.arange(4 * U, device=dev)
                c = col[(step >> 2)[:, None], 4 * q[None, :] + (step & 3)[:, None]]  # [NS, 64]
                rows = (16 * torch.arange(4, device=dev))[:, None, None] + i[None, None, :]  # [4, 1, 64]
                a1 = torch.where(c[None] >= 0, W1[rows.expand(4, 4 * U, 64), c.clamp(min=0)[None].expand(4, 4 * U, 64)], torch.zeros((), device=dev))
                s16 = torch.arange(16, device=dev)
                n_in = 16 * (s16 >> 2)[:, None] + 4 * q[None, :] + (s16 & 3)[:, None]  # [16, 64]
                a2 = W2[rows.expand(4, 16, 64), n_in[None].expand(4, 16, 64)]
                flat = torch.cat([a1.reshape(-1), a2.reshape(-1), b1, b2])
                assert flat.numel() % 4 == 0
                return flat.contiguous()
            f16 = list(range(16))
            stage_cols = [[NUM_NODE_FEATURES + f for f in f16], [NUM_NODE_FEATURES + 16 + f for f in f16], [NUM_NODE_FEATURES + 32 + f for f in f16],
                          [f if f < NUM_NODE_FEATURES else -1 for f in f16]]
            exec_cols = [[NUM_DAG_FEATURES + f for f in f16], [NUM_DAG_FEATURES + 16 + f for f in f16],
                         [f if f < NUM_DAG_FEATURES else (NUM_DAG_FEATURES + 32 if f == NUM_DAG_FEATURES else -1) for f in f16]]
            enc = self.encoder
            w = {"msg16": pack16(enc.node_encoder.mlp_msg), "update16": pack16(enc.node_encoder.mlp_update),
                 "stage16": pack16(self.stage_policy_network.mlp_score), "exec16": pack16(self.exec_policy_network.mlp_score),
                 "stage_mfma": head_image(self.stage_policy_network.mlp_score, stage_cols), "exec_mfma": head_image(self.exec_policy_network.mlp_score, exec_cols),
                 "prep": pack(enc.node_encoder.mlp_prep), "msg": pack(enc.node_encoder.mlp_msg), "update": pack(enc.node_encoder.mlp_update),
                 "dag": pack(enc.dag_encoder.mlp), "glob": pack(enc.global_encoder.mlp),
                 "stage": pack(self.stage_policy_network.mlp_score), "exec": pack(self.exec_policy_network.mlp_score)}
            slope = float(enc.node_encoder.mlp_prep[1].negative_slope)
            self._packed = (ver, w, slope)
        return self._packed[1]

    def invalidate_kernel_weights(self) -> None:
        """call after changing parameters by other means than `update_parameters` / `load_state_dict`"""
        self._packed = None
        self._plist = None

    def load_state_dict(self, *args, **kwargs):
        self._packed = None
        return super().load_state_dict(*args, **kwargs)

    def _apply(self, fn, *args, **kwargs):  # .to() / .cuda() / .float(): new parameter tensors
        self._packed = None
        self._plist = None
        return super()._apply(fn, *args, **kwargs)

    def _launch(self, kind: str, n_rows: int, w: torch.Tensor, layer: int = 0, n_pad: int = 0, n_rows_dev: torch.Tensor | None = None, _stream: int | None = None,
                **ptrs) -> None:
        """`n_rows_dev`: i64[1] on the device holding the real row count (`n_rows` is then a grid-size guess). `_stream`: the stream
        handle, when the caller has looked it up already (and made `w`'s device current).

        Filling a ctypes structure of 30 pointers costs more host time than the launch itself, and an inference loop passes the
        same buffers step after step: the structures are kept, keyed by the pointer values (another buffer anywhere -> another
        structure), and only the row count is written per call."""
        import ctypes

        from .binding import GNN_KINDS, SssGnnArgs, device_of
        names = tuple(ptrs)
        vals = tuple(t.data_ptr() if t is not None and t.numel() else None for t in ptrs.values())
        key = (kind, layer, n_pad, w.data_ptr(), n_rows_dev.data_ptr() if n_rows_dev is not None else None, self._packed[2], names, vals)
        cache = self.__dict__.setdefault("_launch_args", {})
        a = cache.get(key)
        if a is None:
            if len(cache) > 256:  # (graphs that come and go, e.g. training minibatches: do not grow without bound)
                cache.clear()
            a = cache[key] = SssGnnArgs()
            a.w_dev, a.slope, a.num_executors, a.layer, a.n_pad = w.data_ptr(), self._packed[2], self.num_executors, layer, n_pad
            a.n_rows_dev = key[4]
            for k, v in zip(names, vals):
                setattr(a, k + "_dev", v)
        a.n_rows = int(n_rows)
        if _stream is not None:
            self._kb.check(self._kb.lib.sss_gnn_launch(GNN_KINDS[kind], ctypes.byref(a), _stream))
            return
        dev = w.device
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        with device_of(dev):  # (no handle in this entry point: it launches on the current device)
            self._kb.check(self._kb.lib.sss_gnn_launch(GNN_KINDS[kind], ctypes.byref(a), stream))

    def _use_kernels(self, g: dict[str, Any] | None = None) -> bool:
        if getattr(self, "_kb", None) is None or (g is not None and "out_start" not in g):
            return False  # the kernels walk the ranges only the graph kernel (env.decima_graph) provides
        if getattr(self, "_arch_ok", None) is None:
            self._arch_ok = self._kernel_arch_ok()
        return self._arch_ok

    @staticmethod
    def _index_list(mask: torch.Tensor) -> torch.Tensor:
        """indices of the set entries, padded with -1 to the mask's length (no device->host sync)"""
        return torch.nonzero_static(mask, size=mask.numel(), fill_value=-1)[:, 0]

    @torch.no_grad()
    def _encode_kernels(self, g: dict[str, Any], _stream: int | None = None) -> dict[str, torch.Tensor]:
        """`encode(g, per_obs_skip=True)` on the fused kernels. Re-packs the parameters if they changed
        since the last call (the other kernel stages of the same `act` reuse that packing)."""
        w = self._packed_weights()
        x = g["x"]
        dev = x.device
        M, J, B = x.shape[0], g["job_obs"].numel(), g["n_obs"]
        if "max_depth" in g and "layer_cnt" in g and 0 <= g["max_depth"] <= 32:
            return self._encode_one_call(g, w, _stream)
        h_init = torch.empty((M, 16), dtype=torch.float32, device=dev)
        self._launch("prep", M, w["prep"], x=x, out=h_init)
        h = torch.empty_like(h_init)
        self._launch("sink", M, w["update"], h_init=h_init, h=h, out_deg=g["out_deg"], obs_depth=g["obs_depth"], node_obs=g["node_obs"])
        from .vec_env import VecSparkSchedSimEnv
        lists = VecSparkSchedSimEnv.decima_layer_lists(g)  # the pass's device->host sync (sizes of the layer lists)
        tmp = torch.empty((max(M, J), 16), dtype=torch.float32, device=dev)
        # embeddings alternate between `h` and `tmp` per update (include/sss.h node_recv_dev): no COMMIT launch per
        # layer, one MERGE after the last one
        for lvl in range(len(lists) - 1, -1, -1):
            recv = lists[lvl]
            self._launch("layer", recv.numel(), w["msg"], layer=lvl, w2=w["update"], w16=w.get("msg16"), w2_16=w.get("update16"), h_init=h_init, h=h, tmp=tmp, idx0=recv, dst=g["dst"],
                         out_start=g["out_start"], out_deg=g["out_deg"], edge_layers=g["edge_layers"], node_recv=g["node_recv"])
        h_dag = torch.empty((J, 16), dtype=torch.float32, device=dev)
        # (DAGHID brings the embeddings that ended up in `tmp` home to `h` on the fly: the MERGE of include/sss.h)
        self._launch("daghid", M, w["dag"], x=x, h=h, tmp=tmp, node_recv=g["node_recv"] if len(lists) else None)
        self._launch("dagsum", J, w["dag"], tmp=tmp, h_dag=h_dag, job_first=g["job_first"], job_nodes=g["job_nodes"])
        h_glob = torch.empty((B, 16), dtype=torch.float32, device=dev)
        self._launch("globhid", J, w["glob"], h_dag=h_dag, tmp=tmp)
        self._launch("globsum", B, w["glob"], tmp=tmp, h_glob=h_glob, obs_job_off=g["obs_job_off"], obs_jobs=g["obs_jobs"])
        return {"node": h, "dag": h_dag, "glob": h_glob}

    @torch.no_grad()
    def _encode_one_call(self, g: dict[str, Any], w: dict[str, torch.Tensor], _stream: int | None = None) -> dict[str, torch.Tensor]:
        """the encoder through `sss_gnn_encode` (include/sss.h): every launch of the pass enqueued by one call, the layers'
        list sizes stay on the device - no device->host round trip between the graph kernel and the scores"""
        import ctypes

        from .binding import SssGnnEncodeArgs, device_of
        from .vec_env import LateHint  # (list sizes unknown until the first pass's lengths have come back: -1)
        x = g["x"]
        dev = x.device
        M, J, B, D = x.shape[0], g["job_obs"].numel(), g["n_obs"], int(g["max_depth"])
        on_dev = "totals_dev" in g  # a capacity graph (env.decima_graph_on_device): M, J are capacities, the totals live on the device
        stream = _stream if _stream is not None else (torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0)
        f32 = lambda n: torch.empty((n, 16), dtype=torch.float32, device=dev)  # noqa: E731
        if on_dev:  # work buffers at capacity, kept per stream (no allocation per step)
            hb = self.__dict__.setdefault("_enc_cap", {})
            key = (dev, stream, M, J, B)
            if key not in hb:
                hb.clear()
                hb[key] = (f32(M), f32(M), f32(max(M, J)), f32(J), f32(B))
            h_init, h, tmp, h_dag, h_glob = hb[key]
        else:
            h_init, h, tmp, h_dag, h_glob = f32(M), f32(M), f32(max(M, J)), f32(J), f32(B)
        pool = self.__dict__.setdefault("_enc_scratch", {})  # one set of work buffers per stream: passes on different streams overlap
        sc = pool.get((dev, stream))
        need = 1 if on_dev else max(M * D, 1)  # (a capacity graph always comes with the graph kernel's own lists)
        if sc is None or sc["recv"].numel() < need or sc["env_off"].numel() < 32 * B:
            sc = pool[(dev, stream)] = {"recv": torch.empty(max(2 * need, 1 << 16), dtype=torch.int64, device=dev),
                                      "env_off": torch.empty(32 * B, dtype=torch.int64, device=dev), "tot": torch.zeros(32, dtype=torch.int64, device=dev),
                                      "hint": LateHint(32, dev)}
        p = lambda t: t.data_ptr() if t is not None and t.numel() else None  # noqa: E731  (a batch without edges: dst / edge_layers empty)
        # the graph kernel's own lists, if this graph is still the last one its env built on this stream (else: scan + list kernel here)
        ls, epoch = g.get("_layer_lists", (None, -1))
        fresh = ls is not None and ls["epoch"] == epoch and ls["recv"].device == dev and ls["stride"] >= M
        tot_t, recv_t, stride = (g["layer_totals"], ls["recv"], ls["stride"]) if fresh else (sc["tot"], sc["recv"], 0)
        # (the structure is kept per set of buffers - an inference loop passes the same ones step after step; filling its 40
        # fields costs more host time than the launches it describes)
        mode = int(getattr(self, "_layers_mode", 0))  # (include/sss.h sss_gnn_encode_args.layers_mode; 0: the library chooses)
        key = (M, J, B, D, mode, id(w), x.data_ptr(), g["out_deg"].data_ptr(), g["dst"].data_ptr() if g["dst"].numel() else 0, h.data_ptr(), tot_t.data_ptr(), recv_t.data_ptr(),
               recv_t.numel(), stride, g["obs_nodes"].data_ptr(), g["totals_dev"].data_ptr() if on_dev else 0)
        # (graphs with exact sizes bring new buffers every call: nothing to keep, and nothing kept alive; a capacity graph alternates
        # between two sets of list counters: two structures)
        memo = sc.setdefault("args", {}) if on_dev else None
        kept = memo.get(key) if on_dev else None
        if kept is None:
            a = SssGnnEncodeArgs(M, J, B, D, float(self._packed[2]), mode, p(w["prep"]), p(w["update"]), p(w["msg"]), p(w["dag"]), p(w["glob"]), p(w.get("msg16")), p(w.get("update16")),
                                 p(x), p(g["out_deg"]), p(g["obs_depth"]), p(g["node_obs"]), p(g["dst"]), p(g["out_start"]), p(g["edge_layers"]), p(g["node_recv"]),
                                 p(g["job_first"]), p(g["job_nodes"]), p(g["obs_job_off"]), p(g["obs_jobs"]), p(g["obs_node_off"]), p(g["obs_nodes"]), p(g["layer_cnt"]),
                                 p(h_init), p(h), p(tmp), p(h_dag), p(h_glob), p(sc["env_off"]), p(tot_t), p(recv_t), recv_t.numel(), stride,
                                 (ctypes.c_int64 * 32)(),
                                 g["totals_dev"][0:1].data_ptr() if on_dev else None, g["totals_dev"][2:3].data_ptr() if on_dev else None, 0, 0, 0)
            kept = (key, a, (w, x, h_init, h, tmp, h_dag, h_glob))  # (the tensors: kept alive with the pointers)
            if on_dev:
                if len(memo) >= 4:
                    memo.clear()
                memo[key] = kept
        a = kept[1]
        # the list sizes of an EARLIER pass, copied back without waiting (the latest that have arrived: they only size grids);
        # the graph kernel's lists come in pieces per block of envs (i64[32][32] lengths: a layer's rows are the sum over its pieces)
        own_hint = not (fresh and "layer_hint" in g)  # (a capacity graph brings the piece lengths with its totals: one read-back for both)
        if fresh and own_hint and "hint_pieces" not in sc:
            sc["hint_pieces"] = LateHint(33 * 32, dev)
        hint = sc.get("hint_pieces") if fresh else sc["hint"]
        hv = hint.read() if own_hint else g["layer_hint"]
        if fresh and hv[0] >= 0:  # (i64[33][32]: piece lengths per layer and block of envs; row 32: the blocks' largest observation)
            # (a capacity graph's layer launches keep the grid their node capacity gives them: sized by the read-back lengths - a few
            # steps old - the Decima step at 4096 envs was 14 us SLOWER, profiles/r05_hints.txt; the read-back decides the layers' mode)
            a.layer_rows_hint[:] = _NO_HINT if on_dev else hv[:1024].view(32, 32).sum(1).tolist()
            a.max_obs_nodes_hint = int(hv[1024:].max())
        else:
            a.layer_rows_hint[:] = hv[:32].tolist()
            a.max_obs_nodes_hint = 0
        if on_dev:
            a.n_nodes_hint, a.n_jobs_hint = int(g["totals_hint"][0]), int(g["totals_hint"][2])
        if _stream is not None:
            self._kb.check(self._kb.lib.sss_gnn_encode(ctypes.byref(a), stream))
        else:
            with device_of(dev):
                self._kb.check(self._kb.lib.sss_gnn_encode(ctypes.byref(a), stream))
        if own_hint:
            hint.post(tot_t)
        return {"node": h, "dag": h_dag, "glob": h_glob}

    @torch.no_grad()
    def _stage_scores_kernels(self, g: dict[str, Any], h: dict[str, torch.Tensor], _stream: int | None = None, for_draw_only: bool = False) -> torch.Tensor:
        """f32[n_obs, n_pad] stage scores, -inf where the slot is not a schedulable stage.
        `for_draw_only`: the matrix goes to `_sample_kernels` and nowhere else - on a capacity graph the kept matrix is then refilled
        without the -inf pass over all of it (the draw skips the slots that are not schedulable stages by their rank; what an earlier
        pass left in them is never read)"""
        M = g["x"].shape[0]
        rows_dev = None
        if "totals_dev" in g:  # capacity graph: the number of schedulable nodes is on the device; the score matrix is kept and refilled
            ob = self.__dict__.setdefault("_score_cap", {})
            key = (g["x"].device, g["n_obs"], g["n_pad"])
            if key not in ob:
                ob.clear()
                ob[key] = torch.full((g["n_obs"], g["n_pad"]), float("-inf"), dtype=torch.float32, device=g["x"].device)
            out = ob[key] if for_draw_only else ob[key].fill_(float("-inf"))
            out._sss_not_cleared = bool(for_draw_only)  # (read by _sample_kernels: such a matrix must not be handed out as scores)
            hint = int(g["totals_hint"][3])
            rows, idx0, exact, rows_dev = (hint + hint // 4 + 64 if hint > 0 else g["sched_list"].numel()), g["sched_list"], 1, g["totals_dev"][3:4]
            rows = min(rows, g["sched_list"].numel())
        else:
            out = torch.full((g["n_obs"], g["n_pad"]), float("-inf"), dtype=torch.float32, device=g["x"].device)
            if "sched_list" in g:  # the graph kernel's list of the schedulable nodes: exactly the rows to score (layer=1: no padding)
                rows, idx0, exact = g["sched_list"].numel(), g["sched_list"], 1
            else:
                rows, idx0, exact = M, self._index_list(g["stage_mask"]), 0
        self._launch("stage", rows, self._packed[1]["stage"], layer=exact, n_rows_dev=rows_dev, _stream=_stream, w16=self._packed[1].get("stage16"), w2_16=self._packed[1].get("stage_mfma"), n_pad=g["n_pad"], x=g["x"], h=h["node"], h_dag=h["dag"],
                     h_glob=h["glob"], out=out, idx0=idx0, node_job=g["node_job"], node_obs=g["node_obs"], node_loc=g["node_loc"])
        return out

    @torch.no_grad()
    def _sample_kernels(self, g: dict[str, Any], h: dict[str, torch.Tensor], padded: torch.Tensor, generator: torch.Generator | None,
                        scores_out: dict | None = None, _stream: int | None = None) -> dict[str, torch.Tensor]:
        """both draws on the device (include/sss.h sss_decima_sample): stage draw -> executor scores
        of the chosen stage's job -> executor-count draw; Gumbel-max over a counter-based stream
        (seed = the generator's, counter = number of calls so far).
        On a capacity graph (`env.decima_graph_on_device`: act-and-forget inference, its buffers are overwritten by the next
        call anyway) the result tensors and the argument structure are kept and reused from call to call as well."""
        import ctypes

        from .binding import SssDecimaSampleArgs, device_of
        B, E, dev = g["n_obs"], self.num_executors, padded.device
        self._calls = getattr(self, "_calls", 0) + 1
        seed = (generator.initial_seed() if generator is not None else 0) & (2 ** 64 - 1)
        key = (B, E, g["n_pad"], padded.data_ptr(), g["obs_nodes"].data_ptr(), g["obs_node_off"].data_ptr(), g["obs_job_off"].data_ptr(), g["sched_rank"].data_ptr(),
               g["node_job"].data_ptr())
        kept = self.__dict__.get("_sample_ws") if "totals_dev" in g else None
        if kept is None or kept[0] != key:
            i64 = lambda: torch.empty(B, dtype=torch.int64, device=dev)  # noqa: E731
            out = {"stage_sel": i64(), "job_idx": i64(), "exec_sel": i64(), "lgprob": torch.empty(B, dtype=torch.float32, device=dev),
                   "any_stage": torch.empty(B, dtype=torch.bool, device=dev)}
            job_gid = i64()
            stage_idx = torch.empty(B, dtype=torch.int32, device=dev)
            num_exec = torch.empty(B, dtype=torch.int32, device=dev)
            es = torch.empty((B, E), dtype=torch.float32, device=dev)
            a = SssDecimaSampleArgs(g["n_pad"], E, seed, self._calls,
                                    padded.data_ptr(), es.data_ptr(), g["obs_nodes"].data_ptr(), g["obs_node_off"].data_ptr(),
                                    g["obs_job_off"].data_ptr(), g["sched_rank"].data_ptr(), g["node_job"].data_ptr(), job_gid.data_ptr(),
                                    stage_idx.data_ptr(), num_exec.data_ptr(), out["stage_sel"].data_ptr(), out["job_idx"].data_ptr(),
                                    out["exec_sel"].data_ptr(), out["lgprob"].data_ptr(), out["any_stage"].data_ptr())
            kept = (key, out, job_gid, stage_idx, num_exec, es, a)
            if "totals_dev" in g:
                self._sample_ws = kept
        _, out, job_gid, stage_idx, num_exec, es, a = kept
        out = dict(out)
        a.rng_seed, a.rng_counter = seed, self._calls
        if _stream is not None:  # (the caller has made the device current)
            stream = _stream
            self._kb.check(self._kb.lib.sss_decima_sample(B, 0, ctypes.byref(a), stream))
        else:
            stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
            with device_of(dev):
                self._kb.check(self._kb.lib.sss_decima_sample(B, 0, ctypes.byref(a), stream))
        self._launch("exec", B * E, self._packed[1]["exec"], _stream=_stream, w16=self._packed[1].get("exec16"), w2_16=self._packed[1].get("exec_mfma"), x=g["x"], h_dag=h["dag"], h_glob=h["glob"], out=es,
                     idx0=job_gid, job_obs=g["job_obs"], job_first=g["job_first"], job_cap=g["job_cap"])
        if _stream is not None:
            self._kb.check(self._kb.lib.sss_decima_sample(B, 1, ctypes.byref(a), stream))
        else:
            with device_of(dev):
                self._kb.check(self._kb.lib.sss_decima_sample(B, 1, ctypes.byref(a), stream))
        out["env_stage_idx"], out["env_num_exec"] = stage_idx, num_exec
        out["rng"] = (a.rng_seed, a.rng_counter)
        if scores_out is not None:
            # (a capacity graph's kept score matrix is refilled without its -inf pass when it only feeds the draw - `for_draw_only` -
            # and then holds stale finite scores in slots that are not schedulable stages: never hand that one out)
            assert not getattr(padded, "_sss_not_cleared", False), "stage scores were computed for the draw only (stale entries outside the schedulable stages)"
            scores_out["stage_scores"], scores_out["exec_scores"] = padded, es
        return out

    @torch.no_grad()
    def _exec_scores_kernels(self, g: dict[str, Any], h: dict[str, torch.Tensor], job_gid: torch.Tensor) -> torch.Tensor:
        k, E = job_gid.numel(), self.num_executors
        out = torch.empty((k, E), dtype=torch.float32, device=job_gid.device)
        self._launch("exec", k * E, self._packed[1]["exec"], w16=self._packed[1].get("exec16"), w2_16=self._packed[1].get("exec_mfma"), x=g["x"], h_dag=h["dag"], h_glob=h["glob"], out=out,
                     idx0=job_gid.contiguous(), job_obs=g["job_obs"], job_first=g["job_first"], job_cap=g["job_cap"])
        return out

    def encode(self, g: dict[str, Any], per_obs_skip: bool = True) -> dict[str, torch.Tensor]:
        h_node = self.encoder.node_encoder(g, per_obs_skip)
        h_dag = self.encoder.dag_encoder(h_node, g)
        h_glob = self.encoder.global_encoder(h_dag, g)
        return {"node": h_node, "dag": h_dag, "glob": h_glob}

    def stage_scores(self, g: dict[str, Any], h: dict[str, torch.Tensor]):
        """scores of the schedulable stages only (scheduler.py:289-318): (f32[S], global node ids i64[S])"""
        from .train_kernels import concat_rows
        idx = g["stage_mask"].nonzero(as_tuple=True)[0]
        inp = concat_rows([(g["x"], idx), (h["node"], idx), (h["dag"], g["node_job"][idx]), (h["glob"], g["node_obs"][idx])])
        return self.stage_policy_network.mlp_score(inp).squeeze(-1), idx

    def exec_scores(self, g: dict[str, Any], h: dict[str, torch.Tensor], job_gid: torch.Tensor) -> torch.Tensor:
        """f32[k,E] for jobs `job_gid` (global job ids, i64[k]); entry e scores "e+1 executors"; -inf
        where that count is not allowed for the job (scheduler.py:337-385).

        The reference evaluates the network on all k x E (job, count) pairs and masks afterwards. Only the allowed pairs
        (count <= the job's cap) reach the softmax or carry a gradient, so only those rows are built and evaluated here - at
        BASELINE config 5 a job allows 7 of 50 counts on average, and the k x E form was a fifth of a PPO update's device time
        (profiles/r04_ppo.md). A row's score does not depend on the other rows of the call: the values are the same."""
        s, owner, count, caps = self.exec_score_rows(g, h, job_gid)
        return torch.full((job_gid.numel(), self.num_executors), float("-inf"), dtype=s.dtype, device=s.device).index_put((owner, count), s)

    def exec_score_rows(self, g: dict[str, Any], h: dict[str, torch.Tensor], job_gid: torch.Tensor):
        """the allowed (job, count) pairs of `exec_scores` as flat rows, job after job: (scores f32[rows], owner i64[rows] = index into
        job_gid, count i64[rows] = executor count - 1, caps i64[k] = rows per job)"""
        from .train_kernels import concat_rows
        E = self.num_executors
        base = torch.cat([g["x"][g["job_first"][job_gid], :NUM_DAG_FEATURES], h["dag"].index_select(0, job_gid), h["glob"].index_select(0, g["job_obs"][job_gid])], -1)
        k, dev = base.shape[0], base.device
        acts = (torch.arange(E, device=dev) / E).to(base.dtype)
        caps = g["job_cap"][job_gid].clamp(min=0, max=E)
        total = int(caps.sum())  # (device -> host: the number of rows)
        owner = torch.repeat_interleave(torch.arange(k, device=dev), caps, output_size=total)
        count = torch.arange(total, device=dev) - (torch.cumsum(caps, 0) - caps)[owner]
        inp = concat_rows([(base, owner), (acts[:, None], count)])
        return self.exec_policy_network.mlp_score(inp).squeeze(-1), owner, count, caps

    @torch.no_grad()
    def act(self, g: dict[str, Any], generator: torch.Generator | None = None, greedy: bool = False, fresh_outputs: bool = False) -> dict[str, torch.Tensor]:
        """samples one Decima action per observation of a compact graph (scheduler.py:71-99): a stage
        from softmax(stage scores), then an executor count from softmax(exec scores of that stage's
        job). Returns the reference's action tuple entries `stage_sel` (index among the observation's
        schedulable stages = the env's `stage_idx`), `job_idx` (job slot), `exec_sel` (executor
        count - 1), `lgprob`, and `any_stage` (False where nothing is schedulable; the other entries
        are then meaningless).

        OWNERSHIP of the result: on a capacity graph (`env.decima_graph_on_device()`, which `schedule_env` uses by default) the
        returned tensors - `stage_sel`, `job_idx`, `exec_sel`, `lgprob`, `any_stage`, `env_stage_idx`, `env_num_exec` - are work space
        the policy keeps and the NEXT call on a graph of the same shape overwrites, like the observation buffers `env.step`
        returns. Consume them before the next call (the in-tree collector copies them on the stream) or pass
        `fresh_outputs=True` to get copies you own. Graphs with exact sizes (`env.decima_graph()`) return fresh tensors.
        `greedy`: the most probable stage and executor count instead of a draw (evaluation; tensor-op path)."""
        B, N = g["n_obs"], g["n_pad"]
        M, J = g["x"].shape[0], g["job_obs"].numel()
        if self._use_kernels(g) and M > 0 and J > 0 and not greedy:
            # (the stream handle and the current device are looked up once for the pass's seven library calls)
            from .binding import device_of
            dev = g["x"].device
            with device_of(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
                h = self._encode_kernels(g, stream)
                out = self._sample_kernels(g, h, self._stage_scores_kernels(g, h, stream, for_draw_only=True), generator, _stream=stream)
            return {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in out.items()} if fresh_outputs else out
        # tensor-op path (other architectures, graphs built without the graph kernel)
        h = self.encode(g)
        s, idx = self.stage_scores(g, h)
        padded = torch.full((B, N), float("-inf"), dtype=s.dtype, device=s.device)
        padded[g["node_obs"][idx], g["node_loc"][idx]] = s
        any_stage = torch.isfinite(padded).any(1)
        p = torch.softmax(torch.where(any_stage[:, None], padded, torch.zeros_like(padded)), 1)
        col = p.argmax(1) if greedy else torch.multinomial(p, 1, generator=generator)[:, 0]
        node = (_excl_cumsum(g["obs_nodes"]) + col).clamp(max=max(M - 1, 0))
        if M == 0:
            z = torch.zeros(B, dtype=torch.long, device=padded.device)
            return {"stage_sel": z, "job_idx": z, "exec_sel": z, "lgprob": torch.zeros(B, device=padded.device), "any_stage": any_stage}
        stage_sel = g["sched_rank"][node]
        job_gid = g["node_job"][node]
        job_slot = job_gid - _excl_cumsum(g["obs_jobs"])
        es = self.exec_scores(g, h, job_gid.clamp(min=0, max=max(J - 1, 0)))
        any_exec = torch.isfinite(es).any(1) & any_stage
        pe = torch.softmax(torch.where(any_exec[:, None], es, torch.zeros_like(es)), 1)
        k = pe.argmax(1) if greedy else torch.multinomial(pe, 1, generator=generator)[:, 0]
        lg = torch.log(p.gather(1, col[:, None])[:, 0]) + torch.log(pe.gather(1, k[:, None])[:, 0])
        return {"stage_sel": stage_sel, "job_idx": job_slot, "exec_sel": k, "lgprob": lg, "any_stage": any_stage}

    @torch.no_grad()
    def act_env(self, env, counter: int, seed: int = 0, active: torch.Tensor | None = None, want_scores: bool = False,
                want_prof: bool = False):
        """Decima's decision for every env of a `VecSparkSchedSimEnv` in ONE kernel launch
        (include/sss.h sss_decima_policy): transform, GNN, scores and both draws per env inside one
        wavefront, no intermediate graph, no host sync. Slower than the row-parallel pipeline at large
        batches (see `schedule_env`); useful when launches / syncs dominate. Returns (actions for `env.step`, the `act`
        dict [+ "stage_scores" f32[B,node_cap], "exec_scores" f32[B,E] with `want_scores`]). The draws
        are a deterministic function of (seed, counter, env): pass a new `counter` every step."""
        import ctypes

        from .binding import SssDecimaPolicyArgs
        if getattr(self, "_kb", None) is None:
            self.bind_kernels(env._b)
        assert self._use_kernels(), "the fused policy kernel supports the published Decima architecture only"
        w = self._packed_weights()
        B, dev, d = env.num_envs, env.device, env.dims
        ws = getattr(self, "_ws", None)
        if ws is None or ws["key"] != (B, d.node_cap, d.job_cap, str(dev)):
            i32 = lambda: torch.empty(B, dtype=torch.int32, device=dev)  # noqa: E731
            ws = {"key": (B, d.node_cap, d.job_cap, str(dev)),
                  "node": torch.empty((B, d.node_cap, 53), dtype=torch.float32, device=dev),
                  "job": torch.empty((B, d.job_cap, 32), dtype=torch.float32, device=dev),
                  "stage_idx": i32(), "num_exec": i32(), "stage_sel": i32(), "job_idx": i32(), "exec_sel": i32(),
                  "lgprob": torch.empty(B, dtype=torch.float32, device=dev)}
            self._ws = ws
        out = {}
        if want_scores:
            out["stage_scores"] = torch.full((B, d.node_cap), float("-inf"), dtype=torch.float32, device=dev)
            out["exec_scores"] = torch.full((B, self.num_executors), float("-inf"), dtype=torch.float32, device=dev)
        from .vec_env import _mask_u8
        act8 = _mask_u8(active)
        a = SssDecimaPolicyArgs(act8.data_ptr() if act8 is not None else None, 200.0, 1e5, self._packed[2],
                                w["prep"].data_ptr(), w["msg"].data_ptr(), w["update"].data_ptr(), w["dag"].data_ptr(), w["glob"].data_ptr(),
                                w["stage"].data_ptr(), w["exec"].data_ptr(), ws["node"].data_ptr(), ws["job"].data_ptr(),
                                int(seed) & (2 ** 64 - 1), int(counter) & (2 ** 64 - 1), ws["stage_idx"].data_ptr(), ws["num_exec"].data_ptr(),
                                ws["stage_sel"].data_ptr(), ws["job_idx"].data_ptr(), ws["exec_sel"].data_ptr(), ws["lgprob"].data_ptr(),
                                out["stage_scores"].data_ptr() if want_scores else None, out["exec_scores"].data_ptr() if want_scores else None,
                                None)
        if want_prof:  # shader cycles per phase: analysis, prep, layers, summaries, stage, exec; then depth, nodes
            out["prof"] = torch.zeros((B, 8), dtype=torch.int64, device=dev)
            a.prof_dev = out["prof"].data_ptr()
        self._kb.check(self._kb.lib.sss_decima_policy(env._h, ctypes.byref(a), env._stream()))
        out.update(stage_sel=ws["stage_sel"].long(), job_idx=ws["job_idx"].long(), exec_sel=ws["exec_sel"].long(), lgprob=ws["lgprob"],
                   any_stage=ws["stage_idx"] >= 0)
        return {"stage_idx": ws["stage_idx"], "num_exec": ws["num_exec"]}, out

    @staticmethod
    def env_actions(a: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        """`act`'s result in the env's action format (DecimaActWrapper.action, env_wrapper.py:33-34);
        envs without a schedulable stage get stage_idx -1"""
        if "env_stage_idx" in a:  # written by the sampling kernels
            return {"stage_idx": a["env_stage_idx"], "num_exec": a["env_num_exec"]}
        stage_idx = torch.where(a["any_stage"], a["stage_sel"], torch.full_like(a["stage_sel"], -1))
        return {"stage_idx": stage_idx.to(torch.int32), "num_exec": (1 + a["exec_sel"]).to(torch.int32)}

    @torch.no_grad()
    def schedule_env(self, env, generator: torch.Generator | None = None, active: torch.Tensor | None = None, one_launch: bool = False,
                     host_sync: bool | None = None, greedy: bool = False, fresh_outputs: bool = False):
        """Decima in the loop on a `VecSparkSchedSimEnv`: one sampled action per env. Default: the
        graph kernel + the row-parallel GNN kernels + `act` (rows of ALL envs share every launch, so
        the lanes stay full). `one_launch=True` uses the per-env policy kernel (`act_env`; its draw
        counter advances on every call, `generator` only supplies the seed): no host sync and one
        launch, but each env's phases run serially inside one wavefront - on one MI355X it ties the
        pipeline up to ~1024 envs and is 2x slower at 4096 (profiles/r01_decima.md).
        Returns (actions for `env.step`, the `act` dict).

        OWNERSHIP: by default both are views of work space that the next `schedule_env` / `act` call on this policy overwrites (see
        `act`): hand the actions to `env.step` and read `aux["lgprob"]` etc. before calling again, or pass `fresh_outputs=True` for
        tensors you may keep (a rollout recorder of your own, logging). `greedy`: arg-max actions instead of sampled ones."""
        if getattr(self, "_kb", None) is None:
            self.bind_kernels(env._b)
        if greedy:
            a = self.act(env.decima_graph(active), generator, greedy=True)
            return self.env_actions(a), a
        if one_launch:
            self._calls = getattr(self, "_calls", 0) + 1
            return self.act_env(env, self._calls, seed=generator.initial_seed() if generator is not None else 0, active=active)
        # no device->host round trip when the graph kernel and the GNN kernels can do the whole step (the graph's totals stay on
        # the device); else the graph with exact sizes (one read-back of its totals)
        on_dev = host_sync is False or (host_sync is None and self._use_kernels() and env.graph_kernel_fits)
        a = self.act(env.decima_graph_on_device(active) if on_dev else env.decima_graph(active, reuse_buffers=True), generator, fresh_outputs=fresh_outputs)
        return self.env_actions(a), a

    @torch.no_grad()
    def schedule_batch(self, obs, max_depth: int, generator: torch.Generator | None = None):
        """the same from a `BatchedObs` alone, with the observation transform done by tensor ops
        (`decima_observation` + `compact_graph`)"""
        a = self.act(compact_graph(decima_observation(obs, self.num_executors, max_depth)), generator)
        return self.env_actions(a), a

    def evaluate_actions(self, g: dict[str, Any], stage_sel: torch.Tensor, job_idx: torch.Tensor, exec_sel: torch.Tensor):
        """log-probabilities and normalised entropies of recorded actions under the current
        parameters, with gradients (scheduler.py:101-139): `stage_sel` = index among the
        observation's schedulable stages, `job_idx` = job slot inside the observation, `exec_sel` =
        index among the allowed executor counts. Returns {"lgprobs": f32[n_obs], "entropies": f32[n_obs]}."""
        n_obs = g["n_obs"]
        h = self.encode(g, per_obs_skip=False)
        s, idx = self.stage_scores(g, h)
        owner = g["node_obs"][idx]
        job_gid = _excl_cumsum(g["obs_jobs"]) + job_idx
        norm = (self.num_executors * g["obs_nodes"]).to(s.dtype).log()
        from . import train_kernels as tk
        if tk.SEGMENT_CATEGORICAL and s.is_cuda and s.dtype == torch.float32 and s.numel() >= tk.MIN_ROWS:
            # one launch per softmax and pass (csrc/sss_segcat.h): the schedulable stages of an observation are consecutive rows of s (idx
            # ascends), the allowed executor counts of its job consecutive rows of the flat executor scores
            from .train_kernels import segment_offsets
            stage_lg, stage_ent = tk.segment_categorical(s, segment_offsets(owner, n_obs), stage_sel, 1e-16)
            se, _, _, caps = self.exec_score_rows(g, h, job_gid)
            ptr_e = torch.zeros(caps.numel() + 1, dtype=torch.int64, device=s.device)
            ptr_e[1:] = torch.cumsum(caps, 0)
            exec_lg, exec_ent = tk.segment_categorical(se, ptr_e, exec_sel, 0.0)
            return {"lgprobs": stage_lg + exec_lg, "entropies": (stage_ent + exec_ent) / norm}
        p, lp = _segment_log_softmax(s, owner, n_obs)
        n_acts = torch.zeros(n_obs, dtype=torch.long, device=s.device).index_add_(0, owner, torch.ones_like(owner))
        stage_lg = lp[_excl_cumsum(n_acts) + stage_sel]
        stage_ent = -torch.zeros(n_obs, dtype=s.dtype, device=s.device).index_add_(0, owner, lp * p)
        es = self.exec_scores(g, h, job_gid)
        allowed = torch.isfinite(es)
        pe = torch.softmax(es, 1)
        eps = torch.finfo(pe.dtype).eps
        pe = torch.where(allowed, pe.clamp(min=eps, max=1 - eps), torch.ones_like(pe))
        lpe = pe.log()
        exec_lg = lpe.gather(1, exec_sel[:, None])[:, 0]
        exec_ent = -(lpe * pe * allowed).sum(1)
        return {"lgprobs": stage_lg + exec_lg, "entropies": (stage_ent + exec_ent) / norm}

    def update_parameters(self, loss: torch.Tensor | None = None) -> None:
        """scheduler.py (TrainableScheduler) :37-54: backward, clip, optimiser step, zero grads"""
        assert self.optim
        if loss is not None and bool(loss != 0):  # the reference tests `if loss:` (scheduler.py:40)
            loss.backward()
        if self.max_grad_norm:
            torch.nn.utils.clip_grad_norm_(self.parameters(), self.max_grad_norm, error_if_nonfinite=True)
        self.optim.step()
        self.optim.zero_grad()
        self._packed = None  # the inference kernels re-pack the parameters on their next use


# ---- the reference's single-env plugin surface ---------------------------------------------------

class DecimaEnvWrapper:
    """`schedulers/decima/env_wrapper.py:12-34, 37-143` for the single-env facade
    (`spark_sched_sim_amd.SparkSchedSimEnv`): observations gain Decima's node features, `stage_mask`,
    `exec_mask` and `edge_masks`; actions `{"stage_idx", "job_idx", "num_exec"}` (num_exec 0-based)
    are translated back. The transform is the same `decima_observation` that serves the batched env,
    applied to the facade's one-env device tensors."""

    def __init__(self, env, num_tasks_scale: int = 200, work_scale: float = 1e5):
        self.env = env
        self.num_tasks_scale, self.work_scale = num_tasks_scale, work_scale

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def observation(self, obs: dict) -> dict:
        from .spaces import GraphInstance
        vec = self.env.unwrapped._vec
        f = decima_observation(vec._obs(), vec.num_executors, vec.dims.stage_stride, self.num_tasks_scale, self.work_scale, edge_masks=True)
        n, ne, a = int(f["n_nodes"][0]), int(f["n_edges"][0]), int(f["job_valid"][0].sum())
        return {
            "dag_batch": GraphInstance(f["x"][0, :n].cpu().numpy(), obs["dag_batch"].edges, obs["dag_batch"].edge_links),
            "dag_ptr": obs["dag_ptr"],
            "stage_mask": f["stage_mask"][0, :n].cpu().numpy(),
            "exec_mask": f["exec_mask"][0, :a].cpu().numpy(),
            "edge_masks": f["edge_masks"][:, 0, :ne].cpu().numpy(),
        }

    def reset(self, seed=None, options=None):
        obs, info = self.env.reset(seed=seed, options=options)
        return self.observation(obs), info

    def step(self, action: dict):
        obs, rew, term, trunc, info = self.env.step({"stage_idx": action["stage_idx"], "num_exec": 1 + action["num_exec"]})
        return self.observation(obs), rew, term, trunc, info

    def close(self):
        return self.env.close()


def graph_from_wrapped_obs(obs: dict, device=None) -> dict[str, Any]:
    """a one-observation compact graph from Decima's (wrapped) observation dict - what
    `utils.obs_to_pyg` builds for PyG (decima/utils.py:88-115)"""
    import numpy as np
    x = torch.as_tensor(np.asarray(obs["dag_batch"].nodes, dtype=np.float32), device=device)
    M = x.shape[0]
    ptr = torch.as_tensor(np.asarray(obs["dag_ptr"], dtype=np.int64), device=device)
    J = ptr.numel() - 1
    el = torch.as_tensor(np.asarray(obs["dag_batch"].edge_links, dtype=np.int64).reshape(-1, 2), device=device)
    sm = torch.as_tensor(np.asarray(obs["stage_mask"], dtype=bool), device=device)
    em_np = np.asarray(obs["edge_masks"], dtype=bool)
    em = torch.as_tensor(em_np.reshape(em_np.shape[0] if em_np.ndim == 2 else 0, el.shape[0]), device=device)
    z = torch.zeros(M, dtype=torch.long, device=x.device)
    node_job = torch.searchsorted(ptr[1:].contiguous(), torch.arange(M, device=x.device), right=True)
    rank = sm.long().cumsum(0) - 1
    layers = []
    for lvl in range(em.shape[0]):
        e = em[lvl].nonzero(as_tuple=True)[0]
        layers.append((e, torch.zeros(M, dtype=torch.bool, device=x.device).index_fill_(0, el[e, 0], True).nonzero(as_tuple=True)[0]))
    caps = torch.as_tensor(np.asarray(obs["exec_mask"], dtype=bool).reshape(J, -1).sum(1), device=x.device)
    return {"x": x, "node_obs": z, "node_loc": torch.arange(M, device=x.device), "n_pad": max(M, 1), "node_job": node_job,
            "sched_rank": torch.where(sm, rank, torch.full_like(rank, -1)), "gen": z, "stage_mask": sm, "src": el[:, 0], "dst": el[:, 1],
            "edge_obs": torch.zeros(el.shape[0], dtype=torch.long, device=x.device), "job_obs": torch.zeros(J, dtype=torch.long, device=x.device),
            "job_cap": caps, "job_first": ptr[:-1], "n_obs": 1, "obs_nodes": torch.tensor([M], device=x.device),
            "obs_jobs": torch.tensor([J], device=x.device), "obs_depth": torch.tensor([em.shape[0]], device=x.device), "layers": layers}


class DecimaScheduler(DecimaPolicy):
    """the reference's `DecimaScheduler` plugin (schedulers/decima/scheduler.py:16-99) for the
    single-env harness (`examples.run_episode`): `schedule(obs)` on one wrapped observation, actions
    drawn with `random.choices` over the softmax like `utils.sample` (decima/utils.py:19-23)."""

    env_wrapper_cls = DecimaEnvWrapper

    @torch.no_grad()
    def schedule(self, obs: dict) -> tuple[dict, dict]:
        import random

        import numpy as np

        def sample(logits: torch.Tensor):
            pi = torch.softmax(logits, 0).cpu().numpy()
            idx = random.choices(np.arange(pi.size), pi)[0]
            return int(idx), np.log(pi[idx])

        g = graph_from_wrapped_obs(obs, self.device)
        h = self.encode(g)
        s, idx = self.stage_scores(g, h)
        stage_idx, stage_lg = sample(s)
        job_idx = int(g["node_job"][idx[stage_idx]])
        es = self.exec_scores(g, h, torch.tensor([job_idx], device=s.device))[0]
        num_exec, exec_lg = sample(es[torch.isfinite(es)])
        return {"stage_idx": stage_idx, "job_idx": job_idx, "num_exec": num_exec}, {"lgprob": stage_lg + exec_lg}
