"""Decima on the batched env: the observation transform of the reference's `DecimaObsWrapper`
(reference schedulers/decima/env_wrapper.py:37-143, DAG-layer edge masks of decima/utils.py:238-267)
and the Decima GNN policy (decima/scheduler.py:16-385) as plain PyTorch over ALL envs of a
`VecSparkSchedSimEnv` at once - no torch_geometric / torch_sparse / torch_scatter, no host round
trip: the env's observation tensors are consumed where they are (device, padded per env).

This is the first "next" row of SURVEY 8(f); the simulator itself does not depend on it.

Layout: everything is padded per env to the env's capacities (N = node_cap, A = job_cap,
Ed = edge_cap); validity comes from `n_nodes / n_jobs / n_edges`. `DecimaPolicy`'s parameter names
match the reference's `DecimaScheduler.state_dict()` so its checkpoints load unchanged.
"""
from __future__ import annotations

from typing import Any

import torch
import torch.nn as nn

NUM_NODE_FEATURES = 5  # env_wrapper.py:9
NUM_DAG_FEATURES = 3   # scheduler.py:33


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


def compact_graph(f: dict[str, torch.Tensor]) -> dict[str, Any]:
    """the batch as ONE graph over the valid nodes only (what `collate_obsns` / PyG batching does in
    the reference, decima/utils.py:117-160): flat node features, global node / job ids, flat edge
    endpoints, and for every DAG layer the list of edges in that layer's mask. Padding never reaches
    the MLPs. Costs a handful of device->host syncs (sizes of the index lists)."""
    x = f["x"]
    B, N, _ = x.shape
    A = f["job_valid"].shape[1]
    env_n, loc_n = f["node_valid"].nonzero(as_tuple=True)
    off = torch.cumsum(f["n_nodes"], 0) - f["n_nodes"]
    env_e, loc_e = f["edge_valid"].nonzero(as_tuple=True)
    src = off[env_e] + f["edge_src"][env_e, loc_e]
    dst = off[env_e] + f["edge_dst"][env_e, loc_e]
    gen = f["gen"][env_n, loc_n]
    M = env_n.numel()
    depth = int(gen.max()) if M else 0
    layers = []
    for lvl in range(depth):
        in_m = gen == lvl
        hit = torch.zeros(M, dtype=torch.int32, device=x.device).index_add_(0, dst, in_m[src].to(torch.int32))
        in_m = in_m | (hit > 0)
        layers.append((in_m[src] & in_m[dst]).nonzero(as_tuple=True)[0])
    return {"x": x[env_n, loc_n], "env": env_n, "loc": loc_n, "job": env_n * A + f["node_job"][env_n, loc_n],
            "src": src, "dst": dst, "layers": layers, "has_mp": f["has_mp"][env_n], "B": B, "N": N, "A": A}


def make_mlp(input_dim: int, hid_dims: list[int], output_dim: int, act_cls: str, act_kwargs: dict[str, Any] | None = None) -> nn.Sequential:
    """Linear / activation stack with the reference's layer numbering (decima/utils.py:44-64)"""
    act = getattr(torch.nn.modules.activation, act_cls)
    kwargs = dict(act_kwargs or {})
    kwargs.pop("inplace", None)
    layers: list[nn.Module] = []
    prev = input_dim
    dims = list(hid_dims) + [output_dim]
    for i, d in enumerate(dims):
        layers.append(nn.Linear(prev, d))
        if i < len(dims) - 1:
            layers.append(act(**kwargs))
        prev = d
    return nn.Sequential(*layers)


class _NodeEncoder(nn.Module):
    def __init__(self, nf: int, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp_prep = make_mlp(nf, output_dim=emb, **mlp_kwargs)
        self.mlp_msg = make_mlp(emb, output_dim=emb, **mlp_kwargs)
        self.mlp_update = make_mlp(emb, output_dim=emb, **mlp_kwargs)

    def forward(self, g: dict[str, Any]) -> torch.Tensor:
        """child -> parent ("reverse flow") message passing one DAG layer at a time, deepest layer
        first (scheduler.py:192-236). Per layer only the edges of that layer are touched: messages
        are evaluated per edge (a child with two parents in the layer is evaluated twice - in-degree
        is small) and the update is evaluated per edge source and written back (duplicates write the
        same value)."""
        x, src, dst = g["x"], g["src"], g["dst"]
        M = x.shape[0]
        h_init = self.mlp_prep(x)
        if not g["layers"]:
            return h_init  # every observation is a single layer: mlp_prep only (scheduler.py:238-243)
        # nodes that are never the source end of an edge start from update(h_init), the rest from 0
        is_parent = torch.zeros(M, dtype=torch.bool, device=x.device).index_fill_(0, src, True)
        h = torch.where(is_parent[:, None], torch.zeros_like(h_init), self.mlp_update(h_init))
        for e in reversed(g["layers"]):
            s_e, d_e = src[e], dst[e]
            msg = self.mlp_msg(h[d_e])
            agg = torch.zeros_like(h_init).index_add_(0, s_e, msg)
            h = h.index_copy(0, s_e, h_init[s_e] + self.mlp_update(agg[s_e]))
        return torch.where(g["has_mp"][:, None], h, h_init)


class _DagEncoder(nn.Module):
    def __init__(self, nf: int, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp = make_mlp(nf + emb, output_dim=emb, **mlp_kwargs)

    def forward(self, h_node: torch.Tensor, g: dict[str, Any]) -> torch.Tensor:
        """per-job sums (scheduler.py:246-262), returned padded f32[B,A,emb] (zeros for padding)"""
        y = self.mlp(torch.cat([g["x"], h_node], -1))
        out = torch.zeros((g["B"] * g["A"], y.shape[-1]), dtype=y.dtype, device=y.device).index_add_(0, g["job"], y)
        return out.view(g["B"], g["A"], -1)


class _GlobalEncoder(nn.Module):
    def __init__(self, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp = make_mlp(emb, output_dim=emb, **mlp_kwargs)

    def forward(self, h_dag: torch.Tensor, job_valid: torch.Tensor) -> torch.Tensor:
        return (self.mlp(h_dag) * job_valid[..., None]).sum(1)


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


class DecimaPolicy(nn.Module):
    """the reference's Decima architecture (scheduler.py:16-99) with batched-over-envs inference"""

    def __init__(self, num_executors: int, embed_dim: int, gnn_mlp_kwargs: dict[str, Any], policy_mlp_kwargs: dict[str, Any],
                 state_dict_path: str | None = None, **_unused):
        super().__init__()
        self.name = "Decima"
        self.num_executors = num_executors
        self.encoder = _Encoder(NUM_NODE_FEATURES, embed_dim, gnn_mlp_kwargs)
        self.stage_policy_network = _ScoreNet(NUM_NODE_FEATURES + 3 * embed_dim, policy_mlp_kwargs)
        self.exec_policy_network = _ScoreNet(NUM_DAG_FEATURES + 2 * embed_dim + 1, policy_mlp_kwargs)
        for n_, p in self.named_parameters():  # scheduler.py:66-69
            if "bias" in n_:
                p.data.zero_()
        if state_dict_path:
            self.load_state_dict(torch.load(state_dict_path, map_location="cpu"))

    def encode(self, f: dict[str, torch.Tensor], g: dict[str, Any] | None = None) -> dict[str, Any]:
        g = g if g is not None else compact_graph(f)
        h_node = self.encoder.node_encoder(g)
        h_dag = self.encoder.dag_encoder(h_node, g)
        h_glob = self.encoder.global_encoder(h_dag, f["job_valid"])
        return {"node": h_node, "dag": h_dag, "glob": h_glob, "graph": g}

    def stage_scores(self, f: dict[str, torch.Tensor], h: dict[str, Any]) -> torch.Tensor:
        """f32[B,N]; -inf where the node is not a schedulable stage (scheduler.py:289-318)"""
        g = h["graph"]
        h_dag_n = h["dag"].view(g["B"] * g["A"], -1)[g["job"]]
        inp = torch.cat([g["x"], h["node"], h_dag_n, h["glob"][g["env"]]], -1)
        s = self.stage_policy_network.mlp_score(inp).squeeze(-1)
        out = torch.full((g["B"], g["N"]), float("-inf"), dtype=s.dtype, device=s.device)
        sm = f["stage_mask"][g["env"], g["loc"]]
        return out.index_put((g["env"][sm], g["loc"][sm]), s[sm])

    def exec_scores(self, f: dict[str, torch.Tensor], h: dict[str, torch.Tensor], job_idx: torch.Tensor) -> torch.Tensor:
        """f32[B,E]; -inf where the executor count is not allowed for the job (scheduler.py:337-385)"""
        B, N, _ = f["x"].shape
        E = self.num_executors
        start = f["dag_start"].gather(1, job_idx[:, None]).clamp(max=N - 1)
        x_dag = f["x"].gather(1, start[..., None].expand(-1, -1, NUM_NODE_FEATURES))[:, 0, :NUM_DAG_FEATURES]
        h_dag = h["dag"].gather(1, job_idx[:, None, None].expand(-1, -1, h["dag"].shape[-1]))[:, 0]
        base = torch.cat([x_dag, h_dag, h["glob"]], -1)
        acts = (torch.arange(E, device=base.device) / E).to(base.dtype)
        inp = torch.cat([base[:, None, :].expand(-1, E, -1), acts[None, :, None].expand(B, -1, -1)], -1)
        s = self.exec_policy_network.mlp_score(inp).squeeze(-1)
        mask = f["exec_mask"].gather(1, job_idx[:, None, None].expand(-1, -1, E))[:, 0]
        return torch.where(mask, s, torch.full_like(s, float("-inf")))

    @torch.no_grad()
    def schedule_batch(self, obs, max_depth: int, generator: torch.Generator | None = None):
        """one action per env: a stage sampled from softmax(stage scores), then an executor count
        sampled from softmax(exec scores of that stage's job) (scheduler.py:71-99). Returns
        ({"stage_idx": i32[B], "num_exec": i32[B]}, {"lgprob": f32[B], "job_idx": i64[B]})."""
        f = decima_observation(obs, self.num_executors, max_depth)
        h = self.encode(f)
        ss = self.stage_scores(f, h)
        any_stage = f["stage_mask"].any(1)
        ss_safe = torch.where(any_stage[:, None], ss, torch.zeros_like(ss))
        p = torch.softmax(ss_safe, 1)
        node = torch.multinomial(p, 1, generator=generator)[:, 0]
        stage_idx = f["stage_mask"].long().cumsum(1).gather(1, node[:, None])[:, 0] - 1
        job = f["node_job"].gather(1, node[:, None])[:, 0].clamp(max=f["job_valid"].shape[1] - 1)
        es = self.exec_scores(f, h, job)
        any_exec = torch.isfinite(es).any(1)
        es_safe = torch.where(any_exec[:, None], es, torch.zeros_like(es))
        pe = torch.softmax(es_safe, 1)
        k = torch.multinomial(pe, 1, generator=generator)[:, 0]
        lg = torch.log(p.gather(1, node[:, None])[:, 0]) + torch.log(pe.gather(1, k[:, None])[:, 0])
        stage_idx = torch.where(any_stage, stage_idx, torch.full_like(stage_idx, -1))
        return ({"stage_idx": stage_idx.to(torch.int32), "num_exec": (1 + k).to(torch.int32)},
                {"lgprob": lg, "job_idx": job})
