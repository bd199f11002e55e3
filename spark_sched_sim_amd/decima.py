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
                       levels: int | None = None) -> dict[str, torch.Tensor]:
    """`DecimaObsWrapper.observation` for every env of a `BatchedObs`.

    Returns (B = envs): x f32[B,N,5], node_valid / stage_mask bool[B,N], node_job i64[B,N]
    (job slot of each node, A for padding), job_valid bool[B,A], exec_mask bool[B,A,E],
    edge_src / edge_dst i64[B,Ed] (N for padding), edge_masks bool[L,B,Ed] (L = `levels`, or the
    deepest DAG in the batch when None - one device->host sync; levels past an env's own depth
    are empty; `max_depth` = the longest possible path, e.g. the pack's stages-per-job bound), has_mp bool[B] (the reference skips message passing when an observation
    has a single DAG layer, scheduler.py:196-198), commit_caps i64[B,A].
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
    x[..., 0] = (caps_n.double() / E).float()
    x[..., 1] = torch.where(node_job == src[:, None], 1.0, -1.0)
    x[..., 2] = (sup_n.double() / E).float()
    rem, dur = nodes[..., 0], nodes[..., 1]
    x[..., 3] = rem / num_tasks_scale
    x[..., 4] = rem * dur / work_scale
    x = torch.where(node_valid[..., None], x, torch.zeros_like(x))
    stage_mask = (nodes[..., 2] != 0) & node_valid

    exec_mask = torch.arange(E, device=dev)[None, None, :] < commit_caps[..., None]

    # DAG-layer edge masks (decima/utils.py:238-267): topological generations of the active
    # subgraph, mask l = edges with both ends in (generation l) U (its successors)
    edge_valid = torch.arange(Ed, device=dev)[None, :] < n_edges[:, None]
    el = obs["edge_links"].long()
    e_src = torch.where(edge_valid, el[..., 0], torch.full_like(el[..., 0], N))
    e_dst = torch.where(edge_valid, el[..., 1], torch.full_like(el[..., 1], N))
    gen = torch.zeros((B, N + 1), dtype=torch.long, device=dev)
    for _ in range(max_depth):
        cand = gen.gather(1, e_src) + 1
        cand = torch.where(edge_valid, cand, torch.zeros_like(cand))
        gen = gen.scatter_reduce(1, e_dst, cand, "amax", include_self=True)
        gen[:, N] = 0
    gen_n = gen[:, :N]
    max_gen = torch.where(node_valid, gen_n, torch.zeros_like(gen_n)).amax(1)
    masks = []
    if levels is None:
        levels = int(max_gen.max())
    for lvl in range(levels):
        in_lvl = torch.cat([(gen_n == lvl) & node_valid, torch.zeros((B, 1), dtype=torch.bool, device=dev)], 1)
        succ = torch.zeros((B, N + 1), dtype=torch.long, device=dev)
        succ = succ.scatter_reduce(1, e_dst, (in_lvl.gather(1, e_src) & edge_valid).long(), "amax", include_self=True)
        in_m = in_lvl | (succ > 0)
        masks.append(in_m.gather(1, e_src) & in_m.gather(1, e_dst) & edge_valid)
    edge_masks = torch.stack(masks) if masks else torch.zeros((0, B, Ed), dtype=torch.bool, device=dev)
    return {"x": x, "node_valid": node_valid, "stage_mask": stage_mask, "node_job": node_job, "job_valid": job_valid,
            "exec_mask": exec_mask, "edge_src": e_src, "edge_dst": e_dst, "edge_valid": edge_valid,
            "edge_masks": edge_masks, "has_mp": max_gen > 0, "depth": max_gen, "commit_caps": commit_caps,
            "dag_start": obs["dag_ptr"][:, :-1].long()}


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

    def forward(self, f: dict[str, torch.Tensor]) -> torch.Tensor:
        """child -> parent ("reverse flow") message passing one DAG layer at a time (scheduler.py:192-236)"""
        x, e_src, e_dst = f["x"], f["edge_src"], f["edge_dst"]
        B, N, _ = x.shape
        h_init = self.mlp_prep(x)
        F_ = h_init.shape[-1]
        dev = x.device
        zero_row = torch.zeros((B, 1, F_), dtype=h_init.dtype, device=dev)
        # nodes that are never the source end of an edge start from update(h_init)
        is_parent = torch.zeros((B, N + 1), dtype=torch.long, device=dev).scatter_reduce(
            1, e_src, f["edge_valid"].long(), "amax", include_self=True)[:, :N] > 0
        h = torch.where((~is_parent & f["node_valid"])[..., None], self.mlp_update(h_init), torch.zeros_like(h_init))
        for lvl in reversed(range(f["edge_masks"].shape[0])):
            em = f["edge_masks"][lvl]
            msg = torch.cat([self.mlp_msg(h), zero_row], 1)
            contrib = msg.gather(1, e_dst[..., None].expand(-1, -1, F_)) * em[..., None]
            agg = torch.zeros((B, N + 1, F_), dtype=h.dtype, device=dev).scatter_add_(1, e_src[..., None].expand(-1, -1, F_), contrib)[:, :N]
            recv = torch.zeros((B, N + 1), dtype=torch.long, device=dev).scatter_reduce(1, e_src, em.long(), "amax", include_self=True)[:, :N] > 0
            h = torch.where(recv[..., None], h_init + self.mlp_update(agg), h)
        # a single-layer observation gets mlp_prep only (scheduler.py:238-243)
        return torch.where(f["has_mp"][:, None, None], h, h_init)


class _DagEncoder(nn.Module):
    def __init__(self, nf: int, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp = make_mlp(nf + emb, output_dim=emb, **mlp_kwargs)

    def forward(self, h_node: torch.Tensor, f: dict[str, torch.Tensor]) -> torch.Tensor:
        y = self.mlp(torch.cat([f["x"], h_node], -1)) * f["node_valid"][..., None]
        B, N, F_ = y.shape
        A = f["job_valid"].shape[1]
        return torch.zeros((B, A + 1, F_), dtype=y.dtype, device=y.device).scatter_add_(
            1, f["node_job"][..., None].expand(-1, -1, F_), y)[:, :A]


class _GlobalEncoder(nn.Module):
    def __init__(self, emb: int, mlp_kwargs: dict[str, Any]):
        super().__init__()
        self.mlp = make_mlp(emb, output_dim=emb, **mlp_kwargs)

    def forward(self, h_dag: torch.Tensor, f: dict[str, torch.Tensor]) -> torch.Tensor:
        return (self.mlp(h_dag) * f["job_valid"][..., None]).sum(1)


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

    def encode(self, f: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        h_node = self.encoder.node_encoder(f)
        h_dag = self.encoder.dag_encoder(h_node, f)
        h_glob = self.encoder.global_encoder(h_dag, f)
        return {"node": h_node, "dag": h_dag, "glob": h_glob}

    def stage_scores(self, f: dict[str, torch.Tensor], h: dict[str, torch.Tensor]) -> torch.Tensor:
        """f32[B,N]; -inf where the node is not a schedulable stage (scheduler.py:289-318)"""
        B, N, _ = f["x"].shape
        F_ = h["dag"].shape[-1]
        pad = torch.zeros((B, 1, F_), dtype=h["dag"].dtype, device=h["dag"].device)
        h_dag_n = torch.cat([h["dag"], pad], 1).gather(1, f["node_job"][..., None].expand(-1, -1, F_))
        inp = torch.cat([f["x"], h["node"], h_dag_n, h["glob"][:, None, :].expand(-1, N, -1)], -1)
        s = self.stage_policy_network.mlp_score(inp).squeeze(-1)
        return torch.where(f["stage_mask"], s, torch.full_like(s, float("-inf")))

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
    def schedule_batch(self, obs, max_depth: int, generator: torch.Generator | None = None, levels: int | None = None):
        """one action per env: a stage sampled from softmax(stage scores), then an executor count
        sampled from softmax(exec scores of that stage's job) (scheduler.py:71-99). Returns
        ({"stage_idx": i32[B], "num_exec": i32[B]}, {"lgprob": f32[B], "job_idx": i64[B]})."""
        f = decima_observation(obs, self.num_executors, max_depth, levels=levels)
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
