"""shared by the emulator and GPU Decima tests: replays a decima_*.npz fixture on a batched env
(one env per recorded seed) and compares every recorded step."""
import os.path as osp

import numpy as np
import torch

from spark_sched_sim_amd import VecSparkSchedSimEnv
from spark_sched_sim_amd.decima import DecimaPolicy, compact_graph, decima_observation

HERE = osp.dirname(osp.abspath(__file__))
AGENT = dict(embed_dim=16,
             gnn_mlp_kwargs=dict(hid_dims=[32, 16], act_cls="LeakyReLU", act_kwargs=dict(inplace=True, negative_slope=0.2)),
             policy_mlp_kwargs=dict(hid_dims=[64, 64], act_cls="Tanh"))
SCORE_ATOL = 2e-5


def check_decima_fixture(name, device, lib, n_steps, path=None):
    g = np.load(path if path is not None else osp.join(HERE, "golden", f"{name}.npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], [float(v) for v in g["cfg_vals"]]))
    cfg["num_executors"] = int(cfg["num_executors"])
    cfg["job_arrival_cap"] = int(cfg["job_arrival_cap"])
    seeds = [int(s) for s in g["seeds"]]
    E = cfg["num_executors"]
    from spark_sched_sim_amd import workload
    pack = workload.profile_pack(str(g["trace_profile"])) if "trace_profile" in g.files else None  # (None: the frozen default pack)
    if "trace_profile_json" in g.files:  # a regime of the fixture's own (tests/test_oracle_vs_live_reference.py)
        import json
        sizes, n_q = [str(x) for x in g["trace_sizes"]], int(g["trace_queries"])
        pack = workload.build_pack(workload.make_raw_workload(int(g["trace_seed"]), sizes, n_q, profile=json.loads(str(g["trace_profile_json"]))),
                                   query_sizes=sizes, num_queries=n_q)
    env = VecSparkSchedSimEnv(cfg, len(seeds), device=device, _lib=lib, pack=pack)
    dev = env.device
    policy = DecimaPolicy(num_executors=E, **AGENT)
    policy.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_")})
    policy = policy.to(dev).eval()
    max_depth = env.dims.stage_stride
    obs, _ = env.reset(seed=seeds)
    acts = [g[f"s{s}_actions"] for s in seeds]
    T = min(n_steps, min(len(a) for a in acts))
    worst = 0.0
    for t in range(T):
        with torch.no_grad():
            f = decima_observation(obs, E, max_depth, edge_masks=True)
            cg = compact_graph(f)
            kg = env.decima_graph()
            compare_graphs(kg, cg, f)
            h = policy.encode(cg)
            ss_flat, ss_idx = policy.stage_scores(cg, h)
            ss = torch.full(f["x"].shape[:2], float("-inf"), device=dev)
            ss[cg["node_obs"][ss_idx], cg["node_loc"][ss_idx]] = ss_flat
            job_off = torch.cumsum(obs["n_jobs"].long(), 0) - obs["n_jobs"].long()
            es_all = [policy.exec_scores(cg, h, job_off + torch.clamp(torch.full_like(job_off, j), max=obs["n_jobs"].long() - 1))
                      for j in range(int(obs["n_jobs"].max()))]
            # the fused inference kernels against the tensor-op forward pass
            policy.bind_kernels(env._b)
            assert policy._use_kernels()
            hk = policy._encode_kernels(kg)
            for k in ("node", "dag", "glob"):
                # (embeddings are sums over up to ~40 nodes per job, ~1000 per observation: round-off scales with their magnitude)
                assert float((hk[k] - h[k]).abs().max()) <= SCORE_ATOL * max(1.0, float(h[k].abs().max())), (t, k)
            sk = policy._stage_scores_kernels(kg, hk)
            assert torch.equal(torch.isfinite(sk), torch.isfinite(ss)), t
            assert float((sk - ss)[torch.isfinite(ss)].abs().max()) <= SCORE_ATOL, t
            for j, es_t in enumerate(es_all):
                jg = job_off + torch.clamp(torch.full_like(job_off, j), max=obs["n_jobs"].long() - 1)
                ek = policy._exec_scores_kernels(kg, hk, jg)
                assert torch.equal(torch.isfinite(ek), torch.isfinite(es_t)), (t, j)
                fin = torch.isfinite(ek)
                assert float((ek[fin] - es_t[fin]).abs().max()) <= SCORE_ATOL if fin.any() else True, (t, j)
            # the pipeline's sampling kernels: draws are the Gumbel-max of the pipeline's own scores
            sc = {}
            ap = policy._sample_kernels(kg, hk, sk, torch.Generator().manual_seed(123), scores_out=sc)
            check_policy_draw({**ap, **sc}, policy.env_actions(ap), cg, f, es_all, job_off, *ap["rng"])
            # the one-launch policy kernel: same scores, and a draw that is the Gumbel-max of them
            acts_k, ak = policy.act_env(env, counter=1000 + t, seed=77, want_scores=True)
            fin = torch.isfinite(ss)
            assert torch.equal(torch.isfinite(ak["stage_scores"][:, : ss.shape[1]]), fin), t
            assert float((ak["stage_scores"][:, : ss.shape[1]] - ss)[fin].abs().max()) <= SCORE_ATOL, t
            check_policy_draw(ak, acts_k, cg, f, es_all, job_off, 77, 1000 + t)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        for b, s in enumerate(seeds):
            p = f"s{s}_t{t}_"
            n = int(obs["n_nodes"][b]); ne = int(obs["n_edges"][b]); a = int(obs["n_jobs"][b])
            assert n == g[p + "nodes"].shape[0] and a == len(g[p + "dag_ptr"]) - 1, (s, t)
            assert np.array_equal(f["x"][b, :n].cpu().numpy().view(np.uint32), g[p + "nodes"].view(np.uint32)), (s, t, "features")
            assert not f["x"][b, n:].any()
            assert np.array_equal(f["stage_mask"][b, :n].cpu().numpy(), g[p + "stage_mask"]), (s, t)
            assert not f["stage_mask"][b, n:].any()
            assert np.array_equal(f["exec_mask"][b, :a].cpu().numpy(), g[p + "exec_mask"]), (s, t)
            assert np.array_equal(obs["edge_links"][b, :ne].cpu().numpy(), g[p + "edge_links"]), (s, t)
            em = f["edge_masks"][:, b, :].cpu().numpy()
            gm = g[p + "edge_masks"]
            D = gm.shape[0]
            assert int(f["depth"][b]) == D and bool(f["has_mp"][b]) == (D > 0), (s, t)
            assert np.array_equal(em[:D, :ne], gm.reshape(D, ne)), (s, t, "edge masks")
            assert not em[D:].any() and not em[:, ne:].any()
            # scores
            sm = g[p + "stage_mask"]
            got = ss[b, :n].cpu().numpy()
            assert np.all(np.isneginf(got[~sm]))
            d = np.abs(got[sm] - g[p + "stage_scores"]).max() if sm.any() else 0.0
            worst = max(worst, float(d))
            assert d <= SCORE_ATOL, (s, t, "stage scores", d)
            cnt = g[p + "exec_counts"]
            off = np.concatenate([[0], np.cumsum(cnt)])
            for j in range(a):
                want = g[p + "exec_scores"][off[j]: off[j + 1]]
                gj = es_all[j][b].cpu().numpy()
                fin = np.isfinite(gj)
                assert int(fin.sum()) == len(want) and np.array_equal(fin, g[p + "exec_mask"][j]), (s, t, j)
                if len(want):
                    d = np.abs(gj[fin] - want).max()
                    worst = max(worst, float(d))
                    assert d <= SCORE_ATOL, (s, t, j, "exec scores", d)
        stage = torch.tensor([int(acts[b][t][0]) for b in range(len(seeds))], dtype=torch.int32, device=dev)
        nexec = torch.tensor([int(acts[b][t][1]) for b in range(len(seeds))], dtype=torch.int32, device=dev)
        obs, _, term, _, info = env.step({"stage_idx": stage, "num_exec": nexec})
        env.raise_on_error()
    env.close()
    return worst


def compare_graphs(kg, cg, f):
    """the compact graph written by the sss_decima_graph_build kernel against the tensor-op
    construction (decima_observation + compact_graph + graph_layers): identical, field by field"""
    from spark_sched_sim_amd.decima import graph_layers

    assert np.array_equal(kg["x"].cpu().numpy().view(np.uint32), cg["x"].cpu().numpy().view(np.uint32))
    for k in ("node_obs", "node_loc", "node_job", "stage_mask", "src", "dst", "edge_obs", "job_obs", "job_cap", "job_first", "obs_nodes", "obs_jobs"):
        assert torch.equal(kg[k], cg[k]), k
    assert torch.equal(kg["gen"].long(), cg["gen"]) and torch.equal(kg["obs_depth"].long(), cg["obs_depth"])
    rank = f["stage_mask"].long().cumsum(1) - 1
    want = torch.where(cg["stage_mask"], rank[cg["node_obs"], cg["node_loc"]], torch.full_like(cg["node_loc"], -1))
    assert torch.equal(kg["sched_rank"], want)
    layers = graph_layers(dict(cg))
    k_layers = graph_layers(kg)
    assert len(layers) == len(k_layers)
    for (e, r), (ke, kr) in zip(layers, k_layers):
        assert torch.equal(e, ke) and torch.equal(r, kr)
    assert not (kg["edge_layers"] >> len(layers)).any() and not (kg["node_recv"] >> len(layers)).any()
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    lists = VecSparkSchedSimEnv.decima_layer_lists(kg)
    assert len(lists) == len(layers) and all(torch.equal(a, r) for a, (_, r) in zip(lists, layers))
    check_list_pieces(kg, [r for _, r in layers])
    # out-edge ranges: every edge lies in its source's range, ranges tile the edge list
    deg = torch.zeros(kg["x"].shape[0], dtype=torch.long, device=kg["x"].device).index_add_(0, kg["src"], torch.ones_like(kg["src"]))
    assert torch.equal(kg["out_deg"].long(), deg)
    e_ids = torch.arange(kg["src"].numel(), device=deg.device)
    assert bool(((e_ids >= kg["out_start"][kg["src"]]) & (e_ids < kg["out_start"][kg["src"]] + deg[kg["src"]])).all())
    cnt = torch.zeros(kg["job_obs"].numel(), dtype=torch.long, device=deg.device).index_add_(0, kg["node_job"], torch.ones_like(kg["node_job"]))
    assert torch.equal(kg["job_nodes"], cnt)


def check_list_pieces(kg, receivers):
    """the graph kernel's own lists of receiving nodes (include/sss.h sss_decima_graph: per layer a dense piece per block of
    q = ceil(num_envs / 32) consecutive envs, lengths in layer_totals[l][s], piece s of layer l at
    recv[l * stride + node_off[s * q]]) hold exactly the layer's receivers: every piece lies inside its block's node range and
    the pieces of a layer together are the layer's receivers, each once"""
    ls, epoch = kg["_layer_lists"]
    assert ls["epoch"] == epoch, "the lists belong to a later graph of this env"
    B, stride, recv = kg["n_obs"], ls["stride"], ls["recv"]
    tot = kg["layer_totals"].view(33, 32).cpu()
    off = kg["obs_node_off"].cpu()
    M = kg["x"].shape[0] if "totals_dev" not in kg else int(kg["totals_dev"][0])
    q = (B + 31) // 32
    n_sets = (B + q - 1) // q
    assert int(tot[len(receivers):32].sum()) == 0 and int(tot[:, n_sets:].sum()) == 0
    nodes = kg["obs_nodes"].cpu()  # row 32: the largest observation of every block of envs
    assert tot[32, :n_sets].tolist() == [int(nodes[s * q: (s + 1) * q].max()) for s in range(n_sets)]
    for l, want in enumerate(receivers):
        got = []
        for s in range(n_sets):
            lo = int(off[s * q])
            hi = int(off[(s + 1) * q]) if (s + 1) * q < B else M
            n = int(tot[l, s])
            piece = recv[l * stride + lo: l * stride + lo + n]
            assert n <= hi - lo and bool(((piece >= lo) & (piece < hi)).all()), (l, s)
            got.append(piece)
        got = torch.cat(got) if got else recv[:0]
        assert torch.equal(torch.sort(got)[0], torch.sort(want)[0]), l


def check_pieces_on_device(device, lib=None, n_envs=6, steps=40, cfg=None):
    """`env.decima_graph_on_device` (capacity buffers, two sets of list counters used in turn - a launch clears the other set) over
    several consecutive calls, with and without inactive envs: every call's list pieces hold that call's receivers"""
    from spark_sched_sim_amd import VecSparkSchedSimEnv

    cfg = cfg or dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, n_envs, device=device, auto_reset=True, _lib=lib)
    env.reset(seed=21)
    gen = torch.Generator().manual_seed(5)
    seen = 0
    for call in range(5):
        env.rollout("fair", steps)
        active = None if call % 2 == 0 else (torch.rand(n_envs, generator=gen) < 0.7).to(device)
        g = env.decima_graph_on_device(active)
        M = int(g["totals_dev"][0])
        recv_bits = g["node_recv"][:M].long()
        D = int(g["max_depth"])
        assert not bool((recv_bits >> D).any())
        check_list_pieces(g, [((recv_bits >> l) & 1).nonzero(as_tuple=True)[0] for l in range(D)])
        seen += M
    assert seen > 0
    env.close()


def _gumbel(seed, counter, env, idx, draw):
    """tests' restatement of dp_gumbel (csrc/sss_decima_policy.h)"""
    m = (1 << 64) - 1
    z = (seed ^ ((counter * 0x9E3779B97F4A7C15) & m) ^ ((env & 0xFFFFFFFF) << 32) ^ (draw << 28) ^ idx) & m
    z = (z + 0x9E3779B97F4A7C15) & m
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & m
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & m
    z ^= z >> 31
    u = np.float32((np.float32(z >> 40) + np.float32(0.5)) * np.float32(1.0 / 16777216.0))
    return float(-np.log(-np.log(u)))


def check_policy_draw(ak, acts_k, cg, f, es_all, job_off, seed, counter):
    """the kernel's (stage, executor count) must be the Gumbel-max of its own scores under the
    documented counter-based stream, and its log-probability the softmax log-probability"""
    B = f["x"].shape[0]
    ss = ak["stage_scores"].cpu().numpy()
    es = ak["exec_scores"].cpu().numpy()
    sm = f["stage_mask"].cpu().numpy()
    node_job = f["node_job"].cpu().numpy()
    for b in range(B):
        cand = np.flatnonzero(sm[b])
        if cand.size == 0:
            assert int(acts_k["stage_idx"][b]) == -1
            continue
        keys = np.asarray([ss[b, i] + _gumbel(seed, counter, b, int(i), 0) for i in cand])
        sel_rank = int(ak["stage_sel"][b])
        assert keys[sel_rank] >= keys.max() - 1e-4, (b, "stage draw")
        assert int(acts_k["stage_idx"][b]) == sel_rank
        job = int(node_job[b, cand[sel_rank]])
        assert int(ak["job_idx"][b]) == job
        ok = np.isfinite(es[b])
        want_es = es_all[job][b].cpu().numpy() if job < len(es_all) else None
        if want_es is not None:
            assert np.array_equal(ok, np.isfinite(want_es)) and (np.abs(es[b][ok] - want_es[ok]).max() <= SCORE_ATOL if ok.any() else True), (b, "exec scores")
        lg = float(ss[b, cand[sel_rank]] - (np.log(np.exp(ss[b, cand] - ss[b, cand].max()).sum()) + ss[b, cand].max()))
        if ok.any():
            ek = np.asarray([es[b, c] + _gumbel(seed, counter, b, c, 1) if ok[c] else -np.inf for c in range(es.shape[1])])
            c_sel = int(ak["exec_sel"][b])
            assert ek[c_sel] >= ek.max() - 1e-4 and int(acts_k["num_exec"][b]) == c_sel + 1, (b, "exec draw")
            lg += float(es[b, c_sel] - (np.log(np.exp(es[b][ok] - es[b][ok].max()).sum()) + es[b][ok].max()))
        assert abs(float(ak["lgprob"][b]) - lg) <= 1e-4, (b, "lgprob", float(ak["lgprob"][b]), lg)


def check_on_device_step_equals_the_synchronous_one(device, lib=None, n_envs=6, steps=60, cfg=None):
    """`DecimaPolicy.schedule_env` without a device->host round trip (capacity buffers, the graph's totals on the device,
    kernels that read their row counts themselves - env.decima_graph_on_device) against the path that reads the totals back:
    same observations, same generator seed and draw counter -> the same actions, log-probabilities bit for bit, and the same
    embeddings / scores on the meaningful rows. Also with some envs inactive."""
    import torch

    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = cfg or dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, n_envs, device=device, auto_reset=True, _lib=lib)
    torch.manual_seed(7)
    policy = DecimaPolicy(num_executors=cfg["num_executors"], **AGENT).to(device).eval()
    with torch.no_grad():
        for n_, p_ in policy.named_parameters():
            if "bias" in n_:
                p_.normal_(0.0, 0.1)
    gen = torch.Generator(device=device).manual_seed(11)
    env.reset(seed=100)
    for t in range(steps):
        active = None if t % 3 else (torch.arange(n_envs, device=device) % 4 != 1)
        calls = getattr(policy, "_calls", 0)
        act_a, aux_a = policy.schedule_env(env, generator=gen, active=active, host_sync=True)
        act_a = {k: v.clone() for k, v in act_a.items()}
        aux_a = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in aux_a.items()}
        policy._calls = calls  # the same draw counter for the second pass
        act_b, aux_b = policy.schedule_env(env, generator=gen, active=active, host_sync=False)
        sel = torch.ones(n_envs, dtype=torch.bool, device=device) if active is None else active
        for k in ("stage_idx", "num_exec"):
            assert torch.equal(act_a[k][sel], act_b[k][sel]), (t, k)
        assert torch.equal(aux_a["lgprob"][sel & aux_a["any_stage"]], aux_b["lgprob"][sel & aux_b["any_stage"]]), t
        assert torch.equal(aux_a["any_stage"][sel], aux_b["any_stage"][sel]), t
        if active is not None:  # inactive envs are not stepped by this policy: the heuristic takes them along
            fair = env.policy_actions("fair")
            for k in ("stage_idx", "num_exec"):
                act_b[k] = torch.where(sel, act_b[k], fair[k])
        _, _, _, _, info = env.step(act_b)
        assert not info["err"].any(), t
    env.close()


def check_reference_style_episode(device, lib=None, max_steps=None):
    """the single-env harness of the reference (examples.py:84-102) with the Decima plugin:
    env_wrapper_cls(env) + schedule(obs) sampling through `random.choices` under a fixed
    `random.seed` reproduces the recorded reference episode action for action (scores agree to
    ~1e-6, so a draw landing that close to a CDF boundary could differ; none does in this episode)"""
    import os.path as osp
    import random

    import numpy as np
    import torch

    HERE = osp.dirname(osp.abspath(__file__))
    from golden_util import bits
    from spark_sched_sim_amd import SparkSchedSimEnv, make_scheduler

    g = np.load(osp.join(HERE, "golden", "decima_episode.npz"))
    cfg = dict(zip([str(k) for k in g["cfg_keys"]], [float(v) for v in g["cfg_vals"]]))
    cfg["num_executors"], cfg["job_arrival_cap"] = int(cfg["num_executors"]), int(cfg["job_arrival_cap"])
    sched = make_scheduler(dict(AGENT, agent_cls="DecimaScheduler", num_executors=cfg["num_executors"]))
    sched.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w_")})
    sched.eval()
    env = sched.env_wrapper_cls(SparkSchedSimEnv(cfg, device=device, _lib=lib))
    random.seed(int(g["py_seed"]))
    obs, _ = env.reset(seed=int(g["seed"]), options=None)
    terminated = truncated = False
    t = 0
    while not (terminated or truncated) and (max_steps is None or t < max_steps):
        action, info = sched.schedule(obs)
        assert [action["stage_idx"], action["job_idx"], action["num_exec"]] == g["actions"][t].tolist(), t
        assert abs(float(info["lgprob"]) - float(g["lgprobs"][t])) < 1e-4, t
        obs, reward, terminated, truncated, einfo = env.step(action)
        assert bits(float(reward)) == bits(float(g["rewards"][t])) and bits(float(einfo["wall_time"])) == bits(float(g["wall_times"][t])), t
        t += 1
    assert t == (len(g["actions"]) if max_steps is None else min(max_steps, len(g["actions"])))
    env.close()
