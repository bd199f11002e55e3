"""the batched Decima observation transform and GNN forward against fixtures
recorded from the reference's own DecimaEnvWrapper / DecimaScheduler (tests/golden/make_decima_golden.py),
with the simulator running under the CPU wave emulator and the recorded actions replayed.
Node features and masks must be identical; scores agree to float32 round-off (different summation
order in the scatter-adds; tolerance 2e-5 absolute on O(1) scores)."""
import pytest

from decima_util import check_decima_fixture
from emu_util import load_emu


@pytest.mark.parametrize("name,n_steps", [("decima_c1", 50), ("decima_e50", 40), ("decima_e100", 30), ("decima_deep", 30)])
def test_decima_features_and_scores_match_reference(name, n_steps):
    check_decima_fixture(name, "cpu", load_emu(), n_steps)


def test_sampled_actions_are_always_valid():
    """Decima in the loop: sampled (stage, executor count) pairs drive 6 envs for 150 steps
    without a single rejected action, and the log-probabilities are finite"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 6, device="cpu", auto_reset=True, _lib=load_emu())
    torch.manual_seed(7)
    policy = DecimaPolicy(num_executors=10, **AGENT).eval()
    gen = torch.Generator().manual_seed(11)
    obs, _ = env.reset(seed=100)
    for _ in range(150):
        act, aux = policy.schedule_env(env, generator=gen)
        assert torch.isfinite(aux["lgprob"]).all()
        obs, r, term, trunc, info = env.step(act)
        assert not info["err"].any()
    assert int(env.header_field("n_steps").sum()) == 6 * 150
    env.close()


def test_schedule_env_results_are_work_space_unless_fresh_outputs_is_asked_for():
    """the ownership contract of `schedule_env` / `act` (their docstrings): on the default path the returned tensors are the policy's
    work space, overwritten by the next call; `fresh_outputs=True` returns tensors the caller may keep"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 5, device="cpu", auto_reset=True, _lib=load_emu())
    torch.manual_seed(3)
    policy = DecimaPolicy(num_executors=10, **AGENT).eval()
    gen = torch.Generator().manual_seed(5)
    env.reset(seed=40)
    kept = []
    for fresh in (False, False, True, True):
        act, aux = policy.schedule_env(env, generator=gen, fresh_outputs=fresh)
        kept.append((fresh, act, aux, {k: v.clone() for k, v in list(act.items()) + [("lgprob", aux["lgprob"])]}))
        env.step(act)
    (_, a0, x0, c0), (_, a1, x1, c1), (_, a2, x2, c2), (_, a3, x3, c3) = kept
    assert a0["stage_idx"].data_ptr() == a1["stage_idx"].data_ptr() and x0["lgprob"].data_ptr() == x1["lgprob"].data_ptr()   # work space: aliased
    assert len({a1["stage_idx"].data_ptr(), a2["stage_idx"].data_ptr(), a3["stage_idx"].data_ptr()}) == 3
    for act, aux, copy in ((a2, x2, c2), (a3, x3, c3)):   # held across later calls: still what they were
        assert torch.equal(act["stage_idx"], copy["stage_idx"]) and torch.equal(act["num_exec"], copy["num_exec"]) and torch.equal(aux["lgprob"], copy["lgprob"])
    assert not torch.equal(x0["lgprob"], c0["lgprob"])   # (the first result's buffer now holds a later step's values)
    env.close()


def test_greedy_actions_are_the_arg_max_of_the_scores():
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 4, device="cpu", auto_reset=True, _lib=load_emu())
    torch.manual_seed(3)
    policy = DecimaPolicy(num_executors=10, **AGENT).eval()
    env.reset(seed=70)
    for _ in range(25):
        act, aux = policy.schedule_env(env, greedy=True)
        g = env.decima_graph()
        h = policy.encode(g)
        s, idx = policy.stage_scores(g, h)
        for b in range(4):
            mine = g["node_obs"][idx] == b
            assert int(act["stage_idx"][b]) == int(s[mine].argmax())   # index among the observation's schedulable stages
        obs, r, term, trunc, info = env.step(act)
        assert not info["err"].any()
    env.close()


def test_graph_build_refuses_list_counters_of_another_length():
    """include/sss.h sss_decima_graph.layer_totals_len: the i64[33][32] counters must be announced as such"""
    import ctypes as C

    import torch

    from spark_sched_sim_amd import VecSparkSchedSimEnv
    import spark_sched_sim_amd.vec_env as ve

    cfg = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 3, device="cpu", _lib=load_emu())
    env.reset(seed=1)
    env.decima_graph()   # fine
    real = ve.SssDecimaGraph

    def stale(*a):   # a caller built against the round-4 layout: 32 counters
        return real(*a[:-1], 32)
    ve.SssDecimaGraph = stale
    try:
        with pytest.raises(ValueError, match="layer_totals_len"):
            env.decima_graph()
    finally:
        ve.SssDecimaGraph = real
    env.close()


def test_decima_in_the_loop_beyond_64_executors():
    """100 and 128 executors (two executor counts per lane in the sampling kernels; round 4 raised `sss error -28` here):
    sampled actions through the pipeline AND through the one-launch policy kernel drive the wide simulator without a rejected
    action and executor counts above 64 do get drawn (scores of the two paths against each other and against the reference:
    the decima_e100 fixture above)"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    for E, steps in ((100, 60), (128, 40)):
        cfg = dict(num_executors=E, job_arrival_cap=10, job_arrival_rate=1.5e-4, moving_delay=2000.0, warmup_delay=1000.0)
        env = VecSparkSchedSimEnv(cfg, 4, device="cpu", auto_reset=True, _lib=load_emu())
        torch.manual_seed(E)
        policy = DecimaPolicy(num_executors=E, **AGENT).eval()
        gen = torch.Generator().manual_seed(5)
        env.reset(seed=300)
        most = 0
        for t in range(steps):
            act, aux = policy.schedule_env(env, generator=gen, one_launch=(t % 3 == 2))
            assert torch.isfinite(aux["lgprob"]).all()
            most = max(most, int(act["num_exec"].max()))
            _, _, _, _, info = env.step(act)
            assert not info["err"].any()
        assert most > 64, most  # (an empty cluster at the start: the first decisions can hand out nearly all executors)
        env.close()


def test_graph_kernel_lists_and_one_call_encoder():
    """`decima_graph`'s list of schedulable nodes (what the stage-score launch iterates over) is `stage_mask`'s index
    list, with and without frozen envs; and the encoder enqueued by one call (`sss_gnn_encode`: list sizes left on the
    device, PREP and SINK in one pass) produces the embeddings of the launch-by-launch path, bit for bit"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 6, device="cpu", auto_reset=True, _lib=load_emu())
    torch.manual_seed(7)
    policy = DecimaPolicy(num_executors=10, **AGENT).eval()
    policy.bind_kernels(env._b)
    env.reset(seed=100)
    env.rollout("fair", 40)
    for active in (None, torch.tensor([True, False, True, True, False, True])):
        g = env.decima_graph(active)
        assert torch.equal(g["sched_list"], g["stage_mask"].nonzero(as_tuple=True)[0]) and g["sched_list"].numel() > 0
        assert g["max_depth"] == env.max_dag_depth and int(g["obs_depth"].max()) <= g["max_depth"]
        one = policy._encode_kernels(g)
        g2 = {k: v for k, v in g.items() if k != "max_depth"}  # without the bound: launch by launch, list sizes read back
        ref = policy._encode_kernels(g2)
        for k in ("node", "dag", "glob"):
            assert torch.equal(one[k], ref[k]), k
        # the graph kernel's own layer lists live in work space of the env: a graph that is no longer the last one built
        # has its lists rebuilt by the encoder (same embeddings)
        assert g["_layer_lists"][0]["epoch"] == g["_layer_lists"][1]
        env.decima_graph(active)
        assert g["_layer_lists"][0]["epoch"] != g["_layer_lists"][1]
        again = policy._encode_kernels(g)
        for k in ("node", "dag", "glob"):
            assert torch.equal(again[k], ref[k]), k
    env.close()


def test_reference_style_decima_episode():
    """decima_util.check_reference_style_episode on the emulator: the first 150 steps of the recorded episode (the GPU test
    replays all of it)"""
    from decima_util import check_reference_style_episode

    check_reference_style_episode("cpu", load_emu(), max_steps=150)


def test_large_node_capacity_falls_back_to_tensor_op_graph():
    """job capacity 500 -> 9000 node slots: beyond the graph kernel's LDS working set (8 bytes per node slot + 8 per job slot in
    64 KB); the same compact graph then comes from tensor ops and the policy runs its tensor-op forward"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=8, job_arrival_cap=500, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 3, device="cpu", _lib=load_emu())
    assert 8 * env.dims.node_cap > 65536 and not env.graph_kernel_fits
    torch.manual_seed(1)
    policy = DecimaPolicy(num_executors=8, **AGENT).eval()
    gen = torch.Generator().manual_seed(2)
    env.reset(seed=5)
    for _ in range(25):
        act, aux = policy.schedule_env(env, generator=gen)
        obs, r, term, trunc, info = env.step(act)
        assert not info["err"].any() and torch.isfinite(aux["lgprob"]).all()
    g = env.decima_graph(active=torch.tensor([True, False, True]))
    assert "out_start" not in g and int(g["obs_nodes"][1]) == 0 and int(g["obs_nodes"][0]) > 0
    env.close()


def test_reference_checkpoint_loads_unchanged():
    """`models/decima/model.pt` of the reference (present in the build container only): every key
    and shape of its state dict matches `DecimaPolicy`, strictly"""
    import os.path as osp

    import pytest
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd.decima import DecimaPolicy

    path = "/root/reference/models/decima/model.pt"
    if not osp.exists(path):
        pytest.skip("reference checkpoint not available here")
    policy = DecimaPolicy(num_executors=50, **AGENT, state_dict_path=path)
    sd = torch.load(path, map_location="cpu")
    assert set(sd) == set(policy.state_dict()) and sum(v.numel() for v in sd.values()) == 20802
    assert all(torch.equal(policy.state_dict()[k], v) for k, v in sd.items())


def test_prefix_rows_matches_numpy():
    """include/sss.h sss_prefix_rows: exclusive prefix sums, masked counts and totals of strided i32 rows"""
    import ctypes as C

    import numpy as np
    import torch

    lib = load_emu()
    lib.sss_prefix_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    B = 37
    src = torch.from_numpy(rng.integers(0, 1000, size=(B, 8)).astype(np.int32))   # like obs_i32: counts in the columns of per-env rows
    mask = torch.from_numpy((rng.random(B) < 0.7).astype(np.uint8))
    for m in (None, mask):
        off = torch.empty((3, B), dtype=torch.int64)
        cnt = torch.empty((3, B), dtype=torch.int64)
        tot = torch.empty(3, dtype=torch.int64)
        rc = lib.sss_prefix_rows(src.data_ptr(), 1, src.stride(0), m.data_ptr() if m is not None else None, 3, B, off.data_ptr(), cnt.data_ptr(), tot.data_ptr(), None)
        assert rc == 0
        want = src[:, :3].numpy().astype(np.int64).T * (m.numpy()[None, :] if m is not None else 1)
        assert np.array_equal(cnt.numpy(), want)
        assert np.array_equal(off.numpy(), np.cumsum(want, 1) - want)
        assert np.array_equal(tot.numpy(), want.sum(1))
    rows = torch.from_numpy(rng.integers(0, 50, size=(32, B)).astype(np.int32))    # like layer_cnt: contiguous rows, no counts wanted
    off = torch.empty((32, B), dtype=torch.int64)
    tot = torch.empty(32, dtype=torch.int64)
    assert lib.sss_prefix_rows(rows.data_ptr(), rows.stride(0), 1, None, 32, B, off.data_ptr(), None, tot.data_ptr(), None) == 0
    w = rows.numpy().astype(np.int64)
    assert np.array_equal(off.numpy(), np.cumsum(w, 1) - w) and np.array_equal(tot.numpy(), w.sum(1))


def test_decima_step_without_host_round_trip_equals_the_synchronous_one():
    from decima_util import check_on_device_step_equals_the_synchronous_one

    check_on_device_step_equals_the_synchronous_one("cpu", load_emu(), n_envs=5, steps=40)


def test_kernel_weight_images_are_the_documented_permutations():
    """host-side packing of the Decima MLPs for the 16-lanes-per-row kernels (csrc/sss_gnn16.h) and for the matrix-core policy
    heads (csrc/sss_gnn_mfma.h MfmaHead) - `DecimaPolicy._packed_weights` builds them with reshapes / permutes / fancy indexing;
    here every entry is checked against the layout the kernels document, with plain index arithmetic (runs on the CPU: the
    images are only ever EXECUTED by gfx950 kernels, so this is where their construction gets a second opinion)"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd.decima import NUM_DAG_FEATURES, NUM_NODE_FEATURES, DecimaPolicy

    torch.manual_seed(3)
    pol = DecimaPolicy(num_executors=10, **AGENT).eval()
    with torch.no_grad():
        for p_ in pol.parameters():
            p_.normal_(0.0, 1.0)
    w = pol._packed_weights()

    def lin(mlp):
        return [m for m in mlp if isinstance(m, torch.nn.Linear)]

    # 16-lane images: w1[i][g][q] = W1[g + 16 q][i], b1[g][q], w2[jj][q][g][r] = W2[g + 16 r][jj + 16 q], b2[g][r], w3, b3
    for name, mlp in (("msg16", pol.encoder.node_encoder.mlp_msg), ("update16", pol.encoder.node_encoder.mlp_update),
                      ("stage16", pol.stage_policy_network.mlp_score), ("exec16", pol.exec_policy_network.mlp_score)):
        l1, l2, l3 = lin(mlp)
        W1, b1, W2, b2, W3, b3 = (t.detach() for t in (l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias))
        h1, n_in = W1.shape
        h2 = W2.shape[0]
        q1, q2 = h1 // 16, h2 // 16
        img = w[name].tolist()
        o = 0
        for i in range(n_in):
            for g in range(16):
                for q in range(q1):
                    assert img[o] == float(W1[g + 16 * q, i]), (name, "w1", i, g, q)
                    o += 1
        for g in range(16):
            for q in range(q1):
                assert img[o] == float(b1[g + 16 * q]), (name, "b1")
                o += 1
        for jj in range(16):
            for q in range(q1):
                for g in range(16):
                    for r in range(q2):
                        assert img[o] == float(W2[g + 16 * r, jj + 16 * q]), (name, "w2", jj, q, g, r)
                        o += 1
        for g in range(16):
            for r in range(q2):
                assert img[o] == float(b2[g + 16 * r]), (name, "b2")
                o += 1
        if W3.shape[0] == 16:   # w3[k][g] = W3[g][k]
            for k in range(h2):
                for g in range(16):
                    assert img[o] == float(W3[g, k]), (name, "w3")
                    o += 1
        else:                   # one output: w3[g][r] = W3[0][g + 16 r]
            for g in range(16):
                for r in range(q2):
                    assert img[o] == float(W3[0, g + 16 * r]), (name, "w3")
                    o += 1
        for k in range(b3.numel()):
            assert img[o] == float(b3[k])
            o += 1
        assert all(v == 0.0 for v in img[o:]) and len(img) % 4 == 0   # padding to whole float4s

    # matrix-core head images: K-step (u, r) of lane (q, i) holds feature col[u][4 q + r]; A1[tile][step][lane] = W1[16 tile + i][that column] (0 where
    # the column is padding), A2[tile][s][lane] = W2[16 tile + i][16 (s >> 2) + 4 q + (s & 3)], then b1, b2
    f16 = list(range(16))
    stage_cols = [[NUM_NODE_FEATURES + f for f in f16], [NUM_NODE_FEATURES + 16 + f for f in f16], [NUM_NODE_FEATURES + 32 + f for f in f16],
                  [f if f < NUM_NODE_FEATURES else -1 for f in f16]]
    exec_cols = [[NUM_DAG_FEATURES + f for f in f16], [NUM_DAG_FEATURES + 16 + f for f in f16],
                 [f if f < NUM_DAG_FEATURES else (NUM_DAG_FEATURES + 32 if f == NUM_DAG_FEATURES else -1) for f in f16]]
    for name, mlp, cols in (("stage_mfma", pol.stage_policy_network.mlp_score, stage_cols), ("exec_mfma", pol.exec_policy_network.mlp_score, exec_cols)):
        l1, l2, _ = lin(mlp)
        W1, b1, W2, b2 = (t.detach() for t in (l1.weight, l1.bias, l2.weight, l2.bias))
        U = len(cols)
        img = w[name].tolist()
        o = 0
        for tile in range(4):
            for step in range(4 * U):
                for lane in range(64):
                    i, q = lane & 15, lane >> 4
                    c = cols[step >> 2][4 * q + (step & 3)]
                    assert img[o] == (float(W1[16 * tile + i, c]) if c >= 0 else 0.0), (name, "a1", tile, step, lane)
                    o += 1
        for tile in range(4):
            for s in range(16):
                for lane in range(64):
                    i, q = lane & 15, lane >> 4
                    assert img[o] == float(W2[16 * tile + i, 16 * (s >> 2) + 4 * q + (s & 3)]), (name, "a2", tile, s, lane)
                    o += 1
        assert img[o: o + 64] == b1.tolist() and img[o + 64: o + 128] == b2.tolist() and len(img) == o + 128
        # every input column of the first Linear appears exactly once among the K-steps (no feature dropped or doubled)
        used = sorted(c for row in cols for c in row if c >= 0)
        assert used == list(range(W1.shape[1])), name


def test_list_pieces_of_the_graph_on_the_device():
    """decima_util.check_pieces_on_device on the emulator (6 envs: a block per env; 70 envs: blocks of three, the last one short)"""
    from decima_util import check_pieces_on_device

    check_pieces_on_device("cpu", load_emu(), n_envs=6, steps=40)
    check_pieces_on_device("cpu", load_emu(), n_envs=70, steps=25)
