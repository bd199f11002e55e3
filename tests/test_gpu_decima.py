"""GPU leg of tests/test_emu_decima.py: the same fixtures with the simulator on the HIP path and
the Decima transform / GNN running as torch ops on the same device, consuming the env's
observation tensors in place."""
import pytest
import torch

from decima_util import AGENT, check_decima_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.gpu
def test_reference_style_decima_episode_gpu():
    """decima_util.check_reference_style_episode on the GPU: the whole recorded reference episode, action for action"""
    from decima_util import check_reference_style_episode

    check_reference_style_episode("cuda:0")


@pytest.mark.parametrize("name,n_steps", [("decima_c1", 90), ("decima_e50", 90), ("decima_e100", 90), ("decima_deep", 90)])
def test_decima_features_and_scores_match_reference_gpu(name, n_steps):
    check_decima_fixture(name, "cuda:0", None, n_steps)


def test_decima_in_the_loop_256_envs():
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 256, device="cuda:0", auto_reset=True)
    torch.manual_seed(7)
    policy = DecimaPolicy(num_executors=10, **AGENT).to("cuda:0").eval()
    gen = torch.Generator(device="cuda:0").manual_seed(11)
    obs, _ = env.reset(seed=100)
    for _ in range(300):
        act, aux = policy.schedule_env(env, generator=gen)
        obs, r, term, trunc, info = env.step(act)
    torch.cuda.synchronize()
    assert not info["err"].any() and torch.isfinite(aux["lgprob"]).all()
    assert int(env.header_field("n_steps").sum()) == 256 * 300
    env.close()


@pytest.mark.parametrize("cfg,B,T", [
    (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), 64, 400),
    (dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), 24, 500),
    (dict(num_executors=3, job_arrival_cap=20, job_arrival_rate=8.0e-5, moving_delay=2000.0, warmup_delay=1000.0), 24, 300),
    (dict(num_executors=12, job_arrival_cap=40, job_arrival_rate=1.0e-4, moving_delay=0.0, warmup_delay=0.0), 24, 300),
    (dict(num_executors=64, job_arrival_cap=60, job_arrival_rate=2.0e-4, moving_delay=2000.0, warmup_delay=1000.0), 24, 300),
    (dict(num_executors=100, job_arrival_cap=60, job_arrival_rate=2.0e-4, moving_delay=2000.0, warmup_delay=1000.0), 24, 300),
    (dict(num_executors=128, job_arrival_cap=40, job_arrival_rate=3.0e-4, moving_delay=2000.0, warmup_delay=1000.0), 16, 250),
], ids=["c2", "c3", "three_exec", "zero_delays", "sixty_four_exec", "e100", "e128"])
def test_simulator_under_decima_actions_matches_oracle(cfg, B, T, pack):
    """the step kernel under the action distribution a GNN policy produces (many executors per
    decision, any stage of any job, ...): envs driven by sampled Decima actions; every env's
    (action -> reward, wall time, termination, observation sizes) chain is replayed through the C
    oracle and must agree bit for bit - including the step at which the reference's `[step]` stall
    fires, if it does"""
    import numpy as np

    from golden_util import bits
    from oracle_binding import OracleEnv
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    E = cfg["num_executors"]
    env = VecSparkSchedSimEnv(cfg, B, device="cuda:0", pack=pack)
    torch.manual_seed(3)
    policy = DecimaPolicy(num_executors=E, **AGENT).to("cuda:0").eval()
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    obs, _ = env.reset(seed=900)
    rec = []
    for _ in range(T):
        act, _ = policy.schedule_env(env, generator=gen)
        a_s, a_n = act["stage_idx"].cpu().numpy().copy(), act["num_exec"].cpu().numpy().copy()
        obs, rew, term, trunc, info = env.step(act)
        rec.append((a_s, a_n, rew.cpu().numpy().copy(), info["wall_time"].cpu().numpy().copy(), term.cpu().numpy().copy(),
                    obs["n_nodes"].cpu().numpy().copy(), info["err"].cpu().numpy().copy()))
    for b in range(B):
        o = OracleEnv(pack, cfg)
        o.reset(900 + b)
        dead = False
        for t, (a_s, a_n, rew, wall, term, n_nodes, err) in enumerate(rec):
            if dead:  # terminated or failed earlier: the batched env keeps reporting "reset() required"
                assert err[b] in (5, 8), (b, t, err[b])
                continue
            e, r, done = o.step(int(a_s[b]), int(a_n[b]))
            assert e == int(err[b]), (b, t, e, err[b])
            if e:
                assert e == 5
                dead = True
                continue
            info = o.info()
            assert bits(r) == bits(float(rew[b])) and bits(info.wall_time) == bits(float(wall[b])), (b, t)
            assert done == bool(term[b]) and info.n_nodes == int(n_nodes[b]), (b, t)
            dead = done
        o.close()
    env.close()


def test_graph_kernel_at_c3_sizing_matches_tensor_ops():
    """node capacity 3600 (200 jobs x 18 stage slots): 57.6 KB of LDS per workgroup in the graph
    kernel; its output must still equal the tensor-op construction, and the kernel forward the
    tensor-op forward, on busy mid-episode observations"""
    from decima_util import compare_graphs
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy, compact_graph, decima_observation

    cfg = dict(num_executors=50, job_arrival_cap=200, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 32, device="cuda:0")
    torch.manual_seed(5)
    policy = DecimaPolicy(num_executors=50, **AGENT).to("cuda:0").eval().bind_kernels(env._b)
    obs, _ = env.reset(seed=77)
    for chunk in range(4):
        env.rollout("fair", 700)
        obs = env._obs()
        f = decima_observation(obs, 50, env.dims.stage_stride, edge_masks=True)
        cg = compact_graph(f)
        kg = env.decima_graph()
        compare_graphs(kg, cg, f)
        with torch.no_grad():
            h = policy.encode(cg)
            hk = policy._encode_kernels(kg)
            for k in ("node", "dag", "glob"):
                assert float((hk[k] - h[k]).abs().max()) <= 5e-5, (chunk, k)
    assert int(obs["n_nodes"].max()) > 40
    env.close()


@pytest.mark.parametrize("E,J,rate", [(1, 6, 1.0e-4), (2, 12, 1.0e-4), (17, 40, 1.0e-4), (64, 60, 2.0e-4), (65, 60, 2.0e-4), (100, 60, 2.0e-4), (128, 60, 3.0e-4)])
def test_decima_pipeline_at_executor_count_extremes(E, J, rate):
    """1, 2, 17, 64 executors and - two executor counts per lane in the sampling kernels, the simulator's wide
    instantiation - 65, 100, 128: 64 envs stepped by sampled Decima actions for 150 steps without a
    rejected action (only the reference's own `[step]` stall may appear), the kernel forward equal
    to the tensor-op forward, and the one-launch policy kernel consistent with the pipeline's scores"""
    from decima_util import SCORE_ATOL
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy, compact_graph, decima_observation

    cfg = dict(num_executors=E, job_arrival_cap=J, job_arrival_rate=rate, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 64, device="cuda:0", auto_reset=True)
    torch.manual_seed(E)
    policy = DecimaPolicy(num_executors=E, **AGENT).to("cuda:0").eval().bind_kernels(env._b)
    gen = torch.Generator(device="cuda:0").manual_seed(3)
    obs, _ = env.reset(seed=400)
    for t in range(150):
        if t % 50 == 10:
            kg = env.decima_graph()
            cg = compact_graph(decima_observation(obs, E, env.dims.stage_stride))
            with torch.no_grad():
                h, hk = policy.encode(cg), policy._encode_kernels(kg)
            for k in ("node", "dag", "glob"):
                assert float((hk[k] - h[k]).abs().max()) <= 5e-5, (t, k)
            sk = policy._stage_scores_kernels(kg, hk)
            _, one = policy.act_env(env, counter=t, seed=9, want_scores=True)
            fin = torch.isfinite(sk)
            assert torch.equal(torch.isfinite(one["stage_scores"]), fin) and float((one["stage_scores"] - sk)[fin].abs().max()) <= SCORE_ATOL
        act, aux = policy.schedule_env(env, generator=gen)
        obs, r, term, trunc, info = env.step(act)
        live = info["err"] == 0
        assert bool(((info["err"] == 0) | (info["err"] == 5) | (info["err"] == 8)).all()), torch.unique(info["err"])
        assert torch.isfinite(aux["lgprob"][live]).all()
    env.close()


def test_sixteen_lane_kernels_agree_with_one_thread_per_row():
    """csrc/sss_gnn16.h (a row on 16 lanes: DAG layers always, the policy heads at small row counts) against
    csrc/sss_gnn.h (one thread per row) on the same graph: embeddings and scores within the fixtures' 2e-5"""
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 96, device="cuda:0", auto_reset=True)
    torch.manual_seed(3)
    policy = DecimaPolicy(num_executors=10, **AGENT).to("cuda:0").eval()
    env.reset(seed=5)
    env.rollout("hash", 150)
    g = env.decima_graph()
    policy.bind_kernels(env._b)
    w = policy._packed_weights()
    assert {"msg16", "update16", "stage16", "exec16"} <= set(w)

    def forward():
        h = policy._encode_kernels(g)
        s = policy._stage_scores_kernels(g, h)
        return h["node"].clone(), h["dag"].clone(), s.clone()
    wide = forward()
    for k in ("msg16", "update16", "stage16", "exec16"):  # without the images every stage takes the one-thread-per-row kernel
        del w[k]
    narrow = forward()
    for a, b, name in zip(wide, narrow, ("node embeddings", "job summaries", "stage scores")):
        fin = torch.isfinite(b)
        assert torch.equal(fin, torch.isfinite(a)), name
        assert float((a[fin] - b[fin]).abs().max()) <= 2e-5, name
    assert int(torch.isfinite(wide[2]).sum()) > 96  # schedulable stages were scored
    env.close()


def test_prefix_rows_kernel_matches_torch():
    """include/sss.h sss_prefix_rows on the GPU (256 cooperating threads per row) against torch.cumsum"""
    import ctypes as C

    from spark_sched_sim_amd.binding import load_library
    lib = load_library()
    lib.sss_prefix_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    g = torch.Generator().manual_seed(3)
    for B in (1, 255, 256, 4096, 5000):
        src = torch.randint(0, 2000, (B, 8), generator=g, dtype=torch.int32).cuda()
        mask = (torch.rand(B, generator=g) < 0.6).to(torch.uint8).cuda()
        off = torch.empty((3, B), dtype=torch.int64, device="cuda")
        cnt = torch.empty_like(off)
        tot = torch.empty(3, dtype=torch.int64, device="cuda")
        assert lib.sss_prefix_rows(src.data_ptr(), 1, src.stride(0), mask.data_ptr(), 3, B, off.data_ptr(), cnt.data_ptr(), tot.data_ptr(),
                                   torch.cuda.current_stream().cuda_stream) == 0
        want = (src[:, :3].long() * mask[:, None].long()).t()
        assert torch.equal(cnt, want) and torch.equal(off, torch.cumsum(want, 1) - want) and torch.equal(tot, want.sum(1))


@pytest.mark.gpu
def test_decima_step_with_another_current_device(pack):
    """the entry points without an env handle (sss_prefix_rows, sss_gnn_launch, sss_decima_sample, sss_decima_layer_lists)
    launch on the CURRENT device; their callers make the tensors' device current (binding.device_of). With two GPUs in
    one process: an env on cuda:1 stepped by a policy while cuda:0 is current."""
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs in one process")
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    cfg = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    torch.cuda.set_device(0)
    env = VecSparkSchedSimEnv(cfg, 64, device="cuda:1", pack=pack)
    torch.manual_seed(0)
    policy = DecimaPolicy(num_executors=10, **AGENT).to("cuda:1").eval()
    env.reset(seed=0)
    for _ in range(5):
        assert torch.cuda.current_device() == 0
        act, aux = policy.schedule_env(env)
        env.step(act)
    assert torch.isfinite(aux["lgprob"]).all() and int((env.obs_i32[:, 7] != 0).sum()) == 0
    env.close()


@pytest.mark.gpu
def test_matrix_core_gnn_equals_the_vector_unit_forms(tmp_path):
    """csrc/sss_gnn_mfma.h (DAG layers, node / job rows and both policy heads as chained v_mfma_f32_16x16x4_f32) against the
    vector-unit kernels of sss_gnn.h / sss_gnn16.h on the same recorded graphs and parameters: a child process runs the
    same script on a TEST build of the library compiled with -DSSS_TEST_VECTOR_FORMS (tests/gpu_variant.py; the formulation is
    fixed when a library is compiled - the product library has no run-time switch).
    Embeddings, job / observation summaries, stage and executor scores agree within 2e-5 (fp32, different summation order;
    the heads' tanh is exp / rcp based on the matrix-core path: ~1e-7)."""
    import os
    import subprocess
    import sys

    import numpy as np

    script = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[3])
import ctypes
from decima_util import AGENT
from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
from spark_sched_sim_amd.decima import DecimaPolicy
lib = ctypes.CDLL(sys.argv[4]) if len(sys.argv) > 4 else None
out = {}
for name, cfg, n_env, steps in (("c2", dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), 300, 120),
                                 ("e50", dict(num_executors=50, job_arrival_cap=60, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0), 37, 200)):
    env = VecSparkSchedSimEnv(cfg, n_env, device="cuda:0", pack=workload.default_pack(), auto_reset=True, _lib=lib)
    env.reset(seed=5)
    env.rollout("fair", steps)
    torch.manual_seed(9)
    pol = DecimaPolicy(num_executors=cfg["num_executors"], **AGENT).to("cuda:0").eval()
    with torch.no_grad():
        for n_, p_ in pol.named_parameters():
            if "bias" in n_:
                p_.normal_(0.0, 0.1)
    pol.bind_kernels(env._b)
    g = env.decima_graph(None)
    h = pol._encode_kernels(g)
    s = pol._stage_scores_kernels(g, h)
    jobs = torch.arange(0, g["job_obs"].numel(), 3, device="cuda:0")
    es = pol._exec_scores_kernels(g, h, jobs)
    for k, v in (("node", h["node"]), ("dag", h["dag"]), ("glob", h["glob"]), ("stage", s), ("exec", es)):
        out[name + "_" + k] = v.cpu().numpy()
    # the range sums on their own, on the same random hidden vectors in both builds: 16 lanes per row (sss_gnn16.h
    # sss_gnn_sum16_kernel) against one thread per row (sss_gnn.h) - the additions in the same order (multiply-adds may be fused differently)
    w = pol._packed_weights()
    gen = torch.Generator().manual_seed(11)
    M, J, B = g["x"].shape[0], g["job_obs"].numel(), g["n_obs"]
    tmp = torch.randn((max(M, J), 16), generator=gen).cuda()
    hd, hg = torch.empty((J, 16), device="cuda:0"), torch.empty((B, 16), device="cuda:0")
    pol._launch("dagsum", J, w["dag"], tmp=tmp, h_dag=hd, job_first=g["job_first"], job_nodes=g["job_nodes"])
    pol._launch("globsum", B, w["glob"], tmp=tmp, h_glob=hg, obs_job_off=g["obs_job_off"], obs_jobs=g["obs_jobs"])
    out[name + "_dagsum_alone"], out[name + "_globsum_alone"] = hd.cpu().numpy(), hg.cpu().numpy()
    env.close()
np.savez(sys.argv[1], **out)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    from gpu_variant import build_variant

    for tag, extra in (("mfma", []), ("valu", [build_variant("vecforms")])):
        path = str(tmp_path / (tag + ".npz"))
        subprocess.run([sys.executable, "-c", script, path, root, os.path.join(root, "tests")] + extra, check=True, timeout=900)
        res[tag] = dict(np.load(path))
    assert res["mfma"].keys() == res["valu"].keys() and len(res["mfma"]) == 14
    for k, a in res["mfma"].items():
        b = res["valu"][k]
        assert a.shape == b.shape and a.size > 0
        if k.endswith("_alone"):
            assert np.abs(a - b).max() <= 2e-6 * max(1.0, float(np.abs(b).max())), (k, float(np.abs(a - b).max()))
            continue
        fin = np.isfinite(b)
        assert (np.isfinite(a) == fin).all(), k  # (-inf marks slots that are not schedulable / executor counts beyond the cap)
        assert np.abs(a[fin] - b[fin]).max() <= 2e-5 * max(1.0, float(np.abs(b[fin]).max())), (k, float(np.abs(a[fin] - b[fin]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n_envs,steps", [
    (dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0), 512, 150),
    (dict(num_executors=50, job_arrival_cap=60, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0), 96, 120),
])
def test_decima_step_without_host_round_trip_equals_the_synchronous_one(cfg, n_envs, steps):
    """env.decima_graph_on_device + kernels that read their row counts from the device (sss_gnn_args::n_rows_dev,
    sss_gnn_encode_args::n_nodes_dev) against the path that reads the graph's totals back: same actions and log-probabilities"""
    from decima_util import check_on_device_step_equals_the_synchronous_one

    check_on_device_step_equals_the_synchronous_one("cuda:0", None, n_envs=n_envs, steps=steps, cfg=cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("n_envs", [1, 37, 300, 1024])
def test_list_pieces_of_the_graph_on_the_device(n_envs):
    """decima_util.check_pieces_on_device on the GPU: blocks of 1 / 2 / 10 / 32 envs per list counter"""
    from decima_util import check_pieces_on_device

    check_pieces_on_device("cuda:0", n_envs=n_envs, steps=120)


@pytest.mark.gpu
@pytest.mark.parametrize("n_envs,steps,variant", [(300, 25, None), (96, 400, None), (1024, 60, None), (96, 400, "obscap64")])
def test_layers_in_one_launch_equal_the_launches_per_layer(n_envs, steps, variant):
    """include/sss.h sss_gnn_encode_args.layers_mode: all DAG layers in one launch with a wave per observation (2) against a launch
    per layer (1) on the same live observations - small ones early in the episodes, a few hundred nodes later, exact-size graphs
    and the env's capacity graphs: the three encoder outputs are bit-identical (a row's arithmetic does not depend on which rows share
    its tile); the library's own choice (0) is one of the two"""
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import DecimaPolicy

    lib = None
    if variant is not None:  # (a test build of the library: observations with more than 64 list entries take the kernel's chunk-by-chunk path)
        from gpu_variant import load_variant
        lib = load_variant(variant)
    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, n_envs, device="cuda:0", auto_reset=True, _lib=lib)
    torch.manual_seed(3)
    policy = DecimaPolicy(num_executors=10, **AGENT).to("cuda:0").eval().bind_kernels(env._b)
    env.reset(seed=9)
    env.rollout("fair", steps)
    for make in (env.decima_graph, env.decima_graph_on_device):
        outs = {}
        for mode in (1, 2, 0):
            g = make()
            policy._layers_mode = mode
            h = policy._encode_kernels(g)
            M, J = (int(g["totals_dev"][0]), int(g["totals_dev"][2])) if "totals_dev" in g else (g["x"].shape[0], g["job_obs"].numel())
            outs[mode] = (h["node"][:M].clone(), h["dag"][:J].clone(), h["glob"].clone())
        assert outs[1][0].shape[0] > n_envs
        for mode in (2, 0):
            for a, b, name in zip(outs[1], outs[mode], ("node embeddings", "job summaries", "observation summaries")):
                assert torch.equal(a.view(torch.int32), b.view(torch.int32)), (make.__name__, mode, name)
    env.close()
