"""GPU leg of tests/test_emu_training.py: same fixtures, simulator on the HIP path, Decima / PPO
tensors on the same device."""
import pytest
import torch

from training_util import check_async_pipeline, check_sync_pipeline

pytestmark = pytest.mark.gpu


def test_sync_rollouts_returns_baselines_loss_and_step_match_reference_gpu():
    check_sync_pipeline("cuda:0", None)


def test_async_rollouts_match_reference_gpu():
    check_async_pipeline("cuda:0", None)


def test_trainer_two_iterations_gpu(tmp_path):
    from decima_util import AGENT
    from spark_sched_sim_amd.training import Trainer

    train = dict(trainer_cls="PPO", num_iterations=2, num_sequences=4, num_rollouts=4, seed=42, checkpointing_freq=2,
                 num_epochs=3, num_batches=10, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04, beta_discount=5.0e-3,
                 opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5, artifacts_dir=str(tmp_path))
    env = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0,
               mean_time_limit=1.0e6)
    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), env, train, device="cuda:0")
    hist = tr.train(verbose=False)
    assert len(hist) == 2 and all(h["samples"] > 0 for h in hist)
    assert all(torch.isfinite(torch.tensor([h["policy loss"], h["entropy"], h["approx kl div"]])).all() for h in hist)
    assert (tmp_path / "checkpoints" / "2" / "model.pt").exists()
    tr.close()


def test_trainer_iteration_at_100_executors_gpu(tmp_path):
    """one PPO iteration with more than 64 executors (the reference's Decima is parameterised by num_executors throughout,
    schedulers/decima/scheduler.py:24,41,326-385): sampled actions on the wide simulator, the executor head over 100 counts,
    the update; the record made on the device equals the synchronous one at this size too"""
    from decima_util import AGENT
    from spark_sched_sim_amd.training import Trainer
    from training_util import check_record_on_device

    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=2, num_rollouts=3, seed=7, checkpointing_freq=50,
                 num_epochs=2, num_batches=4, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04, beta_discount=5.0e-3,
                 opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5, artifacts_dir=str(tmp_path))
    env = dict(num_executors=100, job_arrival_cap=12, job_arrival_rate=1.2e-4, moving_delay=2000.0, warmup_delay=1000.0,
               mean_time_limit=6.0e5)
    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), env, train, device="cuda:0")
    before = {k: v.clone() for k, v in tr.policy.state_dict().items()}
    hist = tr.train(verbose=False)
    assert len(hist) == 1 and hist[0]["samples"] > 0
    assert torch.isfinite(torch.tensor([hist[0]["policy loss"], hist[0]["entropy"], hist[0]["approx kl div"]])).all()
    assert any(not torch.equal(v, before[k]) for k, v in tr.policy.state_dict().items())
    tr.close()
    check_record_on_device("cuda:0", None, 6, num_executors=100)


def test_trainer_iteration_on_the_deep_trace_set_gpu(tmp_path):
    """one PPO iteration on the "deep" trace set (jobs of up to 40 stages, 12 DAG layers, thousands of tasks per stage): collection
    through the device-side record and the update on the matrix-core kernels; the record made on the device equals the synchronous one"""
    from decima_util import AGENT
    from spark_sched_sim_amd import workload
    from spark_sched_sim_amd.training import Trainer

    train = dict(trainer_cls="PPO", num_iterations=1, num_sequences=8, num_rollouts=4, seed=11, checkpointing_freq=50,
                 num_epochs=2, num_batches=4, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04, beta_discount=5.0e-3,
                 opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5, artifacts_dir=str(tmp_path))
    env = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=2.0e7)
    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), env, train, device="cuda:0", pack=workload.profile_pack("deep"))
    assert tr.env.graph_kernel_fits and tr.env.dims.stage_stride > 24
    before = {k: v.clone() for k, v in tr.policy.state_dict().items()}
    hist = tr.train(verbose=False)
    assert len(hist) == 1 and hist[0]["samples"] > 200
    assert torch.isfinite(torch.tensor([hist[0]["policy loss"], hist[0]["entropy"], hist[0]["approx kl div"]])).all()
    assert any(not torch.equal(v, before[k]) for k, v in tr.policy.state_dict().items())
    tr.close()


def test_train_gpu(tmp_path):
    """reference test/test_train.py:5-7 on the GPU"""
    from spark_sched_sim_amd.training import make_trainer
    from training_util import reference_smoke_test_config

    cfg = reference_smoke_test_config(str(tmp_path))
    tr = make_trainer(cfg, device="cuda:0")
    tr.train(verbose=False)
    assert tr.history[0]["samples"] > 0
    tr.close()


@pytest.mark.gpu
def test_linear_wgrad_kernel_matches_torch():
    """csrc/sss_train.h (MFMA f32 16x16x4, per-wave partials added in a fixed order) against torch's fp32 reference
    dy^T x / column sums at every feature-size combination of the published architecture, row counts that are not
    multiples of four, strided rows; and KernelLinear's backward against nn.Linear's"""
    import torch

    from spark_sched_sim_amd.train_kernels import KernelLinear, linear_wgrad

    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    for K in (1, 3, 4, 1000, 65537, 300001):
        for M, N in ((5, 32), (32, 16), (16, 16), (16, 32), (21, 32), (53, 64), (64, 64), (64, 1), (36, 64), (64, 33), (17, 49)):
            big = torch.randn((K, M + 3), generator=g, device=dev)
            x, dy = big[:, :M], torch.randn((K, N), generator=g, device=dev)
            gw, gb = linear_wgrad(x, dy)
            ref_w, ref_b = (dy.double().t() @ x.double()), dy.double().sum(0)
            tol = 2e-6 * (K ** 0.5) + 1e-5  # fp32 accumulation over K terms of unit variance
            assert (gw.double() - ref_w).abs().max().item() <= tol * max(1.0, ref_w.abs().max().item() ** 0.5), (K, M, N)
            assert (gb.double() - ref_b).abs().max().item() <= tol * max(1.0, ref_b.abs().max().item() ** 0.5), (K, M, N)
    # same run twice: bit-identical (fixed summation order, no atomics)
    a1, b1 = linear_wgrad(x, dy)
    a2, b2 = linear_wgrad(x, dy)
    assert torch.equal(a1, a2) and torch.equal(b1, b2)
    # through autograd
    torch.manual_seed(0)
    lin_k, lin_t = KernelLinear(21, 32).to(dev), torch.nn.Linear(21, 32).to(dev)
    lin_t.load_state_dict(lin_k.state_dict())
    xin = torch.randn((20000, 21), device=dev, requires_grad=True)
    xin2 = xin.detach().clone().requires_grad_(True)
    w = torch.randn((20000, 32), device=dev)
    (lin_k(xin) * w).sum().backward()
    (lin_t(xin2) * w).sum().backward()
    assert torch.allclose(xin.grad, xin2.grad, rtol=1e-5, atol=1e-5)
    assert torch.allclose(lin_k.weight.grad, lin_t.weight.grad, rtol=1e-4, atol=2e-3)
    assert torch.allclose(lin_k.bias.grad, lin_t.bias.grad, rtol=1e-4, atol=2e-3)


@pytest.mark.gpu
def test_mlp_kernels_match_autograd():
    _check_mlp_kernels_match_autograd()


def _check_mlp_kernels_match_autograd():
    """csrc/sss_train16.h (forward and backward of a whole MLP, 16 lanes per row) against torch in fp64 on the same
    parameters, for the five MLP shapes of the published architecture and row counts around the tile sizes; then
    `KernelMLP` (what `make_mlp` builds) through autograd against the plain nn.Sequential, and bit-identical repeats"""
    import torch

    from spark_sched_sim_amd.decima import make_mlp
    from spark_sched_sim_amd.train_kernels import KernelMLP, mlp_backward, mlp_forward, pack_mlp

    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    shapes = (((5, 32, 16, 16), "LeakyReLU", dict(negative_slope=0.2), 0, 0.2), ((16, 32, 16, 16), "LeakyReLU", dict(negative_slope=0.2), 0, 0.2),
              ((21, 32, 16, 16), "LeakyReLU", dict(negative_slope=0.2), 0, 0.2), ((53, 64, 64, 1), "Tanh", {}, 1, 0.0), ((36, 64, 64, 1), "Tanh", {}, 1, 0.0))
    for dims, act_cls, kw, act, slope in shapes:
        mlp = make_mlp(dims[0], [dims[1], dims[2]], dims[3], act_cls, kw).to(dev)
        assert isinstance(mlp, KernelMLP)
        ref = torch.nn.Sequential(*[type(m)(m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else type(m)(**kw) for m in mlp]).to(dev).double()
        ref.load_state_dict({k: v.double() for k, v in mlp.state_dict().items()})
        packed = pack_mlp(mlp[0], mlp[2], mlp[4])
        for rows in (1, 15, 16, 17, 4099, 70001):
            x = torch.randn((rows, dims[0]), device=dev)
            xr = x.double().requires_grad_(True)
            h1 = ref[1](ref[0](xr))
            h2 = ref[3](ref[2](h1))
            y_ref = ref[4](h2)
            dy = torch.randn((rows, dims[3]), device=dev)
            a1, a2, y = mlp_forward(x, packed, dims, act, slope)
            assert (y.double() - y_ref).abs().max().item() <= 2e-5 and (a1.double() - h1).abs().max().item() <= 1e-5 and (a2.double() - h2).abs().max().item() <= 1e-5
            g1, g2, dx = mlp_backward(dy, a1, a2, packed, dims, act, slope)
            y_ref.backward(dy.double())
            assert (dx.double() - xr.grad).abs().max().item() <= 2e-5 * max(1.0, xr.grad.abs().max().item()), (dims, rows)
            # g1 / g2 are the gradients w.r.t. the pre-activations: their products with the layer inputs are the weight gradients
            gw2_ref = ref[2].weight.grad
            assert ((g2.double().t() @ a1.double()) - gw2_ref).abs().max().item() <= 1e-4 * max(1.0, gw2_ref.abs().max().item()), (dims, rows)
            gw1_ref = ref[0].weight.grad
            assert ((g1.double().t() @ x.double()) - gw1_ref).abs().max().item() <= 1e-4 * max(1.0, gw1_ref.abs().max().item()), (dims, rows)
            ref.zero_grad()
            again = mlp_backward(dy, a1, a2, packed, dims, act, slope)
            assert torch.equal(again[0], g1) and torch.equal(again[1], g2) and torch.equal(again[2], dx)
        # through autograd: the module against the same layers evaluated one by one
        assert mlp._fused_spec()
        x = torch.randn((20000, dims[0]), device=dev, requires_grad=True)
        x2 = x.detach().clone().requires_grad_(True)
        w = torch.randn((20000, dims[3]), device=dev)
        (mlp(x) * w).sum().backward()
        got = {k: p.grad.clone() for k, p in mlp.named_parameters()}
        mlp.zero_grad()
        (torch.nn.Sequential.forward(mlp, x2) * w).sum().backward()
        assert torch.allclose(x.grad, x2.grad, rtol=1e-4, atol=1e-5)
        for k, p in mlp.named_parameters():
            assert torch.allclose(got[k], p.grad, rtol=1e-4, atol=5e-3), (dims, k)
        # the packed parameters follow an optimiser step
        before = mlp(x.detach().requires_grad_(True)).detach().clone()
        with torch.no_grad():
            for p in mlp.parameters():
                p.add_(0.01)
        after = mlp(x.detach().requires_grad_(True)).detach()
        assert not torch.allclose(before, after) and torch.allclose(after, torch.nn.Sequential.forward(mlp, x.detach()), rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
def test_message_passing_function_matches_the_tensor_op_form():
    """train_kernels._MessagePassFn (the node encoder's loop over the DAG layers as one autograd node on the MLP kernels)
    against the tensor-op form of `_NodeEncoder.forward` on the same recorded graph: embeddings, and through a scalar of
    them the gradients of every parameter of the encoder and of the node features' embedding"""
    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
    from spark_sched_sim_amd.decima import DecimaPolicy

    dev = torch.device("cuda:0")
    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 512, device="cuda:0", pack=workload.default_pack(), auto_reset=True)
    env.reset(seed=7)
    env.rollout("fair", 150)
    g = env.decima_graph(None)
    torch.manual_seed(3)
    pol = DecimaPolicy(num_executors=10, **AGENT).to(dev)
    enc = pol.encoder.node_encoder
    with torch.no_grad():  # (biases start at zero, scheduler.py:66-69: give them values so that their gradients are exercised)
        for n_, p_ in enc.named_parameters():
            if "bias" in n_:
                p_.normal_(0.0, 0.1)
    assert g["x"].shape[0] >= 8192 and int(g["obs_depth"].max()) >= 3
    # the orders the sums of the update rely on (train_kernels.segment_sum "sorted", _MessagePassFn): an observation's nodes are
    # stored job by job, its jobs together, a layer's edges receiver by receiver - also after a minibatch has been cut out
    from spark_sched_sim_amd.decima import graph_layers, select_observations
    sub = select_observations(g, torch.randperm(g["n_obs"], device=dev)[: g["n_obs"] // 2])
    for gg in (g, sub):
        nondecr = lambda v: bool((v[1:] >= v[:-1]).all())  # noqa: E731
        assert nondecr(gg["node_job"]) and nondecr(gg["job_obs"]) and nondecr(gg["node_obs"])
        for e, recv in graph_layers(gg):
            assert nondecr(gg["src"][e]) and nondecr(recv)
    w = torch.randn((g["x"].shape[0], 16), device=dev)
    out = {}
    for flag in (True, False):
        type(enc).KERNEL_MESSAGE_PASSING = flag
        pol.zero_grad()
        h = enc(g, per_obs_skip=False)
        assert enc._kernel_message_passing(enc.mlp_prep(g["x"])) == flag
        (h * w).sum().backward()
        out[flag] = (h.detach().clone(), {k: p.grad.clone() for k, p in enc.named_parameters()})
    type(enc).KERNEL_MESSAGE_PASSING = True
    assert torch.allclose(out[True][0], out[False][0], rtol=1e-4, atol=2e-5)
    for k in out[True][1]:
        a, b = out[True][1][k], out[False][1][k]
        assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * max(1.0, float(b.abs().max()))), (k, float((a - b).abs().max()), float(b.abs().max()))
    # the same graph with its edges listed in another order (a reference-format observation may do that): a receiver's messages
    # are no longer a range of rows - the encoder notices (decima.edges_grouped_by_source) and sums them the order-free way
    from spark_sched_sim_amd.decima import edges_grouped_by_source
    perm = torch.randperm(g["src"].numel(), device=dev)
    gs = {k: v for k, v in g.items() if k in ("x", "node_obs", "node_loc", "n_pad", "node_job", "sched_rank", "gen", "stage_mask", "job_obs", "job_cap",
                                               "job_first", "n_obs", "obs_nodes", "obs_jobs", "obs_depth")}
    gs.update(src=g["src"][perm].contiguous(), dst=g["dst"][perm].contiguous(), edge_obs=g["edge_obs"][perm].contiguous())
    assert edges_grouped_by_source(g) and not edges_grouped_by_source(gs)
    pol.zero_grad()
    h = enc(gs, per_obs_skip=False)
    (h * w).sum().backward()
    assert torch.allclose(h.detach(), out[False][0], rtol=1e-4, atol=2e-5)
    for k, p_ in enc.named_parameters():
        b = out[False][1][k]
        assert torch.allclose(p_.grad, b, rtol=2e-3, atol=2e-3 * max(1.0, float(b.abs().max()))), k
    env.close()


@pytest.mark.gpu
def test_collection_recorded_on_the_device_equals_the_synchronous_one_gpu():
    """training_util.check_record_on_device on the GPU (events, pinned rings, flags read up to four steps late), also with an
    arena that has to grow"""
    from training_util import check_record_on_device

    check_record_on_device("cuda:0", None, num_envs=64)
    check_record_on_device("cuda:0", None, num_envs=8, tiny_arena=True)


@pytest.mark.gpu
def test_collection_recorded_on_the_device_with_a_failing_env_gpu():
    """training_util.check_record_on_device_with_a_failing_env on the GPU: the error flag of the failing step is read late"""
    from training_util import check_record_on_device_with_a_failing_env

    check_record_on_device_with_a_failing_env("cuda:0", None, num_envs=32)


@pytest.mark.gpu
def test_bit_lists_match_nonzero():
    """sss_bit_lists_kernel on the GPU against nonzero (training_util.check_bit_lists), and `graph_layers` of a recorded graph
    built with it against the per-layer nonzero form"""
    from training_util import check_bit_lists

    from spark_sched_sim_amd import VecSparkSchedSimEnv, workload
    from spark_sched_sim_amd.decima import graph_layers

    check_bit_lists(None, "cuda:0")
    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 512, device="cuda:0", pack=workload.default_pack(), auto_reset=True)
    env.reset(seed=3)
    env.rollout("fair", 120)
    g = env.decima_graph(None)
    assert g["edge_layers"].numel() >= 8192
    got = graph_layers(dict(g))
    depth = int(g["obs_depth"].max())
    assert len(got) == depth >= 3
    for lvl, (e, recv) in enumerate(got):
        assert torch.equal(e, ((g["edge_layers"] >> lvl) & 1).nonzero(as_tuple=True)[0]) and torch.equal(recv, ((g["node_recv"] >> lvl) & 1).nonzero(as_tuple=True)[0])
    env.close()


@pytest.mark.gpu
def test_take_moves_rows_of_any_dtype():
    """decima._take (the minibatch cut of `select_observations` on the row gather kernel) against tensor indexing: int64 / int32 /
    float rows, NaN and negative-zero bit patterns included"""
    from spark_sched_sim_amd.decima import _take

    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(2)
    n, k = 50_000, 20_011
    idx = torch.randint(0, n, (k,), device=dev, generator=gen)
    x = torch.randn((n, 5), device=dev, generator=gen)
    x[::7, 2] = float("nan")
    x[1::7, 3] = -0.0
    for t in (x, torch.randint(-2 ** 62, 2 ** 62, (n,), device=dev, generator=gen), torch.randint(-2 ** 31, 2 ** 31 - 1, (n,), device=dev, generator=gen).int(),
              torch.randn((n, 16), device=dev, generator=gen), torch.rand(n, device=dev, generator=gen) < 0.5):
        got, want = _take(t, idx), t[idx]
        assert got.dtype == want.dtype and got.shape == want.shape
        assert torch.equal(got.view(torch.uint8).reshape(-1) if got.dtype != torch.bool else got, want.view(torch.uint8).reshape(-1) if want.dtype != torch.bool else want)


@pytest.mark.gpu
def test_fused_backward_with_weight_gradients():
    """sss_mlp_mfma_bwdw_kernel / sss_mlp_head_mfma_bwdw_kernel (backward + the six parameter gradients in one pass,
    csrc/sss_train16.h) against fp64 autograd for the three GNN-shaped MLPs and the two policy heads: row counts that are not multiples of 16 / 64 and exceed one grid pass, two calls adding up in one
    accumulator, and the same bits when repeated"""
    from spark_sched_sim_amd.decima import make_mlp
    from spark_sched_sim_amd.train_kernels import mlp_backward_wgrad, mlp_forward, mlp_wgrad_acc, mlp_wgrad_finish, pack_mlp

    dev = torch.device("cuda:0")
    torch.manual_seed(8)
    for in_dim, hid, out, act_cls, kw, act, slope in ((5, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2), 0, 0.2), (16, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2), 0, 0.2),
                                                     (21, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2), 0, 0.2),
                                                     # the two policy heads (sss_mlp_head_mfma_bwdw_kernel)
                                                     (53, [64, 64], 1, "Tanh", {}, 1, 0.0), (36, [64, 64], 1, "Tanh", {}, 1, 0.0)):
        dims = (in_dim, hid[0], hid[1], out)
        mlp = make_mlp(in_dim, hid, out, act_cls, kw).to(dev)
        with torch.no_grad():
            for lin in (mlp[0], mlp[2], mlp[4]):
                lin.bias.normal_(0.0, 0.1)
        ref = make_mlp(in_dim, hid, out, act_cls, kw).to(dev).double()
        ref.load_state_dict({k: v.double() for k, v in mlp.state_dict().items()})
        packed = pack_mlp(mlp[0], mlp[2], mlp[4])
        outs = []
        for rep in range(2):
            acc = mlp_wgrad_acc(in_dim, dev)
            ref.zero_grad()
            for n in (200_003, 1_037, 63):
                gen = torch.Generator(device=dev).manual_seed(100 + n)
                x = torch.randn((n, in_dim), device=dev, generator=gen)
                dy = torch.randn((n, out), device=dev, generator=gen)
                xr = x.double().requires_grad_(True)
                ref(xr).backward(dy.double())
                a1, a2, _ = mlp_forward(x, packed, dims, act, slope)
                dx = mlp_backward_wgrad(dy, x, a1, a2, packed, dims, slope, acc, act=act)
                assert torch.allclose(dx.double(), xr.grad, rtol=1e-4, atol=1e-5)
                assert mlp_backward_wgrad(dy, x, a1, a2, packed, dims, slope, torch.zeros_like(acc), want_dx=False, act=act) is None
            got = mlp_wgrad_finish(dims, acc)
            want = (ref[0].weight.grad, ref[0].bias.grad, ref[2].weight.grad, ref[2].bias.grad, ref[4].weight.grad, ref[4].bias.grad)
            for g_, w_ in zip(got, want):
                assert g_.shape == w_.shape and torch.allclose(g_.double(), w_, rtol=2e-4, atol=2e-3 * max(1.0, float(w_.abs().max()) * 1e-2)), (in_dim, float((g_.double() - w_).abs().max()))
            outs.append(got)
        assert all(torch.equal(a_, b_) for a_, b_ in zip(*outs))  # a fixed order of additions


@pytest.mark.gpu
def test_recomputed_hidden_activations_give_the_stored_forms_bits():
    """sss_mlp_mfma_bwdw_kernel<IN, true>: a forward pass that stores no hidden activations (a1 / a2 NULL) and a backward pass that
    computes them again from x - dx and the six parameter gradients must be BIT-identical to the stored-activation form, y too"""
    from spark_sched_sim_amd.decima import make_mlp
    from spark_sched_sim_amd.train_kernels import mlp_backward_wgrad, mlp_forward, mlp_recompute, mlp_wgrad_acc, mlp_wgrad_finish, pack_mlp

    dev = torch.device("cuda:0")
    torch.manual_seed(9)
    for in_dim in (5, 16, 21):
        assert mlp_recompute(in_dim)
        dims = (in_dim, 32, 16, 16)
        mlp = make_mlp(in_dim, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2)).to(dev)
        with torch.no_grad():
            for lin in (mlp[0], mlp[2], mlp[4]):
                lin.bias.normal_(0.0, 0.1)
        packed = pack_mlp(mlp[0], mlp[2], mlp[4])
        acc_s, acc_r = mlp_wgrad_acc(in_dim, dev), mlp_wgrad_acc(in_dim, dev)
        for n in (300_007, 1_037, 15):
            gen = torch.Generator(device=dev).manual_seed(300 + n)
            x = torch.randn((n, in_dim), device=dev, generator=gen)
            dy = torch.randn((n, 16), device=dev, generator=gen)
            a1, a2, y = mlp_forward(x, packed, dims, 0, 0.2)
            n1, n2, y2 = mlp_forward(x, packed, dims, 0, 0.2, keep_hidden=False)
            assert n1 is None and n2 is None and torch.equal(y, y2)
            dx_s = mlp_backward_wgrad(dy, x, a1, a2, packed, dims, 0.2, acc_s)
            dx_r = mlp_backward_wgrad(dy, x, None, None, packed, dims, 0.2, acc_r)
            assert torch.equal(dx_s, dx_r), (in_dim, n)
        for g_s, g_r in zip(mlp_wgrad_finish(dims, acc_s), mlp_wgrad_finish(dims, acc_r)):
            assert torch.equal(g_s, g_r), in_dim
        if in_dim == 21:  # the input rows in two pieces [x (5) | x2 (16)] (the DAG encoder; sss_mlp_split_supported): the same numbers - the
            # first Linear sums x2's features first (csrc/sss_train16.h MlpSeg), so equal up to the order of fp32 additions, not bit for bit
            from spark_sched_sim_amd.train_kernels import mlp_split
            assert mlp_split(21) and not mlp_split(16)
            acc_p, acc_j = mlp_wgrad_acc(in_dim, dev), mlp_wgrad_acc(in_dim, dev)
            for n in (300_007, 1_037, 15):
                gen = torch.Generator(device=dev).manual_seed(400 + n)
                x = torch.randn((n, in_dim), device=dev, generator=gen)
                dy = torch.randn((n, 16), device=dev, generator=gen)
                xa, xb = x[:, :5].contiguous(), x[:, 5:].contiguous()
                close = lambda u, v: torch.allclose(u, v, rtol=1e-5, atol=1e-5 * max(1.0, float(v.abs().max())))  # noqa: E731
                y_p = mlp_forward(xa, packed, dims, 0, 0.2, keep_hidden=False, x2=xb)[2]
                assert close(y_p, mlp_forward(x, packed, dims, 0, 0.2, keep_hidden=False)[2])
                assert torch.equal(y_p, mlp_forward(xa, packed, dims, 0, 0.2, keep_hidden=False, x2=xb)[2])
                dxb = mlp_backward_wgrad(dy, xa, None, None, packed, dims, 0.2, acc_p, x2=xb)
                assert close(dxb, mlp_backward_wgrad(dy, x, None, None, packed, dims, 0.2, acc_j)[:, 5:]), n
            for g_p, g_j in zip(mlp_wgrad_finish(dims, acc_p), mlp_wgrad_finish(dims, acc_j)):
                assert g_p.shape == g_j.shape and torch.allclose(g_p, g_j, rtol=1e-4, atol=1e-4 * max(1.0, float(g_j.abs().max())))
            # ... and through autograd: KernelMLP.forward_cat against the module on the concatenation
            xa = torch.randn((30_000, 5), device=dev)
            xb = torch.randn((30_000, 16), device=dev, requires_grad=True)
            xc = xb.detach().clone().requires_grad_(True)
            w = torch.randn((30_000, 16), device=dev)
            (mlp.forward_cat(xa, xb) * w).sum().backward()
            got = {k: p.grad.clone() for k, p in mlp.named_parameters()}
            mlp.zero_grad()
            (mlp(torch.cat([xa, xc], -1)) * w).sum().backward()
            assert torch.allclose(xb.grad, xc.grad, rtol=1e-5, atol=1e-5)
            for k, p in mlp.named_parameters():
                assert torch.allclose(got[k], p.grad, rtol=1e-4, atol=1e-4 * max(1.0, float(p.grad.abs().max()))), k


@pytest.mark.gpu
def test_segment_categorical_matches_the_tensor_op_forms():
    """sss_segcat_kernel (csrc/sss_segcat.h) against the tensor-op forms of evaluate_actions; then evaluate_actions itself with and
    without it on a recorded minibatch: log-probabilities, entropies and every parameter's gradient"""
    from training_util import check_segment_categorical

    from spark_sched_sim_amd.binding import Binding

    check_segment_categorical(Binding(), "cuda:0", n_seg=70_001)
    # evaluate_actions on a recorded graph (2048 observations, > 8192 schedulable stages), the kernel form against the tensor-op form
    from decima_util import AGENT
    from spark_sched_sim_amd import VecSparkSchedSimEnv, train_kernels, workload
    from spark_sched_sim_amd.decima import DecimaPolicy

    dev = torch.device("cuda:0")
    cfg = dict(num_executors=10, job_arrival_cap=50, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 2048, device="cuda:0", pack=workload.default_pack(), auto_reset=True)
    env.reset(seed=11)
    env.rollout("fair", 150)
    g = env.decima_graph(None)
    torch.manual_seed(5)
    pol = DecimaPolicy(num_executors=10, **AGENT).to(dev)
    a = pol.act(g, torch.Generator(device=dev).manual_seed(1), fresh_outputs=True)
    keep = a["any_stage"].nonzero(as_tuple=True)[0]
    from spark_sched_sim_amd.decima import select_observations
    sub = select_observations(g, keep)
    acts = [a[k][keep].long() for k in ("stage_sel", "job_idx", "exec_sel")]
    assert int(sub["stage_mask"].sum()) >= train_kernels.MIN_ROWS
    w1, w2 = torch.randn(keep.numel(), device=dev), torch.randn(keep.numel(), device=dev)
    out = {}
    for flag in (True, False):
        train_kernels.SEGMENT_CATEGORICAL = flag
        pol.zero_grad()
        res = pol.evaluate_actions(sub, *acts)
        (res["lgprobs"] * w1 + res["entropies"] * w2).sum().backward()
        out[flag] = (res["lgprobs"].detach().clone(), res["entropies"].detach().clone(), {k: p.grad.clone() for k, p in pol.named_parameters()})
    train_kernels.SEGMENT_CATEGORICAL = True
    assert torch.allclose(out[True][0], out[False][0], rtol=1e-5, atol=2e-5) and torch.allclose(out[True][1], out[False][1], rtol=1e-5, atol=2e-5)
    # (the sampled log-probabilities of pol.act are those of evaluate_actions on the same actions)
    assert torch.allclose(out[True][0], a["lgprob"][keep], rtol=1e-4, atol=1e-4)
    for k in out[True][2]:
        x, y = out[True][2][k], out[False][2][k]
        assert torch.allclose(x, y, rtol=2e-3, atol=2e-3 * max(1.0, float(y.abs().max()))), (k, float((x - y).abs().max()), float(y.abs().max()))
    env.close()


@pytest.mark.gpu
def test_record_kernels_match_the_tensor_op_forms():
    """sss_returns_kernel / sss_baseline_kernel (csrc/sss_returns.h) against the tensor-op forms, bit for bit"""
    from training_util import check_record_kernels

    check_record_kernels(None, "cuda:0")


@pytest.mark.gpu
def test_rows_kernels_match_torch_indexing():
    """`sss_rows_kernel` (csrc/sss_rows.h) against torch indexing, 16-byte and 4-byte forms, and the autograd functions built on
    it (`gather_rows`, `segment_sum`, `concat_rows`) against index_select / index_add_ / cat through autograd"""
    from training_util import check_rows_ops

    from spark_sched_sim_amd.binding import Binding
    from spark_sched_sim_amd.train_kernels import concat_rows, gather_rows, segment_sum

    dev = torch.device("cuda:0")
    check_rows_ops(Binding(), dev, n=100_003)
    torch.manual_seed(5)
    n, rows = 50_000, 9_000
    t1 = torch.randn((rows, 16), device=dev, requires_grad=True)
    t2 = torch.randn((rows // 3, 16), device=dev, requires_grad=True)
    x = torch.randn((rows, 5), device=dev)
    i1, i2 = torch.randint(0, rows, (n,), device=dev), torch.randint(0, rows // 3, (n,), device=dev)
    w = torch.randn((n, 37), device=dev)
    got = concat_rows([(x, i1), (t1, i1), (t2, i2)])
    want = torch.cat([x[i1], t1.index_select(0, i1), t2.index_select(0, i2)], -1)
    assert torch.equal(got, want)
    ga = torch.autograd.grad((got * w).sum(), (t1, t2))
    gb = torch.autograd.grad((want * w).sum(), (t1, t2))
    assert all(torch.allclose(a, b, rtol=1e-4, atol=1e-4) for a, b in zip(ga, gb))
    y = torch.randn((n, 16), device=dev, requires_grad=True)
    seg = torch.sort(torch.randint(0, rows, (n,), device=dev))[0]
    wv = torch.randn((rows, 16), device=dev)
    for mode in ("", "sorted"):
        s1, s2 = segment_sum(y, seg, rows, mode), torch.zeros((rows, 16), device=dev).index_add_(0, seg, y)
        assert torch.allclose(s1, s2, rtol=1e-5, atol=1e-5)
        assert torch.equal(torch.autograd.grad((s1 * wv).sum(), y)[0], torch.autograd.grad((s2 * wv).sum(), y)[0])
    assert torch.equal(segment_sum(y, seg, rows, "sorted"), segment_sum(y, seg, rows, "sorted"))  # a fixed order: same bits
    uq = torch.randperm(rows, device=dev)[: rows // 2]
    yu = torch.randn((rows // 2, 16), device=dev, requires_grad=True)
    assert torch.equal(segment_sum(yu, uq, rows, "unique"), torch.zeros((rows, 16), device=dev).index_copy_(0, uq, yu))
    g1 = gather_rows(t1, i1)
    assert torch.equal(g1, t1.index_select(0, i1))
    tu = torch.randn((rows, 16), device=dev, requires_grad=True)
    big = torch.randperm(rows, device=dev)[:8500]
    wu = torch.randn((8500, 16), device=dev)
    assert torch.equal(torch.autograd.grad((gather_rows(tu, big, unique=True) * wu).sum(), tu)[0], torch.autograd.grad((tu.index_select(0, big) * wu).sum(), tu)[0])


@pytest.mark.gpu
def test_mlp_kernels_16_lane_forms(monkeypatch):
    """the GNN-shaped training MLPs run on the matrix cores in the product library; their 16-lanes-per-row kernels (kept for
    comparisons) go through the same checks on a TEST build of the library compiled with -DSSS_TEST_VECTOR_FORMS
    (tests/gpu_variant.py): the formulation is fixed when a library is compiled, there is no run-time switch"""
    from gpu_variant import load_variant
    from spark_sched_sim_amd import train_kernels
    from spark_sched_sim_amd.binding import Binding

    monkeypatch.setattr(train_kernels, "_BINDING", Binding(load_variant("vecforms")))
    _check_mlp_kernels_match_autograd()
