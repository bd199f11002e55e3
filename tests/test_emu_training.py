"""SURVEY 8(f) next-2 / next-3 under the CPU wave emulator: rollout collection (sync and async),
returns, baselines, evaluate_actions, the PPO CLIP loss and one optimiser step against values
recorded from the reference's own trainer code (tests/golden/make_ppo_golden.py)."""
import pytest

from emu_util import load_emu
from training_util import check_async_pipeline, check_sync_pipeline


def test_sync_rollouts_returns_baselines_loss_and_step_match_reference():
    check_sync_pipeline("cpu", load_emu())


def test_async_rollouts_match_reference():
    check_async_pipeline("cpu", load_emu())


TRAIN = dict(trainer_cls="PPO", num_iterations=2, num_sequences=2, num_rollouts=2, seed=42, checkpointing_freq=2,
             num_epochs=2, num_batches=3, clip_range=0.2, target_kl=0.01, entropy_coeff=0.04, beta_discount=5.0e-3,
             opt_cls="Adam", opt_kwargs=dict(lr=3.0e-4), max_grad_norm=0.5)
ENV = dict(num_executors=5, job_arrival_cap=6, job_arrival_rate=1.0e-4, moving_delay=1500.0, warmup_delay=500.0,
           mean_time_limit=3.0e5)


def test_trainer_runs_and_checkpoints(tmp_path):
    """two PPO iterations end to end (sampled Decima actions, sync rollouts, updates, checkpoint):
    parameters move, the bookkeeping is finite, the checkpoint loads back into a fresh policy"""
    import json

    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd.decima import DecimaPolicy
    from spark_sched_sim_amd.training import Trainer

    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), ENV, dict(TRAIN, artifacts_dir=str(tmp_path)), device="cpu", _lib=load_emu())
    before = {k: v.clone() for k, v in tr.policy.state_dict().items()}
    hist = tr.train(verbose=False)
    assert len(hist) == 2 and all(h["samples"] > 0 for h in hist)
    assert all(torch.isfinite(torch.tensor([h["policy loss"], h["entropy"], h["approx kl div"]])).all() for h in hist)
    assert any(not torch.equal(v, before[k]) for k, v in tr.policy.state_dict().items())
    sd = torch.load(str(tmp_path / "checkpoints" / "2" / "model.pt"))
    DecimaPolicy(num_executors=5, **AGENT).load_state_dict(sd)
    assert "avg_num_jobs" in json.load(open(str(tmp_path / "checkpoints" / "2" / "state.json")))
    tr.close()


def test_trainer_iteration_at_100_executors(tmp_path):
    """the trainer with more than 64 executors (wide simulator, executor head over 100 counts): one iteration, parameters move"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd.training import Trainer

    env = dict(num_executors=100, job_arrival_cap=5, job_arrival_rate=1.5e-4, moving_delay=1500.0, warmup_delay=500.0, mean_time_limit=2.0e5)
    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), env, dict(TRAIN, num_iterations=1, artifacts_dir=str(tmp_path)), device="cpu", _lib=load_emu())
    before = {k: v.clone() for k, v in tr.policy.state_dict().items()}
    hist = tr.train(verbose=False)
    assert len(hist) == 1 and hist[0]["samples"] > 0
    assert torch.isfinite(torch.tensor([hist[0]["policy loss"], hist[0]["entropy"], hist[0]["approx kl div"]])).all()
    assert any(not torch.equal(v, before[k]) for k, v in tr.policy.state_dict().items())
    tr.close()


def test_trainer_iteration_on_the_deep_trace_set(tmp_path):
    """the trainer on the second trace regime (workload.PROFILES["deep"]: jobs of up to 40 stages and 12 DAG layers - more stage slots
    per job than the graph kernel's old limit of 24, which now is the templates' DAG depth): one iteration through the device-side
    record, parameters move"""
    import torch

    from decima_util import AGENT
    from spark_sched_sim_amd import workload
    from spark_sched_sim_amd.training import Trainer

    env = dict(num_executors=10, job_arrival_cap=4, job_arrival_rate=1.0e-4, moving_delay=2000.0, warmup_delay=1000.0, mean_time_limit=6.0e5)
    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), env, dict(TRAIN, num_iterations=1, artifacts_dir=str(tmp_path)), device="cpu", _lib=load_emu(),
                 pack=workload.profile_pack("deep"))
    assert tr.env.graph_kernel_fits and tr.env.dims.stage_stride > 24
    before = {k: v.clone() for k, v in tr.policy.state_dict().items()}
    hist = tr.train(verbose=False)
    assert len(hist) == 1 and hist[0]["samples"] > 0 and hist[0]["env_errors"] == 0
    assert torch.isfinite(torch.tensor([hist[0]["policy loss"], hist[0]["entropy"], hist[0]["approx kl div"]])).all()
    assert any(not torch.equal(v, before[k]) for k, v in tr.policy.state_dict().items())
    tr.close()


def test_train(tmp_path):
    """the reference's one integration test (reference test/test_train.py:5-7): load the YAML,
    `make_trainer(cfg).train()`, pass if nothing raises - same configuration, on the batched env"""
    from spark_sched_sim_amd.training import make_trainer
    from training_util import reference_smoke_test_config

    cfg = reference_smoke_test_config(str(tmp_path))
    tr = make_trainer(cfg, device="cpu", _lib=load_emu())
    tr.train(verbose=False)
    assert tr.history[0]["samples"] > 0
    tr.close()


def test_trainer_async_rollouts(tmp_path):
    """`rollout_duration` in the trainer config selects the asynchronous collection mode
    (trainer.py:60, rollout_worker.py:162-206): fixed simulated time per iteration, episodes
    restarting in place, env state carried over between iterations"""
    from decima_util import AGENT
    from spark_sched_sim_amd.training import Trainer

    tr = Trainer(dict(AGENT, agent_cls="DecimaScheduler"), ENV, dict(TRAIN, artifacts_dir=str(tmp_path), rollout_duration=1.0e5),
                 device="cpu", _lib=load_emu())
    hist = tr.train(verbose=False)
    assert len(hist) == 2 and all(h["samples"] > 0 for h in hist)
    assert int(tr.collector.reset_count.min()) >= 1
    tr.close()


def test_collector_truncates_a_rollout_whose_env_stalls(pack=None):
    """on_env_error="truncate": tests/golden/stall_case.json (valid actions, the reference raises
    AssertionError('[step]') at step 110) replayed through the collector: the rollout keeps its 110
    good steps, the failing one is not recorded, the other env is unaffected; "raise" raises."""
    import json
    import os.path as osp

    import pytest
    import torch

    from golden_util import GOLDEN_DIR
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.training import RolloutCollector, discounted_returns, sequence_baselines

    c = json.load(open(osp.join(GOLDEN_DIR, "stall_case.json")))
    cfg = {k: v for k, v in c["env_cfg"].items() if k != "mean_time_limit"}

    def replay(g, counts):  # env 0 replays the recorded actions, env 1 always takes (stage 0, 1 executor)
        t = int(counts[0]) if int(counts[0]) < len(c["stage_idx"]) else len(c["stage_idx"]) - 1
        z = torch.zeros(2, dtype=torch.long)
        return {"stage_sel": torch.tensor([c["stage_idx"][t], 0]), "job_idx": z, "exec_sel": torch.tensor([c["num_exec"][t] - 1, 0]),
                "lgprob": torch.zeros(2), "any_stage": torch.ones(2, dtype=torch.bool)}

    for mode in ("truncate", "raise"):
        env = VecSparkSchedSimEnv(cfg, 2, device="cpu", _lib=load_emu())
        col = RolloutCollector(env, c["env_cfg"]["mean_time_limit"], [c["seed"], c["seed"]], 1, 10, act_fn=replay, on_env_error=mode)
        # same seed => same sampled time limit as the recorded episode (StochasticTimeLimit's rule)
        if mode == "raise":
            with pytest.raises(RuntimeError, match="simulation stalled") as ei:
                col.collect_sync(with_stats=False)
            assert ei.value.case["stage_idx"] == c["stage_idx"] and ei.value.case["seed"] == c["seed"]
        else:
            ro = col.collect_sync(with_stats=False)
            assert float(col.tl_env.time_limit[0]) == c["time_limit"]
            assert int(ro.lengths[0]) == c["error_step"] and col.env_errors == 1 and int(ro.lengths[1]) > 0
            ret = discounted_returns(ro, cfg["beta"])
            base = sequence_baselines(ro, ret, 1, 2)
            assert torch.isfinite(ret).all() and torch.isfinite(base).all()
        env.close()


def test_two_envs_failing_in_the_same_step_name_a_real_env():
    """flags[3] ("1 + index of a failed env") took an OR over the failing envs - two envs failing in the
    same step decoded to an env that did not fail (or to one beyond the batch). It is a maximum now: envs 2 and 3 both replay
    tests/golden/stall_case.json, the report names env 3 with env 3's seed and action history and the stalled-simulation code."""
    import json
    import os.path as osp

    import torch

    from golden_util import GOLDEN_DIR
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.training import RolloutCollector

    c = json.load(open(osp.join(GOLDEN_DIR, "stall_case.json")))
    cfg = {k: v for k, v in c["env_cfg"].items() if k != "mean_time_limit"}
    B = 4

    def replay(g, counts):  # envs 2 and 3 replay the recorded actions, envs 0 and 1 always take (stage 0, 1 executor)
        t = min(int(counts[2]), len(c["stage_idx"]) - 1)
        z = torch.zeros(B, dtype=torch.long)
        return {"stage_sel": torch.tensor([0, 0, c["stage_idx"][t], c["stage_idx"][t]]), "job_idx": z,
                "exec_sel": torch.tensor([0, 0, c["num_exec"][t] - 1, c["num_exec"][t] - 1]), "lgprob": torch.zeros(B),
                "any_stage": torch.ones(B, dtype=torch.bool)}

    env = VecSparkSchedSimEnv(cfg, B, device="cpu", _lib=load_emu())
    col = RolloutCollector(env, c["env_cfg"]["mean_time_limit"], [c["seed"] + 1, c["seed"] + 2, c["seed"], c["seed"]], 1, 10, act_fn=replay,
                           on_env_error="raise")
    with pytest.raises(RuntimeError, match=r"env 3 .*simulation stalled") as ei:
        col.collect_sync(with_stats=False)
    assert ei.value.case["seed"] == c["seed"] and ei.value.case["stage_idx"] == c["stage_idx"] and ei.value.case["code"] == 5
    assert col.env_errors == 2
    env.close()


def test_linear_wgrad_entry_point_on_the_host_backend():
    """include/sss.h sss_linear_wgrad through the emulator library's host implementation: the argument plumbing of
    spark_sched_sim_amd.train_kernels.linear_wgrad (strided rows, bias on / off, error codes)"""
    import torch

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding
    from spark_sched_sim_amd.train_kernels import linear_wgrad

    b = Binding(load_emu())
    g = torch.Generator().manual_seed(3)
    for K, M, N in ((1, 5, 1), (37, 21, 16), (1000, 53, 64), (513, 64, 64)):
        big = torch.randn((K, M + 3), generator=g)
        x, dy = big[:, :M], torch.randn((K, N), generator=g)  # x: rows strided
        gw, gb = linear_wgrad(x, dy, binding=b)
        assert torch.allclose(gw, dy.t() @ x, rtol=1e-5, atol=1e-4) and torch.allclose(gb, dy.sum(0), rtol=1e-5, atol=1e-4)
        gw2, none = linear_wgrad(x, dy, want_bias=False, binding=b)
        assert none is None and torch.equal(gw2, gw)
    with pytest.raises(ValueError):
        linear_wgrad(torch.zeros((4, 65)), torch.zeros((4, 3)), binding=b)


def test_mlp_entry_points_on_the_host_backend():
    """include/sss.h sss_mlp_forward / sss_mlp_backward through the emulator library's host implementation against
    autograd on the same MLP: the argument plumbing of spark_sched_sim_amd.train_kernels.mlp_forward / mlp_backward
    (packed parameters, activations handed from forward to backward, optional dx) and the shape filter"""
    import torch

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding
    from spark_sched_sim_amd.decima import make_mlp
    from spark_sched_sim_amd.train_kernels import linear_wgrad, mlp_backward, mlp_forward, pack_mlp

    b = Binding(load_emu())
    torch.manual_seed(4)
    for dims, act_cls, kw, act, slope in (((5, 32, 16, 16), "LeakyReLU", dict(negative_slope=0.2), 0, 0.2), ((16, 32, 16, 16), "LeakyReLU", dict(negative_slope=0.2), 0, 0.2),
                                          ((21, 32, 16, 16), "LeakyReLU", dict(negative_slope=0.2), 0, 0.2), ((53, 64, 64, 1), "Tanh", {}, 1, 0.0),
                                          ((36, 64, 64, 1), "Tanh", {}, 1, 0.0)):
        assert b.lib.sss_mlp_supported(*dims, act) == 1
        mlp = make_mlp(dims[0], [dims[1], dims[2]], dims[3], act_cls, kw)
        x = torch.randn((37, dims[0]), requires_grad=True)
        y_ref = mlp(x)
        dy = torch.randn_like(y_ref)
        y_ref.backward(dy)
        packed = pack_mlp(mlp[0], mlp[2], mlp[4])
        a1, a2, y = mlp_forward(x.detach(), packed, dims, act, slope, binding=b)
        assert torch.allclose(y, y_ref.detach(), rtol=1e-5, atol=1e-5)
        g1, g2, dx = mlp_backward(dy, a1, a2, packed, dims, act, slope, binding=b)
        assert torch.allclose(dx, x.grad, rtol=1e-4, atol=1e-5)
        for lin, xin, g in ((mlp[4], a2, dy), (mlp[2], a1, g2), (mlp[0], x.detach(), g1)):
            gw, gb = linear_wgrad(xin, g, binding=b)
            assert torch.allclose(gw, lin.weight.grad, rtol=1e-4, atol=1e-4) and torch.allclose(gb, lin.bias.grad, rtol=1e-4, atol=1e-4)
        assert mlp_backward(dy, a1, a2, packed, dims, act, slope, want_dx=False, binding=b)[2] is None
    assert b.lib.sss_mlp_supported(7, 32, 16, 16, 0) == 0 and b.lib.sss_mlp_supported(16, 32, 16, 16, 1) == 0
    with pytest.raises(ValueError):
        mlp_forward(torch.zeros((4, 7)), torch.zeros(2000), (7, 32, 16, 16), 0, 0.2, binding=b)


def test_fused_backward_entry_points_on_the_host_backend():
    """include/sss.h sss_mlp_backward_wgrad / sss_mlp_wgrad_finish through the emulator library's host loops against autograd on
    the same MLP, with the parameter gradients of two calls adding up in one accumulator (what the layers of the message passing
    do): the argument plumbing of train_kernels.mlp_wgrad_acc / mlp_backward_wgrad / mlp_wgrad_finish and the shape filter"""
    import torch

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding
    from spark_sched_sim_amd.decima import make_mlp
    from spark_sched_sim_amd.train_kernels import mlp_backward_wgrad, mlp_forward, mlp_wgrad_acc, mlp_wgrad_finish, pack_mlp

    b = Binding(load_emu())
    torch.manual_seed(6)
    for in_dim, hid, out, act_cls, kw, act, slope in ((5, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2), 0, 0.2), (16, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2), 0, 0.2),
                                                     (21, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2), 0, 0.2), (53, [64, 64], 1, "Tanh", {}, 1, 0.0),
                                                     (36, [64, 64], 1, "Tanh", {}, 1, 0.0)):  # (the two policy heads: stored activations)
        dims = (in_dim, hid[0], hid[1], out)
        mlp = make_mlp(in_dim, hid, out, act_cls, kw)
        packed = pack_mlp(mlp[0], mlp[2], mlp[4])
        acc = mlp_wgrad_acc(in_dim, "cpu", binding=b)
        xs = [torch.randn((n, in_dim), requires_grad=True) for n in (37, 5)]
        for x in xs:
            y = mlp(x)
            dy = torch.randn_like(y)
            y.backward(dy)
            a1, a2, _ = mlp_forward(x.detach(), packed, dims, act, slope, binding=b)
            dx = mlp_backward_wgrad(dy, x.detach(), a1, a2, packed, dims, slope, acc, binding=b, act=act)
            assert torch.allclose(dx, x.grad, rtol=1e-4, atol=1e-5)
        got = mlp_wgrad_finish(dims, acc, binding=b)
        want = (mlp[0].weight.grad, mlp[0].bias.grad, mlp[2].weight.grad, mlp[2].bias.grad, mlp[4].weight.grad, mlp[4].bias.grad)
        for g_, w_ in zip(got, want):
            assert g_.shape == w_.shape and torch.allclose(g_, w_, rtol=1e-4, atol=1e-4)
    assert mlp_wgrad_acc(7, "cpu", binding=b) is None
    with pytest.raises(ValueError):  # a head without its stored activations: refused (only the GNN-shaped MLPs are recomputed)
        mlp_backward_wgrad(torch.zeros((4, 1)), torch.zeros((4, 53)), None, None, torch.zeros(8000), (53, 64, 64, 1), 0.0, mlp_wgrad_acc(53, "cpu", binding=b), binding=b, act=1)


def test_recompute_entry_points_on_the_host_backend():
    """include/sss.h sss_mlp_recompute_supported: sss_mlp_forward without a1 / a2 and sss_mlp_backward_wgrad recomputing them -
    the same numbers as with stored activations (argument plumbing; the gfx950 kernels' bit identity is a -m gpu test)"""
    import torch

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding
    from spark_sched_sim_amd.decima import make_mlp
    from spark_sched_sim_amd.train_kernels import mlp_backward_wgrad, mlp_forward, mlp_recompute, mlp_wgrad_acc, mlp_wgrad_finish, pack_mlp

    b = Binding(load_emu())
    torch.manual_seed(16)
    assert not b.lib.sss_mlp_recompute_supported(53)
    for in_dim in (5, 16, 21):
        assert mlp_recompute(in_dim, binding=b)
        dims = (in_dim, 32, 16, 16)
        mlp = make_mlp(in_dim, [32, 16], 16, "LeakyReLU", dict(negative_slope=0.2))
        packed = pack_mlp(mlp[0], mlp[2], mlp[4])
        x, dy = torch.randn((41, in_dim)), torch.randn((41, 16))
        a1, a2, y = mlp_forward(x, packed, dims, 0, 0.2, binding=b)
        n1, n2, y2 = mlp_forward(x, packed, dims, 0, 0.2, binding=b, keep_hidden=False)
        assert n1 is None and n2 is None and torch.equal(y, y2)
        acc_s, acc_r = mlp_wgrad_acc(in_dim, "cpu", binding=b), mlp_wgrad_acc(in_dim, "cpu", binding=b)
        assert torch.equal(mlp_backward_wgrad(dy, x, a1, a2, packed, dims, 0.2, acc_s, binding=b), mlp_backward_wgrad(dy, x, None, None, packed, dims, 0.2, acc_r, binding=b))
        for g_s, g_r in zip(mlp_wgrad_finish(dims, acc_s, binding=b), mlp_wgrad_finish(dims, acc_r, binding=b)):
            assert torch.equal(g_s, g_r)
        # the input rows in two pieces (sss_mlp_split_supported: the DAG encoder's 21-wide MLP): the same y, the last 16 columns of dx,
        # the same parameter gradients; refused for the other widths and together with stored activations
        from spark_sched_sim_amd.train_kernels import mlp_split
        assert mlp_split(in_dim, binding=b) == (in_dim == 21)
        if in_dim == 21:
            xa, xb = x[:, :5].contiguous(), x[:, 5:].contiguous()
            assert torch.equal(mlp_forward(xa, packed, dims, 0, 0.2, binding=b, keep_hidden=False, x2=xb)[2], y)
            acc_p = mlp_wgrad_acc(in_dim, "cpu", binding=b)
            dxb = mlp_backward_wgrad(dy, xa, None, None, packed, dims, 0.2, acc_p, binding=b, x2=xb)
            acc_j = mlp_wgrad_acc(in_dim, "cpu", binding=b)
            assert torch.equal(dxb, mlp_backward_wgrad(dy, x, None, None, packed, dims, 0.2, acc_j, binding=b)[:, 5:])
            for g_p, g_j in zip(mlp_wgrad_finish(dims, acc_p, binding=b), mlp_wgrad_finish(dims, acc_j, binding=b)):
                assert torch.equal(g_p, g_j)
            assert mlp_backward_wgrad(dy, xa, None, None, packed, dims, 0.2, acc_p, binding=b, x2=xb, want_dx=False) is None
        else:
            import ctypes

            from spark_sched_sim_amd.train_kernels import _mlp_args
            a = _mlp_args(dims, 0, 0.2, 41, packed, x=x, y=torch.empty((41, 16)), x2=torch.zeros((41, 16)))
            assert b.lib.sss_mlp_forward(ctypes.byref(a), None) == -31


def test_rows_entry_point_on_the_host_backend():
    """include/sss.h sss_rows_op through the emulator library's host implementation against torch indexing: the argument
    plumbing of spark_sched_sim_amd.train_kernels.rows_op (the four operations, a list side that is a column slice, widths that
    are and are not multiples of four, error codes)"""
    from training_util import check_rows_ops

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding

    check_rows_ops(Binding(load_emu()), "cpu", n=300)


def test_segment_categorical_entry_point_on_the_host_backend():
    """include/sss.h sss_segment_categorical through the emulator library's host loop (the per-segment function the gfx950 kernel runs,
    csrc/sss_segcat.h) against the tensor-op forms of evaluate_actions"""
    from training_util import check_segment_categorical

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding

    check_segment_categorical(Binding(load_emu()), "cpu", n_seg=500)


def test_collection_recorded_on_the_device_equals_the_synchronous_one():
    """training_util.check_record_on_device on the emulator library"""
    from training_util import check_record_on_device

    from emu_util import load_emu

    check_record_on_device("cpu", load_emu(), num_envs=6)


def test_collection_recorded_on_the_device_with_a_failing_env():
    """training_util.check_record_on_device_with_a_failing_env on the emulator library"""
    from training_util import check_record_on_device_with_a_failing_env

    from emu_util import load_emu

    check_record_on_device_with_a_failing_env("cpu", load_emu())


def test_graph_arena_grows():
    """`GraphArena` with a starting capacity far too small for the collection: the headroom rule makes it grow (with the
    cursors the host has seen), the record is still the one of the synchronous loop"""
    from training_util import check_record_on_device

    from emu_util import load_emu

    check_record_on_device("cpu", load_emu(), num_envs=3, tiny_arena=True)


def test_bit_lists_under_the_emulator():
    """the layer-list kernel (csrc/sss_decima.h sss_bit_lists_kernel, a wave per chunk) under the wave emulator against nonzero"""
    from training_util import check_bit_lists

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding

    check_bit_lists(Binding(load_emu()), "cpu")


def test_record_kernels_on_the_host_backend():
    """include/sss.h sss_discounted_returns / sss_sequence_baselines through the emulator library's host loops against the
    tensor-op forms (training_util.check_record_kernels)"""
    from training_util import check_record_kernels

    from emu_util import load_emu
    from spark_sched_sim_amd.binding import Binding

    check_record_kernels(Binding(load_emu()), "cpu")


def test_two_groups_of_envs_record_what_one_group_records():
    """`RolloutCollector(groups=2)` - the envs take their steps in two alternating groups (on two streams on the GPU), the
    flags of a group's step are read one step late - against `groups=1`: the same record per env, and every sample's
    observation id (`Rollouts.sample_ids`) points at the same observation of the recorded graph"""
    import torch

    from emu_util import load_emu
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.training import RolloutCollector
    from training_util import counter_act_fn

    cfg = dict(num_executors=10, job_arrival_cap=8, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    res = {}
    for groups in (1, 2):
        for mode in ("sync", "async"):
            env = VecSparkSchedSimEnv(cfg, 5, device="cpu", auto_reset=False, _lib=load_emu())
            col = RolloutCollector(env, 5.0e5, [11, 12, 13, 14, 15], seed_step=5, num_executors=10, act_fn=counter_act_fn, groups=groups)
            ro = col.collect_sync(with_stats=False) if mode == "sync" else col.collect_async(2.0e5, with_stats=False)
            ids = ro.sample_ids()
            res[(groups, mode)] = ([ro.rollout(b) for b in range(5)], ro.graph["obs_nodes"][ids], ro.graph["obs_jobs"][ids], ro.flat(ro.rewards))
            env.close()
    for mode in ("sync", "async"):
        one, two = res[(1, mode)], res[(2, mode)]
        for ra, rb in zip(one[0], two[0]):
            assert ra.keys() == rb.keys() and all((ra[k] == rb[k]).all() and ra[k].shape == rb[k].shape for k in ra)
        assert one[1].numel() > 50 and all(torch.equal(a, b) for a, b in zip(one[1:], two[1:]))


def test_sum_order_of_the_baseline_mean_is_numpys():
    """`training.numpy_sum_order` (and with it `baseline_pairwise` of csrc/sss_returns.h, which the record-kernel test compares with
    it bit for bit) against numpy itself: `y_hat.mean()` over a strided column, as baselines.py:33 takes it - sequential below 8
    rollouts per sequence, pairwise with eight partial sums from 8 on, halved above 128"""
    import numpy as np

    from spark_sched_sim_amd.training import numpy_sum_order

    rng = np.random.default_rng(5)
    for R in (1, 2, 3, 4, 7, 8, 9, 10, 16, 17, 33, 128, 129, 130, 140, 300):
        for _ in range(40):
            M = rng.standard_normal((R, 13)) * 10.0 ** rng.integers(-3, 6)
            for col in M.T[:4]:
                assert np.float64(numpy_sum_order([np.float64(v) for v in col])) / R == col.mean(), R


def test_minibatch_cut_by_ranges_equals_the_general_cut():
    """decima.select_observations: a record stored observation by observation is cut by ranges with arithmetic re-labelling (round 6)
    - the same sub-graph, field by field, as the general form (arena-sized look-up tables), for selections in any order"""
    import torch

    from emu_util import load_emu
    from spark_sched_sim_amd import VecSparkSchedSimEnv
    from spark_sched_sim_amd.decima import concat_graphs, select_observations

    cfg = dict(num_executors=10, job_arrival_cap=20, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
    env = VecSparkSchedSimEnv(cfg, 6, device="cpu", auto_reset=True, _lib=load_emu())
    env.reset(seed=30)
    graphs = []
    for _ in range(5):
        env.rollout("fair", 37)
        graphs.append({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in env.decima_graph().items() if not k.startswith("_")})
    g = concat_graphs(graphs)
    gen = torch.Generator().manual_seed(3)
    for n in (1, 7, g["n_obs"]):
        idx = torch.randperm(g["n_obs"], generator=gen)[:n]
        a = select_observations(dict(g), idx)                       # ranges (checked once per graph, cached in the dict)
        b = select_observations(dict(g, _ranges=False), idx)        # the general form
        assert set(a) == set(b)
        for k in a:
            if isinstance(a[k], torch.Tensor):
                assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
            else:
                assert a[k] == b[k], k
    env.close()
