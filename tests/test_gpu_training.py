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


def test_train_gpu(tmp_path):
    """reference test/test_train.py:5-7 on the GPU"""
    import os.path as osp

    import yaml

    from spark_sched_sim_amd.training import make_trainer

    with open(osp.join(osp.dirname(osp.abspath(__file__)), "test_train.yaml")) as stream:
        cfg = yaml.safe_load(stream)
    cfg["trainer"]["artifacts_dir"] = str(tmp_path)
    tr = make_trainer(cfg, device="cuda:0")
    tr.train(verbose=False)
    assert tr.history[0]["samples"] > 0
    tr.close()
