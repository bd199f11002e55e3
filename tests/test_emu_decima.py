"""the batched Decima observation transform and GNN forward against fixtures
recorded from the reference's own DecimaEnvWrapper / DecimaScheduler (tests/golden/make_decima_golden.py),
with the simulator running under the CPU wave emulator and the recorded actions replayed.
Node features and masks must be identical; scores agree to float32 round-off (different summation
order in the scatter-adds; tolerance 2e-5 absolute on O(1) scores)."""
import pytest

from decima_util import check_decima_fixture
from emu_util import load_emu


@pytest.mark.parametrize("name,n_steps", [("decima_c1", 90), ("decima_e50", 60)])
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
