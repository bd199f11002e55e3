"""GPU leg of tests/test_emu_decima.py: the same fixtures with the simulator on the HIP path and
the Decima transform / GNN running as torch ops on the same device, consuming the env's
observation tensors in place."""
import pytest
import torch

from decima_util import AGENT, check_decima_fixture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,n_steps", [("decima_c1", 90), ("decima_e50", 90)])
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
