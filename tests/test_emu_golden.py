"""The HIP kernel SOURCE (spark_sched_sim_amd/csrc/sss_sim.h), compiled for the host against the
CPU wave emulator (tests/emu) and driven through the real C ABI + Python host, replayed against
the reference's recorded trajectories. This is the no-GPU stand-in for tests/test_gpu_parity.py:
same code path above the launch, same kernel source below it, 64 emulated lanes per env.
It does not replace the GPU run (the emulator is not the product and is never loaded by it)."""
import pytest

from emu_util import load_emu
from replay_util import replay_golden

CASES = [
    ("tiny_hash", list(range(12)) + [21, 22], None),   # 21, 22: the reference's "[step]" assertion
    ("tiny_fair_tlimit", list(range(8)), None),
    ("c1_fair", [1234, 0], None),
    ("c1_hash", [100], None),
    ("bige_hash", [0], 300),
    ("testyaml_fair", [3], None),
    ("c3_fair", [0], 250),
    # more than 64 executors: the wide instantiation of the kernels (csrc/sss_sim.h with SSS_WIDE, tests/emu/emu_wide.cpp)
    ("e100_fair", [0], None),
    ("e100_hash", [2], 600),
    ("q5s2_fair", [0, 1], None),         # a trace set of 5 queries x 2 sizes (the pack header carries the shape)
    ("q5s2_hash", [3], None),
    ("e120_hash", [0, 1, 2, 3], None),   # 120 executors on <= 6 jobs: more than 100 local executors, and all 120 (tpch.py:258-260)
    # the "deep" trace regime (workload.PROFILES: <= 40 stages, in-degree <= 6 over all predecessors, <= 3000 tasks per stage, a 60 MB
    # pack): reference recordings at 10 / 50 / 100 executors; the whole episodes run in tests/test_gpu_parity.py
    ("deep_c1_fair", [0], 150),
    ("deep_c1_hash", [2], 150),
    ("deep_c1_fifo", [3], 150),
    ("deep_e50_fair", [0], 150),
    ("deep_e50_hash", [1], 150),
    ("deep_e100_fair", [0], 150),
    ("deep_e100_hash", [1], 150),
    ("deep_tlimit_hash", [5, 6], None),   # a time limit instead of a job cap: truncated in the middle of long stages
]


@pytest.mark.parametrize("name,seeds,max_steps", CASES)
def test_kernel_source_matches_reference_under_emulation(name, seeds, max_steps, pack):
    bad = replay_golden(name, seeds, pack, device="cpu", lib=load_emu(), full_obs_steps=10, max_steps=max_steps)
    assert not bad, "\n".join(bad[:10])


def test_discounted_rewards_beta_on_the_deep_trace_set(pack):
    bad = replay_golden("deep_c1_fair_beta", [4], pack, device="cpu", lib=load_emu(), reward_rtol=1e-12, max_steps=120)
    assert not bad, "\n".join(bad[:10])


def test_discounted_rewards_beta(pack):
    """beta > 0 (reference spark_sched_sim.py:865-872): the reference sums differences of np.exp values,
    whose last bit is not reproducible, and the subtraction amplifies it - rewards are compared to
    1e-12 relative, every other field bit-exactly; against the C oracle (same FDLIBM exp, same
    summation order) the rewards must agree bit-for-bit."""
    from golden_util import Golden, bits
    from oracle_binding import OracleEnv

    got: list = []
    bad = replay_golden("c1_fair_beta", [11], pack, device="cpu", lib=load_emu(), reward_rtol=1e-12, rewards_out=got, max_steps=200)
    assert not bad, "\n".join(bad[:10])
    g = Golden("c1_fair_beta")
    o = OracleEnv(pack, g.cfg)
    o.reset(11)
    for s, i, r in got:
        e, ro, _ = o.step(int(g.ep(11, "stage_idx")[i]), int(g.ep(11, "num_exec")[i]))
        assert e == 0 and bits(ro) == bits(r), i
