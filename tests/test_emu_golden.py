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
]


@pytest.mark.parametrize("name,seeds,max_steps", CASES)
def test_kernel_source_matches_reference_under_emulation(name, seeds, max_steps, pack):
    bad = replay_golden(name, seeds, pack, device="cpu", lib=load_emu(), full_obs_steps=10, max_steps=max_steps)
    assert not bad, "\n".join(bad[:10])
