"""The oracle and the kernel source against the REFERENCE ITSELF on trace-set regimes nobody has looked at: where /root/reference is
mounted (the build container; never the GPU box) a child process imports the reference, draws generator parameters / an env
configuration / a policy from a seed (tests/golden/make_golden.py random_regime: 2-30 stages, in-degree 1-5 with near or far parents,
1-60 tasks per stage scaled by size, any subset of the executor levels, bases from 20 ms to 20 s, 3-128 executors, zero and non-zero
delays; executor counts on both sides of the 64-lane boundary), records one episode the way the committed fixtures were recorded, and the C oracle (oracle/sss_oracle.c) and the kernel
source under the CPU wave emulator must reproduce it bit for bit. The committed fixtures pin two regimes for good; this keeps probing
others on every run of the CPU suite. Skipped where the reference is not present."""
import os
import os.path as osp
import subprocess
import sys

import pytest

from golden_util import Golden
from oracle_binding import OracleEnv

HERE = osp.dirname(osp.abspath(__file__))
REF = os.environ.get("SSS_REFERENCE", "/root/reference")

pytestmark = pytest.mark.skipif(not osp.isdir(osp.join(REF, "spark_sched_sim")), reason="the reference is only mounted in the build container")


@pytest.mark.parametrize("seed", list(range(10)))
def test_oracle_and_kernel_source_match_the_live_reference_on_a_random_regime(seed, tmp_path):
    import test_oracle_golden as T
    from emu_util import load_emu
    from replay_util import replay_golden

    out = str(tmp_path / f"random_{seed}.npz")
    res = subprocess.run([sys.executable, osp.join(HERE, "golden", "make_golden.py"), "--random", str(seed), out], capture_output=True, text=True, timeout=900,
                         cwd=str(tmp_path))
    assert res.returncode == 0 and osp.exists(out), res.stdout[-1500:] + res.stderr[-3000:]
    g = Golden("random", path=out)
    pack = g.pack(b"")
    s = g.seeds[0]
    # the oracle: every recorded step (rewards / wall times / digests / the first observations in full / job times)
    env = OracleEnv(pack, g.cfg)
    T.replay(env, g, s)
    env.close()
    # the kernel source under the emulator, through the C ABI: the same recording (a reference episode that ended in its own
    # "[step]" assertion is replayed up to that step)
    import golden_util
    keep = golden_util.Golden
    try:
        golden_util.Golden = lambda name: g   # (replay_golden looks its fixture up by name)
        import replay_util
        replay_util.Golden = golden_util.Golden
        bad = replay_golden("random", [s], pack, device="cpu", lib=load_emu(), full_obs_steps=10, max_steps=120)
    finally:
        golden_util.Golden = keep
        replay_util.Golden = keep
    assert not bad, "\n".join(bad[:8])


@pytest.mark.parametrize("seed", [1, 4, 8])
def test_decima_features_and_scores_match_the_live_reference_on_a_random_regime(seed, tmp_path):
    """the same for SURVEY 8(f) next-1: the reference's own DecimaEnvWrapper / DecimaScheduler (functional PyG stand-ins) record 40 steps
    on a random regime; node features, masks, DAG-layer edge masks identical, scores within 2e-5, graph kernel == tensor-op graph"""
    from decima_util import check_decima_fixture
    from emu_util import load_emu

    out = str(tmp_path / f"decima_random_{seed}.npz")
    res = subprocess.run([sys.executable, osp.join(HERE, "golden", "make_decima_golden.py"), "--random", str(seed), out], capture_output=True, text=True,
                         timeout=900, cwd=str(tmp_path))
    assert res.returncode == 0 and osp.exists(out), res.stdout[-1500:] + res.stderr[-3000:]
    check_decima_fixture("random", "cpu", load_emu(), 40, path=out)
