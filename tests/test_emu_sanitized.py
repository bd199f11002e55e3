"""The kernel source under AddressSanitizer + UBSan (CPU wave-emulator build): out-of-bounds LDS /
arena accesses, signed overflow, bad shifts. The GPU pool offers no sanitizers, so this is where
the device code gets them. Runs in a child process so the ASan runtime can be preloaded."""
import glob
import os
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_kernel_source_under_asan_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "emu"), "../_build/libsss_emu_asan.so"], check=True)
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not glob.glob(libasan + "*"):
        pytest.skip("libasan not found")
    code = textwrap.dedent("""
        import sys, ctypes
        sys.path[:0] = [%r, %r]
        from replay_util import replay_golden
        from test_emu_policies import run_policy_episode
        from spark_sched_sim_amd import workload
        lib = ctypes.CDLL(%r)
        pack = workload.default_pack()
        bad = replay_golden("tiny_hash", [0, 21, 22, 3], pack, device="cpu", lib=lib)
        bad += replay_golden("c1_fair", [1234], pack, device="cpu", lib=lib, max_steps=150)
        bad += replay_golden("bige_hash", [0], pack, device="cpu", lib=lib, max_steps=120)
        bad += replay_golden("tiny_fair_tlimit", [0, 1], pack, device="cpu", lib=lib)
        bad += replay_golden("e100_hash", [2], pack, device="cpu", lib=lib, max_steps=250)   # the wide instantiation
        bad += replay_golden("e120_hash", [0], pack, device="cpu", lib=lib)
        bad += run_policy_episode("tiny_hash", "hash", 30, [0, 1], pack, lib=lib, fused=1)
        bad += run_policy_episode("testyaml_fair", "fair", 0, [3], pack, lib=lib, max_steps=80)
        # steps cut at an event budget (sss_step_bounded), then the record's graph arena (sss_arena_append) and the record kernels
        import torch
        from decima_util import AGENT
        from spark_sched_sim_amd import VecSparkSchedSimEnv
        from spark_sched_sim_amd.binding import Binding
        from spark_sched_sim_amd.decima import DecimaPolicy
        from spark_sched_sim_amd.training import RolloutCollector, discounted_returns, sequence_baselines
        cfg = dict(num_executors=10, job_arrival_cap=8, job_arrival_rate=4.0e-5, moving_delay=2000.0, warmup_delay=1000.0)
        env = VecSparkSchedSimEnv(cfg, 4, device="cpu", pack=pack, auto_reset=True, _lib=lib)
        env.reset(seed=[3, 4, 5, 6])
        for _ in range(150):
            a = env.policy_actions("fair")
            env.step_bounded_async(a["stage_idx"], a["num_exec"], 3)
        env.raise_on_error()
        env.close()
        env = VecSparkSchedSimEnv(cfg, 4, device="cpu", pack=pack, auto_reset=False, _lib=lib)
        torch.manual_seed(0)
        col = RolloutCollector(env, 4.0e5, [1, 2, 3, 4], seed_step=4, num_executors=10, policy=DecimaPolicy(num_executors=10, **AGENT))
        ro = col.collect_sync(with_stats=False)
        assert col._arena_sizes is not None and int(ro.active.sum()) > 0
        b = Binding(lib)
        r = discounted_returns(ro, 5.0e-3, binding=b)
        sequence_baselines(ro, r, 2, 2, binding=b)
        env.close()
        print("SANITIZED-OK" if not bad else bad)
    """) % (os.path.dirname(HERE), HERE, os.path.join(HERE, "_build", "libsss_emu_asan.so"))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "SANITIZED-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
