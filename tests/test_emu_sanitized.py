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
        print("SANITIZED-OK" if not bad else bad)
    """) % (os.path.dirname(HERE), HERE, os.path.join(HERE, "_build", "libsss_emu_asan.so"))
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=900)
    assert "SANITIZED-OK" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
    assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
