"""Builds the HIP extension in-tree: spark_sched_sim_amd/csrc/libsss_hip.so (gfx950).

    python -m spark_sched_sim_amd.build            # build if sources are newer than the .so
    python -m spark_sched_sim_amd.build --force
"""
from __future__ import annotations

import os
import os.path as osp
import shutil
import subprocess
import sys

CSRC = osp.join(osp.dirname(osp.abspath(__file__)), "csrc")
ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))


def sources() -> list[str]:
    """everything the library is compiled from: every source / header / table under csrc/ plus the C ABI header"""
    import glob
    out = [osp.join(ROOT, "include", "sss.h"), osp.abspath(__file__)]  # (this file: the compiler flags)
    for pat in ("*.hip", "*.h", "*.inc"):
        out += sorted(glob.glob(osp.join(CSRC, pat)))
    return out


OUT = osp.join(CSRC, "libsss_hip.so")
# translation units: the C ABI with every kernel but the simulator's own, and the simulator kernels (sss_sim.h) - up to 64 executors, and
# the wide instantiation (65..128 executors: the same source with -DSSS_WIDE, csrc/sss_wide.h)
UNITS = ["sss_hip.hip", "sss_hip_sim.hip", "sss_hip_wide.hip"]

# -ffp-contract=off: f64 event times / rewards must round exactly as the reference's do (no FMA fusion)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall", "-Wno-unused-function", "-I", CSRC]
# -mllvm -disable-machine-licm, for the two simulator units ONLY: a simulator kernel is one function (everything is inlined: the launch
#   context lives in the kernel-argument segment) whose event loop spans tens of thousands of instructions; machine LICM hoists dozens of
#   cheap per-lane values (lane masks, LDS addresses) out of that loop and the register allocator then spills them across it: 272 /
#   640 bytes of scratch per lane in sss_step_kernel / sss_rollout_kernel against 32 / 48 without the pass - 16 KB of spill stores
#   per env-step, three times the HBM write traffic, fused C2 -8 % (profiles/r04_bench.md section 4). It is a compiler-internal
#   switch, so it stays off the other kernels, and tests/test_abi.py::test_simulator_kernels_do_not_spill holds the kernels' scratch
#   sizes (tools/isa_counts.py) to what it buys: a toolchain that ignores or renames the flag fails that test instead of silently
#   bringing the spills back.
UNIT_FLAGS = {"sss_hip_sim.hip": ["-mllvm", "-disable-machine-licm"], "sss_hip_wide.hip": ["-mllvm", "-disable-machine-licm"]}


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and osp.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the HIP extension)")


def needs_build(out: str = OUT) -> bool:
    if not osp.exists(out):
        return True
    t = osp.getmtime(out)
    return any(osp.getmtime(s) > t for s in sources())


def build(force: bool = False, verbose: bool = False, out: str = OUT, extra_flags: tuple[str, ...] = ()) -> str:
    """compiles the translation units side by side, links them into `out`. `extra_flags`: test builds only (tests/gpu_variant.py)"""
    if force or needs_build(out):
        from concurrent.futures import ThreadPoolExecutor
        objdir = osp.join(osp.dirname(out), "build", osp.splitext(osp.basename(out))[0])
        os.makedirs(objdir, exist_ok=True)
        cc = hipcc()

        def compile_unit(unit: str) -> str:
            obj = osp.join(objdir, osp.splitext(unit)[0] + ".o")
            cmd = [cc] + FLAGS + UNIT_FLAGS.get(unit, []) + list(extra_flags) + ["-c", "-o", obj, osp.join(CSRC, unit)]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=CSRC)
            return obj

        with ThreadPoolExecutor(len(UNITS)) as pool:
            objs = list(pool.map(compile_unit, UNITS))
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
