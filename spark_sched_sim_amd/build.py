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
    out = [osp.join(ROOT, "include", "sss.h")]
    for pat in ("*.hip", "*.h", "*.inc"):
        out += sorted(glob.glob(osp.join(CSRC, pat)))
    return out


OUT = osp.join(CSRC, "libsss_hip.so")

# -ffp-contract=off: f64 event times / rewards must round exactly as the reference's do (no FMA fusion)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wall", "-Wno-unused-function", "-I", CSRC]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and osp.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the HIP extension)")


def needs_build() -> bool:
    if not osp.exists(OUT):
        return True
    t = osp.getmtime(OUT)
    return any(osp.getmtime(s) > t for s in sources())


def build(force: bool = False, verbose: bool = False) -> str:
    if force or needs_build():
        cmd = [hipcc()] + FLAGS + ["-o", OUT, osp.join(CSRC, "sss_hip.hip")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=CSRC)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
