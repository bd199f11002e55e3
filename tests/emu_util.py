"""builds / loads the CPU wave-emulator build of the kernel source (tests/emu, test infrastructure)"""
import ctypes
import os.path as osp
import subprocess

HERE = osp.dirname(osp.abspath(__file__))
_LIB = {}


def load_emu(variant: str = ""):
    if variant not in _LIB:
        target = f"../_build/libsss_emu{variant}.so"
        subprocess.run(["make", "-s", "-C", osp.join(HERE, "emu"), target], check=True)
        _LIB[variant] = ctypes.CDLL(osp.join(HERE, "_build", f"libsss_emu{variant}.so"))
    return _LIB[variant]
