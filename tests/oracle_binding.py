"""ctypes binding of oracle/_build/liboracle.so for the test-suite (test infrastructure only)."""
from __future__ import annotations

import ctypes as C
import os.path as osp
import subprocess

import numpy as np

ROOT = osp.dirname(osp.dirname(osp.abspath(__file__)))
ORACLE_DIR = osp.join(ROOT, "oracle")


class SsoCfg(C.Structure):
    _fields_ = [("num_executors", C.c_int32), ("job_arrival_cap", C.c_int32),
                ("job_arrival_rate", C.c_double), ("moving_delay", C.c_double),
                ("warmup_delay", C.c_double), ("beta", C.c_double)]


class SsoObsInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("n_edges", C.c_int32), ("n_jobs", C.c_int32),
                ("n_schedulable", C.c_int32), ("num_committable_execs", C.c_int32),
                ("source_job_idx", C.c_int32), ("terminated", C.c_int32), ("num_jobs", C.c_int32),
                ("num_completed", C.c_int32), ("pad_", C.c_int32), ("wall_time", C.c_double)]


_LIBS: dict[str, C.CDLL] = {}


def load_oracle(variant: str = "") -> C.CDLL:
    """builds (if needed) and loads liboracle{variant}.so"""
    if variant in _LIBS:
        return _LIBS[variant]
    target = f"_build/liboracle{variant}.so"
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, target], check=True)
    lib = C.CDLL(osp.join(ORACLE_DIR, target))
    lib.sso_create.restype = C.c_void_p
    lib.sso_create.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(SsoCfg)]
    lib.sso_destroy.argtypes = [C.c_void_p]
    lib.sso_reset.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
    lib.sso_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    lib.sso_obs_sizes.argtypes = [C.c_void_p, C.POINTER(SsoObsInfo)]
    lib.sso_obs_fill.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    lib.sso_obs_digests.argtypes = [C.c_void_p, C.c_void_p]
    lib.sso_job_times.argtypes = [C.c_void_p] + [C.c_void_p] * 4
    lib.sso_active_jobs.argtypes = [C.c_void_p, C.c_void_p]
    lib.sso_duration_buffer.argtypes = [C.c_void_p, C.c_void_p]
    lib.sso_num_jobs.argtypes = [C.c_void_p]
    lib.sso_last_error.argtypes = [C.c_void_p]
    lib.sso_event_count.argtypes = [C.c_void_p]
    lib.sso_event_count.restype = C.c_int64
    lib.sso_step_count.argtypes = [C.c_void_p]
    lib.sso_step_count.restype = C.c_int64
    lib.sso_run_episode.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int64, C.POINTER(C.c_double)]
    lib.sso_run_episode.restype = C.c_int64
    lib.sso_run_episode_tl.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_int, C.c_int64, C.POINTER(C.c_double)]
    lib.sso_run_episode_tl.restype = C.c_int64
    _LIBS[variant] = lib
    return lib


class OracleEnv:
    """single-env handle over the C oracle with numpy observations"""

    def __init__(self, pack: bytes, env_cfg: dict, variant: str = ""):
        self.lib = load_oracle(variant)
        cap = env_cfg.get("job_arrival_cap")
        self.cfg = SsoCfg(int(env_cfg["num_executors"]), int(cap) if cap else 0,
                          float(env_cfg["job_arrival_rate"]), float(env_cfg["moving_delay"]),
                          float(env_cfg["warmup_delay"]), float(env_cfg.get("beta", 0.0)))
        self._pack = pack
        self.h = self.lib.sso_create(pack, len(pack), C.byref(self.cfg))
        assert self.h, "sso_create failed"

    def close(self):
        if self.h:
            self.lib.sso_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def reset(self, seed: int, time_limit: float = float("inf")) -> int:
        return self.lib.sso_reset(self.h, seed, time_limit)

    def step(self, stage_idx: int, num_exec: int):
        r, t = C.c_double(), C.c_int()
        err = self.lib.sso_step(self.h, int(stage_idx), int(num_exec), C.byref(r), C.byref(t))
        return err, r.value, bool(t.value)

    def info(self) -> SsoObsInfo:
        info = SsoObsInfo()
        self.lib.sso_obs_sizes(self.h, C.byref(info))
        return info

    def obs(self):
        info = self.info()
        nodes = np.zeros((info.n_nodes, 3), np.float32)
        el = np.zeros((info.n_edges, 2), np.int32)
        ptr = np.zeros(info.n_jobs + 1, np.int32)
        sup = np.zeros(info.n_jobs, np.int32)
        self.lib.sso_obs_fill(self.h, nodes.ctypes.data, el.ctypes.data, ptr.ctypes.data, sup.ctypes.data)
        return info, nodes, el, ptr, sup

    def digests(self):
        out = np.zeros(4, np.uint64)
        self.lib.sso_obs_digests(self.h, out.ctypes.data)
        return out

    def job_times(self):
        J = self.lib.sso_num_jobs(self.h)
        ta, tc = np.zeros(J), np.zeros(J)
        tm, co = np.zeros(J, np.int32), np.zeros(J, np.int32)
        self.lib.sso_job_times(self.h, ta.ctypes.data, tc.ctypes.data, tm.ctypes.data, co.ctypes.data)
        return ta, tc, tm, co

    def active_jobs(self):
        J = self.lib.sso_num_jobs(self.h)
        out = np.zeros(J + 1, np.int32)
        n = self.lib.sso_active_jobs(self.h, out.ctypes.data)
        return out[:n].copy()

    def duration_buffer(self):
        out = np.zeros(200)
        n = self.lib.sso_duration_buffer(self.h, out.ctypes.data)
        return out[:n].copy()
