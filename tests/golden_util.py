"""helpers shared by the golden-fixture tests"""
from __future__ import annotations

import os.path as osp

import numpy as np

GOLDEN_DIR = osp.join(osp.dirname(osp.abspath(__file__)), "golden")
ALL_SETS = ["tiny_hash", "tiny_fair_tlimit", "c1_fair", "c1_hash", "c1_fifo", "c3_fair", "c3_hash",
            "testyaml_fair", "bige_hash", "e100_fair", "e100_hash", "e120_hash", "q5s2_fair", "q5s2_hash"]
# recorded on the "deep" trace regime (workload.PROFILES["deep"], a 60 MB pack)
DEEP_SETS = ["deep_c1_fair", "deep_c1_hash", "deep_c1_fifo", "deep_e50_fair", "deep_e50_hash", "deep_e100_fair", "deep_e100_hash",
             "deep_tlimit_hash", "deep_c1_fair_beta"]


_PACKS: dict = {}


class Golden:
    def __init__(self, name: str, path: str | None = None):
        self.name = name
        self.z = np.load(path if path is not None else osp.join(GOLDEN_DIR, f"{name}.npz"))
        keys = self.z["cfg_keys"].tolist()
        vals = self.z["cfg_vals"].tolist()
        self.cfg = {}
        for k, v in zip(keys, vals):
            if k in ("num_executors", "job_arrival_cap"):
                self.cfg[k] = None if np.isnan(v) else int(v)
            else:
                self.cfg[k] = float(v)
        self.time_limit = float(self.z["time_limit"])
        self.seeds = [int(s) for s in self.z["seeds"]]
        self.policy = str(self.z["policy"])
        self.pack_sha256 = str(self.z["pack_sha256"])

    def pack(self, default: bytes) -> bytes:
        """the workload pack the set was recorded on: the frozen default, or - sets that name a trace-set shape (make_golden.py:
        q5s2_*, deep_*) - the pack of that many sizes x queries from that generator seed and profile"""
        if "trace_sizes" not in self.z:
            return default
        from spark_sched_sim_amd import workload
        if "trace_profile_json" in self.z:  # generator parameters of the set's own (tests/test_oracle_vs_live_reference.py)
            import json
            prof = json.loads(str(self.z["trace_profile_json"]))
            prof = {k: (tuple(v) if isinstance(v, list) and k != "levels" else v) for k, v in prof.items()}
            sizes, n_q, seed = [str(x) for x in self.z["trace_sizes"]], int(self.z["trace_queries"]), int(self.z["trace_seed"])
            return workload.build_pack(workload.make_raw_workload(seed, sizes, n_q, profile=prof), query_sizes=sizes, num_queries=n_q)
        profile = str(self.z["trace_profile"]) if "trace_profile" in self.z else "default"
        key = (tuple(str(x) for x in self.z["trace_sizes"]), int(self.z["trace_queries"]), int(self.z["trace_seed"]), profile)
        if key not in _PACKS:
            _PACKS[key] = workload.profile_pack(profile, key[2], list(key[0]), key[1])
        return _PACKS[key]

    def ep(self, seed: int, key: str):
        return self.z[f"s{seed}_{key}"]


def bits(x) -> int:
    return int(np.float64(x).view(np.uint64))
