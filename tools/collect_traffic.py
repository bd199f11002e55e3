#!/usr/bin/env python3
"""Runs on the GPU box: measures HBM traffic per launch of the simulator kernels with rocprofv3
PMC passes (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes:
TCC has 4 slots, FETCH_SIZE costs 3 and WRITE_SIZE 2), calibrates the counters on a torch copy of
known size in the same access regime (wide coalesced streaming), and writes
gpurun_out/traffic.json (copy it to profiles/traffic.json; bench.py reads it for roofline.traffic).

    python tools/collect_traffic.py            # from the repo root, on the GPU box
"""
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
CALIB = """
import torch
x = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device='cuda')
x.fill_(1)
y = torch.empty_like(x)
for _ in range(5):
    y.copy_(x)
torch.cuda.synchronize()
"""


def pmc_run(counter: str, tag: str, cmd: list[str]) -> dict[str, list[float]]:
    d = os.path.join(OUT, f"traffic_{tag}_{counter}")
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "t", "--"] + cmd,
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=ROOT)
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                rows.append((int(row.get("Dispatch_Id", len(rows))), row["Kernel_Name"], float(row["Counter_Value"])))
    vals: dict[str, list[float]] = {}
    for _, name, v in sorted(rows):  # launch order: the timed launches are the last ones of a bench run
        vals.setdefault(name, []).append(v)
    return vals


def main():
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("TMPDIR", "/tmp")
    calib_py = os.path.join(OUT, "traffic_calib.py")
    open(calib_py, "w").write(CALIB)
    nbytes = 256 * 1024 * 1024
    cal = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = pmc_run(counter, "calib", ["python3", calib_py])
        copies = [v for k, vs in vals.items() if "copy" in k.lower() or "Copy" in k for v in vs]
        big = [v for v in copies if v > 0]
        cal[counter] = (sum(big) / len(big)) if big else float("nan")
    # counter units are KiB; factor = true bytes / reported bytes for a 256 MiB streaming copy
    f_fetch = nbytes / (cal["FETCH_SIZE"] * 1024.0)
    f_write = nbytes / (cal["WRITE_SIZE"] * 1024.0)
    recs = []
    for config, envs in (("c2", 4096), ("c3", 4096)):
        for mode, kernel in (("step", "sss_step_kernel"), ("fused", "sss_rollout_kernel")):
            cmd = ["python3", "bench.py", "--config", config, "--envs", str(envs), "--mode", mode, "--single-mode",
                   "--no-cpu-baseline", "--no-decima", "--no-c3", "--steps", "200", "--warmup", "50"]
            # the regime the counters are collected in (steady state: bench.py pre-rolls every env), from an unprofiled run
            line = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()[-1]
            ref = json.loads(line)
            per = {}
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                vals = pmc_run(counter, f"{config}_{mode}", cmd)
                v = vals.get(kernel, [])
                if mode == "fused":  # only the timed launches (200 steps in chunks of 50); the pre-roll launches are longer
                    v = v[-4:]
                else:
                    v = v[-200:]
                per[counter] = sum(v) / len(v) if v else float("nan")
            hbm = (per["FETCH_SIZE"] * f_fetch + per["WRITE_SIZE"] * f_write) * 1024.0
            recs.append({"kernel": kernel, "config": config, "envs": envs, "mode": mode,
                         "fetch_size_kib_raw": per["FETCH_SIZE"], "write_size_kib_raw": per["WRITE_SIZE"],
                         "calibration": {"fetch_factor": f_fetch, "write_factor": f_write,
                                         "how": "256 MiB torch copy_ kernel in the same rocprofv3 setup"},
                         "events_per_step": ref["events_per_step"], "algorithmic_bytes_per_launch": ref["roofline"]["bytes_per_launch"],
                         "hbm_bytes_per_launch": hbm})
            print(recs[-1], flush=True)
    json.dump(recs, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
