#!/usr/bin/env python3
"""Summarises a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats` writes
`<name>_results.db` on ROCm 7.2) as the per-kernel stats CSV the judge reads from profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_step/r01_step_results.db > profiles/r01_step_kernel_stats.csv
"""
import sqlite3
import sys


def main(path: str) -> None:
    con = sqlite3.connect(path)
    print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,VGPRs,SGPRs,LDSBytes,ScratchBytes")
    rows = con.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
        "max(sgpr_count), max(lds_size), max(scratch_size) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    for name, calls, tot, avg, mn, mx, vg, sg, lds, scr in rows:
        name = name.replace('"', "'")
        print(f'"{name}",{calls},{tot},{avg:.1f},{100.0 * tot / total:.4f},{mn},{mx},{vg},{sg},{lds},{scr}')


if __name__ == "__main__":
    main(sys.argv[1])
