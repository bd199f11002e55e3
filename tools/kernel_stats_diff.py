#!/usr/bin/env python3
"""kernel_stats(A) - kernel_stats(B), kernel by kernel: what run A launched beyond run B (rocprofv3 --kernel-trace --stats CSVs).
tools/profile_ppo_rocprof.sh: A = one collection + one PPO update, B = the same collection alone -> the update's kernels.

    python tools/kernel_stats_diff.py iter_kernel_stats.csv collect_kernel_stats.csv > train_kernel_stats.csv
"""
import csv
import sys


def load(path):
    out = {}
    for row in csv.DictReader(open(path)):
        out[row["Name"]] = (int(row["Calls"]), int(float(row["TotalDurationNs"])))
    return out


def main(a_path, b_path):
    a, b = load(a_path), load(b_path)
    rows = []
    for name, (calls, tot) in a.items():
        bc, bt = b.get(name, (0, 0))
        dc, dt = calls - bc, tot - bt
        if dc > 0 and dt > 0:
            rows.append((name, dc, dt))
    total = sum(r[2] for r in rows) or 1
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage"')
    for name, dc, dt in sorted(rows, key=lambda r: -r[2]):
        q = name.replace('"', "'")
        print(f'"{q}",{dc},{dt},{dt / dc:.1f},{100.0 * dt / total:.4f}')


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
