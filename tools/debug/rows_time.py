#!/usr/bin/env python3
"""debug: sss_rows_op against torch's index_select / index_add_ at the sizes of a PPO minibatch at the BASELINE config-5 share
(21 M rows of 16 floats). usage: rows_time.py [variant]   (variant: a tests/gpu_variant.py build, e.g. vecatom)"""
import os.path as osp
import sys
import time

import torch

ROOT = osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, osp.join(ROOT, "tests"))
from spark_sched_sim_amd import train_kernels as tk  # noqa: E402
from spark_sched_sim_amd.binding import Binding, load_library  # noqa: E402

b = Binding()
if len(sys.argv) > 1:
    import gpu_variant
    b = Binding(load_library(gpu_variant.build_variant(sys.argv[1])))
dev = torch.device("cuda:0")
n, rows = 21_000_000, 21_000_000


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


torch.manual_seed(0)
table = torch.randn((rows, 16), device=dev)
src = torch.randn((n, 16), device=dev)
out = torch.empty((n, 16), device=dev)
local = (torch.arange(n, device=dev) + torch.randint(-20, 20, (n,), device=dev)).clamp(0, rows - 1)   # a child a few rows away
seg = torch.sort(torch.randint(0, rows // 6, (n,), device=dev))[0]                                       # ~6 rows per segment
ptr = tk.segment_offsets(seg, rows // 6)
acc6 = torch.zeros((rows // 6, 16), device=dev)
uniq = torch.randperm(rows, device=dev)[: n // 4]
for name, fn in (
    ("gather local: kernel", lambda: tk.rows_op(tk.ROWS_GATHER, local, out, table, binding=b)),
    ("gather local: torch", lambda: torch.index_select(table, 0, local, out=out)),
    ("scatter-add local: kernel", lambda: tk.rows_op(tk.ROWS_SCATTER_ADD, local, src, table, binding=b)),
    ("scatter-add local: torch", lambda: table.index_add_(0, local, src)),
    ("scatter-add sorted segments (6 rows): kernel atomics", lambda: tk.rows_op(tk.ROWS_SCATTER_ADD, seg, src, acc6, binding=b)),
    ("segment sum (6 rows): kernel, offsets given", lambda: tk.rows_op(tk.ROWS_SEGMENT_SUM, ptr, src, acc6, binding=b)),
    ("segment offsets (searchsorted)", lambda: tk.segment_offsets(seg, rows // 6)),
    ("scatter-add sorted segments: torch", lambda: acc6.index_add_(0, seg, src)),
    ("scatter unique (n/4 rows): kernel", lambda: tk.rows_op(tk.ROWS_SCATTER, uniq, src[: n // 4], table, binding=b)),
):
    print(f"{name:55s} {timed(fn):8.3f} ms", flush=True)
