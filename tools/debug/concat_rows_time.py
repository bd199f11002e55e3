"""the stage head's input rows (train_kernels.concat_rows: four gathers into column slices of one [S, 53] matrix) and their backward
pass (three scatter-adds from column slices), at a config-5 minibatch's sizes. python tools/debug/concat_rows_time.py"""
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
from spark_sched_sim_amd.train_kernels import concat_rows  # noqa: E402

dev = torch.device("cuda:0")
M, J, O, S = 21_000_000, 2_400_000, 570_000, 5_000_000
torch.manual_seed(0)
x = torch.randn((M, 5), device=dev)
hn = torch.randn((M, 16), device=dev, requires_grad=True)
hd = torch.randn((J, 16), device=dev, requires_grad=True)
hg = torch.randn((O, 16), device=dev, requires_grad=True)
idx = torch.sort(torch.randperm(M, device=dev)[:S]).values
node_job = (idx.double() / M * J).long().clamp(max=J - 1)
node_obs = (idx.double() / M * O).long().clamp(max=O - 1)
w = torch.randn((S, 53), device=dev)


def fwd():
    return concat_rows([(x, idx), (hn, idx), (hd, node_job), (hg, node_obs)])


def fwd_bwd():
    out = fwd()
    out.backward(w)
    hn.grad = hd.grad = hg.grad = None


for name, fn in (("forward", fwd), ("forward + backward", fwd_bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
