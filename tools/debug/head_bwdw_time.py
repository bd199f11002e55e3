"""the two policy heads' backward pass: one kernel with the parameter gradients (sss_mlp_head_mfma_bwdw_kernel) against the backward
launch + three weight-gradient launches it replaces, at a config-5 minibatch's row counts. python tools/debug/head_bwdw_time.py [rows]"""
import os.path as osp
import sys
import time

import torch

sys.path.insert(0, osp.dirname(osp.dirname(osp.dirname(osp.abspath(__file__)))))
from spark_sched_sim_amd.decima import make_mlp  # noqa: E402
from spark_sched_sim_amd.train_kernels import (linear_wgrad, mlp_backward, mlp_backward_wgrad, mlp_forward, mlp_wgrad_acc, mlp_wgrad_finish,  # noqa: E402
                                               pack_mlp)

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
for in_dim in (53, 36):
    dims = (in_dim, 64, 64, 1)
    mlp = make_mlp(in_dim, [64, 64], 1, "Tanh", {}).to(dev)
    packed = pack_mlp(mlp[0], mlp[2], mlp[4])
    x = torch.randn((rows, in_dim), device=dev)
    dy = torch.randn((rows, 1), device=dev)
    a1, a2, _ = mlp_forward(x, packed, dims, 1, 0.0)

    def fused():
        acc = mlp_wgrad_acc(in_dim, dev)
        dx = mlp_backward_wgrad(dy, x, a1, a2, packed, dims, 0.0, acc, act=1)
        return (dx,) + tuple(mlp_wgrad_finish(dims, acc))

    def unfused():
        g1, g2, dx = mlp_backward(dy, a1, a2, packed, dims, 1, 0.0)
        gw3, gb3 = linear_wgrad(a2, dy)
        gw2, gb2 = linear_wgrad(a1, g2)
        gw1, gb1 = linear_wgrad(x, g1)
        return dx, gw1, gb1, gw2, gb2, gw3, gb3

    for name, fn in (("fused", fused), ("unfused", unfused)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        print(f"in_dim {in_dim} rows {rows} {name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms")
    f, u = fused(), unfused()
    print("  max abs diff", [float((a_ - b_).abs().max()) for a_, b_ in zip(f, u)])
