"""Hand-written kernels of the PPO update behind torch.autograd.

`KernelLinear` is `torch.nn.Linear` (same parameters, same `state_dict`, same forward) whose backward pass computes
the weight / bias gradient with `sss_linear_wgrad` (include/sss.h; csrc/sss_train.h) when the input is a large
minibatch on the GPU: gw = dy^T x with 10^5 .. 10^7 rows and at most 64 x 64 outputs is the shape the BLAS library
handles worst - 68 % of a PPO update's device time before (profiles/r03_ppo.md). What the reference runs here is
autograd's AddmmBackward inside `loss.backward()` (trainers/ppo.py:129-131, schedulers/scheduler.py:44-54).
Small inputs and CPU tensors take torch's own path.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

MIN_ROWS = 8192  # below this the library call is as fast as two extra launches

_BINDING = None
_SCRATCH: dict = {}


def _binding():
    global _BINDING
    if _BINDING is None:
        from .binding import Binding
        _BINDING = Binding()  # raises if the HIP extension is missing: there is no fallback on a GPU box
    return _BINDING


def linear_wgrad(x: torch.Tensor, dy: torch.Tensor, want_bias: bool = True, binding=None):
    """(gw f32[N, M], gb f32[N] | None) for x f32[K, M], dy f32[K, N] on the GPU (rows may be strided)"""
    from .binding import device_of
    b = binding if binding is not None else _binding()
    assert x.dim() == 2 and dy.dim() == 2 and x.shape[0] == dy.shape[0] and x.dtype == dy.dtype == torch.float32
    if x.stride(1) != 1:
        x = x.contiguous()
    if dy.stride(1) != 1:
        dy = dy.contiguous()
    K, M = x.shape
    N = dy.shape[1]
    dev = x.device
    gw = torch.empty((N, M), dtype=torch.float32, device=dev)
    gb = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    need = int(b.lib.sss_linear_wgrad_scratch(M, N))
    if need <= 0:
        raise ValueError(f"sss_linear_wgrad supports 1..64 features, got M={M}, N={N}")
    key = (dev.type, dev.index)
    sc = _SCRATCH.get(key)
    if sc is None or sc.numel() < need:
        sc = _SCRATCH[key] = torch.empty(max(need, int(b.lib.sss_linear_wgrad_scratch(64, 64))), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    with device_of(dev):
        b.check(b.lib.sss_linear_wgrad(x.data_ptr(), x.stride(0) if K > 1 else max(M, x.stride(0)), dy.data_ptr(), dy.stride(0) if K > 1 else max(N, dy.stride(0)),
                                       K, M, N, gw.data_ptr(), gb.data_ptr() if gb is not None else None, sc.data_ptr(), stream))
    return gw, gb


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gy @ weight if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = linear_wgrad(x.reshape(-1, x.shape[-1]), gy.reshape(-1, gy.shape[-1]), want_bias=ctx.has_bias)
        return gx, gw, gb


class KernelLinear(nn.Linear):
    """nn.Linear with the hand-written weight-gradient kernel in its backward pass (see the module docstring)"""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and self.weight.requires_grad
                and x.numel() // max(1, x.shape[-1]) >= MIN_ROWS and self.in_features <= 64 and self.out_features <= 64):
            return _LinearFn.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
