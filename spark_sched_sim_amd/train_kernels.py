"""Hand-written kernels of the PPO update behind torch.autograd.

`KernelLinear` is `torch.nn.Linear` (same parameters, same `state_dict`, same forward) whose backward pass computes
the weight / bias gradient with `sss_linear_wgrad` (include/sss.h; csrc/sss_train.h) when the input is a large
minibatch on the GPU: gw = dy^T x with 10^5 .. 10^7 rows and at most 64 x 64 outputs is the shape the BLAS library
handles worst - 68 % of a PPO update's device time before (profiles/r03_ppo.md). What the reference runs here is
autograd's AddmmBackward inside `loss.backward()` (trainers/ppo.py:129-131, schedulers/scheduler.py:44-54).
Small inputs and CPU tensors take torch's own path.

`KernelMLP` is the `nn.Sequential` the reference builds for every network of the architecture (decima/utils.py:44-64:
Linear - act - Linear - act - Linear, same module numbering, same `state_dict`) evaluated by ONE forward kernel and, in the
backward pass, one kernel for the three input-side gradients plus three `sss_linear_wgrad` calls (`sss_mlp_forward` /
`sss_mlp_backward`, csrc/sss_train16.h) - instead of three addmm, two activations, three mm, three wgrad and two activation
backward launches with their intermediate tensors. What the reference runs here: `nn.Sequential.forward` under autograd.
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
    stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
    # one partial-sum buffer per (device, stream): the partial and the reduce kernel of two backward passes on different
    # streams must not meet in it
    key = (dev.type, dev.index, stream)
    sc = _SCRATCH.get(key)
    if sc is None or sc.numel() < need:
        sc = _SCRATCH[key] = torch.empty(max(need, int(b.lib.sss_linear_wgrad_scratch(64, 64))), dtype=torch.float32, device=dev)
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


ROWS_GATHER, ROWS_SCATTER_ADD, ROWS_UPDATE, ROWS_TAKE, ROWS_SCATTER, ROWS_SEGMENT_SUM = 0, 1, 2, 3, 4, 5  # include/sss.h SSS_ROWS_*


def rows_op(op: int, idx: torch.Tensor, a: torch.Tensor, b: torch.Tensor, c: torch.Tensor | None = None, binding=None) -> None:
    """`sss_rows_op` (include/sss.h; csrc/sss_rows.h): `a` f32[n, width] is the list side (a column slice of a wider row-major
    matrix is allowed), `b` / `c` f32[rows, width] contiguous tables, `idx` i64[n]"""
    import ctypes

    from .binding import SssRowsArgs, device_of
    bnd = binding if binding is not None else _binding()
    width = a.shape[1]
    n = b.shape[0] if op == ROWS_SEGMENT_SUM else a.shape[0]  # (SEGMENT_SUM: idx = the n + 1 row offsets of b's n segments)
    if n == 0:
        return
    idx = idx.contiguous()
    assert idx.dtype == torch.int64 and idx.numel() == (n + 1 if op == ROWS_SEGMENT_SUM else n)
    assert a.dtype == b.dtype == torch.float32 and a.stride(1) == 1 and b.is_contiguous() and b.dim() == 2 and b.shape[1] == width
    assert c is None or (c.dtype == torch.float32 and c.is_contiguous() and c.shape == b.shape)
    dev = b.device
    args = SssRowsArgs(n, a.stride(0) if a.shape[0] > 1 else max(width, a.stride(0)), width, op, idx.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr() if c is not None else None)
    with device_of(dev):
        bnd.check(bnd.lib.sss_rows_op(ctypes.byref(args), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))


def _rows_ok(t: torch.Tensor, n: int) -> bool:
    return t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and 1 <= t.shape[1] <= 64 and n >= MIN_ROWS


class _GatherRowsFn(torch.autograd.Function):
    """table[idx] (rows); backward: the gradient rows are added into a zero table with float atomics"""

    @staticmethod
    def forward(ctx, table, idx, unique):
        table = table.contiguous()
        out = torch.empty((idx.numel(), table.shape[1]), dtype=torch.float32, device=table.device)
        rows_op(ROWS_GATHER, idx, out, table)
        ctx.save_for_backward(idx)
        ctx.rows, ctx.unique = table.shape[0], unique
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        if g.stride(1) != 1 or (g.stride(0) < g.shape[1] and g.shape[0] > 1):  # (an expanded gradient)
            g = g.contiguous()
        gt = torch.zeros((ctx.rows, g.shape[1]), dtype=torch.float32, device=g.device)
        rows_op(ROWS_SCATTER if ctx.unique else ROWS_SCATTER_ADD, idx, g, gt)
        return gt, None, None


def segment_offsets(idx: torch.Tensor, n_seg: int) -> torch.Tensor:
    """i64[n_seg + 1] row offsets of the segments of a NON-DECREASING owner array (segment s = rows ptr[s] .. ptr[s + 1])"""
    return torch.searchsorted(idx, torch.arange(n_seg + 1, device=idx.device))


class _SegmentSumFn(torch.autograd.Function):
    """out[idx[i]] += y[i] into a zero [n_seg, width] table; backward: the gradient of a row is its segment's.
    mode "sorted": idx is non-decreasing - every segment is a range of rows, summed in row order without atomics;
    mode "unique": idx has no repeats - plain stores; else float atomics"""

    @staticmethod
    def forward(ctx, y, idx, n_seg, mode):
        if y.stride(1) != 1:
            y = y.contiguous()
        if mode == "sorted":
            out = torch.empty((n_seg, y.shape[1]), dtype=torch.float32, device=y.device)
            rows_op(ROWS_SEGMENT_SUM, segment_offsets(idx, n_seg), y, out)
        else:
            out = torch.zeros((n_seg, y.shape[1]), dtype=torch.float32, device=y.device)
            rows_op(ROWS_SCATTER if mode == "unique" else ROWS_SCATTER_ADD, idx, y, out)
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        gy = torch.empty((idx.numel(), g.shape[1]), dtype=torch.float32, device=g.device)
        rows_op(ROWS_GATHER, idx, gy, g)
        return gy, None, None, None


def gather_rows(table: torch.Tensor, idx: torch.Tensor, unique: bool = False) -> torch.Tensor:
    """`table.index_select(0, idx)` for a 2-D table, on the row kernels when it is a large float32 gather on the GPU
    (`unique`: idx has no repeats - the backward pass stores instead of adding)"""
    if _rows_ok(table, idx.numel()) and torch.is_grad_enabled():
        return _GatherRowsFn.apply(table, idx.contiguous(), unique)
    return table.index_select(0, idx)


def segment_sum(y: torch.Tensor, idx: torch.Tensor, n_seg: int, mode: str = "") -> torch.Tensor:
    """`zeros(n_seg, width).index_add_(0, idx, y)`, on the row kernels when it is a large float32 sum on the GPU. `mode`: what
    the caller knows about idx - "sorted" (non-decreasing), "unique" (no repeats) or "" (see `_SegmentSumFn`)"""
    if _rows_ok(y, y.shape[0]) and torch.is_grad_enabled():
        return _SegmentSumFn.apply(y, idx.contiguous(), n_seg, mode)
    return torch.zeros((n_seg, y.shape[-1]), dtype=y.dtype, device=y.device).index_add_(0, idx, y)


CONCAT_ONE_LAUNCH = True  # the rows of a concatenation written / read back by ONE launch over the flat output (sss_rows_concat)


def rows_concat(op: int, out: torch.Tensor, tables, idxs, binding=None) -> None:
    """`sss_rows_concat` (include/sss.h; csrc/sss_rows.h sss_concat_kernel): op 0 out[i] = cat_k(tables[k][idxs[k][i]]);
    op 1 tables[k][idxs[k][i]] += the k-th column range of out[i] (tables[k] an int - the part's width - instead of a tensor: that part
    gets no gradient, its columns are skipped). out f32[n, sum of widths] contiguous"""
    import ctypes

    from .binding import SssConcatArgs, device_of
    bnd = binding if binding is not None else _binding()
    n = out.shape[0]
    if n == 0:
        return
    assert out.dtype == torch.float32 and out.is_contiguous() and 1 <= len(tables) <= 4 and len(tables) == len(idxs)
    a = SssConcatArgs()
    a.n, a.n_parts, a.op, a.out_dev = n, len(tables), op, out.data_ptr()
    for k, (t, ix) in enumerate(zip(tables, idxs)):
        width = t.shape[1] if not isinstance(t, int) else t
        if not isinstance(t, int):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.dim() == 2
        assert ix is None or (ix.dtype == torch.int64 and ix.is_contiguous() and ix.numel() == n)
        a.parts[k].table_dev, a.parts[k].idx_dev, a.parts[k].width = (None if isinstance(t, int) else t.data_ptr()), (ix.data_ptr() if ix is not None else None), width
    dev = out.device
    with device_of(dev):
        bnd.check(bnd.lib.sss_rows_concat(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))


class _ConcatRowsFn(torch.autograd.Function):
    """cat([t_0[idx_0], t_1[idx_1], ...], -1) written by one launch over the flat result (no intermediate rows, no copy into the
    concatenation, whole memory transactions); backward: one launch that reads the gradient rows once and adds every part's
    columns into its table. (`CONCAT_ONE_LAUNCH = False`, or more than four parts: one gather / scatter-add per part on column
    slices - the form of rounds 4-5.)"""

    @staticmethod
    def forward(ctx, n_parts, *args):
        tables, idxs = [t.contiguous() for t in args[:n_parts]], list(args[n_parts:])
        n = idxs[0].numel()
        out = torch.empty((n, sum(t.shape[1] for t in tables)), dtype=torch.float32, device=tables[0].device)
        ctx.one = CONCAT_ONE_LAUNCH and n_parts <= 4
        if ctx.one:
            rows_concat(0, out, tables, idxs)
        else:
            off = 0
            for t, ix in zip(tables, idxs):
                rows_op(ROWS_GATHER, ix, out[:, off:off + t.shape[1]], t)
                off += t.shape[1]
        ctx.save_for_backward(*idxs)
        ctx.shapes = [tuple(t.shape) for t in tables]
        return out

    @staticmethod
    def backward(ctx, g):
        idxs = ctx.saved_tensors
        if ctx.one:
            g = g.contiguous()
            grads = [torch.zeros(shape, dtype=torch.float32, device=g.device) if ctx.needs_input_grad[1 + k] else None for k, shape in enumerate(ctx.shapes)]
            if any(t is not None for t in grads):
                rows_concat(1, g, [t if t is not None else shape[1] for t, shape in zip(grads, ctx.shapes)], list(idxs))
            return (None, *grads, *([None] * len(idxs)))
        if g.stride(1) != 1:
            g = g.contiguous()
        grads, off = [], 0
        for k, (shape, ix) in enumerate(zip(ctx.shapes, idxs)):
            if ctx.needs_input_grad[1 + k]:
                gt = torch.zeros(shape, dtype=torch.float32, device=g.device)
                rows_op(ROWS_SCATTER_ADD, ix, g[:, off:off + shape[1]], gt)
                grads.append(gt)
            else:
                grads.append(None)
            off += shape[1]
        return (None, *grads, *([None] * len(idxs)))


def concat_rows(parts) -> torch.Tensor:
    """`torch.cat([t.index_select(0, idx) for t, idx in parts], -1)`; on the row kernels when large, float32, on the GPU"""
    n = parts[0][1].numel()
    if all(_rows_ok(t, n) for t, _ in parts) and torch.is_grad_enabled():
        return _ConcatRowsFn.apply(len(parts), *[t for t, _ in parts], *[ix.contiguous() for _, ix in parts])
    return torch.cat([t.index_select(0, ix) for t, ix in parts], -1)


SEGMENT_CATEGORICAL = True  # evaluate_actions' log-probabilities and entropies by one launch per pass over the segments (sss_segment_categorical)


def _segcat_call(backward: int, scores, ptr, chosen, den_eps, lg=None, ent=None, g_lg=None, g_ent=None, g_scores=None, binding=None) -> None:
    import ctypes

    from .binding import SssSegcatArgs, device_of
    b = binding if binding is not None else _binding()
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    a = SssSegcatArgs(chosen.numel(), p(scores), p(ptr), p(chosen), den_eps, 0, p(lg), p(ent), p(g_lg), p(g_ent), p(g_scores))
    dev = scores.device
    with device_of(dev):
        b.check(b.lib.sss_segment_categorical(ctypes.byref(a), backward, torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))


class _SegmentCategoricalFn(torch.autograd.Function):
    """(log of the clamped probability of row ptr[s] + chosen[s], -sum p log p) of the softmax inside every segment s of `scores`
    (include/sss.h sss_segment_categorical; csrc/sss_segcat.h) - decima/utils.py:26-41 `evaluate` with torch.distributions' clamp"""

    @staticmethod
    def forward(ctx, scores, ptr, chosen, den_eps, binding):
        scores = scores.contiguous()
        n = chosen.numel()
        lg, ent = torch.empty(n, dtype=torch.float32, device=scores.device), torch.empty(n, dtype=torch.float32, device=scores.device)
        _segcat_call(0, scores, ptr, chosen, den_eps, lg=lg, ent=ent, binding=binding)
        ctx.save_for_backward(scores, ptr, chosen)
        ctx.den_eps, ctx.binding = den_eps, binding
        return lg, ent

    @staticmethod
    def backward(ctx, g_lg, g_ent):
        scores, ptr, chosen = ctx.saved_tensors
        g = torch.zeros_like(scores)  # (rows outside every segment - there are none when ptr covers the array - keep 0)
        _segcat_call(1, scores, ptr, chosen, ctx.den_eps, g_lg=g_lg.contiguous().float(), g_ent=g_ent.contiguous().float(), g_scores=g, binding=ctx.binding)
        return g, None, None, None, None


def segment_categorical(scores: torch.Tensor, ptr: torch.Tensor, chosen: torch.Tensor, den_eps: float, binding=None):
    """(lg f32[n_seg], ent f32[n_seg]) for scores f32[rows], ptr i64[n_seg + 1] (row offsets), chosen i64[n_seg]"""
    assert scores.dtype == torch.float32 and scores.dim() == 1 and ptr.dtype == chosen.dtype == torch.int64 and ptr.numel() == chosen.numel() + 1
    return _SegmentCategoricalFn.apply(scores, ptr.contiguous(), chosen.contiguous(), float(den_eps), binding)


def pack_mlp(lin1: nn.Linear, lin2: nn.Linear, lin3: nn.Linear) -> torch.Tensor:
    """[W1, b1, W2^T, b2, W3, b3] as one flat tensor (include/sss.h sss_gnn_launch / sss_mlp_forward)"""
    parts = [lin1.weight, lin1.bias, lin2.weight.t(), lin2.bias, lin3.weight, lin3.bias]
    return torch.cat([t.detach().float().contiguous().reshape(-1) for t in parts]).contiguous()


def _mlp_args(dims, act, slope, rows, w, x=None, a1=None, a2=None, y=None, dy=None, g1=None, g2=None, dx=None, x2=None, dx2=None):
    from .binding import SssMlpArgs
    p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    return SssMlpArgs(rows, dims[0], dims[1], dims[2], dims[3], act, slope, p(w), p(x), p(a1), p(a2), p(y), p(dy), p(g1), p(g2), p(dx), p(x2), p(dx2))


def _out(t, rows, width, dev):
    if t is None:
        return torch.empty((rows, width), dtype=torch.float32, device=dev)
    assert t.shape == (rows, width) and t.is_contiguous() and t.dtype == torch.float32
    return t


_RECOMPUTE: dict = {}


def mlp_recompute(in_dim: int, binding=None) -> bool:
    """whether the library evaluates the (in_dim) -> 32 -> 16 -> 16 MLP without storing its hidden activations and recomputes them
    in `mlp_backward_wgrad` (include/sss.h sss_mlp_recompute_supported)"""
    b = binding if binding is not None else _binding()
    key = (id(b.lib), in_dim)
    if key not in _RECOMPUTE:
        _RECOMPUTE[key] = bool(RECOMPUTE_HIDDEN and FUSED_WGRAD and b.lib.sss_mlp_recompute_supported(in_dim))
    return _RECOMPUTE[key]


_SPLIT: dict = {}


def mlp_split(in_dim: int, binding=None) -> bool:
    """whether the library takes the (in_dim) -> 32 -> 16 -> 16 MLP's input rows in two pieces [x (in_dim - 16) | x2 (16)]
    (include/sss.h sss_mlp_split_supported) - the concatenation is then never built"""
    b = binding if binding is not None else _binding()
    key = (id(b.lib), in_dim)
    if key not in _SPLIT:
        _SPLIT[key] = bool(SPLIT_INPUT and mlp_recompute(in_dim, b) and b.lib.sss_mlp_split_supported(in_dim))
    return _SPLIT[key]


def mlp_forward(x: torch.Tensor, packed: torch.Tensor, dims, act: int, slope: float, binding=None, a1=None, a2=None, y=None, keep_hidden: bool = True, x2=None):
    """(a1 f32[rows, H1], a2 f32[rows, H2], y f32[rows, OUT]) of `sss_mlp_forward` for x f32[rows, IN] (contiguous); written
    into the tensors given, or into new ones. `keep_hidden=False` (only where `mlp_recompute` says so): a1 / a2 are not stored
    (returned as None) - the backward pass computes them again. `x2` (only where `mlp_split` says so, with keep_hidden=False): the
    input rows are [x f32[rows, IN - 16] | x2 f32[rows, 16]]"""
    import ctypes

    from .binding import device_of
    b = binding if binding is not None else _binding()
    rows, dev = x.shape[0], x.device
    assert x.dim() == 2 and x.shape[1] == dims[0] - (16 if x2 is not None else 0) and x.is_contiguous() and x.dtype == torch.float32
    assert x2 is None or (x2.shape == (rows, 16) and x2.is_contiguous() and x2.dtype == torch.float32 and not keep_hidden)
    if keep_hidden:
        a1, a2 = _out(a1, rows, dims[1], dev), _out(a2, rows, dims[2], dev)
    else:
        assert a1 is None and a2 is None
    y = _out(y, rows, dims[3], dev)
    a = _mlp_args(dims, act, slope, rows, packed, x=x, a1=a1, a2=a2, y=y, x2=x2)
    with device_of(dev):
        b.check(b.lib.sss_mlp_forward(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))
    return a1, a2, y


def mlp_backward(dy: torch.Tensor, a1: torch.Tensor, a2: torch.Tensor, packed: torch.Tensor, dims, act: int, slope: float, want_dx: bool = True, binding=None,
                 g1=None, g2=None):
    """(g1 f32[rows, H1], g2 f32[rows, H2], dx f32[rows, IN] | None) of `sss_mlp_backward`: the gradients w.r.t. the two hidden
    layers' pre-activations and the input, for dy f32[rows, OUT]"""
    import ctypes

    from .binding import device_of
    b = binding if binding is not None else _binding()
    rows, dev = a1.shape[0], a1.device
    dy = dy.contiguous()
    assert dy.shape == (rows, dims[3]) and dy.dtype == torch.float32
    g1, g2 = _out(g1, rows, dims[1], dev), _out(g2, rows, dims[2], dev)
    dx = torch.empty((rows, dims[0]), dtype=torch.float32, device=dev) if want_dx else None
    a = _mlp_args(dims, act, slope, rows, packed, a1=a1, a2=a2, dy=dy, g1=g1, g2=g2, dx=dx)
    with device_of(dev):
        b.check(b.lib.sss_mlp_backward(ctypes.byref(a), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))
    return g1, g2, dx


FUSED_WGRAD = True  # the GNN-shaped MLPs' backward pass with the parameter gradients in the same kernel (sss_mlp_backward_wgrad)
# ... and without stored hidden activations: the forward kernels write y only, the backward kernel recomputes a1 / a2 from x on the
# matrix cores (bit-identical). The kernels are bound by exactly that traffic (profiles/r06_ppo.md), and the stored activations were
# ~25 GB of a config-5 minibatch's peak memory.
RECOMPUTE_HIDDEN = True
# the two policy heads' backward pass likewise in one kernel with their parameter gradients (sss_mlp_head_mfma_bwdw_kernel) instead of
# a backward launch that writes g1 / g2 and three weight-gradient launches that read them back
FUSED_HEAD_WGRAD = True
# the DAG encoder's MLP reads its input rows [x | h_node] from the two tensors (the 21-wide concatenation and the slices of its gradient
# are never built)
SPLIT_INPUT = True


def mlp_wgrad_acc(in_dim: int, dev, binding=None) -> torch.Tensor | None:
    """a zeroed accumulator for `mlp_backward_wgrad` calls whose parameter gradients belong together, or None when the shape has
    no fused kernel"""
    b = binding if binding is not None else _binding()
    n = int(b.lib.sss_mlp_wgrad_scratch(in_dim))
    return torch.zeros(n, dtype=torch.float32, device=dev) if n > 0 else None


def mlp_backward_wgrad(dy: torch.Tensor, x: torch.Tensor, a1: torch.Tensor, a2: torch.Tensor, packed: torch.Tensor, dims, slope: float, acc: torch.Tensor,
                       want_dx: bool = True, binding=None, act: int = 0, x2=None):
    """`sss_mlp_backward_wgrad`: dx f32[rows, IN] | None for dy f32[rows, OUT]; the six parameter gradients are added to `acc`
    (`act`: 0 LeakyReLU - the GNN-shaped MLPs; 1 Tanh - the two policy heads, stored activations required). `x2` (see
    `mlp_forward`): the input rows in two pieces; what is returned is then the gradient w.r.t. x2, f32[rows, 16]"""
    import ctypes

    from .binding import device_of
    b = binding if binding is not None else _binding()
    rows, dev = x.shape[0], x.device   # (a1 / a2 None: recomputed from x by the kernel)
    dy = dy.contiguous()
    assert dy.shape == (rows, dims[3]) and x.shape == (rows, dims[0] - (16 if x2 is not None else 0)) and x.is_contiguous() and dy.dtype == x.dtype == torch.float32
    assert (a1 is None) == (a2 is None) and (x2 is None or (a1 is None and x2.shape == (rows, 16) and x2.is_contiguous() and x2.dtype == torch.float32))
    dx = torch.empty((rows, 16 if x2 is not None else dims[0]), dtype=torch.float32, device=dev) if want_dx else None
    if rows == 0:
        return dx
    if x2 is not None:
        a = _mlp_args(dims, act, slope, rows, packed, x=x, dy=dy, x2=x2, dx2=dx)
    else:
        a = _mlp_args(dims, act, slope, rows, packed, x=x, a1=a1, a2=a2, dy=dy, dx=dx)
    with device_of(dev):
        b.check(b.lib.sss_mlp_backward_wgrad(ctypes.byref(a), acc.data_ptr(), torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))
    return dx


def mlp_wgrad_finish(dims, acc: torch.Tensor, binding=None):
    """(gw1, gb1, gw2, gb2, gw3, gb3) from an accumulator: the per-workgroup slots added in a fixed order"""
    from .binding import device_of
    b = binding if binding is not None else _binding()
    dev = acc.device
    f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
    gw1, gb1, gw2, gb2, gw3, gb3 = f(dims[1], dims[0]), f(dims[1]), f(dims[2], dims[1]), f(dims[2]), f(dims[3], dims[2]), f(dims[3])
    with device_of(dev):
        b.check(b.lib.sss_mlp_wgrad_finish(dims[0], acc.data_ptr(), gw1.data_ptr(), gb1.data_ptr(), gw2.data_ptr(), gb2.data_ptr(), gw3.data_ptr(), gb3.data_ptr(),
                                           torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0))
    return gw1, gb1, gw2, gb2, gw3, gb3


class _MlpFn(torch.autograd.Function):
    """y = W3 act(W2 act(W1 x + b1) + b2) + b3 for x f32[rows, IN]; saves x and the two hidden activations"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3, packed, act, slope):
        dims = (w1.shape[1], w1.shape[0], w2.shape[0], w3.shape[0])
        ctx.recompute = act == 0 and tuple(dims[1:]) == (32, 16, 16) and mlp_recompute(dims[0])
        a1, a2, y = mlp_forward(x, packed, dims, act, slope, keep_hidden=not ctx.recompute)
        if ctx.recompute:
            ctx.save_for_backward(x, packed)
        else:
            ctx.save_for_backward(x, a1, a2, packed)
        ctx.dims, ctx.act, ctx.slope = dims, act, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.recompute:
            (x, packed), a1, a2 = ctx.saved_tensors, None, None
        else:
            x, a1, a2, packed = ctx.saved_tensors
        dy = dy.contiguous()
        fused_shape = (ctx.act == 0 and tuple(ctx.dims[1:]) == (32, 16, 16)) or (FUSED_HEAD_WGRAD and ctx.act == 1 and tuple(ctx.dims[1:]) == (64, 64, 1))
        acc = mlp_wgrad_acc(ctx.dims[0], dy.device) if (FUSED_WGRAD and fused_shape) else None
        if acc is not None:  # one kernel: dx and the six parameter gradients (g1 / g2 never leave the chip)
            dx = mlp_backward_wgrad(dy, x, a1, a2, packed, ctx.dims, ctx.slope, acc, want_dx=ctx.needs_input_grad[0], act=ctx.act)
            return (dx, *mlp_wgrad_finish(ctx.dims, acc), None, None, None)
        g1, g2, dx = mlp_backward(dy, a1, a2, packed, ctx.dims, ctx.act, ctx.slope, want_dx=ctx.needs_input_grad[0])
        gw3, gb3 = linear_wgrad(a2, dy)
        gw2, gb2 = linear_wgrad(a1, g2)
        gw1, gb1 = linear_wgrad(x, g1)
        return dx, gw1, gb1, gw2, gb2, gw3, gb3, None, None, None


class _MlpSplitFn(torch.autograd.Function):
    """`_MlpFn` on rows [xa | xb] that exist as two tensors (xa f32[rows, IN - 16] without a gradient, xb f32[rows, 16]): no stored
    hidden activations, the backward pass returns the gradient w.r.t. xb only"""

    @staticmethod
    def forward(ctx, xa, xb, w1, b1, w2, b2, w3, b3, packed, slope):
        dims = (w1.shape[1], w1.shape[0], w2.shape[0], w3.shape[0])
        _, _, y = mlp_forward(xa, packed, dims, 0, slope, keep_hidden=False, x2=xb)
        ctx.save_for_backward(xa, xb, packed)
        ctx.dims, ctx.slope = dims, slope
        return y

    @staticmethod
    def backward(ctx, dy):
        xa, xb, packed = ctx.saved_tensors
        acc = mlp_wgrad_acc(ctx.dims[0], dy.device)
        dxb = mlp_backward_wgrad(dy.contiguous(), xa, None, None, packed, ctx.dims, ctx.slope, acc, want_dx=ctx.needs_input_grad[1], x2=xb)
        return (None, dxb, *mlp_wgrad_finish(ctx.dims, acc), None, None)


class KernelMLP(nn.Sequential):
    """Linear - act - Linear - act - Linear with the reference's module numbering; large float32 minibatches on the GPU go
    through the fused forward / backward kernels when the shape is one of the architecture's (`sss_mlp_supported`)"""

    # The two policy heads (53 / 36 -> 64 -> 64 -> 1, Tanh) go through the kernels as well since they run on the matrix cores
    # (csrc/sss_train16.h sss_mlp_head_mfma_*: 600 k rows forward + backward 0.77 / 0.65 ms against 1.46 / 1.28 ms for the three
    # library GEMMs with the MFMA weight-gradient kernel; their 16-lanes-per-row form was LDS-bound and slower, 2.50 / 1.54 ms).
    # The GNN's 32 / 16-wide MLPs: 2.5 M rows forward + backward ~0.9 ms against 2.1 - 2.3 ms (profiles/r03_ppo.md).
    FUSE_WIDE = True

    def _fused_spec(self):
        spec = getattr(self, "_spec", None)
        if spec is None:
            spec = False
            mods = list(self)
            if len(mods) == 5 and all(isinstance(mods[i], nn.Linear) and mods[i].bias is not None for i in (0, 2, 4)) and type(mods[1]) is type(mods[3]):
                act = mods[1]
                kind = (0, float(act.negative_slope)) if isinstance(act, nn.LeakyReLU) else (1, 0.0) if isinstance(act, nn.Tanh) else None
                dims = (mods[0].in_features, mods[0].out_features, mods[2].out_features, mods[4].out_features)
                if (kind is not None and mods[2].in_features == dims[1] and mods[4].in_features == dims[2] and (dims[1] <= 32 or self.FUSE_WIDE)
                        and _binding().lib.sss_mlp_supported(*dims, kind[0])):
                    spec = (kind[0], kind[1])
            self._spec = spec
        return spec

    def _packed_for_step(self) -> torch.Tensor:
        """the packed parameters, re-packed when an optimiser step (or anything else) has written to them"""
        l1, l2, l3 = self[0], self[2], self[4]
        ver = tuple(t._version for t in (l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias)) + (l1.weight.data_ptr(),)
        if getattr(self, "_pack_ver", None) != ver:
            self._pack, self._pack_ver = pack_mlp(l1, l2, l3), ver
        return self._pack

    def forward_cat(self, xa: torch.Tensor, xb: torch.Tensor) -> torch.Tensor:
        """`self(torch.cat([xa, xb], -1))` for 2-D xa (no gradient) and xb f32[rows, 16]: on the kernels without building the
        concatenation where the library can (`mlp_split`), else exactly that"""
        if (xa.is_cuda and xa.dtype == xb.dtype == torch.float32 and xa.dim() == xb.dim() == 2 and xb.shape[1] == 16 and xa.shape[0] >= MIN_ROWS and torch.is_grad_enabled()
                and self[0].weight.requires_grad and not xa.requires_grad and FUSED_WGRAD):
            spec = self._fused_spec()
            dims = (self[0].in_features, self[0].out_features, self[2].out_features, self[4].out_features)
            if spec and spec[0] == 0 and tuple(dims[1:]) == (32, 16, 16) and dims[0] == xa.shape[1] + 16 and mlp_split(dims[0]):
                l1, l2, l3 = self[0], self[2], self[4]
                return _MlpSplitFn.apply(xa.contiguous(), xb.contiguous(), l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, self._packed_for_step(), spec[1])
        return self(torch.cat([xa, xb], -1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and x.numel() // max(1, x.shape[-1]) >= MIN_ROWS and torch.is_grad_enabled()
                and self[0].weight.requires_grad):
            spec = self._fused_spec()
            if spec:
                l1, l2, l3 = self[0], self[2], self[4]
                y = _MlpFn.apply(x.reshape(-1, x.shape[-1]).contiguous(), l1.weight, l1.bias, l2.weight, l2.bias, l3.weight, l3.bias, self._packed_for_step(),
                                 spec[0], spec[1])
                return y.reshape(x.shape[:-1] + (y.shape[-1],))
        return super().forward(x)


class _MessagePassFn(torch.autograd.Function):
    """The node encoder's message passing (scheduler.py:209-232: for every DAG layer, deepest first,
    h[recv] = h_init[recv] + update(sum over the layer's edges (recv -> child) of msg(h[child]))) as ONE autograd node.

    The tensor-op form builds a full-size aggregate and a new copy of the whole embedding tensor per layer, forward and
    backward (`index_copy` of a functional graph). Here the embeddings are updated in place in a working buffer, the
    aggregates are compact (one row per receiving node), the backward pass walks the layers the other way on ONE gradient
    buffer (rows of receivers are taken out and cleared, gather gradients are added in), both MLPs run on the MLP kernels
    into slices of per-call buffers - so that their six weight gradients are six `sss_linear_wgrad` calls over all layers
    together. `layers`: [(child ids of the layer's edges, position of each edge's receiver in `recv`, recv)], in
    processing order."""

    DIMS = (16, 32, 16, 16)

    @staticmethod
    def forward(ctx, h_init, h0, layers, slope, packed_msg, packed_upd, *params):
        dev, D = h_init.device, _MessagePassFn.DIMS
        ne = sum(int(c.numel()) for c, _, _ in layers)
        nr = sum(int(r.numel()) for _, _, r in layers)
        f = lambda n, w: torch.empty((n, w), dtype=torch.float32, device=dev)  # noqa: E731
        keep = not mlp_recompute(16)  # (else the backward kernels recompute the hidden activations from mx / ux)
        mx, ma1, ma2 = f(ne, 16), f(ne, 32) if keep else None, f(ne, 16) if keep else None  # message MLP: inputs (and hidden activations) of every edge, layer after layer
        ux, ua1, ua2 = f(nr, 16), f(nr, 32) if keep else None, f(nr, 16) if keep else None  # update MLP: the same per receiving node
        sl = lambda t, lo, n: t[lo:lo + n] if t is not None else None  # noqa: E731
        # the embeddings are updated IN h0 (the caller's fresh per-leaf sums: marked dirty for autograd, returned as the result) - a copy of
        # the whole [M, 16] table per minibatch otherwise
        if h0.is_contiguous():
            ctx.mark_dirty(h0)
            h = h0
        else:
            h = h0.contiguous()
        h_init = h_init.contiguous()
        eo = ro = 0
        for child, pos, recv in layers:
            n_e, n_r = int(child.numel()), int(recv.numel())
            if n_e == 0 or n_r == 0:
                continue
            xs = mx[eo:eo + n_e]
            rows_op(ROWS_GATHER, child, xs, h)
            _, _, my = mlp_forward(xs, packed_msg, D, 0, slope, a1=sl(ma1, eo, n_e), a2=sl(ma2, eo, n_e), keep_hidden=keep)
            agg = ux[ro:ro + n_r]
            # the layer's edges are listed receiver by receiver (graph_layers: edge ids ascending, edges stored source node by
            # source node), so a receiver's messages are a range of rows: summed in order, no atomics
            rows_op(ROWS_SEGMENT_SUM, segment_offsets(pos, n_r), my, agg)
            _, _, uy = mlp_forward(agg, packed_upd, D, 0, slope, a1=sl(ua1, ro, n_r), a2=sl(ua2, ro, n_r), keep_hidden=keep)
            rows_op(ROWS_UPDATE, recv, uy, h, h_init)  # h[recv] = uy + h_init[recv] (every read of the layer came before this write)
            eo, ro = eo + n_e, ro + n_r
        ctx.layers, ctx.slope, ctx.keep = layers, slope, keep
        if keep:
            ctx.save_for_backward(mx, ma1, ma2, ux, ua1, ua2, packed_msg, packed_upd)
        else:
            ctx.save_for_backward(mx, ux, packed_msg, packed_upd)
        return h

    @staticmethod
    def backward(ctx, gh_in):
        if ctx.keep:
            mx, ma1, ma2, ux, ua1, ua2, packed_msg, packed_upd = ctx.saved_tensors
        else:
            (mx, ux, packed_msg, packed_upd), ma1, ma2, ua1, ua2 = ctx.saved_tensors, None, None, None, None
        sl = lambda t, lo, n: t[lo:lo + n] if t is not None else None  # noqa: E731
        layers, slope, D = ctx.layers, ctx.slope, _MessagePassFn.DIMS
        dev = gh_in.device
        f = lambda t: torch.empty_like(t)  # noqa: E731
        acc_m = mlp_wgrad_acc(16, dev) if FUSED_WGRAD else None
        acc_u = mlp_wgrad_acc(16, dev) if FUSED_WGRAD else None
        fused = acc_m is not None and acc_u is not None
        mdy, udy = f(mx), f(ux)
        assert fused or ctx.keep
        mg1, mg2, ug1, ug2 = (None,) * 4 if fused else (f(ma1), f(ma2), f(ua1), f(ua2))
        gh = gh_in.contiguous().clone()  # gradient w.r.t. the embeddings as they were before the layer being undone
        g_init = torch.zeros_like(gh)
        eo, ro = int(mx.shape[0]), int(ux.shape[0])
        for child, pos, recv in reversed(layers):
            n_e, n_r = int(child.numel()), int(recv.numel())
            if n_e == 0 or n_r == 0:
                continue
            eo, ro = eo - n_e, ro - n_r
            g_new = udy[ro:ro + n_r]
            # g_new = gh[recv]; gh[recv] = 0 (the receivers' previous embeddings were overwritten); g_init[recv] += g_new
            rows_op(ROWS_TAKE, recv, g_new, gh, g_init)
            if fused:  # the parameter gradients of every layer add up in the two accumulators
                g_agg = mlp_backward_wgrad(g_new, ux[ro:ro + n_r], sl(ua1, ro, n_r), sl(ua2, ro, n_r), packed_upd, D, slope, acc_u)
            else:
                _, _, g_agg = mlp_backward(g_new, ua1[ro:ro + n_r], ua2[ro:ro + n_r], packed_upd, D, 0, slope, g1=ug1[ro:ro + n_r], g2=ug2[ro:ro + n_r])
            g_msg = mdy[eo:eo + n_e]
            rows_op(ROWS_GATHER, pos, g_msg, g_agg)
            if fused:
                g_xs = mlp_backward_wgrad(g_msg, mx[eo:eo + n_e], sl(ma1, eo, n_e), sl(ma2, eo, n_e), packed_msg, D, slope, acc_m)
            else:
                _, _, g_xs = mlp_backward(g_msg, ma1[eo:eo + n_e], ma2[eo:eo + n_e], packed_msg, D, 0, slope, g1=mg1[eo:eo + n_e], g2=mg2[eo:eo + n_e])
            rows_op(ROWS_SCATTER_ADD, child, g_xs, gh)
        if fused:
            return (g_init, gh, None, None, None, None) + tuple(mlp_wgrad_finish(D, acc_m)) + tuple(mlp_wgrad_finish(D, acc_u))
        grads = []
        for x, a1, a2, dy, g1, g2 in ((mx, ma1, ma2, mdy, mg1, mg2), (ux, ua1, ua2, udy, ug1, ug2)):
            if x.shape[0] == 0:
                grads += [None] * 6
                continue
            gw1, gb1 = linear_wgrad(x, g1)
            gw2, gb2 = linear_wgrad(a1, g2)
            gw3, gb3 = linear_wgrad(a2, dy)
            grads += [gw1, gb1, gw2, gb2, gw3, gb3]
        return (g_init, gh, None, None, None, None) + tuple(grads)


def message_passing(h_init: torch.Tensor, h0: torch.Tensor, layers, mlp_msg: "KernelMLP", mlp_update: "KernelMLP") -> torch.Tensor:
    """`_MessagePassFn` for the two MLPs' current parameters (16 -> 32 -> 16 -> 16, LeakyReLU)"""
    slope = float(mlp_msg[1].negative_slope)
    pm, pu = mlp_msg._packed_for_step(), mlp_update._packed_for_step()
    params = [t for m in (mlp_msg, mlp_update) for lin in (m[0], m[2], m[4]) for t in (lin.weight, lin.bias)]
    return _MessagePassFn.apply(h_init, h0, layers, slope, pm, pu, *params)
