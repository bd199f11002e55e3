"""functional stand-in for torch_scatter (not installed): only what the reference's Decima code
calls. Test infrastructure for tests/golden/make_decima_golden.py."""
import torch


def segment_csr(src, indptr, out=None, reduce="sum"):
    assert reduce == "sum"
    n = indptr.numel() - 1
    counts = (indptr[1:] - indptr[:-1]).to(torch.long)
    seg = torch.repeat_interleave(torch.arange(n, device=src.device), counts.to(src.device))
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    res.index_add_(0, seg, src[: seg.numel()])
    return res
