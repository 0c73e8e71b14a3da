"""Test-only stand-in for torch_scatter.scatter (sum / mean / max / min along one dim).

mean = sum / clamp(count, min=1) (empty segment -> 0); max/min over present entries, empty -> 0
(torch_scatter documentation).
"""
import torch


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    assert out is None
    if dim < 0:
        dim = src.dim() + dim
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() > 0 else 0
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        idx = index.view(shape).expand_as(src)
    else:
        idx = index
    oshape = list(src.shape)
    oshape[dim] = dim_size
    if reduce in ("sum", "add"):
        return torch.zeros(oshape, dtype=src.dtype, device=src.device).scatter_add(dim, idx, src)
    if reduce == "mean":
        s = torch.zeros(oshape, dtype=src.dtype, device=src.device).scatter_add(dim, idx, src)
        cnt = torch.zeros(oshape, dtype=src.dtype, device=src.device).scatter_add(dim, idx, torch.ones_like(src))
        return s / cnt.clamp(min=1)
    if reduce in ("max", "min"):
        red = "amax" if reduce == "max" else "amin"
        fill = float("-inf") if reduce == "max" else float("inf")
        o = torch.full(oshape, fill, dtype=src.dtype, device=src.device).scatter_reduce(dim, idx, src, reduce=red, include_self=True)
        return torch.where(torch.isinf(o), torch.zeros_like(o), o)
    raise ValueError(reduce)
