"""Stand-in for torch-scatter 2.0.5 (only what the reference calls)."""
import torch
from . import composite  # noqa: F401


def _expand_index(index, src, dim):
    dim = dim % src.dim()
    if index.dim() == 1 and src.dim() > 1:
        shape = [1] * src.dim()
        shape[dim] = -1
        index = index.view(shape)
    return index.expand_as(src), dim


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce="sum"):
    idx, dim = _expand_index(index, src, dim)
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    if reduce in ("sum", "add"):
        res = src.new_zeros(shape)
        return res.scatter_add_(dim, idx, src)
    if reduce == "mean":
        res = src.new_zeros(shape).scatter_add_(dim, idx, src)
        cnt = src.new_zeros(shape).scatter_add_(dim, idx, torch.ones_like(src))
        return res / cnt.clamp(min=1)
    if reduce in ("min", "max"):
        res = src.new_zeros(shape)
        return res.scatter_reduce(dim, idx, src, reduce="a" + reduce, include_self=False)
    raise ValueError(reduce)


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "sum")


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "mean")


def scatter_min(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "min"), None


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    return scatter(src, index, dim, out, dim_size, "max"), None
