"""scatter_softmax as documented for torch-scatter 2.0.5: per segment and per
column, subtract the segment max, exponentiate, divide by (segment sum + eps)."""
import torch


def scatter_softmax(src, index, dim=-1, eps=1e-12):
    dim = dim % src.dim()
    shape = [1] * src.dim()
    shape[dim] = -1
    idx = index.view(shape).expand_as(src)
    size = list(src.shape)
    size[dim] = int(index.max()) + 1
    mx = src.new_zeros(size).scatter_reduce(dim, idx, src, reduce="amax", include_self=False)
    ex = (src - mx.gather(dim, idx)).exp()
    den = src.new_zeros(size).scatter_add_(dim, idx, ex)
    return ex / (den.gather(dim, idx) + eps)
