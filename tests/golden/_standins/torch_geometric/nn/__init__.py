import torch
from . import conv, inits  # noqa: F401


def global_add_pool(x, batch, size=None):
    size = int(batch.max()) + 1 if size is None else size
    out = x.new_zeros((size,) + tuple(x.shape[1:]))
    return out.index_add_(0, batch, x)
