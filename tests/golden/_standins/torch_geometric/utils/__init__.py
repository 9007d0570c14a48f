import torch


def degree(index, num_nodes=None, dtype=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    out = torch.zeros(n, dtype=dtype or torch.float)
    return out.scatter_add_(0, index, torch.ones(index.numel(), dtype=out.dtype))
