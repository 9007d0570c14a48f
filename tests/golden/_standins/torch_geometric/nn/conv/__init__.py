"""MessagePassing with the documented PyG 1.6.1 semantics the reference relies
on: flow source_to_target, node_dim=-2; `foo_j` gathers with edge_index[0],
`foo_i` with edge_index[1]; aggregate = scatter(reduce=aggr) onto
edge_index[1] with dim_size = number of nodes; update = identity."""
import inspect
import torch
from torch_scatter import scatter


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super().__init__()
        self.aggr = aggr
        self.flow = flow
        self.node_dim = node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = (0, 1) if self.flow == "source_to_target" else (1, 0)
        n_nodes = None
        for v in kwargs.values():
            if torch.is_tensor(v) and v.dim() >= 2 and n_nodes is None:
                n_nodes = v.size(self.node_dim)
        if size is not None and size[1] is not None:
            n_nodes = size[1]
        wanted = inspect.signature(self.message).parameters
        args = {}
        for name in wanted:
            if name.endswith("_j") or name.endswith("_i"):
                base = kwargs[name[:-2]]
                sel = edge_index[src] if name.endswith("_j") else edge_index[dst]
                args[name] = base.index_select(self.node_dim, sel)
            else:
                args[name] = kwargs.get(name)
        msg = self.message(**args)
        out = self.aggregate(msg, edge_index[dst], dim_size=n_nodes)
        return self.update(out)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, dim_size=None):
        return scatter(inputs, index, dim=self.node_dim, dim_size=dim_size, reduce=self.aggr)

    def update(self, inputs):
        return inputs
