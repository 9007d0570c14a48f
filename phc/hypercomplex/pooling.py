"""Graph read-out (reference: phc/hypercomplex/pooling.py).  The per-graph sum
runs on the engine's segment walker (``phc_aggregate_fwd`` over rows grouped by
``batch``); its gradient is a row gather (``phc_gather_rows_i64``)."""
from typing import Optional, Union

import torch
import torch.nn as nn

from phc.hypercomplex.layers import PHMLinear, RealTransformer
from phc.hypercomplex.utils import any_forward_hooks
from phc_gnn_amd import ops as _engine


class PHMGlobalSumPooling(nn.Module):
    def __init__(self, phm_dim: int):
        super().__init__()
        self.phm_dim = phm_dim
        self.module = _engine.global_add_pool

    def forward(self, x: torch.Tensor, batch, num_graphs: Optional[int] = None, is_sorted: bool = False) -> torch.Tensor:
        # x: [num_nodes, phm_dim * feats];  batch: int64 graph id per node
        return self.module(x, batch, num_graphs, is_sorted)

    def reset_parameters(self):
        pass

    def __repr__(self):
        return f"{self.__class__.__name__}(phm_dim={self.phm_dim})"


class PHMSoftAttentionPooling(nn.Module):
    """gate = sigmoid(RealTransformer(PHMLinear(x)))  [N, k]; every component of
    x is scaled by the gate, then summed per graph   (ref :29-66)."""

    def __init__(self, embed_dim: int, phm_dim: int, phm_rule: Union[None, nn.ParameterList],
                 learn_phm: bool = True, bias: bool = True, w_init: str = "phm", c_init: str = "standard",
                 real_trafo: str = "linear"):
        super().__init__()
        self.embed_dim = embed_dim
        self.phm_dim = phm_dim
        self.w_init = w_init
        self.c_init = c_init
        self.phm_rule = phm_rule
        self.learn_phm = learn_phm
        self.real_trafo_type = real_trafo
        self.bias = bias
        self.linear = PHMLinear(in_features=embed_dim, out_features=embed_dim, phm_dim=phm_dim, phm_rule=phm_rule,
                                learn_phm=learn_phm, w_init=w_init, c_init=c_init, bias=bias)
        self.real_trafo = RealTransformer(type=real_trafo, phm_dim=phm_dim, in_features=embed_dim, bias=True)
        self.sigmoid = nn.Sigmoid()
        self.sum_pooling = PHMGlobalSumPooling(phm_dim=phm_dim)
        self.reset_parameters()

    def reset_parameters(self):
        self.real_trafo.reset_parameters()
        self.linear.reset_parameters()

    def forward(self, x: torch.Tensor, batch, num_graphs: Optional[int] = None, is_sorted: bool = False) -> torch.Tensor:
        # x has two consumers (the gate's PHM layer and the gated sum): the sum's share of dx rides into the layer's dx GEMM
        # instead of meeting it in an autograd add (engine: DxHandoff; sorted batches, i.e. the one-kernel read-out)
        handoff = _engine.DxHandoff() if (is_sorted and x.is_cuda and x.dim() == 2 and torch.is_grad_enabled() and x.requires_grad) \
            else None
        aff = getattr(self.real_trafo, "affine", None)
        if _engine.GATE_COMPOSED and isinstance(aff, nn.Linear) and x.is_cuda and x.dim() == 2 \
                and _engine.gate_composable(self.linear.W, aff.weight) and not any_forward_hooks(self.linear, self.real_trafo):
            # Linear(PHMLinear(x)) has nothing between its two maps: one GEMM with the composed [embed, embed / phm_dim] operand
            # (built by the step's operand launch), no [N, embed] intermediate — the same numbers up to fp32 round-off
            gate = _engine.gate_linear(x, self.linear.phm_rule, self.linear.W, self.linear.b, aff.weight, aff.bias, dx_handoff=handoff,
                                       lazy=bool(is_sorted))
        else:
            _engine.materialize_pending(x)
            gate = self.real_trafo(self.linear(x, dx_handoff=handoff))         # pre-sigmoid [N, embed_dim / phm_dim]
        # sigmoid, broadcast over the components, multiply and per-graph sum in one engine pass
        return _engine.gated_segment_sum(x, gate, batch, self.phm_dim, num_graphs=num_graphs, is_sorted=is_sorted,
                                         dx_handoff=handoff)

    def __repr__(self):
        return "{}(embed_dim={}, phm_dim={}, phm_rule={}, learn_phm={}, bias={}, w_init='{}', c_init='{}', " \
               "real_trafo='{}')".format(self.__class__.__name__, self.embed_dim, self.phm_dim, self.phm_rule,
                                         self.learn_phm, self.bias, self.w_init, self.c_init, self.real_trafo_type)
