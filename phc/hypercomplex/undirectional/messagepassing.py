"""PHC graph-convolution operators (reference:
phc/hypercomplex/undirectional/messagepassing.py).

The reference subclasses PyG's ``MessagePassing(aggr='add')``:
``propagate`` gathers ``x[edge_index[0]]``, ``message`` applies
``msg_encoder(x_j + edge_attr)``, ``aggregate`` scatter-adds onto
``edge_index[1]``.  Here those three steps plus the self-loop add are ONE engine
call (``phc_aggregate_fwd``, destination-sorted segments, one wavefront per
node) and the transform is the PHM kernel; no PyG / torch-scatter involved.
Constructor signatures, attribute names (``transform``, ``msg_encoder`` ...) and
``forward(x, edge_index, edge_attr, size)`` are the reference's."""
from typing import Optional, Union

import torch
import torch.nn as nn

from phc.hypercomplex.layers import PHMLinear, PHMMLP
from phc.quaternion.activations import get_module_activation
from phc_gnn_amd import ops as _engine

Adj = torch.Tensor
Size = Optional[tuple]

_FUSED_MSG_ENCODERS = ("identity", "relu", "lrelu", "elu", "swish", "selu")


def _check_sum_aggr(aggr: str, who: str) -> None:
    if aggr not in ("add", "sum"):
        raise NotImplementedError(f"{who}: aggr='{aggr}' is not implemented by the MI355X engine "
                                  "(scatter-sum only; mean/min/max are out of scope, SURVEY.md §2.1 #7)")


class _SumConvBase(nn.Module):
    """Shared plumbing of the two sum-aggregation operators."""

    flow = "source_to_target"        # PyG defaults the reference relies on
    node_dim = -2

    def _setup_message(self, msg_encoder: str) -> None:
        self.msg_encoder_str = msg_encoder
        self.msg_encoder = get_module_activation(activation=msg_encoder)
        if msg_encoder.lower() not in _FUSED_MSG_ENCODERS:
            raise ValueError(f"msg_encoder '{msg_encoder}' unknown")

    def propagate(self, edge_index: Adj, x: torch.Tensor, edge_attr: torch.Tensor, size: Size = None,
                  add_self: bool = False) -> torch.Tensor:
        """sum_{j->i} msg_encoder(x_j + edge_attr_ji)  (+ x_i when add_self)."""
        assert x.size(-1) == edge_attr.size(-1)                 # reference message() assert (:73,:145)
        num_dst = size[1] if size is not None and size[1] is not None else None
        return _engine.aggregate_sum(x, edge_index, edge_attr, self.msg_encoder_str, add_self, num_dst)

    def propagate_encoded(self, edge_index: Adj, x: torch.Tensor, edge_idx: torch.Tensor, encoder, size: Size = None,
                          add_self: bool = False, dx_handoff=None, pending=None, lazy: bool = False, dx_link=None) -> torch.Tensor:
        """Same as ``propagate`` with edge_attr = encoder(edge_idx), evaluated inside the
        aggregation kernel from LDS-staged embedding tables (no [E, h] tensor).  ``dx_handoff`` (engine: DxHandoff): another
        consumer of ``x`` may park its gradient there; this op's backward then adds it inside its own dx kernel.  ``dx_link``
        (engine: DxLink): this op is x's only autograd consumer and x's producer may build this op's dx inside its own backward launch."""
        tables = encoder.engine_tables
        num_dst = size[1] if size is not None and size[1] is not None else None
        # pending (engine: PendingNorm): x = act(norm(y)) (+ skip) of the layer in front has not been evaluated yet; the
        # aggregation kernel does it on the fly (or runs the stand-alone pass first where it cannot)
        return _engine.aggregate_sum_encoded(x, edge_index, edge_idx, _engine.tableset_for(encoder, tables), tables,
                                             self.msg_encoder_str, add_self, num_dst, dx_handoff=dx_handoff, pending=pending, lazy=lazy,
                                             dx_link=dx_link)

    def message(self, x_j: torch.Tensor, edge_attr: torch.Tensor) -> torch.Tensor:
        """Per-edge message (API parity; the engine fuses this into the aggregation)."""
        assert x_j.size(-1) == edge_attr.size(-1)
        return self.msg_encoder(x_j + edge_attr)


class PHMConv(_SumConvBase):
    """Sum aggregation followed by ONE PHMLinear   (ref :19-88).
    same_dim=True:  y = transform(agg) + x      (self loop after the transform)
    same_dim=False: y = transform(agg + x)."""

    def __init__(self, in_features: int, out_features: int, phm_dim: int, phm_rule, learn_phm: True,
                 bias: bool = True, add_self_loops: bool = True, w_init: str = "phm", c_init: str = "standard",
                 aggr: str = "add", same_dim: bool = True, msg_encoder: str = "identity") -> None:
        super().__init__()
        _check_sum_aggr(aggr, "PHMConv")
        self.in_features = in_features
        self.out_features = out_features
        self.phm_dim = phm_dim
        self.phm_rule = phm_rule
        self.learn_phm = learn_phm
        self.bias = bias
        self.add_self_loops = add_self_loops
        self.w_init = w_init
        self.c_init = c_init
        self.aggr = aggr
        self.same_dim = same_dim
        self.transform = PHMLinear(in_features=in_features, out_features=out_features, phm_rule=phm_rule,
                                   phm_dim=phm_dim, bias=bias, w_init=w_init, c_init=c_init, learn_phm=learn_phm)
        self._setup_message(msg_encoder)
        self.reset_parameters()

    def reset_parameters(self):
        self.transform.reset_parameters()

    def forward(self, x: torch.Tensor, edge_index: Adj, edge_attr: torch.Tensor, size: Size = None) -> torch.Tensor:
        if self.same_dim:
            agg = self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size, add_self=False)
            lin = self.transform
            # residual (the self loop) rides in the GEMM epilogue
            return _engine.phm_linear(agg, lin.phm_rule, lin.W, lin.b, x if self.add_self_loops else None)
        agg = self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size, add_self=self.add_self_loops)
        return self.transform(agg)

    def forward_encoded(self, x: torch.Tensor, edge_index: Adj, edge_idx: torch.Tensor, encoder, size: Size = None,
                        n_valid=None, bn_stats: bool = False):
        """forward() with edge_attr given as integer bond features + their PHMEncoder.  ``bn_stats`` (engine): the model's norm
        follows (models.py:208-215) — the transform's GEMM also emits the batch-norm statistics of its output, so that norm is one
        apply pass."""
        stats = _engine.BNStatsRequest(n_valid, consumer="apply") if (bn_stats and x.is_cuda and x.dim() == 2) else None
        if self.same_dim:
            # x is read twice — by the aggregation and as the self loop behind the transform; the self loop's share of dx (= dy of the
            # transform) is parked and added inside the aggregation's backward walk instead of by an add launch
            share = _engine.DxHandoff() if (_engine.RESIDUAL_HANDOFF and self.add_self_loops and x.is_cuda and x.requires_grad
                                            and torch.is_grad_enabled()) else None
            agg = self.propagate_encoded(edge_index, x, edge_idx, encoder, size, add_self=False, dx_handoff=share)
            lin = self.transform
            return _engine.phm_linear(agg, lin.phm_rule, lin.W, lin.b, x if self.add_self_loops else None, bn_stats=stats,
                                      res_handoff=share)
        agg = self.propagate_encoded(edge_index, x, edge_idx, encoder, size, add_self=self.add_self_loops)
        return self.transform(agg, bn_stats=stats) if stats is not None else self.transform(agg)

    def __repr__(self):
        return "{}(in_features={}, out_features={}, phm_dim={}, phm_rule={}, learn_phm={}, bias={}, " \
               "add_self_loops={}, w_init='{}', c_init='{}', aggr='{}')".format(
                   self.__class__.__name__, self.in_features, self.out_features, self.phm_dim, self.phm_rule,
                   self.learn_phm, self.bias, self.add_self_loops, self.w_init, self.c_init, self.aggr)


class PHMGINEConv(_SumConvBase):
    """GIN-style operator: y = PHMMLP(agg + x)   (ref :91-161)."""

    def __init__(self, in_features: int, out_features: int, phm_dim: int, phm_rule, learn_phm: bool = True,
                 bias: bool = True, add_self_loops: bool = True, norm: Optional[str] = None,
                 activation: str = "relu", w_init: str = "phm", c_init: str = "standard", aggr: str = "add",
                 msg_encoder: str = "identity") -> None:
        super().__init__()
        _check_sum_aggr(aggr, "PHMGINEConv")
        self.in_features = in_features
        self.out_features = out_features
        self.phm_dim = phm_dim
        self.bias = bias
        self.add_self_loops = add_self_loops
        self.learn_phm = learn_phm
        self.phm_rule = phm_rule
        self.norm = norm
        self.activation_str = activation
        self.w_init = w_init
        self.c_init = c_init
        self.aggr = aggr
        self.transform = PHMMLP(in_features=in_features, out_features=out_features, phm_dim=phm_dim,
                                phm_rule=phm_rule, w_init=w_init, c_init=c_init, factor=1, bias=bias,
                                activation=activation, norm=norm)
        self._setup_message(msg_encoder)
        self.reset_parameters()

    def reset_parameters(self):
        self.transform.reset_parameters()

    def forward(self, x: torch.Tensor, edge_index: Adj, edge_attr: torch.Tensor, size: Size = None,
                n_valid=None) -> torch.Tensor:
        agg = self.propagate(edge_index, x=x, edge_attr=edge_attr, size=size, add_self=self.add_self_loops)
        return self.transform(agg, n_valid=n_valid)

    def forward_encoded(self, x: torch.Tensor, edge_index: Adj, edge_idx: torch.Tensor, encoder, size: Size = None,
                        n_valid=None):
        """forward() with edge_attr given as integer bond features + their PHMEncoder."""
        agg = self.propagate_encoded(edge_index, x, edge_idx, encoder, size, add_self=self.add_self_loops)
        return self.transform(agg, n_valid=n_valid)

    def __repr__(self):
        return "{}(in_features={}, out_features={}, phm_dim={}, phm_rule={}, learn_phm={}, bias={}, " \
               "add_self_loops={}, norm='{}', activation='{}', w_init='{}', c_init='{}', aggr='{}')".format(
                   self.__class__.__name__, self.in_features, self.out_features, self.phm_dim, self.phm_rule,
                   self.learn_phm, self.bias, self.add_self_loops, self.norm, self.activation_str, self.w_init,
                   self.c_init, self.aggr)


class _SoftmaxMixin:
    """beta handling + softmax aggregate shared by the two softmax operators   (ref :194-214, 280-300)."""

    def _setup_beta(self, kwargs) -> None:
        self.initial_beta = kwargs.get("initial_beta")
        self.learn_beta = kwargs.get("learn_beta")
        if self.initial_beta is None:
            self.initial_beta = 1.0
        self.beta = nn.Parameter(torch.tensor(float(self.initial_beta)), requires_grad=bool(self.learn_beta))

    def propagate(self, edge_index: Adj, x: torch.Tensor, edge_attr: torch.Tensor, size: Size = None,
                  add_self: bool = False) -> torch.Tensor:
        assert x.size(-1) == edge_attr.size(-1)
        num_dst = size[1] if size is not None and size[1] is not None else None
        # the self loop (x_c added to the aggregate, ref :222-231 / :303-309) rides in the aggregation's own launches
        fused_self = add_self and (num_dst is None or num_dst == x.size(0))
        out = _engine.aggregate_softmax(x, edge_index, edge_attr, self.beta, self.msg_encoder_str, num_dst, add_self=fused_self)
        return out + x if (add_self and not fused_self) else out

    def propagate_encoded(self, edge_index, x, edge_idx, encoder, size=None, add_self=False, dx_handoff=None, pending=None, lazy=False, dx_link=None):
        # (dx_handoff: the skips' shares of dx are added inside the softmax backward walk — engine phc_aggregate_softmax_enc_bwd_res;
        # `pending`: the un-applied norm of the layer in front rides in the walk — engine PendingNorm; `lazy`: the walk may be left to
        # the fused MLP + norm node's one-launch forward — engine PendingSoftmaxAggregate)
        """Softmax aggregation with edge_attr = encoder(edge_idx): the kernels add the F table rows of every edge
        themselves (packed codes, tables from L2), so no [E, h] embedding is built in the forward pass."""
        tables = encoder.engine_tables
        ts = _engine.tableset_for(encoder, tables).vector_form(tables)     # (k % 4 != 0: the flat one-component image)
        num_dst = size[1] if size is not None and size[1] is not None else None
        if ts.F <= 4 and ts.R <= 256 and ts.k % 4 == 0 and (num_dst is None or num_dst == x.size(0) or not add_self):
            return _engine.aggregate_softmax_encoded(x, edge_index, edge_idx, ts, tables, self.beta, self.msg_encoder_str,
                                                     add_self=add_self, num_dst=num_dst, pending=pending, lazy=lazy,
                                                     dx_handoff=dx_handoff)
        _engine.materialize_pending(x)
        ea = encoder(edge_idx)                               # wider vocabularies: embed the bonds and take the tensor path
        return self.propagate(edge_index, x=x, edge_attr=ea.reshape(ea.size(0), -1), size=size, add_self=add_self)

    def aggregate(self, inputs: torch.Tensor, index: torch.Tensor, dim_size: Optional[int] = None) -> torch.Tensor:
        """API parity with the reference's override (scatter_softmax + scatter_sum on explicit messages)."""
        n = int(index.max()) + 1 if dim_size is None else dim_size
        ei = torch.stack([torch.arange(index.numel(), device=index.device), index])
        zeros = torch.zeros_like(inputs)
        return _engine.aggregate_softmax(inputs, _engine.EdgeStructure(ei, index.numel(), n), zeros, self.beta, "identity")


class PHMConvSoftmax(_SoftmaxMixin, PHMConv):
    """PHMConv with softmax aggregation and a learnable inverse temperature beta   (ref :164-245)."""

    def __init__(self, in_features: int, out_features: int, phm_dim: int, phm_rule, learn_phm: True, bias: bool = True,
                 add_self_loops: bool = True, w_init: str = "phm", c_init: str = "standard", aggr: str = "softmax",
                 same_dim: bool = True, msg_encoder: str = "identity", **kwargs) -> None:
        PHMConv.__init__(self, in_features, out_features, phm_dim, phm_rule, learn_phm, bias, add_self_loops, w_init, c_init,
                         "add", same_dim, msg_encoder)
        self.aggr = aggr
        self._setup_beta(kwargs)

    def reset_parameters(self):
        self.transform.reset_parameters()
        if hasattr(self, "beta"):
            self.beta.data.fill_(float(self.initial_beta))


class PHMGINEConvSoftmax(_SoftmaxMixin, PHMGINEConv):
    """PHMGINEConv with softmax aggregation   (ref :248-327)."""

    def __init__(self, in_features: int, out_features: int, phm_dim: int, phm_rule, learn_phm: bool = True, bias: bool = True,
                 add_self_loops: bool = True, norm: Optional[str] = None, activation: str = "relu", w_init: str = "phm",
                 c_init: str = "standard", aggr: str = "softmax", msg_encoder: str = "identity", **kwargs) -> None:
        PHMGINEConv.__init__(self, in_features, out_features, phm_dim, phm_rule, learn_phm, bias, add_self_loops, norm,
                             activation, w_init, c_init, "add", msg_encoder)
        self.aggr = aggr
        self._setup_beta(kwargs)

    def reset_parameters(self):
        self.transform.reset_parameters()
        if hasattr(self, "beta"):
            self.beta.data.fill_(float(self.initial_beta))


class PHMMessagePassing(nn.Module):
    """Dispatcher over the graph-conv operators   (ref :456-518): ``mlp`` picks
    PHMGINEConv over PHMConv; ``aggr`` in {"add", "sum"} is served by the engine."""

    def __init__(self, in_features: int, out_features: int, phm_dim: int, phm_rule, learn_phm: bool = True,
                 bias: bool = True, add_self_loops: bool = True, norm: Optional[str] = None,
                 activation: str = "relu", w_init: str = "phm", c_init: str = "standard", aggr: str = "add",
                 mlp: bool = True, same_dim: bool = True, msg_encoder: str = "identity", **kwargs):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.phm_dim = phm_dim
        self.bias = bias
        self.add_self_loops = add_self_loops
        self.norm = norm
        self.learn_phm = learn_phm
        self.phm_rule = phm_rule
        self.activation_str = activation
        self.w_init = w_init
        self.c_init = c_init
        self.aggr = aggr
        self.mlp = mlp
        self.same_dim = same_dim
        self.msg_encoder_str = msg_encoder

        if aggr == "pna":
            raise NotImplementedError("aggr='pna' (PHMPNAConvSimple) is outside the MI355X engine's scope")
        if aggr == "softmax":
            if mlp:
                self.transform = PHMGINEConvSoftmax(in_features, out_features, phm_dim, phm_rule, learn_phm, bias,
                                                    add_self_loops, norm, activation, w_init, c_init, aggr, msg_encoder,
                                                    **kwargs)
            else:
                self.transform = PHMConvSoftmax(in_features, out_features, phm_dim, phm_rule, learn_phm, bias,
                                                add_self_loops, w_init, c_init, aggr, same_dim, msg_encoder, **kwargs)
        elif mlp:
            self.transform = PHMGINEConv(in_features, out_features, phm_dim, phm_rule, learn_phm, bias,
                                         add_self_loops, norm, activation, w_init, c_init, aggr, msg_encoder)
        else:
            self.transform = PHMConv(in_features, out_features, phm_dim, phm_rule, learn_phm, bias, add_self_loops,
                                     w_init, c_init, aggr, same_dim, msg_encoder)
        self.reset_parameters()

    def reset_parameters(self):
        self.transform.reset_parameters()

    def get_num_params(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def forward(self, x: torch.Tensor, edge_index: Adj, edge_attr: torch.Tensor, size: Size = None) -> torch.Tensor:
        return self.transform(x, edge_index, edge_attr, size)

    def forward_encoded(self, x: torch.Tensor, edge_index: Adj, edge_idx: torch.Tensor, encoder, size: Size = None,
                        n_valid=None, bn_stats: bool = False):
        """Engine extension: like forward(), but takes the raw integer edge features and the
        PHMEncoder that embeds them, so the embedding is fused into the aggregation.  ``n_valid``
        marks the padding rows of a fixed-shape batch (device int32 scalar)."""
        if bn_stats and isinstance(self.transform, PHMConv):
            return self.transform.forward_encoded(x, edge_index, edge_idx, encoder, size, n_valid=n_valid, bn_stats=True)
        return self.transform.forward_encoded(x, edge_index, edge_idx, encoder, size, n_valid=n_valid)
