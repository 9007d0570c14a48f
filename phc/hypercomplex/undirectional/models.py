"""Graph-property models (reference: phc/hypercomplex/undirectional/models.py).

``PHMSkipConnectAdd`` is the harness BASELINE.json's metric is measured
through: encoders -> L x [conv -> norm -> activation -> dropout -> + skip] ->
pooling -> downstream head.  Sub-module names (``atomencoder``,
``bondencoders``, ``convs``, ``norms``, ``pooling``, ``downstream``) are read by
the training scripts (benchmarks/train_hiv.py:266-282) and are preserved."""
from typing import Optional, Union

import torch
import torch.nn as nn

from phc.hypercomplex.downstream import PHMDownstreamNet
from phc.hypercomplex.encoder import NaivePHMEncoder, PHMEncoder
from phc.hypercomplex.layers import PHMMLP, phm_dropout
from phc.hypercomplex.norm import PHMNorm
from phc.hypercomplex.pooling import PHMGlobalSumPooling, PHMSoftAttentionPooling
from phc.hypercomplex.undirectional.messagepassing import PHMMessagePassing
from phc.hypercomplex.utils import any_forward_hooks as _any_hooks, get_multiplication_matrices
from phc.quaternion.activations import get_module_activation
from phc_gnn_amd import ops as _engine

# ogb 1.2.4 feature vocabularies (ogb.utils.features.get_{atom,bond}_feature_dims);
# the reference imports them from ogb at module import time (ref :18-21).
ATOM_FEAT_DIMS = [119, 4, 12, 12, 10, 6, 6, 2, 2]
BOND_FEAT_DIMS = [5, 6, 2]


class PHMSkipConnectAdd(nn.Module):
    """Undirectional message-passing network with additive skip connections.

    Observability contract: on the engine's fast path a layer's last element-wise pass may be PARKED
    (``ops.PendingNorm`` / ``PendingAggregate`` / ``PendingGate``) and evaluated inside the next launch, and sub-modules are driven
    through fused entry points rather than ``__call__``.  That is invisible to callers of ``model(data)``.  Whoever wants to
    LOOK between the layers registers forward hooks on ``convs[i]`` / ``norms[i]`` / ``pooling`` as with the reference — the
    model then runs the reference's own sequence through every module's ``__call__`` (ref :200-217), and every hook sees
    finished tensors."""

    def __init__(self, phm_dim: int = 4, learn_phm: bool = True, phm_rule=None,
                 atom_input_dims: Union[int, list] = ATOM_FEAT_DIMS, atom_encoded_dim: int = 196,
                 bond_input_dims: Union[int, list] = BOND_FEAT_DIMS, naive_encoder: bool = False,
                 w_init: str = "phm", c_init: str = "standard", same_dropout: bool = False,
                 mp_layers: list = [196, 196, 196], bias: bool = True, dropout_mpnn: list = [0.0, 0.0, 0.0],
                 norm_mp: Optional[str] = "naive-batch-norm", add_self_loops: bool = True, msg_aggr: str = "add",
                 node_aggr: str = "sum", mlp: bool = False, pooling: str = "softattention",
                 activation: str = "relu", real_trafo: str = "linear", downstream_layers: list = [256, 128],
                 target_dim: int = 1, dropout_dn: Union[list, float] = [0.2, 0.1],
                 norm_dn: Optional[str] = "naive-batch-norm", msg_encoder: str = "identity",
                 sc_type: str = "first", **kwargs) -> None:
        super().__init__()
        assert all(x == atom_encoded_dim == mp_layers[0] for x in mp_layers), "dimensionalities need to match for model"
        assert activation.lower() in ["relu", "lrelu", "elu", "selu", "swish"]
        assert len(dropout_mpnn) == len(mp_layers)
        assert pooling in ["globalsum", "softattention"], f"pooling variable '{pooling}' wrong."
        assert norm_mp in [None, "naive-batch-norm", "None", "naive-naive-batch-norm"]
        assert norm_dn in [None, "naive-batch-norm", "None", "naive-naive-batch-norm"]
        assert w_init in ["phm", "glorot_uniform", "glorot_normal"], f"w_init variable '{w_init}' wrong."
        assert c_init in ["standard", "random"], f"c_init variable '{c_init}' wrong."
        assert sc_type in ["first", "last"], f"sc_type variable '{sc_type}' wrong."

        if msg_aggr == "sum":
            msg_aggr = "add"
        self.msg_encoder_str = msg_encoder
        self.phm_rule = phm_rule
        self.variable_phm = phm_rule is None
        self.phm_dim = phm_dim
        self.learn_phm = learn_phm
        self.atom_input_dims = atom_input_dims
        self.bond_input_dims = bond_input_dims
        self.atom_encoded_dim = atom_encoded_dim // phm_dim          # per component, as in the reference (:72)
        self.naive_encoder = naive_encoder
        self.w_init = w_init
        self.c_init = c_init
        self.same_dropout = same_dropout
        self.mp_layers = mp_layers
        self.bias = bias
        self.dropout_mpnn = dropout_mpnn
        self.norm_mp = None if norm_mp == "None" else norm_mp
        self.add_self_loops = add_self_loops
        self.msg_aggr_type = msg_aggr
        self.node_aggr_type = node_aggr
        self.mlp_mp = mlp
        self.pooling_type = pooling
        self.activation_str = activation
        self.real_trafo_type = real_trafo
        self.downstream_layers = downstream_layers
        self.target_dim = target_dim
        self.dropout_dn = dropout_dn
        self.norm_dn_type = None if norm_dn == "None" else norm_dn
        self.input_dim = atom_encoded_dim
        self.f_act = get_module_activation(self.activation_str)
        self.sc_type = sc_type
        # registration order follows the reference (convs, norms, encoders, pooling, downstream)
        # so that parameters() / state_dict() enumerate identically
        self.convs = nn.ModuleList()
        self.norms = nn.ModuleList()

        Encoder = NaivePHMEncoder if naive_encoder else PHMEncoder
        self.atomencoder = Encoder(out_dim=self.atom_encoded_dim, input_dims=atom_input_dims, phm_dim=phm_dim,
                                   combine="sum")
        self.bondencoders = nn.ModuleList([Encoder(out_dim=d // phm_dim, input_dims=bond_input_dims,
                                                   phm_dim=phm_dim, combine="sum") for d in mp_layers])
        for i, out_dim in enumerate(mp_layers):
            in_dim = self.input_dim if i == 0 else mp_layers[i - 1]
            self.convs.append(PHMMessagePassing(in_features=in_dim, out_features=out_dim, bias=bias, phm_dim=phm_dim,
                                           learn_phm=learn_phm, phm_rule=self.phm_rule, norm=self.norm_mp,
                                           activation=activation, w_init=w_init, c_init=c_init, aggr=msg_aggr,
                                           mlp=mlp, add_self_loops=add_self_loops, same_dim=True,
                                           msg_encoder=msg_encoder, **kwargs))
            self.norms.append(PHMNorm(num_features=out_dim, phm_dim=phm_dim, type=norm_mp) if self.norm_mp else None)
            if self.norm_mp:
                # the model normalises the conv's MLP output directly: let that MLP's last GEMM emit the statistics
                for m in self.convs[-1].modules():
                    if isinstance(m, PHMMLP):
                        m.out_norm_follows = True

        if pooling == "globalsum":
            self.pooling = PHMGlobalSumPooling(phm_dim=phm_dim)
        else:
            self.pooling = PHMSoftAttentionPooling(embed_dim=mp_layers[-1], phm_dim=phm_dim, learn_phm=learn_phm,
                                                   phm_rule=self.phm_rule, w_init=w_init, c_init=c_init, bias=bias,
                                                   real_trafo=real_trafo)
        self.downstream = PHMDownstreamNet(in_features=mp_layers[-1], hidden_layers=downstream_layers,
                                           out_features=target_dim, phm_rule=self.phm_rule, phm_dim=phm_dim,
                                           learn_phm=learn_phm, activation=activation, bias=bias,
                                           norm=self.norm_dn_type, w_init=w_init, c_init=c_init, dropout=dropout_dn,
                                           same_dropout=same_dropout, real_trafo=real_trafo)
        self.reset_parameters()

    def reset_parameters(self):
        if not self.variable_phm:
            for i, init in enumerate(get_multiplication_matrices(phm_dim=self.phm_dim)):
                self.phm_rule[i].data = init
        self.atomencoder.reset_parameters()
        for enc in self.bondencoders:
            enc.reset_parameters()
        for conv, norm in zip(self.convs, self.norms):
            conv.reset_parameters()
            if norm is not None:
                norm.reset_parameters()
        self.pooling.reset_parameters()
        self.downstream.reset_parameters()

    def get_number_of_params_(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def compute_hidden_layer_embedding(self, conv, norm, x, edge_index, edge_attr, dropout_mpnn: float,
                                       size=None, edge_encoder=None, n_valid=None, skip_sum=None, first_layer=False,
                                       defer_norm=False) -> torch.Tensor:
        """x = [input of the conv, skip tensor]   (ref :200-217).  With ``edge_encoder`` given,
        ``edge_attr`` holds the raw integer bond features and the embedding is evaluated
        inside the aggregation kernel."""
        if getattr(self, "_observable_now", False):
            # somebody registered forward hooks: the reference's sequence (ref :200-217), every module through its __call__
            _engine.materialize_pending(x[0])
            if n_valid is not None:
                raise NotImplementedError("forward hooks are not available on fixed-shape (padded / hipGraph-replayed) batches: "
                                          "a replay does not run Python; remove the hooks or run the plain batch")
            if edge_encoder is not None:
                e = edge_encoder(edge_attr)
                edge_attr = e.reshape(e.size(0), -1)
            h = conv(x=x[0], edge_index=edge_index, edge_attr=edge_attr, size=size)
            if norm is not None:
                h = norm(h)
            h = self.f_act(h)
            h = phm_dropout(x=h, p=dropout_mpnn, phm_dim=self.phm_dim, training=self.training, same=self.same_dropout)
            return h + x[1]
        dropout_active = self.training and dropout_mpnn > 0.0
        tr = getattr(conv, "transform", None)
        # the layer in front may have parked its last element-wise pass (engine: PendingNorm on x[0]): only the fused branch below
        # passes it on to the aggregation kernel, every other path evaluates it now
        pending = getattr(x[0], "_phc_pending", None)
        # (with dropout the mask rides in the fused node's norm passes: engine HS_DR, round 5; MLP_NORM_DROPOUT=0 restores the separate nodes)
        fused_branch = self.training and norm is not None and (not dropout_active or _engine.MLP_NORM_DROPOUT) and self.mlp_mp and \
            x[0].is_cuda and hasattr(tr, "propagate_encoded") and hasattr(getattr(tr, "transform", None), "linear2")
        if pending is not None and not (fused_branch and edge_encoder is not None and
                                        type(tr).__name__ in ("PHMGINEConv", "PHMGINEConvSoftmax")):
            _engine.materialize_pending(x[0])
            pending = None
        if fused_branch:
            # PHMGINEConv + model norm, training mode: aggregation, then the MLP and the norm as ONE autograd node whose
            # backward folds both batch-norm backwards into the dx GEMMs (engine: mlp_norm_fused); None = not applicable
            # the conv's input is also the layer's skip tensor (sc_type "last"): the skip's share of dx is handed to the
            # aggregation's backward, which adds it inside its dx kernel — x then has ONE gradient and autograd launches no add
            handoff = _engine.DxHandoff() if (x[0] is x[1] and edge_encoder is not None and x[0].requires_grad) else None
            # sc_type "first": every layer's skip is the encoded atoms, which also feed the first conv — the skips' shares of their
            # gradient are parked in `skip_sum` by every layer and added inside the FIRST layer's aggregation backward
            agg_handoff = handoff
            if handoff is None and skip_sum is not None and edge_encoder is not None and x[0].requires_grad:
                agg_handoff = skip_sum if first_layer else None
                handoff = skip_sum
            if edge_encoder is not None:
                # lazy: the aggregate goes straight into the fused node below, whose first GEMM may build the rows itself
                # (engine: PendingAggregate; mlp_norm_fused launches the aggregation on its own when it does not apply)
                lazy = type(tr).__name__ in ("PHMGINEConv", "PHMGINEConvSoftmax")
                # the aggregation is x[0]'s ONLY autograd consumer (the skip's share travels through a hand-off, or the skip is another
                # tensor): its dx may be built inside the backward launch of the fused node that produced x[0] (engine: DxLink)
                single = (handoff is not None and agg_handoff is handoff and x[0] is x[1]) or \
                    (skip_sum is not None and x[0] is not x[1] and not first_layer)
                dx_link = getattr(x[0], "_phc_dx_link", None) if (single and lazy) else None
                if pending is not None:
                    agg = tr.propagate_encoded(edge_index, x[0], edge_attr, edge_encoder, size, add_self=tr.add_self_loops,
                                               dx_handoff=agg_handoff, pending=pending, lazy=lazy, dx_link=dx_link)
                    x[0]._phc_pending = None                 # evaluated (or materialised) by that call, or riding with the aggregate
                elif lazy:
                    agg = tr.propagate_encoded(edge_index, x[0], edge_attr, edge_encoder, size, add_self=tr.add_self_loops,
                                               dx_handoff=agg_handoff, lazy=True, dx_link=dx_link)
                else:
                    agg = tr.propagate_encoded(edge_index, x[0], edge_attr, edge_encoder, size, add_self=tr.add_self_loops,
                                               dx_handoff=agg_handoff)
            else:
                agg = tr.propagate(edge_index, x=x[0], edge_attr=edge_attr, size=size, add_self=tr.add_self_loops)
            site = next((i for i, nm in enumerate(self.norms) if nm is norm), 0)
            fused = _engine.mlp_norm_fused(agg, x[1], n_valid, tr.transform, norm, self.activation_str, skip_handoff=handoff,
                                           defer_apply=defer_norm and not dropout_active,
                                           dropout=(dropout_mpnn, self.same_dropout, site) if dropout_active else None)
            if fused is not None:
                return fused
            h = tr.transform(agg, n_valid=n_valid)
        elif edge_encoder is not None:
            # (PHMConv: the transform's GEMM emits the statistics of the norm that follows, unless hooks want the reference's sequence)
            h = conv.forward_encoded(x[0], edge_index, edge_attr, edge_encoder, size, n_valid=n_valid,
                                     bn_stats=norm is not None and self.training and not getattr(self, "_observable_now", False))
        elif n_valid is not None and self.mlp_mp:
            h = conv.transform(x[0], edge_index, edge_attr, size, n_valid=n_valid)
        else:
            h = conv(x=x[0], edge_index=edge_index, edge_attr=edge_attr, size=size)
        if norm is not None and (not dropout_active or h.is_cuda):
            # norm -> activation -> (dropout ->) (+ skip) ride in one engine pass; the dropout mask is regenerated in the
            # backward from (seed, step, site) — site = the layer's position in self.norms
            site = next((i for i, nm in enumerate(self.norms) if nm is norm), 0)
            return norm.forward_fused(h, activation=self.activation_str, residual=x[1], n_valid=n_valid,
                                      dropout=(dropout_mpnn, self.same_dropout, site) if dropout_active else None)
        if norm is not None:
            h = norm.forward_fused(h, activation=self.activation_str, n_valid=n_valid)
        else:
            h = self.f_act(h)
        h = phm_dropout(x=h, p=dropout_mpnn, phm_dim=self.phm_dim, training=self.training, same=self.same_dropout)
        return h + x[1]

    def forward(self, data, size=None) -> torch.Tensor:
        # Called outside the step's contexts (the reference's own loop body: model(data); loss.backward()): a training-mode forward on
        # a ROCm device opens them itself and closes them behind its backward (engine: auto_step_forward) — one operand-build launch,
        # statistics through accumulators, the batched step tail.
        if _engine.AUTO_STEP and self.training and torch.is_grad_enabled() and getattr(data.x, "is_cuda", False) and \
                (not _engine.step_context_active() or _engine.auto_step_open()):      # (an auto-opened block whose backward never came is replaced)
            return _engine.auto_step_forward(self, data, lambda: self._forward(data, size))
        return self._forward(data, size)

    def _forward(self, data, size=None) -> torch.Tensor:
        x, edge_index, edge_attr, batch = data.x, data.edge_index, data.edge_attr, data.batch
        if isinstance(self.bond_input_dims, list):
            edge_attr = edge_attr.to(torch.long)
        # fixed-shape (padded) batches carry device-side valid counts (engine extension, see phc_gnn_amd/graphed.py)
        n_valid = getattr(data, "n_valid_nodes", None)
        n_valid_graphs = getattr(data, "n_valid_graphs", None)
        atom = self.atomencoder(x)
        atom = atom.reshape(atom.size(0), self.phm_dim * self.atom_encoded_dim)     # component-major [N, h]
        L_ = len(self.mp_layers)
        # unfinished tensors never leave this method, and nobody inside can look at them unless hooked in: with hooks anywhere
        # below, every layer takes the reference's module-by-module sequence (class docstring)
        self._observable_now = observable = _any_hooks(self.convs, self.norms, self.pooling)
        # the encoded atoms feed the first conv and (sc_type "first") every layer's skip: one fan-out node, so that
        # their gradients are summed in one pass
        # ("last": the conv input IS the skip tensor — one tensor object, see compute_hidden_layer_embedding's hand-off)
        uses = _engine.fanout(atom, 1 + L_) if (atom.is_cuda and self.sc_type == "first") else None
        # (their gradients meet in the first layer's aggregation backward when that layer takes the hand-off; the fan-out node then
        # sees one gradient and launches nothing)
        skip_sum = _engine.DxSumHandoff() if (uses is not None and atom.requires_grad and torch.is_grad_enabled()
                                              and L_ <= _engine.DxSumHandoff.MAX and _engine.SKIP_SUM_HANDOFF) else None
        h = uses[0] if uses is not None else atom
        # (a sorted batch's soft-attention read-out applies the last layer's norm itself: engine PendingGate / phc_readout_fused_fwd)
        readout_takes_norm = isinstance(self.pooling, PHMSoftAttentionPooling) and bool(getattr(data, "batch_is_sorted", False)) \
            and not observable
        for i in range(L_):
            if uses is not None:
                skip = uses[1 + i]
            else:
                skip = atom if (i == 0 or self.sc_type == "first") else h
            enc = self.bondencoders[i]
            if getattr(enc, "engine_tables", None) is not None:
                # integer bond features: embedding fused into the aggregation (no [E, h] tensor)
                h = self.compute_hidden_layer_embedding(conv=self.convs[i], norm=self.norms[i], x=[h, skip],
                                                        edge_index=edge_index, edge_attr=edge_attr,
                                                        dropout_mpnn=self.dropout_mpnn[i], size=size, edge_encoder=enc,
                                                        n_valid=n_valid, skip_sum=skip_sum, first_layer=(i == 0),
                                                        # the layer's last element-wise pass rides in the NEXT layer's aggregation
                                                        # ... the last layer's in the soft-attention read-out's one launch
                                                        defer_norm=not observable and (
                                                            getattr(self.bondencoders[i + 1], "engine_tables", None) is not None
                                                            if i + 1 < L_ else readout_takes_norm))
                continue
            e = enc(edge_attr)
            e = e.reshape(e.size(0), self.mp_layers[i])
            h = self.compute_hidden_layer_embedding(conv=self.convs[i], norm=self.norms[i], x=[h, skip],
                                                    edge_index=edge_index, edge_attr=e,
                                                    dropout_mpnn=self.dropout_mpnn[i], size=size, n_valid=n_valid)
        if not readout_takes_norm:
            _engine.materialize_pending(h)                   # (a guard for subclasses: only that read-out knows about a parked norm)
        num_graphs = getattr(data, "num_graphs", None)
        # The reference's scatter-based global_add_pool does not care about the order of `batch`; the binary-search
        # read-out does.  Only a batch object that SAYS its `batch` vector is non-decreasing (harness.to_device and the
        # fixed-shape batches of graphed.py do; PyG collation produces such vectors) takes that path, anything else is
        # grouped by a stable sort first.
        out = self.pooling(x=h, batch=batch, num_graphs=num_graphs, is_sorted=bool(getattr(data, "batch_is_sorted", False)))
        if n_valid_graphs is not None:
            return self.downstream(out, n_valid=n_valid_graphs)
        return self.downstream(out)

    def __repr__(self):
        return "{}(phm_dim={}, learn_phm={}, phm_rule={}, atom_input_dim={}, atom_encoded_dim={}, " \
               "bond_input_dims={}, naive_encoder={}, w_init='{}', c_init='{}', same_dropout={}, mp_layers={}, " \
               "bias={}, dropout_mpnn={}, norm_mp='{}', add_self_loops={}, msg_aggr='{}', node_aggr={}, mlp={}, " \
               "pooling='{}', activation='{}', real_trafo='{}', downstream_layers={}, target_dim={}, " \
               "dropout_dn={}, norm_dn={})".format(
                   self.__class__.__name__, self.phm_dim, self.learn_phm, self.phm_rule, self.atom_input_dims,
                   self.atom_encoded_dim, self.bond_input_dims, self.naive_encoder, self.w_init, self.c_init,
                   self.same_dropout, self.mp_layers, self.bias, self.dropout_mpnn, self.norm_mp,
                   self.add_self_loops, self.msg_aggr_type, self.node_aggr_type, self.mlp_mp, self.pooling_type,
                   self.activation_str, self.real_trafo_type, self.downstream_layers, self.target_dim,
                   self.dropout_dn, self.norm_dn_type)


class PHMSkipConnectConcat(nn.Module):
    """Import-compatibility stub (benchmarks/train_hiv.py:32-33 imports the name):
    the concatenating variant uses the same kernels at different widths and is
    not part of BASELINE.json's configs."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("PHMSkipConnectConcat is not provided by phc_gnn_amd yet; "
                                  "use PHMSkipConnectAdd (--type undirectional-phm-sc-add)")
