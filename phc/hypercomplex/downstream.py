"""Downstream prediction head (reference: phc/hypercomplex/downstream.py):
PHMLinear -> norm -> activation -> dropout stack on the pooled [B, h] graph
embeddings, a final PHMLinear to phm_dim*out_features and a RealTransformer."""
from typing import Union

import torch
import torch.nn as nn

from phc.hypercomplex.layers import PHMLinear, RealTransformer, phm_dropout
from phc.hypercomplex.norm import PHMNorm
from phc_gnn_amd import ops as _engine
from phc.quaternion.activations import get_module_activation


def _any_local_hooks(*roots) -> bool:
    """A forward (pre-)hook registered on one of these modules or their sub-modules (global hooks do not count: they would take
    every model off its fused head)."""
    for root in roots:
        for sub in root.modules():
            if sub._forward_hooks or sub._forward_pre_hooks:
                return True
    return False


class PHMDownstreamNet(nn.Module):
    def __init__(self, in_features: int, phm_dim: int, phm_rule, hidden_layers: list, out_features: int,
                 activation: str, bias: bool, norm: str, w_init: str, c_init: str, dropout: Union[float, list],
                 learn_phm: bool = True, same_dropout: bool = False, real_trafo: str = "linear") -> None:
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.learn_phm = learn_phm
        self.phm_rule = phm_rule
        self.phm_dim = phm_dim
        self.hidden_layers = hidden_layers
        self.activation_str = activation
        self.activation_func = get_module_activation(activation)
        self.w_init = w_init
        self.c_init = c_init
        self.bias = bias
        self.dropout = [dropout] * len(hidden_layers) if isinstance(dropout, float) else dropout
        assert len(self.dropout) == len(self.hidden_layers), "dropout list must be of the same size " \
                                                             "as number of hidden layer"
        self.norm_type = norm
        self.same_dropout = same_dropout

        widths = [in_features] + list(hidden_layers) + [phm_dim * out_features]
        self.affine = nn.ModuleList([
            PHMLinear(in_features=widths[i], out_features=widths[i + 1], phm_dim=phm_dim, phm_rule=phm_rule,
                      learn_phm=learn_phm, bias=bias, w_init=w_init, c_init=c_init)
            for i in range(len(widths) - 1)])
        self.real_trafo_type = real_trafo
        self.real_trafo = RealTransformer(type=real_trafo, in_features=phm_dim * out_features, phm_dim=phm_dim, bias=True)
        self.norm_flag = bool(self.norm_type)
        if self.norm_flag:
            self.norm = nn.ModuleList([PHMNorm(num_features=d, phm_dim=phm_dim, type=self.norm_type)
                                       for d in hidden_layers])
        self.reset_parameters()

    def reset_parameters(self):
        for module in self.affine:
            module.reset_parameters()
        if self.norm_flag:
            for module in self.norm:
                module.reset_parameters()
        self.real_trafo.reset_parameters()

    def forward(self, x: torch.Tensor, verbose=False, n_valid=None, **kwargs) -> torch.Tensor:
        last = len(self.affine) - 1
        engine = self.norm_flag and self.training and x.is_cuda and x.dim() == 2
        # (forward hooks on the head's layers: the one-launch head never calls them — the per-layer sequence below does)
        if engine and last == 2 and self.real_trafo_type == "linear" and all(hasattr(nm.bn, "bn") for nm in self.norm) and not verbose \
                and not _any_local_hooks(self.affine, self.real_trafo):
            # the whole head (three PHM layers, two norms, the dense map to the targets) in ONE engine launch each way; with
            # dropout (:104-107) the masks of the stand-alone fused passes (sites 64 + i) are drawn inside it
            y = _engine.downstream_head(x, n_valid, list(self.affine), [nm.bn for nm in self.norm], self.real_trafo.affine,
                                        self.activation_str,
                                        dropout=((self.dropout[0], self.dropout[1]), self.same_dropout, (64, 65))
                                        if any(p > 0.0 for p in self.dropout[:2]) else None)
            if y is not None:
                return y
        i = 0
        x = self.affine[0](x, bn_stats=_engine.BNStatsRequest(n_valid) if (engine and last > 0) else None)
        while i < last:
            if verbose:
                print(f"layer {i}: {self.affine[i]} -> {tuple(x.shape)}")
            nxt = self.affine[i + 1]
            nxt_stats = _engine.BNStatsRequest(n_valid) if (engine and i + 1 != last) else None
            no_dropout = not (self.training and self.dropout[i] > 0.0)
            if engine and no_dropout and not _any_local_hooks(nxt):
                # norm -> activation -> next layer inside the next GEMM's operand staging (no normalised tensor; never with a forward
                # hook on that layer: the fused form does not call it)
                y = self.norm[i].forward_fused_linear(x, self.activation_str, n_valid, nxt, nxt_stats)
                if y is not None:
                    x = y
                    i += 1
                    continue
            if self.norm_flag and not no_dropout and x.is_cuda:
                # norm -> activation -> dropout in one engine pass (mask regenerated in the backward; sites 64 + i)
                x = self.norm[i].forward_fused(x, activation=self.activation_str, n_valid=n_valid,
                                               dropout=(self.dropout[i], self.same_dropout, 64 + i))
            else:
                if self.norm_flag:
                    x = self.norm[i].forward_fused(x, activation=self.activation_str, n_valid=n_valid)
                else:
                    x = self.activation_func(x)
                if not no_dropout:
                    x = phm_dropout(x, p=self.dropout[i], phm_dim=self.phm_dim, training=self.training,
                                    same=self.same_dropout)
            x = nxt(x, bn_stats=nxt_stats)
            i += 1
        if verbose:
            print(f"layer {last}: {self.affine[last]} -> {tuple(x.shape)}")
        return self.real_trafo(x)

    def __repr__(self):
        return "{}(in_features={}, phm_dim={}, phm_rule={}, hidden_layers={}, out_features={}, activation='{}', " \
               "bias={}, norm='{}', w_init='{}', c_init='{}', dropout={}, learn_phm={}, same_dropout={}, " \
               "real_trafo='{}')".format(self.__class__.__name__, self.in_features, self.phm_dim, self.phm_rule,
                                         self.hidden_layers, self.out_features, self.activation_str, self.bias,
                                         self.norm_type, self.w_init, self.c_init, self.dropout, self.learn_phm,
                                         self.same_dropout, self.real_trafo_type)
