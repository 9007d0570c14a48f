"""PHM layers with the reference's module API (reference:
phc/hypercomplex/layers.py), computing on the MI355X engine.

``PHMLinear.forward`` is one call into libphc_hip.so (``phc_phm_linear_fwd``:
contribution tensor in LDS, H tiles assembled in LDS, fp32 MFMA, fused bias);
its backward is ``phc_phm_linear_bwd``.  Parameter names and shapes are the
reference's: ``W [n, in/n, out/n]``, ``phm_rule [n, n, n]``, ``b [out]``."""
from typing import Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

from phc.hypercomplex.inits import phm_init
from phc.hypercomplex.norm import PHMNorm
from phc.hypercomplex.utils import get_multiplication_matrices
from phc.quaternion.activations import get_module_activation
from phc.quaternion.inits import glorot_normal, glorot_uniform
from phc_gnn_amd import ops as _engine


def get_bernoulli_mask(x: torch.Tensor, p: float) -> torch.Tensor:
    return torch.bernoulli(torch.full(x.size(), 1 - p, device=x.device))


def torch_dropout(x: torch.Tensor, p: float, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    if mask is None:
        mask = get_bernoulli_mask(x, p)
    return (mask * x) * (1 / (1 - p))


def phm_dropout(x: torch.Tensor, phm_dim: int, p: float = 0.2, training: bool = True, same: bool = False) -> torch.Tensor:
    """Dropout on a component-major [N, n*k] tensor (ref :31-55).  ``same=True``
    draws ONE [N, k] mask and applies it to every component; kept entries are
    scaled by 1/(1-p)."""
    assert 0.0 <= p <= 1.0, f"dropout rate must be in [0.0 ; 1.0]. {p} was inserted!"
    if not (training and p > 0.0):
        return x
    _engine.warn_slow_path("phm-dropout", "phm_dropout as a stand-alone op")
    if not same:
        return F.dropout(input=x, p=p, training=training)
    k = x.size(-1) // phm_dim
    xc = x.reshape(x.size(0), phm_dim, k)
    mask = get_bernoulli_mask(xc[:, 0], p=p).unsqueeze(1)
    return torch_dropout(xc, p=p, mask=mask).reshape(x.size(0), -1)


def matvec_product_new(W: torch.Tensor, x: torch.Tensor, bias: Optional[torch.Tensor],
                       phm_rule: torch.Tensor) -> torch.Tensor:
    """y = x . (sum_i phm_rule[i] (x) W[i]) + bias   (ref :198-219), on the HIP engine."""
    return _engine.phm_linear(x, phm_rule, W, bias)


class PHMLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, phm_dim: int,
                 phm_rule: Union[None, torch.Tensor] = None, bias: bool = True, w_init: str = "phm",
                 c_init: str = "random", learn_phm: bool = True) -> None:
        super().__init__()
        assert w_init in ["phm", "glorot-normal", "glorot-uniform"]
        assert c_init in ["standard", "random"]
        assert in_features % phm_dim == 0, f"Argument `in_features`={in_features} is not divisble be `phm_dim`{phm_dim}"
        assert out_features % phm_dim == 0, f"Argument `out_features`={out_features} is not divisble be `phm_dim`{phm_dim}"
        self.in_features = in_features
        self.out_features = out_features
        self.learn_phm = learn_phm
        self.phm_dim = phm_dim
        self._in_feats_per_axis = in_features // phm_dim
        self._out_feats_per_axis = out_features // phm_dim
        self.bias_flag = bias
        self.w_init = w_init
        self.c_init = c_init

        rule = phm_rule if phm_rule is not None else get_multiplication_matrices(phm_dim, type=c_init)
        self.phm_rule = nn.Parameter(torch.stack([*rule], dim=0).detach().clone(), requires_grad=learn_phm)
        self.W = nn.Parameter(torch.empty(phm_dim, self._in_feats_per_axis, self._out_feats_per_axis))
        if bias:
            self.b = nn.Parameter(torch.empty(out_features))
        else:
            self.register_parameter("b", None)
        self.reset_parameters()

    def reset_parameters(self):
        n, kin, kout = self.phm_dim, self._in_feats_per_axis, self._out_feats_per_axis
        with torch.no_grad():
            if self.w_init == "phm":
                self.W.copy_(phm_init(phm_dim=n, in_features=kin, out_features=kout, transpose=False))
            elif self.w_init == "glorot-normal":
                for i in range(n):
                    glorot_normal(self.W[i])
            elif self.w_init == "glorot-uniform":
                for i in range(n):
                    glorot_uniform(self.W[i])
            else:
                raise ValueError(self.w_init)
            if self.bias_flag:
                # reference (:276-278): b[:k] = 0, b[k+1:] = 0.2 and b[k] left UNINITIALISED;
                # here the first component is 0 and every other entry 0.2 (documented divergence)
                self.b.fill_(0.2)
                self.b[:kout] = 0.0
            # as in the reference (:281) the contribution matrices are re-drawn from c_init,
            # which also overwrites a rule handed to the constructor
            self.phm_rule.copy_(torch.stack(get_multiplication_matrices(n, type=self.c_init), dim=0))

    def forward(self, x: torch.Tensor, phm_rule=None, bn_stats=None, dx_handoff=None) -> torch.Tensor:
        # the second argument is accepted and ignored, as in the reference (:284-286); ``bn_stats`` (engine
        # extension): an ops.BNStatsRequest when a batch-norm consumes this layer's output directly
        return _engine.phm_linear(x, self.phm_rule, self.W, self.b, bn_stats=bn_stats, dx_handoff=dx_handoff)

    def __repr__(self):
        return "{}(in_features={}, out_features={}, phm_dim={}, bias={}, w_init={}, c_init={}, learn_phm={})".format(
            self.__class__.__name__, self.in_features, self.out_features, self.phm_dim, self.bias_flag,
            self.w_init, self.c_init, self.learn_phm)


class PHMMLP(nn.Module):
    """linear1 -> (PHMNorm) -> activation -> linear2   (ref :304-369)."""

    def __init__(self, in_features: int, out_features: int, phm_dim: int, phm_rule, bias: bool = True,
                 learn_phm: bool = True, activation: str = "relu", norm: Union[None, str] = None,
                 w_init: str = "phm", c_init: str = "standard", factor: float = 1, **kwargs) -> None:
        super().__init__()
        assert norm in ["None", None, "naive-batch-norm", "naive-naive-batch-norm"]
        self.in_features = in_features
        self.out_features = out_features
        self.phm_dim = phm_dim
        self.bias_flag = bias
        self.learn_phm = learn_phm
        self.phm_rule = phm_rule
        self.activation_str = activation
        hidden = int(factor * out_features)
        self.linear1 = PHMLinear(in_features=in_features, out_features=hidden, phm_dim=phm_dim, phm_rule=phm_rule,
                                 learn_phm=learn_phm, bias=bias, w_init=w_init, c_init=c_init)
        self.linear2 = PHMLinear(in_features=hidden, out_features=out_features, phm_dim=phm_dim, phm_rule=phm_rule,
                                 learn_phm=learn_phm, bias=bias, w_init=w_init, c_init=c_init)
        self.activation = get_module_activation(activation)
        self.norm_type = norm
        self.factor = factor
        self.w_init = w_init
        self.c_init = c_init
        self.norm_flag = norm in ["naive-batch-norm", "naive-naive-batch-norm"]
        if self.norm_flag:
            self.norm = PHMNorm(num_features=hidden, phm_dim=phm_dim, type=norm, **kwargs)
        self.reset_parameters()

    def reset_parameters(self):
        self.linear1.reset_parameters()
        self.linear2.reset_parameters()
        if self.norm_flag:
            self.norm.reset_parameters()

    def forward(self, x: torch.Tensor, n_valid=None) -> torch.Tensor:
        """``n_valid`` (engine extension): device int32 scalar; rows past it are padding of a
        fixed-shape batch and stay out of the batch-norm statistics."""
        want = self.training and x.is_cuda and x.dim() == 2
        x = self.linear1(x, bn_stats=_engine.BNStatsRequest(n_valid, "apply") if (want and self.norm_flag) else None)
        out_req = _engine.BNStatsRequest(n_valid, "apply") if (want and getattr(self, "out_norm_follows", False)) else None
        if self.norm_flag and want:
            # norm -> activation -> linear2 in the second GEMM's operand staging: the normalised activation never exists
            y = self.norm.forward_fused_linear(x, self.activation_str, n_valid, self.linear2, out_req)
            if y is not None:
                return y
        if self.norm_flag:
            x = self.norm.forward_fused(x, activation=self.activation_str, n_valid=n_valid)   # norm + activation in one pass
        else:
            x = self.activation(x)
        # ``out_norm_follows`` is set by the owning model when it normalises this MLP's output directly
        return self.linear2(x, bn_stats=_engine.BNStatsRequest(n_valid, "apply") if (want and getattr(self, "out_norm_follows", False))
                            else None)

    def __repr__(self):
        return '{}(in_features={}, out_features={}, phm_dim={}, phm_rule={}, bias={}, learn_phm={}, ' \
               'activation="{}", norm="{}", w_init="{}", c_init={}, factor={})'.format(
                   self.__class__.__name__, self.in_features, self.out_features, self.phm_dim, self.phm_rule,
                   self.bias_flag, self.learn_phm, self.activation_str, self.norm_type, self.w_init, self.c_init,
                   self.factor)


class RealTransformer(nn.Module):
    """Hypercomplex -> real head (ref :372-420): a dense ``Linear(in, in/n)`` or a
    reduction over equal feature chunks."""

    def __init__(self, type: str, in_features: int, phm_dim: int, bias: bool = True) -> None:
        super().__init__()
        assert type in ["linear", "sum", "mean", "norm"]
        self.type = type
        self.in_features = in_features
        self.phm_dim = phm_dim
        self.bias_flag = bias
        self.affine = nn.Linear(in_features, in_features // phm_dim, bias=bias) if type == "linear" else None
        self.reset_parameters()

    def reset_parameters(self):
        if self.type == "linear":
            nn.init.xavier_uniform_(self.affine.weight)
            if self.bias_flag:
                self.affine.bias.data.fill_(0.0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.type == "linear":
            if x.is_cuda and self.in_features >= 32 and x.dim() == 2:
                return _engine.dense_linear(x, self.affine.weight, self.affine.bias)      # phm_dim = 1 PHM kernel
            if x.is_cuda and x.dim() == 2 and self.affine.weight.numel() <= 64:
                return _engine.small_linear(x, self.affine.weight, self.affine.bias)      # the n*target -> target head
            return self.affine(x)
        # the reference splits the last dim into chunks of `in_features` columns (:405)
        parts = torch.stack(x.split(self.in_features, dim=-1), dim=0)
        if self.type == "sum":
            return parts.sum(dim=0)
        if self.type == "mean":
            return parts.mean(dim=0)
        return parts.norm(dim=0)

    def __repr__(self):
        return '{}(type="{}", in_features={}, phm_dim={}, bias={})'.format(
            self.__class__.__name__, self.type, self.in_features, self.phm_dim, self.bias_flag)
