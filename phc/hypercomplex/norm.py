"""Hypercomplex batch normalisation (reference: phc/hypercomplex/norm.py).

"naive" PHC batch-norm keeps one ``BatchNorm1d(k)`` per hypercomplex component
(parameter names ``bn.bn.<c>.{weight,bias,running_mean,running_var,
num_batches_tracked}`` — part of the checkpoint contract).  Because batch-norm
is per column, the n component-wise norms over [N, k] slices are exactly one
batch-norm over the whole component-major [N, n*k] matrix; the forward below
runs that single pass instead of reshape -> permute -> n norms -> cat."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from phc_gnn_amd import ops as _engine

_FUSABLE_ACTS = ("identity", "relu", "lrelu", "elu", "swish", "selu")


class NaivePHMNorm(nn.Module):
    def __init__(self, num_features: int, phm_dim: int, momentum: float = 0.1, eps: float = 1e-5,
                 affine: bool = True, track_running_stats: bool = True) -> None:
        super().__init__()
        self.phm_dim = phm_dim
        self.num_features = num_features // phm_dim          # per component, as in the reference (:15)
        self.momentum = momentum
        self.affine = affine
        self.eps = eps
        self.track_running_stats = track_running_stats
        self.bn = nn.ModuleList([nn.BatchNorm1d(self.num_features, eps, momentum, affine, track_running_stats)
                                 for _ in range(phm_dim)])
        self.reset_parameters()
        self._lay_out_back_to_back()

    def reset_parameters(self):
        for module in self.bn:
            module.reset_parameters()

    def _lay_out_back_to_back(self):
        """Engine extension: the n components' weights (biases, running means, running variances) become views of ONE storage each,
        component c at offset c * k.  Names, shapes, values and ``state_dict`` are the reference's; what changes is that the engine's
        kernels find column ``c * k + j`` of a parameter at ``weight_0 + c * k + j`` — one load in their prologue instead of a pointer
        hop per component table (``BnParams.flat``, csrc/phc_common.h).  Re-done whenever the module's tensors are moved
        (``_apply``: .to / .cuda / .float); any other re-pointing of a parameter (a deep copy, ``load_state_dict(assign=True)``)
        simply leaves the kernels on their pointer tables."""
        k = self.num_features
        for name in ("weight", "bias", "running_mean", "running_var"):
            ts = [getattr(m, name, None) for m in self.bn]
            if any(t is None for t in ts) or len(ts) < 2:
                continue
            t0 = ts[0]
            if all(t.device == t0.device and t.dtype == t0.dtype and t.is_contiguous() and
                   t.data_ptr() == t0.data_ptr() + c * k * t0.element_size() for c, t in enumerate(ts)):
                continue
            if any(t.device != t0.device or t.dtype != t0.dtype for t in ts):
                continue
            flat = torch.cat([t.detach().reshape(-1) for t in ts])
            for c, t in enumerate(ts):
                t.data = flat[c * k:(c + 1) * k]

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._lay_out_back_to_back()
        return out

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.forward_fused(x)

    def _momentum(self) -> float:
        if self.momentum is None:
            # BatchNorm1d(momentum=None) = cumulative moving average with factor 1 / num_batches_tracked (a device
            # counter): the engine's kernels take the factor as a launch argument, so this mode is refused rather than
            # silently run with 0.1.  The reference never uses it (norm.py:10: momentum = 0.1).
            raise NotImplementedError("NaivePHMNorm(momentum=None) (cumulative average) is not supported by phc_gnn_amd")
        return float(self.momentum)

    def forward_fused(self, x: torch.Tensor, activation: str = "identity", residual=None, n_valid=None, dropout=None) -> torch.Tensor:
        """``dropout`` = (p, same, site): phm_dropout behind the activation inside the same pass (engine extension).
        act(norm(x)) (+ residual) in one engine pass (``phc_bn_fwd``): the activation and the
        skip add that follow the norm in the model (models.py:208-215) and in PHMMLP
        (layers.py:351-353) ride in the normalisation kernel.  x: [N, phm_dim * k]."""
        mods = list(self.bn)
        track = self.track_running_stats
        use_batch_stats = self.training or not track
        if not track:
            # no running statistics: batch statistics in both modes, nothing to update
            return _engine.batch_norm_act(x, [m.weight for m in mods] if self.affine else None,
                                          [m.bias for m in mods] if self.affine else None, None, None, None, True,
                                          0.0, self.eps, activation, residual, n_valid) if self.affine else \
                self._plain(x, activation, residual)
        momentum = self._momentum()
        return _engine.batch_norm_act(
            x, [m.weight for m in mods] if self.affine else None, [m.bias for m in mods] if self.affine else None,
            [m.running_mean for m in mods], [m.running_var for m in mods],
            [m.num_batches_tracked for m in mods] if self.training else None, use_batch_stats, momentum, self.eps,
            activation, residual, n_valid, dropout if self.training else None)

    def forward_fused_linear(self, x: torch.Tensor, activation: str, n_valid, linear, out_stats=None):
        """``linear(act(norm(x)))`` (training mode) without writing the normalised activation (engine:
        ``bn_act_linear``); ``None`` when the engine cannot fuse here and the caller must run the two steps."""
        if not (self.training and self.track_running_stats and x.is_cuda):
            return None
        mods = list(self.bn)
        momentum = self._momentum()
        return _engine.bn_act_linear(
            x, [m.weight for m in mods] if self.affine else None, [m.bias for m in mods] if self.affine else None,
            [m.running_mean for m in mods], [m.running_var for m in mods], [m.num_batches_tracked for m in mods], momentum, self.eps,
            activation, n_valid, linear.phm_rule, linear.W, linear.b, out_stats)

    def _plain(self, x, activation, residual):
        _engine.warn_slow_path("norm-plain", "NaivePHMNorm without running statistics")
        out = F.batch_norm(x, None, None, None, None, True, 0.0, self.eps)
        out = _engine_act(out, activation)
        return out if residual is None else out + residual

    def __repr__(self):
        return "{num_features}, phm_dim={phm_dim}, eps={eps}, momentum={momentum}, affine={affine}, " \
               "track_running_stats={track_running_stats})".format(**self.__dict__)


def _engine_act(x, activation):
    from phc.quaternion.activations import get_module_activation
    return get_module_activation(activation)(x)


class PHMNorm(nn.Module):
    """Wrapper selecting the norm flavour (reference :45-74)."""

    def __init__(self, num_features: int, phm_dim: int, type: str = "naive-batch-norm", **kwargs):
        super().__init__()
        assert type in ["naive-batch-norm", "naive-naive-batch-norm"]
        self.type = type
        self.num_features = num_features
        self.phm_dim = phm_dim
        self.kwargs = kwargs
        if type == "naive-batch-norm":
            self.bn = NaivePHMNorm(num_features=num_features, phm_dim=phm_dim, **kwargs)
        else:
            # the reference sizes this variant as num_features * phm_dim (:61)
            self.bn = nn.BatchNorm1d(num_features=num_features * phm_dim, **kwargs)
        self.reset_parameters()

    def reset_parameters(self):
        self.bn.reset_parameters()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.bn(x)

    def forward_fused(self, x: torch.Tensor, activation: str = "identity", residual=None, n_valid=None, dropout=None) -> torch.Tensor:
        """act(norm(x)) (+ dropout) (+ residual); one engine pass for the naive-batch-norm flavour."""
        if isinstance(self.bn, NaivePHMNorm) and activation.lower() in _FUSABLE_ACTS and self.bn.track_running_stats and \
                (dropout is None or (x.shape[-1] % 4 == 0 and (not dropout[1] or (x.shape[-1] // self.phm_dim) % 4 == 0))):
            return self.bn.forward_fused(x, activation, residual, n_valid, dropout)
        _engine.warn_slow_path("norm-act-dropout", f"norm + activation '{activation}'" + (" + dropout" if dropout is not None else "")
                               + f" at width {x.shape[-1]} / phm_dim {self.phm_dim}")
        out = _engine_act(self.bn(x), activation)
        if dropout is not None and self.training and dropout[0] > 0.0:
            from phc.hypercomplex.layers import phm_dropout
            out = phm_dropout(out, phm_dim=self.phm_dim, p=dropout[0], training=True, same=bool(dropout[1]))
        return out if residual is None else out + residual

    def forward_fused_linear(self, x: torch.Tensor, activation: str, n_valid, linear, out_stats=None):
        if isinstance(self.bn, NaivePHMNorm) and activation.lower() in _FUSABLE_ACTS:
            return self.bn.forward_fused_linear(x, activation, n_valid, linear, out_stats)
        return None

    def __repr__(self):
        return f"{self.__class__.__name__}:(num_features={self.num_features}, phm_dim={self.phm_dim} type={self.type})"
