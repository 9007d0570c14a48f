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

    def reset_parameters(self):
        for module in self.bn:
            module.reset_parameters()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        # x: [N, phm_dim * k], component-major
        weight = bias = None
        if self.affine:
            weight = torch.cat([m.weight for m in self.bn])
            bias = torch.cat([m.bias for m in self.bn])
        use_batch_stats = self.training or not self.track_running_stats
        if not self.track_running_stats:
            return F.batch_norm(x, None, None, weight, bias, True, self.momentum, self.eps)
        mean = torch.cat([m.running_mean for m in self.bn])
        var = torch.cat([m.running_var for m in self.bn])
        out = F.batch_norm(x, mean, var, weight, bias, use_batch_stats, self.momentum, self.eps)
        if self.training:
            k = self.num_features
            with torch.no_grad():
                torch._foreach_copy_([m.running_mean for m in self.bn], list(mean.split(k)))
                torch._foreach_copy_([m.running_var for m in self.bn], list(var.split(k)))
                torch._foreach_add_([m.num_batches_tracked for m in self.bn], 1)
        return out

    def __repr__(self):
        return "{num_features}, phm_dim={phm_dim}, eps={eps}, momentum={momentum}, affine={affine}, " \
               "track_running_stats={track_running_stats})".format(**self.__dict__)


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

    def __repr__(self):
        return f"{self.__class__.__name__}:(num_features={self.num_features}, phm_dim={self.phm_dim} type={self.type})"
