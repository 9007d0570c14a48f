"""Weight initialisers for PHM layers (reference: phc/hypercomplex/inits.py).
Host-side numpy/scipy sampling; parity with the reference is statistical (its
numpy/scipy RNG streams are not reproducible across versions)."""
import numpy as np
import torch
from scipy.stats import chi


def unitary_init(phm_dim: int, in_features: int, out_features: int, low: int = 0, high: int = 1) -> torch.Tensor:
    """[phm_dim, in, out]: zero "real" part, uniform "imaginary" parts, unit 2-norm over dim 0   (ref :6-13)."""
    v = torch.zeros(phm_dim, in_features, out_features)
    if phm_dim > 1:
        v[1:].uniform_(low, high)
    return v / v.norm(p=2, dim=0)


def phm_init(phm_dim: int, in_features: int, out_features: int, low: int = 0, high: int = 1,
             criterion: str = "glorot", transpose: bool = True) -> torch.Tensor:
    """Polar-form init   (ref :16-44): magnitude ~ chi(phm_dim) scaled by the
    glorot/he factor, phase ~ U(-pi, pi), direction = unit purely-imaginary vector.
    Component 0 gets |w| cos(theta), component i>0 gets |w| v_i sin(theta)."""
    if criterion == "glorot":
        scale = np.sqrt(2.0 / (phm_dim * (in_features + out_features)))
    elif criterion == "he":
        scale = np.sqrt(2.0 / (phm_dim * in_features))
    else:
        raise ValueError("Invalid criterion: " + criterion)
    shape = (in_features, out_features)
    mag = torch.from_numpy(chi.rvs(phm_dim, loc=0, scale=scale, size=shape)).float()
    direction = unitary_init(phm_dim, in_features, out_features, low, high)
    theta = torch.from_numpy(np.random.uniform(-np.pi, np.pi, size=shape)).float()
    weight = torch.empty(phm_dim, in_features, out_features)
    weight[0] = mag * torch.cos(theta)
    if phm_dim > 1:
        weight[1:] = mag * torch.sin(theta) * direction[1:]
    return weight.permute(0, 2, 1) if transpose else weight
