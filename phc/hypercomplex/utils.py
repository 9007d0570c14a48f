"""Contribution-matrix ("phm_rule") initialisers and small tensor helpers with
the reference's names (reference: phc/hypercomplex/utils.py)."""
import warnings
from typing import List

import numpy as np
import torch

# Hamilton product in matrix form: row a, column c of A_i is the coefficient with
# which input component a of W_i contributes to output component c (ref :5-22).
_QUAT = [
    [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
    [[0, -1, 0, 0], [1, 0, 0, 0], [0, 0, 0, -1], [0, 0, 1, 0]],
    [[0, 0, -1, 0], [0, 0, 0, 1], [1, 0, 0, 0], [0, -1, 0, 0]],
    [[0, 0, 0, -1], [0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0]],
]
QUATERNION_MULTIPLICATION_RULE = {f"A{i + 1}": torch.tensor(m, dtype=torch.float32) for i, m in enumerate(_QUAT)}


def get_quaternion_mul_rule(device: str = "cpu") -> list:
    return [m.to(device) for m in QUATERNION_MULTIPLICATION_RULE.values()]


def get_complex_mul_rule() -> list:
    return [torch.tensor([[1., 0.], [0., 1.]]), torch.tensor([[0., -1.], [1., 0.]])]


def get_right_shift_permutation_matrix(phm_dim: int) -> torch.Tensor:
    return torch.from_numpy(np.roll(np.eye(phm_dim, dtype=np.float32), shift=1, axis=1))


def get_left_shift_permutation_matrix(phm_dim: int) -> torch.Tensor:
    return torch.from_numpy(np.roll(np.eye(phm_dim, dtype=np.float32), shift=-1, axis=1))


def right_cyclic_permutation(arr: list, power: int = 1) -> list:
    if not arr:
        return arr
    s = power % len(arr)
    return arr[len(arr) - s:] + arr[:len(arr) - s]


def left_cyclic_permutation(arr: list, power: int = 1) -> list:
    if not arr:
        return arr
    s = power % len(arr)
    return arr[s:] + arr[:s]


def get_multiplication_matrices(phm_dim: int, type="standard") -> List[torch.Tensor]:
    """List of phm_dim matrices [phm_dim, phm_dim]   (ref :61-85).
    "standard": complex rule for 2, quaternion rule for 4, otherwise A_0 = I and
    A_i = diag(+1, -1, +1, ...) . P^i with P the right-shift permutation.
    "random": iid U(-1, 1)."""
    assert type in ["standard", "random"]
    if type == "random":
        return list(torch.empty(phm_dim, phm_dim, phm_dim, dtype=torch.float32).uniform_(-1, 1))
    assert phm_dim >= 1
    if phm_dim == 2:
        return get_complex_mul_rule()
    if phm_dim == 4:
        return get_quaternion_mul_rule()
    shift = get_right_shift_permutation_matrix(phm_dim)
    signs = torch.diag(torch.tensor([1.0 if j % 2 == 0 else -1.0 for j in range(phm_dim)], dtype=torch.float32))
    mats = [torch.eye(phm_dim, dtype=torch.float32)]
    cur = signs
    for _ in range(1, phm_dim):
        cur = cur @ shift
        mats.append(cur.clone())
    return mats


def ensure_first_ax(x: torch.Tensor, type_dim: int = 4, type: str = "phm") -> torch.Tensor:
    """Move the (first) axis of length ``type_dim`` of a 3-D tensor to the front   (ref :96-119)."""
    assert type in ["phm", "batch"]
    assert x.dim() == 3
    hits = [ax for ax, s in enumerate(x.shape) if s == type_dim]
    if len(hits) > 1:
        warnings.warn(f"{type} occurs {len(hits)} times in shape {tuple(x.shape)}; using the first occurrence.")
    ax = hits[0]
    if ax == 0:
        return x
    return x.permute(ax, *[a for a in range(3) if a != ax])


def phm_cat(tensors: list, phm_dim: int, dim=-1) -> torch.Tensor:
    """Concatenate component-major tensors [N, n*k_t] so that the result is again
    component-major: [c0 of every tensor | c1 of every tensor | ...]   (ref :122-135)."""
    split = [t.reshape(t.shape[0], phm_dim, t.shape[1] // phm_dim) for t in tensors]
    out = torch.cat(split, dim=2).reshape(tensors[0].shape[0], -1)
    assert out.shape[1] == sum(t.shape[1] for t in tensors)
    return out


def any_forward_hooks(*roots) -> bool:
    """True if a forward (pre-)hook is registered on any of the modules or their sub-modules (or globally)."""
    from torch.nn.modules import module as _m
    if _m._global_forward_hooks or _m._global_forward_pre_hooks:
        return True
    for root in roots:
        if root is None:
            continue
        for sub in root.modules():
            if sub._forward_hooks or sub._forward_pre_hooks:
                return True
    return False
