"""Shared helpers for the parity tests (golden loading, error metric)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# SURVEY.md §7 "hard parts": element-wise relative error explodes near zero when
# the summation order differs, so parity is judged per tensor as
# max|Δ| / max|ref|  (north_star tolerance: 1e-5 relative fp32).
TOL = 1e-5


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def to_t(a):
    """numeric arrays -> torch tensors (0-dim too); strings stay numpy."""
    return torch.from_numpy(np.asarray(a)) if a.dtype.kind in "fiub" else a


def tensors(z):
    return {k: to_t(z[k]) for k in z.files}


def case(z, ci):
    """All arrays of case ``ci`` as torch tensors, keyed without the prefix."""
    pre = f"c{ci}::"
    return {k[len(pre):]: to_t(z[k]) for k in z.files if k.startswith(pre)}


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def relerr(got, ref):
    got = torch.as_tensor(got).detach().double().cpu()
    ref = torch.as_tensor(ref).detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.numel() == 0:
        return 0.0
    scale = float(ref.abs().max())
    return float((got - ref).abs().max()) / max(scale, 1e-30)


def assert_close(got, ref, tol=TOL, what=""):
    e = relerr(got, ref)
    assert e <= tol, f"{what}: max|Δ|/max|ref| = {e:.3e} > {tol:.1e}"
