"""Kronecker-product helpers with the reference's names and index layout
(reference: phc/hypercomplex/kronecker.py).  Host-side utilities: the engine
never materialises these products on the hot path (H tiles are assembled in LDS
inside the PHM kernels), but regularisers, tests and user code may call them."""
import torch


def kronecker_product_single(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """2-D Kronecker product: out[a*Bh + k, c*Bw + p] = A[a, c] * B[k, p]   (ref :4-25)."""
    if A.dim() != 2 or B.dim() != 2:
        raise AssertionError("kronecker_product_single expects two matrices")
    return torch.kron(A, B)


def _kronecker_product(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Same product through a broadcasted outer product   (ref :28-32)."""
    assert A.dim() == B.dim() == 2
    return (A[:, None, :, None] * B[None, :, None, :]).reshape(A.shape[0] * B.shape[0], A.shape[1] * B.shape[1])


def kronecker_product_einsum_batched(A: torch.Tensor, B: torch.Tensor) -> torch.Tensor:
    """Batched product (b,a,c) x (b,k,p) -> (b, a*k, c*p)   (ref :35-48).
    Row index a*k_dim + k, column index c*p_dim + p: this is the layout of the
    PHM weight H = sum_b A_b (x) W_b that the kernels reproduce."""
    assert A.dim() == 3 and B.dim() == 3
    b, a, c = A.shape
    k, p = B.shape[1], B.shape[2]
    prod = A.reshape(b, a, 1, c, 1) * B.reshape(b, 1, k, 1, p)
    return prod.reshape(b, a * k, c * p)


def kronecker_product(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Kronecker product over the last two dims with broadcast leading dims   (ref :53-65)."""
    rows = a.shape[-2] * b.shape[-2]
    cols = a.shape[-1] * b.shape[-1]
    prod = a.unsqueeze(-1).unsqueeze(-3) * b.unsqueeze(-2).unsqueeze(-4)
    return prod.reshape(prod.shape[:-4] + (rows, cols))
