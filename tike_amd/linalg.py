"""Linear algebra with broadcasting and complex support.

Mirror of reference src/tike/linalg.py:12-137; every function accepts NumPy
arrays or torch tensors (device arrays of this package) and returns the same
kind.
"""
import numpy as np
import torch


def _is_t(x):
    return isinstance(x, torch.Tensor)


def _axis(axis):
    if isinstance(axis, list):
        return tuple(axis)
    return axis


def _sum(x, axis=None, keepdims=False):
    axis = _axis(axis)
    if _is_t(x):
        if axis is None:
            r = x.sum()
            return r.reshape((1,) * x.ndim) if keepdims else r
        return x.sum(dim=axis, keepdim=keepdims)
    return np.sum(x, axis=axis, keepdims=keepdims)


def _mean(x, axis=None, keepdims=False):
    axis = _axis(axis)
    if _is_t(x):
        if axis is None:
            r = x.mean()
            return r.reshape((1,) * x.ndim) if keepdims else r
        return x.mean(dim=axis, keepdim=keepdims)
    return np.mean(x, axis=axis, keepdims=keepdims)


def _sqrt(x):
    return torch.sqrt(x) if _is_t(x) else np.sqrt(x)


def mnorm(x, axis=None, keepdims=False):
    """Vector 2-norm with the sum replaced by a mean (linalg.py:12-14)."""
    return _sqrt(_mean((x * x.conj()).real, axis=axis, keepdims=keepdims))


def norm(x, axis=None, keepdims=False):
    """Vector 2-norm along the given axes (linalg.py:17-19)."""
    return _sqrt(_sum((x * x.conj()).real, axis=axis, keepdims=keepdims))


def inner(x, y, axis=None, keepdims=False):
    """Complex inner product <x, y> = sum(x * conj(y)) (linalg.py:28-30)."""
    return _sum(x * y.conj(), axis=axis, keepdims=keepdims)


def projection(a, b, axis=None):
    """Complex projection of a onto b (linalg.py:22-25)."""
    bh = b / inner(b, b, axis=axis, keepdims=True)
    return inner(a, b, axis=axis, keepdims=True) * bh


def hermitian(x):
    return x.conj().swapaxes(-1, -2)


def lstsq(a, b, weights=None):
    """Least-squares solution x of a @ x = b for stacks of matrices, a
    (..., M, N), b (..., M, K) -> x (..., N, K), through the normal equations
    x = (a^H a)^-1 a^H b; rows optionally weighted by `weights` (..., M)
    (linalg.py:33-61)."""
    rows = tuple(a.shape[:-1])
    if rows != tuple(b.shape[:-1]):
        raise AssertionError(f"Leading dims of a {tuple(a.shape)} and b "
                             f"{tuple(b.shape)} must be same!")
    if weights is not None and tuple(weights.shape) != rows:
        raise AssertionError("one weight per row of a")
    if weights is not None:
        a, b = (m * _sqrt(weights)[..., None] for m in (a, b))
    # Evaluated as (inverse(a^H a) a^H) b in the working precision, like the
    # reference: with float32 pixel coordinates (the affine position fit) the
    # normal equations are ill-conditioned enough (cond ~ 1e4-1e5) for a
    # differently ordered solve to move the translation by tenths of a pixel
    # -- the reference-run fixtures pin this order
    # (lstsq_recon_positions_masked.npz).
    inverse = torch.linalg.inv if _is_t(a) else np.linalg.inv
    adjoint = hermitian(a)
    return inverse(adjoint @ a) @ adjoint @ b


def orthogonalize_gs(x, axis=-1, N=None):
    """Gram-Schmidt for complex arrays (linalg.py:64-105): the vectors live
    on `axis` (one or several dimensions) and are counted along dimension N
    (default: the last dimension outside `axis`); all other dimensions
    broadcast.  Vector k + 1 onwards lose their component along the already
    orthogonal vector k, for k = 0, 1, ..."""
    dims = x.ndim
    along = tuple(d % dims for d in (axis if isinstance(axis, (tuple, list))
                                     else (axis,)))
    if N is None:
        N = max((d for d in range(dims) if d not in along), default=-1)
    N %= dims
    if N in along:
        raise ValueError("Cannot orthogonalize a single vector.")
    move = torch.movedim if _is_t(x) else np.moveaxis
    original = move(x, N, 0)
    basis = original.clone() if _is_t(x) else original.copy()
    for k in range(len(original) - 1):
        pivot = basis[k:k + 1]
        overlap = inner(original[k + 1:], pivot, axis=along, keepdims=True)
        basis[k + 1:] -= pivot * (overlap /
                                  inner(pivot, pivot, axis=along, keepdims=True))
    return move(basis, 0, N)


def cov(x):
    x0 = x - _mean(x, axis=-2, keepdims=True)
    return hermitian(x0) @ x0


def pca_eig(data, k):
    """k principal components via eigen-decomposition (linalg.py:118-137)."""
    eigh = torch.linalg.eigh if _is_t(data) else np.linalg.eigh
    S, U = eigh(cov(data))
    idx = list(range(S.shape[-1] - 1, S.shape[-1] - 1 - k, -1))
    return S[..., idx], U[..., idx]
