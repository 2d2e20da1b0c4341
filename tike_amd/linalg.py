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
    """Batched least squares a @ x = b (linalg.py:33-61)."""
    assert a.shape[:-1] == b.shape[:-1]
    if weights is not None:
        assert weights.shape == a.shape[:-1]
        w = _sqrt(weights[..., None])
        a, b = a * w, b * w
    aT = hermitian(a)
    inv = torch.linalg.inv if _is_t(a) else np.linalg.inv
    return inv(aT @ a) @ aT @ b


def orthogonalize_gs(x, axis=-1, N=None):
    """Gram-Schmidt orthogonalisation (linalg.py:64-105)."""
    try:
        axis = tuple(a % x.ndim for a in axis)
    except TypeError:
        axis = (axis % x.ndim,)
    if N is None:
        N = x.ndim - 1
        while N in axis:
            N -= 1
    N = N % x.ndim
    if N in axis:
        raise ValueError("Cannot orthogonalize a single vector.")
    if _is_t(x):
        x = torch.movedim(x, N, 0)
        u = x.clone()
    else:
        x = np.moveaxis(x, N, 0)
        u = x.copy()
    for i in range(1, len(x)):
        u[i:] -= projection(x[i:], u[i - 1:i], axis=axis)
    return torch.movedim(u, 0, N) if _is_t(u) else np.moveaxis(u, 0, N)


def cov(x):
    x0 = x - _mean(x, axis=-2, keepdims=True)
    return hermitian(x0) @ x0


def pca_eig(data, k):
    """k principal components via eigen-decomposition (linalg.py:118-137)."""
    eigh = torch.linalg.eigh if _is_t(data) else np.linalg.eigh
    S, U = eigh(cov(data))
    idx = list(range(S.shape[-1] - 1, S.shape[-1] - 1 - k, -1))
    return S[..., idx], U[..., idx]
