"""NumPy/SciPy-backed stand-in for ``cupyx`` (fixture generation only)."""
import numpy as _np

from . import scipy  # noqa: F401


def empty_pinned(shape, dtype=float, order='C'):
    # plain ndarray on purpose: stream_and_modify2 rejects cupy arrays in
    # ind_args (reference communicators/stream.py:364-370)
    return _np.empty(shape, dtype=dtype, order=order)


def zeros_pinned(shape, dtype=float, order='C'):
    return _np.zeros(shape, dtype=dtype, order=order)


def empty_like_pinned(a):
    return _np.empty_like(_np.asarray(a))
