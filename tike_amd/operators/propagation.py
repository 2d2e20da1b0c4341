"""Far-field propagation = batched 2-D FFT
(reference operators/cupy/propagation.py:13-118; cuFFT replaced by
csrc/fft2.hip)."""
import numpy as np
import torch

from .. import _arrays as A
from .._lib import check, lib
from .operator import Operator


def fft_scales(n, norm):
    """(forward, inverse) scale factors of an n x n transform for `norm`."""
    if norm == "ortho":
        return 1.0 / n, 1.0 / n
    if norm == "forward":
        return 1.0 / (n * n), 1.0
    if norm == "backward" or norm is None:
        return 1.0, 1.0 / (n * n)
    raise ValueError(f"unknown FFT normalization {norm!r}")


class Propagation(Operator):
    """Fourier-based free-space propagation over the last two dimensions."""

    def __init__(self, detector_shape, norm="ortho", **kwargs):
        self.detector_shape = detector_shape
        self.norm = norm

    def _check_shape(self, x):
        shape = (-1, self.detector_shape, self.detector_shape)
        if tuple(x.shape[-2:]) != shape[-2:]:
            raise ValueError(f"waves must have shape {shape} not {x.shape}.")

    def _run(self, x, overwrite, inverse):
        kind = x
        self._check_shape(x)
        xt = A.to_device(x, np.complex64)
        out = xt if (overwrite and A.is_device(x) and
                     xt.data_ptr() == x.data_ptr()) else torch.empty_like(xt)
        n = self.detector_shape
        scale = fft_scales(n, self.norm)[1 if inverse else 0]
        ntile = xt.numel() // (n * n)
        check(
            lib.tike_fft2(A.ptr(xt), A.ptr(out), ntile, n, int(inverse), scale,
                          A.stream_ptr()), "Propagation")
        return A.like_input(out, kind)

    def fwd(self, nearplane, overwrite=False, **kwargs):
        return self._run(nearplane, overwrite, inverse=False)

    def adj(self, farplane, overwrite=False, **kwargs):
        return self._run(farplane, overwrite, inverse=True)


class ZeroPropagation(Propagation):
    """Zero-distance propagation: the identity."""

    def fwd(self, nearplane, overwrite=False, **kwargs):
        return nearplane

    def adj(self, farplane, overwrite=False, **kwargs):
        return farplane
