"""Fresnel spectrum (short-range, near-field) propagation
(reference operators/cupy/fresnelspectprop.py:15-137): the free-space step
between the slices of a multislice object.  Two passes of the HIP FFT engine
around one spectral multiply (``tike_fresnel_spect_prop``)."""
import numpy as np
import torch

from .. import _arrays as A
from .._lib import check, lib
from .operator import Operator
from .propagation import fft_scales


def fresnel_spectrum_propagator(N, probe_FOV, distance, wavelength):
    """exp(i z sqrt(k^2 - kx^2 - ky^2)) on the FFT's own frequency order,
    complex64 (N[0], N[1]) -- what fresnelspectprop.py:115-137 builds centred
    and then fft-shifts.  The reference samples the frequencies half a bin off
    centre, (m + 1/2 - n/2) * 2 pi / FOV for m = 0..n-1; here each axis is put
    into FFT order (a roll by n // 2, the 1-D form of the 2-D shift) before
    the outer sum, in float64 like the reference's grids."""

    def squared_frequencies(n, fov):
        centred = (np.arange(n, dtype=np.float64) + 0.5 - 0.5 * n) / n
        return np.roll(2 * np.pi * n * centred / fov, n // 2)**2

    k2 = (2 * np.pi / wavelength)**2
    kz = np.sqrt(k2 - squared_frequencies(N[0], probe_FOV[0])[:, None]
                 - squared_frequencies(N[1], probe_FOV[1])[None, :])
    return np.exp(1j * distance * kz).astype(np.complex64)


class FresnelSpectProp(Operator):
    """nearplane (..., W, H) complex64 -> the wavefront `distance` further."""

    def __init__(self, norm="ortho", pixel_size=1e-7, probe_FOV=(1e-6, 1e-6),
                 distance=1e-6, wavelength=1e-9, **kwargs):
        vars(self).update(norm=norm, pixel_size=pixel_size,
                          probe_FOV=probe_FOV, distance=distance,
                          wavelength=wavelength)
        self._cache = {}  # propagator per (shape, device)

    def _propagator(self, shape, device):
        key = (tuple(shape), str(device))
        if key not in self._cache:
            self._cache[key] = A.to_device(
                fresnel_spectrum_propagator(shape, self.probe_FOV,
                                            self.distance, self.wavelength),
                np.complex64, device)
        return self._cache[key]

    def _run(self, x, overwrite, adjoint):
        kind = x
        xt = A.to_device(x, np.complex64)
        n = xt.shape[-1]
        if xt.shape[-2] != n:
            raise ValueError(f"waves must be square, not {tuple(xt.shape)}.")
        out = xt if (overwrite and A.is_device(x) and
                     xt.data_ptr() == x.data_ptr()) else torch.empty_like(xt)
        fwd_scale, inv_scale = fft_scales(n, self.norm)
        check(
            lib.tike_fresnel_spect_prop(
                A.ptr(xt), A.ptr(out),
                A.ptr(self._propagator((n, n), xt.device)),
                xt.numel() // (n * n), n, int(adjoint), fwd_scale, inv_scale,
                A.stream_ptr()), "FresnelSpectProp")
        return A.like_input(out, kind)

    def fwd(self, nearplane, overwrite=False, **kwargs):
        return self._run(nearplane, overwrite, adjoint=False)

    def adj(self, farplane, overwrite=False, **kwargs):
        return self._run(farplane, overwrite, adjoint=True)
