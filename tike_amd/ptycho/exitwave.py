"""Exit-wave options (reference src/tike/ptycho/exitwave.py:22-119)."""
from __future__ import annotations

import dataclasses

import numpy as np

from .. import _arrays as A


@dataclasses.dataclass
class ExitWaveOptions:
    """Settings of the exit-wave (far-plane) update; same fields and defaults
    as the reference."""

    measured_pixels: np.ndarray
    """Boolean (det, det) mask: True = measured pixel, False = bad pixel."""

    noise_model: str = "gaussian"
    """'gaussian' or 'poisson'."""

    step_length_weight: float = 0.5
    step_length_usemodes: str = "all_modes"
    step_length_start: float = 0.5

    unmeasured_pixels_scaling: float = 1.00
    """Scaling of the far-plane in unmeasured regions; 1.0 = none."""

    propagation_normalization: str = "ortho"
    """'ortho', 'forward' or 'backward' FFT scaling of the forward model."""

    def _copy(self, measured_pixels):
        """The same settings around another mask array."""
        return dataclasses.replace(self, measured_pixels=measured_pixels)

    def resample(self, factor: float) -> "ExitWaveOptions":
        """The mask cropped in Fourier space to the rescaled detector
        (exitwave.py:106-119)."""
        mask = np.asarray(A.to_host(self.measured_pixels))
        return self._copy(crop_fourier_space(mask,
                                             int(mask.shape[-1] * factor)))

    def copy_to_device(self) -> "ExitWaveOptions":
        return self._copy(A.to_device(np.asarray(A.to_host(
            self.measured_pixels), dtype=bool)))

    def copy_to_host(self) -> "ExitWaveOptions":
        return self._copy(np.asarray(A.to_host(self.measured_pixels),
                                     dtype=bool))


def _low_frequencies(n: int, w: int):
    """Where the w lowest frequencies sit on a corner-centred (unshifted FFT)
    axis of length n: the first w - w // 2 and the last w // 2 samples."""
    upper = w // 2
    return np.r_[0:w - upper, n - upper:n]


def crop_fourier_space(x, w: int):
    """The w x w lowest frequencies of the last two (corner-centred) axes
    (exitwave.py:237-249, options.py:366-377)."""
    assert x.shape[-2] == x.shape[-1], "Only works on square arrays right now."
    keep = _low_frequencies(x.shape[-1], w)
    return x[..., keep[:, None], keep[None, :]]


def pad_fourier_space(x, w: int):
    """Inverse of crop_fourier_space: the spectrum embedded in a w x w one
    whose higher frequencies are zero (options.py:380-388)."""
    assert x.shape[-2] == x.shape[-1], "Only works on square arrays right now."
    slots = _low_frequencies(w, x.shape[-1])
    wide = np.zeros((*x.shape[:-2], w, w), dtype=x.dtype)
    wide[..., slots[:, None], slots[None, :]] = x
    return wide
