"""Solver options and the parameter container
(reference src/tike/ptycho/solvers/options.py:19-330)."""
from __future__ import annotations

import abc
import copy
import dataclasses
import typing

import numpy as np

from ... import _arrays as A
from ... import precision
from ..exitwave import (ExitWaveOptions, crop_fourier_space,
                        pad_fourier_space)
from ..object import ObjectOptions
from ..position import PositionOptions, check_allowed_positions
from ..probe import ProbeOptions


@dataclasses.dataclass
class IterativeOptions(abc.ABC):
    """Options shared by the iterative solvers (options.py:19-78)."""

    name: str = dataclasses.field(default="", init=False)
    num_batch: int = 1
    batch_method: str = "wobbly_center"
    rescale_method: str = "mean_of_abs_object"
    rescale_period: int = 10
    costs: typing.List[typing.List[float]] = dataclasses.field(
        init=False, default_factory=list)
    num_iter: int = 1
    times: typing.List[float] = dataclasses.field(init=False,
                                                  default_factory=list)
    convergence_window: int = 0
    time_limit: float = np.inf


@dataclasses.dataclass
class LstsqOptions(IterativeOptions):
    name: str = dataclasses.field(default="lstsq_grad", init=False)


@dataclasses.dataclass
class RpieOptions(IterativeOptions):
    """Options of the rpie solver (options.py:82-90): `alpha` = 1 is ePIE."""
    name: str = dataclasses.field(default="rpie", init=False)
    num_batch: int = 5
    alpha: float = 0.05


@dataclasses.dataclass
class CgradOptions(IterativeOptions):
    """Conjugate-gradient solver composed from tike.opt.conjugate_gradient
    (the reference snapshot has no ptychography cgrad)."""
    name: str = dataclasses.field(default="cgrad", init=False)
    batch_method: str = "compact"
    cg_iter: int = 4
    step_length: float = 1.0


@dataclasses.dataclass
class PtychoParameters():
    """Forward-model parameters (options.py:98-330).

    probe (1, 1, SHARED, WIDE, HIGH) complex64; psi (DEPTH, WIDE, HIGH)
    complex64; scan (POSI, 2) float32; eigen_probe (1, EIGEN, SHARED', W, H);
    eigen_weights (POSI, EIGEN + 1, SHARED) float32.
    """
    probe: typing.Any
    psi: typing.Any
    scan: typing.Any
    eigen_probe: typing.Any = None
    eigen_weights: typing.Any = None
    algorithm_options: IterativeOptions = dataclasses.field(
        default_factory=RpieOptions)
    exitwave_options: ExitWaveOptions = None
    probe_options: typing.Union[ProbeOptions, None] = None
    object_options: typing.Union[ObjectOptions, None] = None
    position_options: typing.Union[PositionOptions, None] = None

    def __post_init__(self):
        if (self.scan.ndim != 2 or self.scan.shape[1] != 2
                or np.any(np.asarray(self.scan.shape) < 1)):
            raise ValueError(f"scan shape {self.scan.shape} is incorrect. "
                             "It should be (N, 2) "
                             "where N >= 1 is the number of scan positions.")
        if (self.probe.ndim != 5 or tuple(self.probe.shape[:2]) != (1, 1)
                or np.any(np.asarray(self.probe.shape) < 1)
                or self.probe.shape[-2] != self.probe.shape[-1]):
            raise ValueError(f"probe shape {self.probe.shape} is incorrect. "
                             "It should be (1, 1, S, W, H) "
                             "where S >=1 is the number of probes, and "
                             "W, H >= 1 are the square probe grid dimensions.")
        if (self.psi.ndim != 3 or np.any(
                np.asarray(self.psi.shape[-2:]) <= np.asarray(
                    self.probe.shape[-2:]))):
            raise ValueError(
                f"psi shape {self.psi.shape} is incorrect. "
                "It should be (D, W, H) where W, H > probe.shape[-2:].")
        check_allowed_positions(self.scan, self.psi, self.probe.shape)
        if self.exitwave_options is None:
            self.exitwave_options = ExitWaveOptions(measured_pixels=np.ones(
                tuple(self.probe.shape[-2:]), dtype=np.bool_))

    def _map(self, f, fo):
        return PtychoParameters(
            probe=f(self.probe, precision.cfloating),
            psi=f(self.psi, precision.cfloating),
            scan=f(self.scan, precision.floating),
            eigen_probe=f(self.eigen_probe, precision.cfloating)
            if self.eigen_probe is not None else None,
            eigen_weights=f(self.eigen_weights, precision.floating)
            if self.eigen_weights is not None else None,
            algorithm_options=self.algorithm_options,
            exitwave_options=fo(self.exitwave_options),
            probe_options=fo(self.probe_options),
            object_options=fo(self.object_options),
            position_options=fo(self.position_options),
        )

    def resample(self, factor: float, interp=None) -> "PtychoParameters":
        """Host copy of the parameters on a grid rescaled by `factor`
        (options.py:170-196): probes by `interp` (Fourier interpolation by
        default), the object by a cubic spline, positions scaled."""
        interp = _resize_fft if interp is None else interp
        h = A.to_host
        return PtychoParameters(
            probe=interp(h(self.probe), factor),
            psi=_resize_spline(h(self.psi), factor),
            scan=h(self.scan) * factor,
            eigen_probe=interp(h(self.eigen_probe), factor)
            if self.eigen_probe is not None else None,
            eigen_weights=None if self.eigen_weights is None else h(
                self.eigen_weights),
            algorithm_options=self.algorithm_options,
            probe_options=self.probe_options.resample(factor, interp)
            if self.probe_options is not None else None,
            object_options=self.object_options.resample(factor, interp)
            if self.object_options is not None else None,
            position_options=self.position_options.copy_to_host().resample(
                factor) if self.position_options is not None else None,
            exitwave_options=self.exitwave_options.resample(factor)
            if self.exitwave_options is not None else None,
        )

    def copy_to_device(self) -> "PtychoParameters":
        return self._map(
            lambda x, dt: A.to_device(x, dt),
            lambda o: None if o is None else o.copy_to_device())

    def copy_to_host(self) -> "PtychoParameters":
        return self._map(
            lambda x, dt: A.to_host(x),
            lambda o: None if o is None else o.copy_to_host())

    @staticmethod
    def split(indices, *, x: "PtychoParameters") -> "PtychoParameters":
        """Host copy keeping only the positions in `indices`
        (options.py:266-290)."""
        return PtychoParameters(
            probe=np.asarray(x.probe).astype(precision.cfloating),
            psi=np.asarray(x.psi).astype(precision.cfloating),
            scan=np.asarray(x.scan)[indices].astype(precision.floating),
            eigen_probe=np.asarray(x.eigen_probe).astype(precision.cfloating)
            if x.eigen_probe is not None else None,
            eigen_weights=np.asarray(x.eigen_weights)[indices].astype(
                precision.floating) if x.eigen_weights is not None else None,
            algorithm_options=copy.deepcopy(x.algorithm_options),
            exitwave_options=x.exitwave_options,
            probe_options=x.probe_options,
            object_options=x.object_options,
            position_options=x.position_options.split(indices)
            if x.position_options is not None else None,
        )


def _resize_spline(x, f: float):
    """Cubic-spline zoom of the last two axes (options.py:332-338)."""
    import scipy.ndimage
    return scipy.ndimage.zoom(x, zoom=[1] * (x.ndim - 2) + [f, f],
                              grid_mode=True, prefilter=False)


def _resize_fft(x, f: float):
    """Fourier interpolation of the last two axes (options.py:391-409)."""
    if f == 1:
        return x
    crop_or_pad = crop_fourier_space if f < 1 else pad_fourier_space
    return np.fft.ifft2(
        crop_or_pad(np.fft.fft2(x, norm="ortho", axes=(-2, -1)),
                    w=int(x.shape[-1] * f)), norm="ortho", axes=(-2, -1))
