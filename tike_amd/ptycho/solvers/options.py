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
        """Shape rules of the forward model (options.py:141-168): scan (N, 2);
        probe (1, 1, S, W, W); psi (D, H', W') larger than the probe window;
        every patch inside psi.  All pixels count as measured unless
        `exitwave_options` says otherwise."""
        scan, probe, psi = (tuple(int(n) for n in x.shape)
                            for x in (self.scan, self.probe, self.psi))
        rules = (
            (len(scan) == 2 and scan[1:] == (2,) and scan[0] >= 1,
             f"scan shape {self.scan.shape} is incorrect. "
             "It should be (N, 2) "
             "where N >= 1 is the number of scan positions."),
            (len(probe) == 5 and probe[:2] == (1, 1) and min(probe) >= 1
             and probe[3] == probe[4],
             f"probe shape {self.probe.shape} is incorrect. "
             "It should be (1, 1, S, W, H) "
             "where S >=1 is the number of probes, and "
             "W, H >= 1 are the square probe grid dimensions."),
            (len(psi) == 3 and len(probe) == 5
             and all(o > w for o, w in zip(psi[1:], probe[3:])),
             f"psi shape {self.psi.shape} is incorrect. "
             "It should be (D, W, H) where W, H > probe.shape[-2:]."),
        )
        for holds, complaint in rules:
            if not holds:
                raise ValueError(complaint)
        check_allowed_positions(self.scan, self.psi, self.probe.shape)
        self.exitwave_options = self.exitwave_options or ExitWaveOptions(
            measured_pixels=np.ones(probe[3:], dtype=np.bool_))

    _ARRAYS = (("probe", "cfloating"), ("psi", "cfloating"),
               ("scan", "floating"), ("eigen_probe", "cfloating"),
               ("eigen_weights", "floating"))
    _OPTIONS = ("exitwave_options", "probe_options", "object_options",
                "position_options")

    def _map(self, f, fo):
        """A copy with every array field through f(array, dtype) and every
        option object through fo(options); absent arrays stay absent."""
        changes = {}
        for name, kind in self._ARRAYS:
            value = getattr(self, name)
            if value is not None:
                changes[name] = f(value, getattr(precision, kind))
        changes.update((name, fo(getattr(self, name)))
                       for name in self._OPTIONS)
        return dataclasses.replace(self, **changes)

    def resample(self, factor: float, interp=None) -> "PtychoParameters":
        """Host copy of the parameters on a grid rescaled by `factor`
        (options.py:170-196): probes by `interp` (Fourier interpolation by
        default), the object by a cubic spline, positions scaled; the option
        objects rescale themselves."""
        interp = _resize_fft if interp is None else interp
        rescaled = {
            "probe": lambda x: interp(A.to_host(x), factor),
            "psi": lambda x: _resize_spline(A.to_host(x), factor),
            "scan": lambda x: A.to_host(x) * factor,
            "eigen_probe": lambda x: interp(A.to_host(x), factor),
            "eigen_weights": A.to_host,
            "probe_options": lambda o: o.resample(factor, interp),
            "object_options": lambda o: o.resample(factor, interp),
            "position_options": lambda o: o.copy_to_host().resample(factor),
            "exitwave_options": lambda o: o.resample(factor),
        }
        return dataclasses.replace(
            self, **{
                name: change(getattr(self, name))
                for name, change in rescaled.items()
                if getattr(self, name) is not None
            })

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
        """Host copy of `x` that keeps only the positions `indices`: the
        per-position arrays (scan, eigen weights, position options) are
        sliced, the shared ones cast to the working precision
        (options.py:266-290)."""

        def cast(array, dtype, rows=slice(None)):
            if array is None:
                return None
            return np.asarray(array)[rows].astype(dtype)

        positions = x.position_options
        return dataclasses.replace(
            x,
            probe=cast(x.probe, precision.cfloating),
            psi=cast(x.psi, precision.cfloating),
            eigen_probe=cast(x.eigen_probe, precision.cfloating),
            scan=cast(x.scan, precision.floating, indices),
            eigen_weights=cast(x.eigen_weights, precision.floating, indices),
            algorithm_options=copy.deepcopy(x.algorithm_options),
            position_options=(None if positions is None else
                              positions.split(indices)))


def _resize_spline(x, f: float):
    """Cubic-spline zoom of the last two axes (options.py:332-338)."""
    import scipy.ndimage
    return scipy.ndimage.zoom(x, zoom=[1] * (x.ndim - 2) + [f, f],
                              grid_mode=True, prefilter=False)


def _resize_fft(x, f: float):
    """Fourier interpolation of the last two axes to int(width * f) samples:
    the spectrum cropped (f < 1) or zero-padded (f > 1) (options.py:391-409)."""
    if f == 1:
        return x
    width = int(x.shape[-1] * f)
    spectrum = np.fft.fft2(x, norm="ortho")
    spectrum = (crop_fourier_space(spectrum, width) if f < 1 else
                pad_fourier_space(spectrum, width))
    return np.fft.ifft2(spectrum, norm="ortho")
