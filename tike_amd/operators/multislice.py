"""Object-probe interaction (reference operators/cupy/multislice.py:18-279).

Only the single-slice object (``len(psi) == 1``) is on the accelerated path;
for it ``Multislice`` reduces to ``Convolution`` (multislice.py:86 never
enters the slice loop) and the adjoint's division by ``nslices`` is by one.
"""
from .convolution import Convolution
from .operator import Operator
from .propagation import ZeroPropagation


class Multislice(Operator):
    """Multiple-slice wavefield propagation, restricted to one slice."""

    def __init__(self, detector_shape, probe_shape, nz, n,
                 probe_wavelength=float("nan"),
                 probe_FOV_lengths=(float("nan"), float("nan")),
                 multislice_propagation_distance=1e-9, propagation=None,
                 diffraction=Convolution, norm="ortho", **kwargs):
        self.diffraction = diffraction(probe_shape=probe_shape,
                                       detector_shape=detector_shape, nz=nz,
                                       n=n, **kwargs)
        # near-field propagator between slices: unused for one slice
        self.propagation = ZeroPropagation(detector_shape=probe_shape)
        self.probe_shape = probe_shape
        self.detector_shape = detector_shape
        self.nz = nz
        self.n = n
        self.probe_wavelength = probe_wavelength
        self.probe_FOV_lengths = probe_FOV_lengths
        self.multislice_propagation_distance = multislice_propagation_distance

    def __enter__(self):
        self.propagation.__enter__()
        self.diffraction.__enter__()
        return self

    def __exit__(self, type, value, traceback):
        self.propagation.__exit__(type, value, traceback)
        self.diffraction.__exit__(type, value, traceback)

    @staticmethod
    def _one_slice(psi):
        assert psi.ndim == 3
        if psi.shape[0] != 1:
            raise NotImplementedError(
                "tike_amd accelerates single-slice objects only "
                f"(psi.shape[0] == 1); got {tuple(psi.shape)}.")

    def fwd(self, probe, scan, psi, **kwargs):
        self._one_slice(psi)
        return self.diffraction.fwd(psi=psi[0], scan=scan, probe=probe)

    def adj(self, nearplane, probe, scan, psi, overwrite=False, **kwargs):
        self._one_slice(psi)
        psi_adj = self.diffraction.adj(nearplane=nearplane, probe=probe,
                                       scan=scan, overwrite=False)[None, ...]
        probe_adj = self.diffraction.adj_probe(nearplane=nearplane, scan=scan,
                                               psi=psi[0])
        return psi_adj, probe_adj

    @property
    def patch(self):
        return self.diffraction.patch

    @property
    def pad(self):
        return self.diffraction.pad

    @property
    def end(self):
        return self.diffraction.end


class SingleSlice(Multislice):
    """Single-slice wavefield propagation (multislice.py:209-279)."""
