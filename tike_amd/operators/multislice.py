"""Object-probe interaction through one or several object slices
(reference operators/cupy/multislice.py:18-279).

A single-slice object (``len(psi) == 1``) reduces to ``Convolution`` and is
what the fused solver kernels accelerate.  With several slices the exit wave
of slice s, propagated by ``FresnelSpectProp``, is the probe of slice s+1
(multislice.py:86-91); this composition runs the same HIP kernels slice by
slice and requires ``detector_shape == probe_shape`` like the reference (the
propagated exit wave must be probe-shaped)."""
import numpy as np
import torch

from .. import _arrays as A
from .convolution import Convolution
from .fresnelspectprop import FresnelSpectProp
from .operator import Operator
from .propagation import ZeroPropagation


class Multislice(Operator):
    """Multiple-slice wavefield propagation."""

    def __init__(self, detector_shape, probe_shape, nz, n,
                 probe_wavelength=float("nan"),
                 probe_FOV_lengths=(float("nan"), float("nan")),
                 multislice_propagation_distance=1e-9,
                 propagation=FresnelSpectProp, diffraction=Convolution,
                 norm="ortho", **kwargs):
        self.diffraction = diffraction(probe_shape=probe_shape,
                                       detector_shape=detector_shape, nz=nz,
                                       n=n, **kwargs)
        self.propagation = propagation(
            norm=norm, probe_shape=probe_shape, wavelength=probe_wavelength,
            probe_FOV=probe_FOV_lengths,
            distance=multislice_propagation_distance, **kwargs)
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
                "this fused path handles single-slice objects only "
                f"(psi.shape[0] == 1); got {tuple(psi.shape)}.")

    def _check_slices(self, psi):
        assert psi.ndim == 3
        if len(psi) > 1 and self.detector_shape != self.probe_shape:
            raise ValueError(
                "a multislice object needs detector_shape == probe_shape "
                "(the propagated exit wave is the next slice's probe)")

    def fwd(self, probe, scan, psi, **kwargs):
        """multislice.py:69-92."""
        self._check_slices(psi)
        exitwave = self.diffraction.fwd(psi=psi[0], scan=scan, probe=probe)
        for s in range(1, len(psi)):
            exitwave = self.diffraction.fwd(
                psi=psi[s], scan=scan, probe=self.propagation.fwd(exitwave))
        return exitwave

    def fwd_return_intermediate_probes(self, probe, scan, psi, **kwargs):
        """Exit wave plus the probe incident on every slice, (D, N, S, pw, pw)
        (multislice.py:97-141)."""
        self._check_slices(psi)
        kind = psi
        psi = A.to_device(psi, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        N = scan.shape[-2]
        probes = torch.zeros((psi.shape[0], N, *probe.shape[-3:]),
                             dtype=torch.complex64, device=psi.device)
        probes[0] = probe[..., 0, :, :, :] if probe.ndim == 5 else probe
        exitwave = None
        for t in range(len(psi)):
            exitwave = self.diffraction.fwd(psi=psi[t], scan=scan,
                                            probe=probes[t])
            if t == len(psi) - 1:
                break
            probes[t + 1] = self.propagation.fwd(nearplane=exitwave)
        return A.like_input(exitwave, kind), A.like_input(probes, kind)

    def adj(self, nearplane, probe, scan, psi, overwrite=False, **kwargs):
        """multislice.py:144-194 (including the division of psi_adj by the
        number of slices)."""
        self._check_slices(psi)
        kind = nearplane
        psi = A.to_device(psi, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        nearplane = A.to_device(nearplane, np.complex64)
        nslices = len(psi)
        probes = [None] * nslices
        probes[0] = probe
        for s in range(1, nslices):
            probes[s] = self.propagation.fwd(
                self.diffraction.fwd(psi=psi[s - 1], scan=scan,
                                     probe=probes[s - 1]))
        psi_adj = torch.zeros_like(psi)
        psi_adj[nslices - 1] = self.diffraction.adj(
            nearplane=nearplane, probe=probes[nslices - 1], scan=scan,
            overwrite=False)
        probe_adj = self.diffraction.adj_probe(nearplane=nearplane, scan=scan,
                                               psi=psi[nslices - 1])
        for s in range(nslices - 2, -1, -1):
            probe_adj = self.propagation.adj(probe_adj)
            psi_adj[s] = self.diffraction.adj(nearplane=probe_adj,
                                              probe=probes[s], scan=scan,
                                              overwrite=False)
            probe_adj = self.diffraction.adj_probe(nearplane=probe_adj,
                                                   scan=scan, psi=psi[s])
        return (A.like_input(psi_adj / nslices, kind),
                A.like_input(probe_adj, kind))

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

    def __init__(self, detector_shape, probe_shape, nz, n,
                 propagation=ZeroPropagation, diffraction=Convolution,
                 norm="ortho", **kwargs):
        self.diffraction = diffraction(probe_shape=probe_shape,
                                       detector_shape=detector_shape, nz=nz,
                                       n=n, **kwargs)
        self.propagation = propagation(detector_shape=detector_shape)
        self.probe_shape = probe_shape
        self.detector_shape = detector_shape
        self.nz = nz
        self.n = n

    def fwd(self, probe, scan, psi, **kwargs):
        assert psi.shape[0] == 1 and psi.ndim == 3
        return self.diffraction.fwd(psi=psi[0], scan=scan, probe=probe)

    def adj(self, nearplane, probe, scan, psi=None, overwrite=False,
            **kwargs):
        assert psi is None or (psi.shape[0] == 1 and psi.ndim == 3)
        psi_adj = self.diffraction.adj(nearplane=nearplane, probe=probe,
                                       scan=scan, overwrite=False)[None, ...]
        probe_adj = self.diffraction.adj_probe(nearplane=nearplane, scan=scan,
                                               psi=psi[0], overwrite=False)
        return psi_adj, probe_adj
