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


def fused_slices(probe_shape, detector_shape, S):
    """Shapes whose slice-to-slice step runs on the two-pass kernels
    (`next_incident_probe`): 128^2, 256^2 or 512^2 tiles, probe window =
    detector, <= 8 modes."""
    return (probe_shape == detector_shape and detector_shape in (128, 256, 512)
            and S <= 8)


def next_incident_probe(psi_slice, scan, beam, scratch, out, propagator,
                        scale, patches=None, amplitude=None, keep=True):
    """The probe incident on the NEXT slice, fused (multislice.py:86-91 =
    Convolution.fwd + FresnelSpectProp.fwd): `tike_fwd_pass1` forms patch x
    incident probe on the fly and runs the first pass of the transform,
    `tike_fresnel_colpass` the column pass x propagator -> inverse pass 1,
    `tike_fft2_pass2_inplace` finishes in `out` (n, S, pw, pw).  beam
    (1|n, ..., S, pw, pw) device tensor; scratch: a workspace shaped like
    `out`; scale = forward x inverse normalisation; patches (n, pw, pw), if
    given, receives the object patches of this slice.  amplitude (n, pw, pw)
    float32, if given, receives sum_s |out_s|^2 from the last pass
    (`tike_fft2_pass2_intensity`); with keep=False that pass does not write
    the wave and `out` is scratch."""
    from .._lib import check, lib
    n = scan.shape[0]
    S, pw = out.shape[-3], out.shape[-1]
    H, W = psi_slice.shape[-2:]
    st = A.stream_ptr()
    check(
        lib.tike_fwd_pass1(A.ptr(psi_slice), A.ptr(scan), A.ptr(beam),
                           int(beam.shape[0] != 1), None, None, None, 0, 0,
                           A.ptr(scratch), A.ptr(patches), n, S, pw, pw, H, W,
                           st), "slice exit wave, pass 1")
    check(
        lib.tike_fresnel_colpass(A.ptr(scratch), A.ptr(propagator), 0,
                                 A.ptr(out), n * S, pw, scale, st),
        "Fresnel step: column passes")
    if amplitude is None:
        check(lib.tike_fft2_pass2_inplace(A.ptr(out), n * S, pw, 1, 1.0, st),
              "Fresnel step: pass 2")
    else:
        check(
            lib.tike_fft2_pass2_intensity(A.ptr(out), A.ptr(amplitude), n, S,
                                          pw, 1, 1.0, int(keep), st),
            "Fresnel step: pass 2 + illumination")
    return out


class Multislice(Operator):
    """Multiple-slice wavefield propagation."""

    def __init__(self, detector_shape, probe_shape, nz, n,
                 probe_wavelength=float("nan"),
                 probe_FOV_lengths=(float("nan"), float("nan")),
                 multislice_propagation_distance=1e-9,
                 propagation=FresnelSpectProp, diffraction=Convolution,
                 norm="ortho", **kwargs):
        geometry = dict(probe_shape=probe_shape, detector_shape=detector_shape,
                        nz=nz, n=n)
        optics = dict(
            probe_wavelength=probe_wavelength,
            probe_FOV_lengths=probe_FOV_lengths,
            multislice_propagation_distance=multislice_propagation_distance)
        vars(self).update(geometry, **optics)  # the reference's attributes
        self.diffraction = diffraction(**geometry, **kwargs)
        self.propagation = propagation(
            norm=norm, probe_shape=probe_shape, probe_FOV=probe_FOV_lengths,
            wavelength=probe_wavelength,
            distance=multislice_propagation_distance, **kwargs)

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

    def _incident_probes(self, probe, scan, psi):
        """The probe that falls on slice 0, 1, ...: the exit wave of a slice,
        carried on by `propagation`, is the next slice's probe
        (multislice.py:86-91)."""
        beam = probe
        for depth, layer in enumerate(psi):
            yield beam
            if depth + 1 < len(psi):
                beam = self.propagation.fwd(
                    self.diffraction.fwd(psi=layer, scan=scan, probe=beam))

    def fwd(self, probe, scan, psi, **kwargs):
        """Exit wave behind the last slice (multislice.py:69-92)."""
        self._check_slices(psi)
        *_, last = self._incident_probes(probe, scan, psi)
        return self.diffraction.fwd(psi=psi[-1], scan=scan, probe=last)

    def fwd_return_intermediate_probes(self, probe, scan, psi, **kwargs):
        """Exit wave plus the probe incident on every slice, (D, N, S, pw, pw)
        (multislice.py:97-141)."""
        self._check_slices(psi)
        kind = psi
        psi = A.to_device(psi, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        first = probe[..., 0, :, :, :] if probe.ndim == 5 else probe
        beams = list(self._incident_probes(first, scan, psi))
        exitwave = self.diffraction.fwd(psi=psi[-1], scan=scan,
                                        probe=beams[-1])
        every = (scan.shape[-2], *probe.shape[-3:])
        stacked = torch.stack([b.expand(every) for b in beams]).contiguous()
        return A.like_input(exitwave, kind), A.like_input(stacked, kind)

    def adj(self, nearplane, probe, scan, psi, overwrite=False, **kwargs):
        """Adjoint of `fwd` with respect to the object slices and the probe
        (multislice.py:144-194): the incident probes are recomputed front to
        back, then the wave is taken back slice by slice -- each slice yields
        its object gradient from (wave, incident probe) and hands
        conj(object) x wave, back-propagated, to the slice in front.  The
        object gradient is divided by the number of slices, as the reference
        does."""
        self._check_slices(psi)
        kind = nearplane
        psi = A.to_device(psi, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        wave = A.to_device(nearplane, np.complex64)
        beams = list(self._incident_probes(probe, scan, psi))
        psi_adj = torch.empty_like(psi)
        for depth in reversed(range(len(psi))):
            if depth < len(psi) - 1:
                wave = self.propagation.adj(wave)
            psi_adj[depth] = self.diffraction.adj(
                nearplane=wave, probe=beams[depth], scan=scan, overwrite=False)
            wave = self.diffraction.adj_probe(nearplane=wave, scan=scan,
                                              psi=psi[depth])
        return (A.like_input(psi_adj / len(psi), kind),
                A.like_input(wave, kind))

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
        geometry = dict(probe_shape=probe_shape, detector_shape=detector_shape,
                        nz=nz, n=n)
        vars(self).update(geometry)
        self.diffraction = diffraction(**geometry, **kwargs)
        self.propagation = propagation(detector_shape=detector_shape)

    def fwd(self, probe, scan, psi, **kwargs):
        assert psi.shape[0] == 1 and psi.ndim == 3
        return self.diffraction.fwd(psi=psi[0], scan=scan, probe=probe)

    def adj(self, nearplane, probe, scan, psi=None, overwrite=False,
            **kwargs):
        assert psi is None or (psi.shape[0] == 1 and psi.ndim == 3)
        common = dict(nearplane=nearplane, scan=scan, overwrite=False)
        return (self.diffraction.adj(probe=probe, **common)[None, ...],
                self.diffraction.adj_probe(psi=psi[0], **common))
