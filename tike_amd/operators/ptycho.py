"""The ptychography operator (reference operators/cupy/ptycho.py:26-204).

``fwd`` and ``adj`` keep the reference's signatures and shapes but run as
fused HIP kernels: patch gather * probe -> FFT2 in one launch
(``tike_ptycho_fwd``), IFFT2 -> crop (``tike_ifft2_crop``) followed by the
object scatter-add and the probe gradient.
"""
import numpy as np
import torch

from .. import _arrays as A
from .._lib import check, lib
from . import objective
from .multislice import Multislice
from .operator import Operator
from .propagation import Propagation, fft_scales


class Ptycho(Operator):
    """Compose diffraction and far-field propagation.

    farplane (POSI, 1, SHARED, det, det) complex64;
    probe (1|POSI, 1, SHARED, pw, pw) complex64; psi (DEPTH, H, W) complex64;
    scan (POSI, 2) float32; data (FRAME, det, det) float32.

    ``probe_wavelength``, ``probe_FOV_lengths`` and
    ``multislice_propagation_distance`` are only used by multi-slice objects
    and default to the values ``reconstruct`` passes for one slice.
    """

    def __init__(self, detector_shape, probe_shape,
                 probe_wavelength=float("nan"),
                 probe_FOV_lengths=(float("nan"), float("nan")), nz=None,
                 n=None, multislice_propagation_distance=1e-9,
                 propagation=Propagation, diffraction=Multislice, norm="ortho",
                 **kwargs):
        if nz is None or n is None:
            raise TypeError("Ptycho requires nz and n (object height, width)")
        shapes = dict(probe_shape=probe_shape, detector_shape=detector_shape,
                      nz=nz, n=n)
        optics = dict(
            probe_wavelength=probe_wavelength,
            probe_FOV_lengths=probe_FOV_lengths,
            multislice_propagation_distance=multislice_propagation_distance)
        vars(self).update(shapes, norm=norm, **optics)
        self.diffraction = diffraction(**shapes, **optics, **kwargs)
        self.propagation = propagation(detector_shape=detector_shape,
                                       norm=norm, **kwargs)

    def __enter__(self):
        self.propagation.__enter__()
        self.diffraction.__enter__()
        return self

    def __exit__(self, type, value, traceback):
        self.propagation.__exit__(type, value, traceback)
        self.diffraction.__exit__(type, value, traceback)

    # -- device-level entry points used by the solvers ---------------------
    def fwd_device(self, probe, scan, psi, eigen_probe=None,
                   eigen_weights=None, out=None):
        """Fused forward on device tensors; optional on-the-fly eigen probes.

        probe (1|N,1,S,pw,pw); eigen_probe (1,C,Sm,pw,pw) and eigen_weights
        (N,C+1,S) select the varying probe of probe.py:272-303.
        """
        Multislice._one_slice(psi)
        N = scan.shape[0]
        S, pw, det = probe.shape[-3], self.probe_shape, self.detector_shape
        assert probe.shape[-1] == pw and probe.shape[-2] == pw
        assert probe.shape[0] in (1, N)
        if out is None:
            out = torch.empty((N, 1, S, det, det), dtype=torch.complex64,
                              device=psi.device)
        C = Sm = 0
        if eigen_weights is not None:
            assert probe.shape[0] == 1
            assert eigen_weights.shape[0] == N and eigen_weights.shape[2] == S
            if eigen_probe is not None:
                C, Sm = eigen_probe.shape[-4], eigen_probe.shape[-3]
            assert eigen_weights.shape[1] >= C + 1
            if eigen_weights.shape[1] != C + 1:
                eigen_weights = eigen_weights[:, :C + 1].contiguous()
        check(
            lib.tike_ptycho_fwd(
                A.ptr(psi), A.ptr(scan), A.ptr(probe),
                int(probe.shape[0] != 1), A.ptr(eigen_probe),
                A.ptr(eigen_weights), C, Sm, A.ptr(out), N, S, pw, det,
                psi.shape[-2], psi.shape[-1],
                fft_scales(det, self.norm)[0], A.stream_ptr()), "Ptycho.fwd")
        return out

    def fwd(self, probe, scan, psi, **kwargs):
        kind = psi
        psi = A.to_device(psi, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        assert probe.ndim == 5 and probe.shape[1] == 1, probe.shape
        if psi.shape[0] > 1:
            # several object slices (ptycho.py:114-129 over multislice.py:
            # 69-92): slice-by-slice composition, then the far-field transform
            far = self.propagation.fwd(self.diffraction.fwd(
                psi=psi, scan=scan, probe=probe[..., 0, :, :, :]),
                                       overwrite=True)[..., None, :, :, :]
            return A.like_input(far, kind)
        return A.like_input(self.fwd_device(probe, scan, psi), kind)

    def fwd_return_intermediate_probes(self, probe, scan, psi, **kwargs):
        """(farplane, probes incident on every slice) -- ptycho.py:131-146."""
        on_device = (A.to_device(psi, np.complex64),
                     A.to_device(scan, np.float32),
                     A.to_device(probe, np.complex64))
        exitwave, beams = self.diffraction.fwd_return_intermediate_probes(
            psi=on_device[0], scan=on_device[1], probe=on_device[2])
        far = self.propagation.fwd(nearplane=exitwave, overwrite=True)
        return (A.like_input(far.unsqueeze(-4), psi),
                A.like_input(beams, psi))

    def adj(self, farplane, probe, scan, psi, overwrite=False, **kwargs):
        kind = farplane
        farplane_in = farplane
        farplane = A.to_device(farplane, np.complex64)
        psi = A.to_device(psi, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        if psi.shape[0] > 1:  # ptycho.py:148-176 over multislice.py:144-194
            psi_adj, probe_adj = self.diffraction.adj(
                nearplane=self.propagation.adj(
                    farplane, overwrite=False)[..., 0, :, :, :],
                probe=probe[..., 0, :, :, :], scan=scan, psi=psi)
            return (A.like_input(psi_adj, kind),
                    A.like_input(probe_adj[..., None, :, :, :], kind))
        N, S = scan.shape[0], farplane.shape[-3]
        pw, det = self.probe_shape, self.detector_shape
        assert tuple(farplane.shape) == (N, 1, S, det, det), farplane.shape
        assert probe.shape[0] in (1, N) and probe.shape[-3] == S
        in_place = (overwrite and A.is_device(farplane_in)
                    and farplane.data_ptr() == farplane_in.data_ptr())
        work = farplane if in_place else torch.empty_like(farplane)
        chi = work if pw == det else torch.empty(
            (N, 1, S, pw, pw), dtype=torch.complex64, device=psi.device)
        check(
            lib.tike_ifft2_crop(A.ptr(farplane), A.ptr(work), A.ptr(chi),
                                N * S, det, pw,
                                fft_scales(det, self.norm)[1],
                                A.stream_ptr()), "Ptycho.adj (ifft2)")
        psi_adj = torch.zeros_like(psi)
        check(
            lib.tike_conv_adj(A.ptr(chi), A.ptr(scan), A.ptr(probe),
                              int(probe.shape[0] != 1), A.ptr(psi_adj), N, S,
                              pw, pw, psi.shape[-2], psi.shape[-1],
                              A.stream_ptr()), "Ptycho.adj (object)")
        probe_adj = torch.empty((N, 1, S, pw, pw), dtype=torch.complex64,
                                device=psi.device)
        check(
            lib.tike_conv_adj_probe(A.ptr(chi), A.ptr(scan), A.ptr(psi),
                                    A.ptr(probe_adj), N, S, pw, pw,
                                    psi.shape[-2], psi.shape[-1],
                                    A.stream_ptr()), "Ptycho.adj (probe)")
        return A.like_input(psi_adj, kind), A.like_input(probe_adj, kind)

    def _compute_intensity(self, data, psi, scan, probe):
        """(intensity (N,det,det), farplane) -- ptycho.py:178-191."""
        kind = psi
        far = self.fwd_device(A.to_device(probe, np.complex64),
                              A.to_device(scan, np.float32),
                              A.to_device(psi, np.complex64))
        N, _, S, det, _ = far.shape
        intensity = torch.empty((N, det, det), dtype=torch.float32,
                                device=far.device)
        check(
            lib.tike_intensity(A.ptr(far), A.ptr(intensity), N, S, det * det,
                               A.stream_ptr()), "Ptycho._compute_intensity")
        return A.like_input(intensity, kind), A.like_input(far, kind)

    def cost(self, data, psi, scan, probe, *, model):
        """ptycho.py:193-204."""
        intensity, _ = self._compute_intensity(data, psi, scan, probe)
        return getattr(objective, model)(data, intensity)
