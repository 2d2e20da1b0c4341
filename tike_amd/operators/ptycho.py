"""The ptychography operator (reference operators/cupy/ptycho.py:26-204).

``fwd`` and ``adj`` keep the reference's signatures and shapes but run as
fused HIP kernels: patch gather * probe -> FFT2 (``tike_ptycho_fwd``) and,
for the adjoint, the two-pass inverse transform whose second pass forms both
products in place (``tike_ptycho_adj``; other shapes: ``tike_ifft2_crop``
followed by the unfused object scatter-add and probe product).
"""

import numpy as np
import torch

from .. import _tuning
from .. import _arrays as A
from .._lib import check, lib
from . import objective
from .multislice import Multislice
from .operator import Operator
from .propagation import Propagation, fft_scales


def sub_batch_positions(S, det, mib=None):
    """Positions per sub-batch of the two-kernel 256^2 / 512^2 operators: the
    ``sub_batch`` argument of ``tike_ptycho_fwd`` / ``tike_ptycho_adj``
    (0 = the library's default -- 256 MiB of far plane for the forward
    operator, one batch for the adjoint --, -1 = one batch).
    ``TIKE_FWD_SUB_MIB`` (`tike_amd._tuning`) is read on the host side of
    the C ABI, never inside the library."""
    if mib is None:
        mib = _tuning.fwd_sub_mib
    if mib is None:
        return 0
    mib = float(mib)
    if mib <= 0:
        return -1
    return max(1, int(mib * 2**20) // (8 * S * det * det))


def _positions_allowed(scan, psi_shape, pw):
    """check_allowed_positions as a predicate (position.py:600-628): every
    patch and its +1 taps inside the object.  Host arrays are tested on the
    host; a device tensor costs one small reduction and a read-back."""
    hi_y, hi_x = psi_shape[-2] - pw - 1, psi_shape[-1] - pw - 1
    if isinstance(scan, torch.Tensor):
        c = torch.floor(scan)
        lo = c.amin(dim=0)
        hi = c.amax(dim=0)
        ok = (lo >= 1).all() & (hi[0] <= hi_y) & (hi[1] <= hi_x)
        return bool(ok.item())
    c = np.floor(scan)
    return bool(c.min() >= 1 and c[..., 0].max() <= hi_y
                and c[..., 1].max() <= hi_x)


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
                   eigen_weights=None, out=None, sub_batch=None):
        """Fused forward on device tensors; optional on-the-fly eigen probes.

        probe (1|N,1,S,pw,pw); eigen_probe (1,C,Sm,pw,pw) and eigen_weights
        (N,C+1,S) select the varying probe of probe.py:272-303.  sub_batch:
        positions per sub-batch at 256^2 / 512^2 (None: sub_batch_positions).
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
                psi.shape[-2], psi.shape[-1], fft_scales(det, self.norm)[0],
                sub_batch_positions(S, det) if sub_batch is None else
                int(sub_batch), A.stream_ptr()), "Ptycho.fwd")
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

    def fused_adjoint_shapes(self, S):
        """Shapes the fused adjoint (``tike_ptycho_adj``) takes: probe window =
        detector in {128, 256, 512}, at most 8 modes."""
        return (self.probe_shape == self.detector_shape
                and self.detector_shape in (128, 256, 512) and S <= 8)

    def adj_device(self, farplane, probe, scan, psi, psi_adj=None,
                   probe_adj=None, sub_batch=None):
        """Fused adjoint on device tensors (``tike_ptycho_adj``): inverse pass 1
        into ``probe_adj``, pass 2 in place with both products, grouped
        footprint scatter.  Requires ``fused_adjoint_shapes`` and scan positions
        that satisfy ``check_allowed_positions`` (position.py:600-628: what the
        solvers guarantee; ``adj`` checks it).  ``farplane`` is not modified.
        """
        Multislice._one_slice(psi)
        N, S = scan.shape[0], farplane.shape[-3]
        pw, det = self.probe_shape, self.detector_shape
        assert tuple(farplane.shape) == (N, 1, S, det, det), farplane.shape
        assert probe.shape[0] in (1, N) and probe.shape[-3] == S
        H, W = psi.shape[-2:]
        dev = psi.device
        if probe_adj is None:
            probe_adj = torch.empty((N, 1, S, pw, pw), dtype=torch.complex64,
                                    device=dev)
        if psi_adj is None:
            psi_adj = torch.empty_like(psi)
        objproj = torch.empty((max(N, 1), pw, pw), dtype=torch.complex64,
                              device=dev)
        acc = torch.empty((2, H, W), dtype=torch.float32, device=dev)
        check(
            lib.tike_ptycho_adj(
                A.ptr(farplane), A.ptr(probe), int(probe.shape[0] != 1),
                A.ptr(scan), A.ptr(psi), A.ptr(psi_adj), A.ptr(probe_adj),
                A.ptr(objproj), A.ptr(acc), N, S, pw, det, H, W,
                fft_scales(det, self.norm)[1],
                sub_batch_positions(S, det) if sub_batch is None else
                int(sub_batch), A.stream_ptr()), "Ptycho.adj")
        return psi_adj, probe_adj

    def adj(self, farplane, probe, scan, psi, overwrite=False, **kwargs):
        kind = farplane
        farplane_in = farplane
        host_scan = None if A.is_device(scan) else np.asarray(scan)
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
        if N > 0 and self.fused_adjoint_shapes(S) and _positions_allowed(
                scan if host_scan is None else host_scan, psi.shape, pw):
            psi_adj, probe_adj = self.adj_device(farplane, probe, scan, psi)
            return A.like_input(psi_adj, kind), A.like_input(probe_adj, kind)
        # general shapes (probe window narrower than the detector, other
        # sizes, more than 8 modes) and positions whose taps leave the image
        # (the reference's linear tap addressing, convolution.cu:113-133):
        # IFFT2 + crop, then the unfused Convolution adjoints
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
