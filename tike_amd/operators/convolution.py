"""2-D convolution with linear interpolation
(reference operators/cupy/convolution.py:11-154)."""
import numpy as np
import torch

from .. import _arrays as A
from .._lib import check, lib
from .operator import Operator
from .patch import Patch


class Convolution(Operator):
    """Product of probe and object patches at scan positions.

    psi (..., nz, n) complex64; probe (..., nscan|1, nprobe, pw, pw);
    nearplane (..., nscan, nprobe, det, det); scan (..., nscan, 2) float32.
    """

    def __init__(self, probe_shape, nz, n, ntheta=None, detector_shape=None,
                 **kwargs):
        det = probe_shape if detector_shape is None else detector_shape
        margin = (det - probe_shape) // 2  # probe window centred in detector
        vars(self).update(probe_shape=probe_shape, detector_shape=det, nz=nz,
                          n=n, pad=margin, end=margin + probe_shape,
                          patch=Patch())

    @staticmethod
    def _flat(x, keep):
        """Collapse leading dims to one; keep the last `keep` dims."""
        return x.reshape(-1, *x.shape[-keep:])

    def fwd(self, psi, scan, probe):
        kind = psi
        psi = A.to_device(psi, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        assert psi.shape[:-2] == scan.shape[:-2], (psi.shape, scan.shape)
        assert probe.shape[:-4] == scan.shape[:-2], (probe.shape, scan.shape)
        assert probe.shape[-4] == 1 or probe.shape[-4] == scan.shape[-2]
        N, S = scan.shape[-2], probe.shape[-3]
        det, pw = self.detector_shape, self.probe_shape
        out = torch.empty((*scan.shape[:-1], S, det, det),
                          dtype=torch.complex64, device=psi.device)
        psi_f, scan_f = self._flat(psi, 2), self._flat(scan, 2)
        probe_f, out_f = self._flat(probe, 4), self._flat(out, 4)
        for i in range(psi_f.shape[0]):
            check(
                lib.tike_conv_fwd(A.ptr(psi_f[i]), A.ptr(scan_f[i]),
                                  A.ptr(probe_f[i]),
                                  int(probe.shape[-4] != 1), A.ptr(
                                      out_f[i]), N, S, pw, det, psi.shape[-2],
                                  psi.shape[-1], A.stream_ptr()),
                "Convolution.fwd")
        return A.like_input(out, kind)

    def adj(self, nearplane, scan, probe, psi=None, overwrite=False):
        kind = nearplane
        nearplane = A.to_device(nearplane, np.complex64)
        scan = A.to_device(scan, np.float32)
        probe = A.to_device(probe, np.complex64)
        assert probe.shape[:-4] == scan.shape[:-2], (probe.shape, scan.shape)
        assert probe.shape[-4] == 1 or probe.shape[-4] == scan.shape[-2]
        assert nearplane.shape[:-3] == scan.shape[:-1], (nearplane.shape,
                                                         scan.shape)
        N, S = scan.shape[-2], nearplane.shape[-3]
        det, pw = self.detector_shape, self.probe_shape
        if psi is None:
            psi_t = torch.zeros((*scan.shape[:-2], self.nz, self.n),
                                dtype=torch.complex64, device=nearplane.device)
        else:
            psi_t = A.to_device(psi, np.complex64)
        assert psi_t.shape[:-2] == scan.shape[:-2]
        near_f, scan_f = self._flat(nearplane, 4), self._flat(scan, 2)
        probe_f, psi_f = self._flat(probe, 4), self._flat(psi_t, 2)
        for i in range(psi_f.shape[0]):
            check(
                lib.tike_conv_adj(A.ptr(near_f[i]), A.ptr(scan_f[i]),
                                  A.ptr(probe_f[i]),
                                  int(probe.shape[-4] != 1),
                                  A.ptr(psi_f[i]), N, S, pw, det,
                                  psi_t.shape[-2], psi_t.shape[-1],
                                  A.stream_ptr()), "Convolution.adj")
        if psi is not None and A.is_device(psi):
            if psi_t.data_ptr() != psi.data_ptr():
                psi.copy_(psi_t)
            return psi
        return A.like_input(psi_t, kind)

    def adj_probe(self, nearplane, scan, psi, overwrite=False):
        kind = nearplane
        nearplane = A.to_device(nearplane, np.complex64)
        scan = A.to_device(scan, np.float32)
        psi = A.to_device(psi, np.complex64)
        assert nearplane.shape[:-3] == scan.shape[:-1], (nearplane.shape,
                                                         scan.shape)
        assert psi.shape[:-2] == scan.shape[:-2], (psi.shape, scan.shape)
        N, S = scan.shape[-2], nearplane.shape[-3]
        det, pw = self.detector_shape, self.probe_shape
        out = torch.empty((*scan.shape[:-1], S, pw, pw), dtype=torch.complex64,
                          device=psi.device)
        near_f, scan_f = self._flat(nearplane, 4), self._flat(scan, 2)
        psi_f, out_f = self._flat(psi, 2), self._flat(out, 4)
        for i in range(psi_f.shape[0]):
            check(
                lib.tike_conv_adj_probe(A.ptr(near_f[i]), A.ptr(scan_f[i]),
                                        A.ptr(psi_f[i]), A.ptr(out_f[i]), N, S,
                                        pw, det, psi.shape[-2], psi.shape[-1],
                                        A.stream_ptr()),
                "Convolution.adj_probe")
        return A.like_input(out, kind)
