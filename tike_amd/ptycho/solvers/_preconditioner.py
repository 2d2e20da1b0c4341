"""Object / probe illumination preconditioners
(reference src/tike/ptycho/solvers/_preconditioner.py:48-209).

Device work: ``tike_psi_preconditioner`` (one atomic per object pixel and
position) and ``tike_probe_preconditioner``.  Unlike the reference (whose all-reduce is
commented out, :185,201, because every GPU owns a spatial stripe) positions
are sharded across ranks here, so both preconditioners are summed over ranks.
"""

import torch

from ... import _tuning
from ... import _arrays as A
from ..._lib import check, lib
from ...operators.multislice import fused_slices, next_incident_probe
from ...operators.propagation import fft_scales


MULTISLICE_CHUNK = _tuning.precond_chunk
"""Positions per launch of the fused multislice object preconditioner
(c3rpie2 at 64 / 128 / 256 / 512: 49.1 / 49.5 / 50.2 / 50.6 k patterns/s)."""


def _psi_preconditioner(parameters, operator):
    """sum_s |probe_s|^2 scattered at every position (:48-104)."""
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    if psi.shape[0] > 1:
        return _psi_preconditioner_multislice(parameters, operator)
    out = torch.zeros(tuple(psi.shape), dtype=torch.float32, device=psi.device)
    pw = probe.shape[-1]
    # sum_s |probe_s|^2 (probe-sized; _preconditioner.py:40-45)
    probe_amp = torch.sum(torch.square(probe.abs()), dim=-3)[0, 0].contiguous()
    check(
        lib.tike_psi_preconditioner(A.ptr(probe_amp), A.ptr(scan), A.ptr(out),
                                    scan.shape[0], pw, psi.shape[-2],
                                    psi.shape[-1], A.stream_ptr()),
        "psi preconditioner")
    return out


def _psi_preconditioner_multislice(parameters, operator):
    """Several object slices: slice i sees the probe propagated through the
    slices in front of it (:82-95), so its illumination differs per position
    (Patch.adj of one patch per position)."""
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    pw = probe.shape[-1]
    D, H, W = psi.shape
    N = scan.shape[0]
    st = A.stream_ptr()
    out = torch.zeros((D, H, W), dtype=torch.float32, device=psi.device)
    probe1 = probe[:, 0]  # (1, S, pw, pw)
    # first slice: the shared probe's amplitude at every position
    amp = torch.sum(torch.square(probe1.abs()), dim=-3)[0].contiguous()
    check(
        lib.tike_psi_preconditioner(A.ptr(amp), A.ptr(scan), A.ptr(out[0]), N,
                                    pw, H, W, st), "psi preconditioner")
    S = probe.shape[-3]
    if fused_slices(pw, operator.detector_shape, S):
        # slice by slice on the two-pass kernels, a chunk of positions at a
        # time: the incident probes of all N positions never exist at once
        fwd_scale, inv_scale = fft_scales(pw, operator.norm)
        prop = operator.diffraction.propagation._propagator((pw, pw),
                                                            psi.device)
        chunk = max(1, min(N, MULTISLICE_CHUNK))
        bufs = [torch.empty((chunk, S, pw, pw), dtype=torch.complex64,
                            device=psi.device) for _ in range(3)]
        amp = torch.empty((chunk, pw, pw), dtype=torch.float32,
                          device=psi.device)
        for lo in range(0, N, chunk):
            n = min(chunk, N - lo)
            sc = scan[lo:lo + n]
            beam = probe1
            for i in range(1, D):
                # (the illumination comes out of the step's last pass; the
                # wave itself is only written when another slice follows)
                beam = next_incident_probe(
                    psi[i - 1], sc, beam, bufs[2][:n], bufs[i % 2][:n], prop,
                    fwd_scale * inv_scale, amplitude=amp, keep=i + 1 < D)
                check(
                    lib.tike_scatter_amplitudes(A.ptr(amp), A.ptr(sc),
                                                A.ptr(out[i]), n, pw, H, W,
                                                st),
                    "psi preconditioner (slice)")
        return out
    acc = torch.empty((2, H, W), dtype=torch.float32, device=psi.device)
    for i in range(1, D):
        probe1 = operator.diffraction.propagation.fwd(
            operator.diffraction.diffraction.fwd(probe=probe1, scan=scan,
                                                 psi=psi[i - 1]))
        # one amplitude patch per position: the grouped footprint scatter
        # (8 neighbouring positions summed on chip, one atomic per box pixel:
        # 0.2 ms per 1000 positions) instead of Patch.adj's atomic per pixel
        # and position (7.7 ms per 2000 positions at 256^2)
        amp = torch.sum(probe1 * probe1.conj(), dim=-3).contiguous()
        acc.zero_()
        check(
            lib.tike_scatter_patches(A.ptr(amp), A.ptr(scan), A.ptr(acc), N,
                                     pw, H, W, st),
            "psi preconditioner (slice)")
        out[i] = acc[0]
    return out


def _probe_preconditioner(parameters, operator):
    """sum_n |patch_n(psi)|^2 -> (D, pw, pw) complex (:116-167)."""
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    pw = probe.shape[-1]
    out = torch.zeros((psi.shape[0], pw, pw), dtype=probe.dtype,
                      device=probe.device)
    for i in range(psi.shape[0]):  # every slice (:136-144)
        check(
            lib.tike_probe_preconditioner(A.ptr(scan), A.ptr(psi[i]),
                                          A.ptr(out[i]), scan.shape[0], pw,
                                          psi.shape[-2], psi.shape[-1],
                                          A.stream_ptr()),
            "probe preconditioner")
    return out


def update_preconditioners(comm, parameters, operator, *, psi=True,
                           probe=True):
    """Refresh both preconditioners once per epoch (:170-209).  `psi` /
    `probe` = False skips one (cgrad reads neither while it iterates).  With
    several ranks the probe preconditioner's sum travels while the object
    preconditioner's kernel runs."""
    probe_sum = probe_pre = None
    if parameters.probe_options and probe:
        probe_pre = _probe_preconditioner(parameters, operator)
        probe_sum = comm.Allreduce_start(
            torch.view_as_real(probe_pre).reshape(-1))
    if parameters.object_options and psi:
        # accumulated as float32 (real-valued), stored complex64 like the
        # reference's array (object.py:69-72)
        parameters.object_options.preconditioner = comm.Allreduce(
            _psi_preconditioner(parameters, operator)).to(torch.complex64)
    if probe_pre is not None:
        probe_sum.wait()
        parameters.probe_options.preconditioner = probe_pre
    return parameters
