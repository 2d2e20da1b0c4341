"""Object / probe illumination preconditioners
(reference src/tike/ptycho/solvers/_preconditioner.py:48-209).

Device work: ``tike_patch_adj`` with a single broadcast patch (K = 1) and
``tike_probe_preconditioner``.  Unlike the reference (whose all-reduce is
commented out, :185,201, because every GPU owns a spatial stripe) positions
are sharded across ranks here, so both preconditioners are summed over ranks.
"""
import torch

from ... import _arrays as A
from ..._lib import check, lib


def _psi_preconditioner(parameters, operator):
    """sum_s |probe_s|^2 scattered at every position (:48-104)."""
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    assert psi.shape[0] == 1, "single-slice objects only"
    out = torch.zeros_like(psi)
    probe_amp = torch.sum(probe * probe.conj(), dim=-3)[:, 0].contiguous()
    pw = probe.shape[-1]
    check(
        lib.tike_patch_adj(A.ptr(out), A.ptr(probe_amp), A.ptr(scan), 1,
                           psi.shape[-2], psi.shape[-1], scan.shape[0], 1, pw,
                           pw, 1, A.stream_ptr()), "psi preconditioner")
    return out


def _probe_preconditioner(parameters, operator):
    """sum_n |patch_n(psi)|^2 -> (D, pw, pw) complex (:116-167)."""
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    pw = probe.shape[-1]
    out = torch.zeros((psi.shape[0], pw, pw), dtype=probe.dtype,
                      device=probe.device)
    check(
        lib.tike_probe_preconditioner(A.ptr(scan), A.ptr(psi), A.ptr(out),
                                      scan.shape[0], pw, psi.shape[-2],
                                      psi.shape[-1], A.stream_ptr()),
        "probe preconditioner")
    return out


def update_preconditioners(comm, parameters, operator):
    """Refresh both preconditioners once per epoch (:170-209)."""
    if parameters.object_options:
        parameters.object_options.preconditioner = comm.Allreduce(
            _psi_preconditioner(parameters, operator))
    if parameters.probe_options:
        parameters.probe_options.preconditioner = comm.Allreduce(
            _probe_preconditioner(parameters, operator))
    return parameters
