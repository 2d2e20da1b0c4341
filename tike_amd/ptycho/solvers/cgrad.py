"""Conjugate-gradient ptychography solver.

The reference snapshot ships no ptychography ``cgrad``; it is
composed here, as BASELINE's configs ask, from the reference's own pieces:
``tike.opt.conjugate_gradient`` (opt.py:312-380: Dai-Yuan direction,
backtracking line search), the gaussian cost ``Ptycho.cost`` (ptycho.py:193-204)
and the gradient ``Ptycho.adj(gaussian_grad(...))`` (objective.py:31-44),
following the multi-GPU pattern of lamino/solvers/cgrad.py:58-92: the cost and
the gradient are summed over ranks, every rank then takes the same step.
"""
import numpy as np
import torch

from ... import _arrays as A
from ... import opt
from ..._lib import check, lib
from ...operators.propagation import fft_scales
from .lstsq import (NO_FARPLANE_SIZES, _workspace, chunk_positions,
                    global_count)


def _cost_and_grad(op, comm, data, psi, scan, probe, *, want_psi, want_probe,
                   want_grad):
    """Global gaussian cost (mean over all positions and pixels) and,
    optionally, d cost / d psi and d cost / d probe (unnormalised adjoints)."""
    dev = psi.device
    N = scan.shape[0]
    S, pw = probe.shape[-3], probe.shape[-1]
    det = op.detector_shape
    H, W = psi.shape[-2:]
    ws = _workspace(op)
    st = A.stream_ptr()
    fwd_scale, inv_scale = fft_scales(det, op.norm)
    # 256^2: the far-plane-free kernels of lstsq_grad serve here too -- a
    # line-search probe (cost only) then never writes the far plane at all
    lean = det in NO_FARPLANE_SIZES
    chunk = chunk_positions(S, det, lean)
    gscale = (ws.get("gscale", (min(chunk, max(N, 1)), det, det),
                     torch.float32, dev) if lean else None)
    far = ws.get("far", (min(chunk, max(N, 1)), 1, S, det, det),
                 torch.complex64, dev)
    mid = ws.get("mid", tuple(far.shape), torch.complex64, dev)
    chi_ws = mid if pw == det else ws.get(
        "chi", (min(chunk, max(N, 1)), 1, S, pw, pw), torch.complex64, dev)
    costs = ws.get("costs", (max(N, 1),), torch.float32, dev)
    gacc = (torch.zeros((2, H, W), dtype=torch.float32, device=dev)
            if (want_grad and want_psi) else None)
    gprobe = torch.zeros_like(probe) if (want_grad and want_probe) else None
    objproj = (ws.get("objproj", (min(chunk, max(N, 1)), pw, pw),
                      torch.complex64, dev) if gacc is not None else None)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        n = hi - lo
        chi = chi_ws
        if lean:
            # cost (and the -gradient factor) from the intensity in registers
            check(
                lib.tike_ptycho_fwd_gradient_scale(
                    A.ptr(psi), A.ptr(scan[lo:hi]), A.ptr(probe), 0, None,
                    None, 0, 0, A.ptr(far), None, None, A.ptr(data[lo:hi]), None,
                    A.ptr(gscale), A.ptr(costs[lo:hi]), n, S, pw, det, H, W,
                    fwd_scale, 0, 1.0, det * det, st), "cgrad cost")
            if not want_grad:
                continue
            check(
                lib.tike_grad_ifft2_crop(A.ptr(far), A.ptr(gscale), None, None,
                                         S, A.ptr(mid), A.ptr(chi), n * S, det,
                                         pw, fwd_scale, inv_scale, st),
                "cgrad gradient + ifft2")
        else:
            op.fwd_device(probe, scan[lo:hi], psi, out=far[:n])
            # gaussian cost per pattern; with the gradient requested the
            # farplane becomes -grad (sign flipped back below)
            check(
                lib.tike_farplane_gradient(A.ptr(far), A.ptr(data[lo:hi]),
                                           None, None, A.ptr(costs[lo:hi]), n,
                                           S, det, 0, int(want_grad), 1.0,
                                           det * det, st), "cgrad cost")
            if not want_grad:
                continue
            check(
                lib.tike_ifft2_crop(A.ptr(far), A.ptr(mid), A.ptr(chi), n * S,
                                    det, pw, inv_scale, st), "cgrad ifft2")
        check(
            lib.tike_lstsq_gradients(A.ptr(chi), A.ptr(scan[lo:hi]), A.ptr(psi),
                                     A.ptr(probe), None, None, 0, 0, None, None,
                                     A.ptr(gprobe), A.ptr(objproj), n, S, pw,
                                     H, W, st), "cgrad gradients")
        if gacc is not None:
            check(
                lib.tike_scatter_patches(A.ptr(objproj), A.ptr(scan[lo:hi]),
                                         A.ptr(gacc), n, pw, H, W, st),
                "cgrad object scatter")
    tot = comm.Allreduce_scalars([costs[:N].sum()], dev)
    cost = float((tot[0] / global_count(comm, op, 0, N)).item())
    grads = [t for t in (gacc, gprobe) if t is not None]
    if grads and comm.collective:
        comm.Allreduce(*grads)
    gpsi = None
    if gacc is not None:
        gpsi = -torch.complex(gacc[0], gacc[1])[None]
    if gprobe is not None:
        gprobe = -gprobe
    return cost, gpsi, gprobe


def cgrad(parameters, data, batches, comm, *, op, epoch):
    """One epoch: for every minibatch, `cg_iter` CG iterations on psi and
    then (when probe recovery is on) on the probe."""
    if data.dtype != torch.float32:
        data = data.to(torch.float32)  # 16-bit resident data: cgrad reads f32
    o = parameters.algorithm_options
    if parameters.eigen_probe is not None or parameters.eigen_weights is not None:
        raise NotImplementedError("cgrad does not support eigen probes")
    recover_psi = parameters.object_options is not None
    recover_probe = (parameters.probe_options is not None
                     and epoch >= parameters.probe_options.update_start)
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    batch_cost = []
    for b in batches:
        lo = int(b[0]) if len(b) else 0
        hi = lo + len(b)
        d, s = data[lo:hi], scan[lo:hi]
        cost = None
        if recover_psi:
            psi, cost = opt.conjugate_gradient(
                torch, x=psi,
                cost_function=lambda x: _cost_and_grad(
                    op, comm, d, x, s, probe, want_psi=True, want_probe=False,
                    want_grad=False)[0],
                grad=lambda x: [_cost_and_grad(
                    op, comm, d, x, s, probe, want_psi=True, want_probe=False,
                    want_grad=True)[1]],
                dir_multi=lambda x: x[0], num_iter=o.cg_iter,
                step_length=o.step_length)
        if recover_probe:
            probe, cost = opt.conjugate_gradient(
                torch, x=probe,
                cost_function=lambda x: _cost_and_grad(
                    op, comm, d, psi, s, x, want_psi=False, want_probe=True,
                    want_grad=False)[0],
                grad=lambda x: [_cost_and_grad(
                    op, comm, d, psi, s, x, want_psi=False, want_probe=True,
                    want_grad=True)[2]],
                dir_multi=lambda x: x[0], num_iter=o.cg_iter,
                step_length=o.step_length)
        if cost is None:
            cost = _cost_and_grad(op, comm, d, psi, s, probe, want_psi=False,
                                  want_probe=False, want_grad=False)[0]
        batch_cost.append(cost)
    o.costs.append([float(np.mean(batch_cost))])
    parameters.psi, parameters.probe = psi, probe
    return parameters
