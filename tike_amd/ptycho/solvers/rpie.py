"""Regularised ptychographical iterative engine
(reference src/tike/ptycho/solvers/rpie.py:26-612, as that snapshot has it).

The gradients of a minibatch are the ones ``lstsq_grad`` needs --
``psi_update_numerator`` is its object gradient divided by the number of
modes (rpie.py:466) and ``probe_update_numerator`` its un-normalised probe
gradient -- so a single-slice object runs through the same fused HIP kernels
(position-major forward, far-plane-free gradient + inverse, pixel-major
pass 2 with both gradients, grouped footprint scatter).  A multislice object
(``len(psi) > 1``) goes slice by slice through ``Multislice`` /
``FresnelSpectProp`` (rpie.py:464-495).

Quirks of the reference kept on purpose, because the bar is its iterates:
the probe numerator is re-zeroed by every minibatch (rpie.py:349), so in
'compact' mode only the last minibatch moves the probe; it is only formed
when the object is recovered (:462); the probe step is ``alpha *
max(preconditioner)`` alone (:271-280); position correction is commented out
(:163-176, 521-563).
"""
import logging

import numpy as np
import torch

from ... import _tuning
from ... import _arrays as A
from ... import linalg
from ... import opt
from ... import random as trandom
from ..._lib import check, lib
from ...operators.multislice import fused_slices, next_incident_probe
from ...operators.propagation import fft_scales
from ..probe import get_varying_probe
from . import lstsq as L

logger = logging.getLogger(__name__)


def rpie(parameters, data, batches, comm, *, op, epoch):
    """Advance psi / probe / eigen weights by one epoch (rpie.py:26-214)."""
    o = parameters.algorithm_options
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    eigen_probe, eigen_weights = parameters.eigen_probe, parameters.eigen_weights
    exitwave_options = parameters.exitwave_options
    object_options = parameters.object_options
    probe_options = parameters.probe_options
    if exitwave_options.noise_model not in L._MODELS:
        raise ValueError(
            f"unknown noise model {exitwave_options.noise_model!r}")
    recover_probe = (probe_options is not None
                     and epoch >= probe_options.update_start)
    recover_psi = object_options is not None
    compact = o.batch_method == "compact"
    order = (range(o.num_batch) if compact else
             trandom.randomizer_np.permutation(o.num_batch))
    if hasattr(data, "hint"):  # patterns streamed from pinned host memory
        data.hint([(int(batches[b][0]), int(batches[b][0]) + len(batches[b]))
                   for b in order if len(batches[b])])
    dev = psi.device
    psi_num = probe_num = None
    batch_cost = torch.zeros(o.num_batch, dtype=torch.float32, device=dev)
    for n in order:
        lo = int(batches[n][0]) if len(batches[n]) else 0
        hi = lo + len(batches[n])
        comm.minibatch = int(n)
        if psi_num is None:
            psi_num = torch.zeros_like(psi)
        cost, probe_num = _get_nearplane_gradients(
            data, psi, scan, probe, eigen_probe, eigen_weights, lo, hi, comm,
            psi_num, op=op, exitwave_options=exitwave_options,
            recover_psi=recover_psi, recover_probe=recover_probe)
        batch_cost[n] = cost
        if not compact:
            psi, probe = _update(psi, probe, psi_num, probe_num,
                                 object_options, probe_options, recover_probe,
                                 o)
            psi_num = probe_num = None
    # one device->host scalar per epoch, as in the reference (rpie.py:161)
    o.costs.append([float(batch_cost.mean().item())])
    if compact:
        psi, probe = _update(psi, probe, psi_num, probe_num, object_options,
                             probe_options, recover_probe, o,
                             errors=[float(x[0]) for x in o.costs[-3:]])
    if eigen_weights is not None:
        # rpie.py:209-214 (mean over the positions of ALL ranks)
        sq = torch.sum(eigen_weights * eigen_weights, dim=-3, keepdim=True)
        n_local = torch.tensor(float(eigen_weights.shape[0]),
                               dtype=torch.float32, device=dev)
        if comm.collective:
            packed = torch.cat([sq.reshape(-1), n_local.reshape(1)])
            comm.Allreduce(packed)
            sq, n_local = packed[:-1].reshape(sq.shape), packed[-1]
        eigen_weights = eigen_weights / torch.sqrt(sq / n_local)
    parameters.psi, parameters.probe = psi, probe
    parameters.eigen_weights = eigen_weights
    return parameters


def _second_step(direction, options, errors):
    """The accelerated part of an update: checked momentum once epoch costs
    exist, ADAM before that.  Keeps the moments in `options`."""
    if errors:
        step, options.v, options.m = L._momentum_checked(
            g=direction, v=options.v, m=options.m, mdecay=options.mdecay,
            errors=errors, memory_length=3)
    else:
        step, options.v, options.m = opt.adam(
            g=direction, v=options.v, m=options.m, vdecay=options.vdecay,
            mdecay=options.mdecay)
    return step


def _peak(weight):
    return torch.amax(weight.real, dim=(-2, -1), keepdim=True)


def _update(psi, probe, psi_num, probe_num, object_options, probe_options,
            recover_probe, o, errors=None):
    """Apply the accumulated rPIE numerators (rpie.py:217-307).  Object:
    numerator over the alpha-regularised preconditioner
    (1 - alpha) P + alpha max P; probe: numerator of the first slice over
    alpha max P only, as the reference does.  With
    `use_adaptive_moment` the accelerated direction is applied as a second
    step on top (probe: main mode only, the other modes step twice)."""
    if object_options:
        weight = object_options.preconditioner
        damping = (1 - o.alpha) * weight + o.alpha * _peak(weight)
        psi = psi + psi_num / damping
        if object_options.use_adaptive_moment:
            psi = psi + _second_step(psi_num, object_options,
                                     errors) / damping
    if recover_probe:
        step = probe_num[0]  # (1, 1, S, pw, pw): the first slice's probe
        damping = o.alpha * _peak(probe_options.preconditioner[0])
        probe = probe + step / damping
        if probe_options.use_adaptive_moment:
            step[0, 0, 0] = _second_step(step[0, 0, 0], probe_options, errors)
            probe = probe + step / damping
    return psi, probe.contiguous()


def _get_nearplane_gradients(data, psi, scan, probe, eigen_probe,
                             eigen_weights, lo, hi, comm, psi_num, *, op,
                             exitwave_options, recover_psi, recover_probe):
    """Cost of the minibatch; psi_num (accumulated in place) and a fresh
    probe numerator (D, 1, 1, S, pw, pw) (rpie.py:310-548)."""
    dev = psi.device
    D = psi.shape[0]
    S = probe.shape[-3]
    probe_num = torch.zeros((D, *probe.shape), dtype=probe.dtype, device=dev)
    count = L.global_count(comm, op, lo, hi)
    if D == 1:
        # the lstsq_grad kernels: object gradient (planar accumulator), the
        # un-normalised probe gradient (num_batch = 1), costs, O_n, chi mode 0
        g = L._get_nearplane_gradients(
            data, psi, scan, probe, eigen_probe, eigen_weights, lo, hi, comm,
            num_batch=1, exitwave_options=exitwave_options, op=op,
            recover_psi=recover_psi, recover_probe=recover_psi)
        if recover_psi:
            psi_num += L.object_upd_sum(g) / S
            probe_num[0] = g["m_probe_update"]
        costs, chi0, chi_modes, patches = (g["costs"], g["chi0"],
                                           g["chi_modes"], g["patches"])
    else:
        costs, chi0, patches = _gradients_multislice(
            data, psi, scan, probe, eigen_probe, eigen_weights, lo, hi, comm,
            psi_num, probe_num, op=op, exitwave_options=exitwave_options,
            recover_psi=recover_psi)
        chi_modes = 1
    if recover_probe and eigen_weights is not None:
        # weights of the shared probe, mode 0 (rpie.py:505-519): the last two
        # sums of the step-statistics kernel, with no update directions
        B = hi - lo
        stats = torch.empty((max(B, 1), 8), dtype=torch.float32, device=dev)
        ep, w_old, C, Sm = L._eigen_args(eigen_probe, eigen_weights[lo:hi])
        check(
            lib.tike_lstsq_step_stats(
                A.ptr(chi0), A.ptr(scan[lo:hi]), A.ptr(psi[:1]), None,
                A.ptr(probe), A.ptr(ep), A.ptr(w_old), C, Sm, None, None,
                A.ptr(patches), A.ptr(stats), B, S, chi_modes,
                probe.shape[-1], psi.shape[-2], psi.shape[-1], None, None,
                A.stream_ptr()), "eigen weight sums")
        Cw = eigen_weights.shape[-2] - 1
        norms = torch.empty(max(Cw, 1), dtype=torch.float32, device=dev)
        check(
            lib.tike_eigen_weights0(A.ptr(eigen_weights[lo:hi]), A.ptr(stats),
                                    B, Cw, S, 0, A.ptr(norms),
                                    A.stream_ptr()), "eigen weights")
    tot = comm.Allreduce_scalars([costs.sum()], dev)
    return (tot[0] / count).to(torch.float32), probe_num


FUSED_MULTISLICE = True
"""128^2, 256^2 or 512^2 tiles with probe window = detector, at most 8 modes
and the object being recovered: a multislice minibatch runs on the two-pass
kernels (`_gradients_multislice_fused`).  False: the slice-by-slice
composition of the general operators, which remains the path of every other
configuration."""


SLICE_STEP_FUSED = _tuning.multislice_slice_step
"""The last pass of a Fresnel step and the first pass of the next slice's
transform in one launch (`tike_slice_step`)."""

FIRST_SLICE_STORED_PATCHES = _tuning.multislice_first_stored
"""The numerators of the first slice by `tike_ifft2_pass2_gradients` on the
patches pass 1 stored (shared probe only)."""

STEP_BACK_IN_FREQUENCY = _tuning.multislice_step_back
"""The steps back through the slices of the fused multislice path as extra
outputs of the last slice's gradient pass (see _gradients_multislice_fused)."""


def _fused_multislice_shapes(op, S, pw, exitwave_options, recover_psi, data):
    return (FUSED_MULTISLICE and recover_psi
            and fused_slices(pw, op.detector_shape, S)
            and isinstance(data, torch.Tensor))


def _gradients_multislice_fused(data, psi, scan, probe, eigen_probe,
                                eigen_weights, lo, hi, comm, psi_num,
                                probe_num, *, op, exitwave_options):
    """rpie.py:367-495 for an object of several slices on the fused kernels.

    Way forward, per slice d < D - 1: `tike_fwd_pass1` (patch of slice d x
    incident probe, formed on the fly, through the first pass of the
    transform) -> `tike_fresnel_colpass` (column pass x propagator -> inverse
    pass 1) -> `tike_fft2_pass2_inplace`: the probe incident on slice d + 1,
    kept for the way back -- or, SLICE_STEP_FUSED, that last pass together
    with pass 1 of slice d + 1 (`tike_slice_step`).  Last slice: `tike_fwd_pass1` ->
    `tike_fwd_grad_ifft2_pass1` (far field, cost, gradient factor and the
    inverse's first pass in one launch; the far plane is never stored).
    Way back: the reference hands `diff = propagation.adj(diff)` to the slice
    in front (:470); with diff = IFFT2(G) that is IFFT2(conj(H) FFT2(IFFT2(G)))
    = IFFT2(conj(H)^b G) -- the forward transform of every step back cancels
    (probe window = detector).  `tike_fwd_grad_ifft2_pass1_slices` therefore
    emits the inverse's first pass of conj(H)^b G for every slice while G is
    in its registers, and `tike_ifft2_pass2_products` of slice D - 1 - b
    finishes it and forms both numerators of the slice (object: through the
    grouped scatter).  STEP_BACK_IN_FREQUENCY = False runs the step as the
    reference writes it (`tike_fft2_pass1` -> `tike_fresnel_colpass` with the
    conjugated propagator).

    That one-launch last slice exists at 256^2 (512^2: without the steps back)
    for the gaussian model.  The Poisson model (its step lengths need the whole
    far field of a position first, exitwave.py:122-234) and 128^2 tiles store
    the far field instead: `tike_fft2_pass2_inplace` -> [`tike_intensity` ->
    `tike_poisson_steps`] -> `tike_farplane_gradient` [-> `tike_scale_modes`]
    -> `tike_fft2_pass1` (inverse), two more trips of the last slice's waves
    through memory, and the steps back as the reference writes them."""
    dev = psi.device
    B = hi - lo
    D = psi.shape[0]
    S, pw = probe.shape[-3], probe.shape[-1]
    det = op.detector_shape
    H, W = psi.shape[-2:]
    st = A.stream_ptr()
    fwd_scale, inv_scale = fft_scales(det, op.norm)
    nmeasured, mask_u8 = L.mask_info(exitwave_options, det)
    unmeasured = float(exitwave_options.unmeasured_pixels_scaling)
    ws = L._workspace(op)
    poisson = exitwave_options.noise_model == "poisson"
    # the last slice's far field in one launch, never stored
    one_launch = not poisson and det in (256, 512)
    step_back_in_frequency = (STEP_BACK_IN_FREQUENCY and one_launch
                              and det == 256)
    costs = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
    chi0 = torch.empty((max(B, 1), pw, pw), dtype=torch.complex64, device=dev)
    patches0 = (torch.empty_like(chi0) if eigen_weights is not None else None)
    # (a chunk holds far + mid + D - 1 sets of incident probes: 1.5 GiB per
    # 256 positions at 8 modes and two slices)
    # Measured (c3rpie2, same box): 64 / 128 / 256 / 512 / 1000 positions per
    # chunk -> 45.2 / 47.4 / 49.5 / 51.2 / 52.4 k patterns/s: these stages
    # hand nothing over through the Infinity Cache, longer launches win.
    # ... within HALF the HBM that is free right now (ranks that share a GPU,
    # smaller parts, a resident dataset), at most 16 GiB, at least 64 positions
    budget = 1 << 34
    if dev.type == "cuda":
        free = torch.cuda.mem_get_info(dev)[0]
        held = sum(t.numel() * t.element_size() for name, t in
                   getattr(ws, "buffers", {}).items() if name.startswith("ms_"))
        budget = min(budget, (free + held) // 2)
    chunk = (L.chunk_positions(S, det) if L.CHUNK_POSITIONS_OVERRIDE else
             max(64, budget // (2 * D * S * det * det * 8)))
    nmax = max(1, min(chunk, B))
    far = ws.get("ms_far", (nmax, S, det, det), torch.complex64, dev)
    nback = D if step_back_in_frequency else 1
    mids = ws.get("ms_mid", (nback, nmax, S, det, det), torch.complex64, dev)
    beams = ws.get("ms_beams", (max(D - 1, 1), nmax, S, pw, pw),
                   torch.complex64, dev)
    objproj = ws.get("ms_objproj", (nmax, pw, pw), torch.complex64, dev)
    acc = torch.zeros((D, 2, H, W), dtype=torch.float32, device=dev)
    pacc = torch.zeros((D, S, pw, pw), dtype=torch.complex64, device=dev)
    prop = op.diffraction.propagation._propagator((pw, pw), dev)
    u16 = int(data.dtype == torch.uint16)
    for clo in range(lo, hi, chunk):
        chi_hi = min(hi, clo + chunk)
        n = chi_hi - clo
        blo = clo - lo
        sc = scan[clo:chi_hi]
        w_c = None if eigen_weights is None else eigen_weights[clo:chi_hi]
        unique = get_varying_probe(probe, eigen_probe, w_c).contiguous()
        # the probe incident on slice d: (tensor, one per position?)
        incident = [(unique, int(unique.shape[0] != 1))]
        # a shared probe on the first slice: its numerators come from the
        # single-slice solver's kernel (`tike_ifft2_pass2_gradients`, which
        # reads the stored patches: 1.09 ms per 1000 positions against 1.59 for
        # `tike_ifft2_pass2_products`, which gathers them again)
        first_stored = (FIRST_SLICE_STORED_PATCHES and SLICE_STEP_FUSED
                        and incident[0][1] == 0)
        stored = patches0[blo:blo + n] if patches0 is not None else (
            ws.get("ms_patches", (nmax, pw, pw), torch.complex64, dev)[:n]
            if first_stored else None)
        if SLICE_STEP_FUSED:
            # pass 1 of slice 0, then per slice behind it: column passes of
            # the Fresnel step -> `tike_slice_step` (the step's last pass,
            # x the slice's patch, pass 1 of the next transform)
            check(
                lib.tike_fwd_pass1(
                    A.ptr(psi[0]), A.ptr(sc), A.ptr(unique), incident[0][1],
                    None, None, None, 0, 0, A.ptr(far), A.ptr(stored),
                    n, S, pw, det, H, W, st), "first slice, pass 1")
            for d in range(1, D):
                check(
                    lib.tike_fresnel_colpass(
                        A.ptr(far), A.ptr(prop), 0, A.ptr(beams[d - 1, :n]),
                        n * S, det, fwd_scale * inv_scale, st),
                    "Fresnel step: column passes")
                check(
                    lib.tike_slice_step(
                        A.ptr(beams[d - 1, :n]), A.ptr(psi[d]), A.ptr(sc),
                        A.ptr(far), n, S, det, H, W, 1.0, st),
                    "Fresnel step: last pass + next slice, pass 1")
                incident.append((beams[d - 1, :n], 1))
        else:
            for d in range(D - 1):
                nxt = next_incident_probe(
                    psi[d], sc, incident[d][0], far[:n], beams[d, :n], prop,
                    fwd_scale * inv_scale,
                    patches=patches0[blo:blo + n]
                    if d == 0 and patches0 is not None else None)
                incident.append((nxt, 1))
            beam, per = incident[D - 1]
            check(
                lib.tike_fwd_pass1(A.ptr(psi[D - 1]), A.ptr(sc), A.ptr(beam),
                                   per, None, None, None, 0, 0, A.ptr(far),
                                   None, n, S, pw, det, H, W, st),
                "last slice, pass 1")
        # (the outputs of a chunk packed: (nback, n, S, det, det))
        midv = mids.view(-1)[:nback * n * S * det * det].view(
            nback, n, S, det, det)
        if one_launch:
            check(
                lib.tike_fwd_grad_ifft2_pass1_slices(
                    A.ptr(far), A.ptr(data[clo:chi_hi]), u16, A.ptr(mask_u8),
                    A.ptr(costs[blo:blo + n]), A.ptr(midv), n, S, det,
                    fwd_scale, 0, unmeasured, nmeasured, A.ptr(prop), nback,
                    st),
                "far field + gradient + inverse pass 1 (every slice)")
        else:
            _stored_farplane_gradient(
                far[:n], midv[0], data, clo, chi_hi, mask_u8,
                costs[blo:blo + n], fwd_scale, unmeasured, nmeasured,
                exitwave_options)
        mid = midv[0]
        for tt in range(D - 1, -1, -1):
            beam, per = incident[tt]
            if step_back_in_frequency:
                mid = midv[D - 1 - tt]
            if tt == 0 and first_stored:
                check(
                    lib.tike_ifft2_pass2_gradients(
                        A.ptr(mid), A.ptr(stored), A.ptr(beam), None, None, 0,
                        0, A.ptr(objproj), A.ptr(chi0[blo:blo + n]),
                        A.ptr(pacc[0]), 1.0, n, S, det, inv_scale, st),
                    "inverse pass 2 + numerators (first slice)")
                check(
                    lib.tike_scatter_patches(A.ptr(objproj), A.ptr(sc),
                                             A.ptr(acc[0]), n, pw, H, W, st),
                    "object numerator")
                continue
            check(
                lib.tike_ifft2_pass2_products(
                    A.ptr(mid), A.ptr(psi[tt]), A.ptr(sc), A.ptr(beam), per,
                    A.ptr(objproj), A.ptr(pacc[tt]), 1.0,
                    A.ptr(chi0[blo:blo + n]) if tt == 0 else None,
                    int(tt > 0 and not step_back_in_frequency), n, S, det, H,
                    W, inv_scale, st),
                "inverse pass 2 + numerators")
            check(
                lib.tike_scatter_patches(A.ptr(objproj), A.ptr(sc),
                                         A.ptr(acc[tt]), n, pw, H, W, st),
                "object numerator")
            if tt == 0 or step_back_in_frequency:
                continue
            check(lib.tike_fft2_pass1(A.ptr(mid), A.ptr(far), n * S, det, 0,
                                      st), "Fresnel step back: pass 1")
            # (the inverse transform's normalisation is applied by the pass
            # 2 that follows: tike_ifft2_pass2_products)
            check(
                lib.tike_fresnel_colpass(A.ptr(far), A.ptr(prop), 1,
                                         A.ptr(mid), n * S, det, fwd_scale,
                                         st),
                "Fresnel step back: column passes")
    if comm.collective:
        comm.Allreduce(acc, pacc)
    psi_num += torch.complex(acc[:, 0], acc[:, 1]) / S
    probe_num[:, 0, 0] = pacc
    return costs[:B], chi0[:B], None if patches0 is None else patches0[:B]


def _stored_farplane_gradient(far, mid, data, lo, hi, mask_u8, costs,
                              fwd_scale, unmeasured, nmeasured,
                              exitwave_options):
    """The last slice with its far field stored: `far` (n, S, det, det), the
    hand-off of `tike_fwd_pass1`, becomes the far field, then its gradient
    (rpie.py:420-442; the Poisson model's per-mode step lengths,
    exitwave.py:122-234), and `mid` receives the first pass of the inverse
    transform."""
    n, S, det = far.shape[0], far.shape[1], far.shape[-1]
    st = A.stream_ptr()
    poisson = exitwave_options.noise_model == "poisson"
    dchunk = A.data_f32(data, lo, hi)
    check(lib.tike_fft2_pass2_inplace(A.ptr(far), n * S, det, 0, fwd_scale, st),
          "last slice: far field")
    if poisson:
        inten = torch.empty((n, det, det), dtype=torch.float32,
                            device=far.device)
        steps = torch.empty((n, S), dtype=torch.float32, device=far.device)
        check(lib.tike_intensity(A.ptr(far), A.ptr(inten), n, S, det * det,
                                 st), "intensity")
        check(
            lib.tike_poisson_steps(
                A.ptr(far), A.ptr(inten), A.ptr(dchunk), A.ptr(mask_u8),
                A.ptr(steps), n, S, det,
                float(exitwave_options.step_length_start),
                float(exitwave_options.step_length_weight),
                int(exitwave_options.step_length_usemodes == "dominant_mode"),
                st), "poisson step lengths")
    check(
        lib.tike_farplane_gradient(
            A.ptr(far), A.ptr(dchunk), A.ptr(mask_u8), None, A.ptr(costs), n,
            S, det, L._MODELS[exitwave_options.noise_model], 1, unmeasured,
            nmeasured, st), "farplane gradient")
    if poisson:
        check(lib.tike_scale_modes(A.ptr(far), A.ptr(steps), A.ptr(mask_u8),
                                   n * S, det, st), "poisson step scaling")
    check(lib.tike_fft2_pass1(A.ptr(far), A.ptr(mid), n * S, det, 1, st),
          "last slice: inverse pass 1")


def _gradients_multislice(data, psi, scan, probe, eigen_probe, eigen_weights,
                          lo, hi, comm, psi_num, probe_num, *, op,
                          exitwave_options, recover_psi):
    """rpie.py:367-495 for an object of several slices, slice by slice."""
    if _fused_multislice_shapes(op, probe.shape[-3], probe.shape[-1],
                                exitwave_options, recover_psi, data):
        return _gradients_multislice_fused(
            data, psi, scan, probe, eigen_probe, eigen_weights, lo, hi, comm,
            psi_num, probe_num, op=op, exitwave_options=exitwave_options)
    dev = psi.device
    B = hi - lo
    D = psi.shape[0]
    S, pw = probe.shape[-3], probe.shape[-1]
    det = op.detector_shape
    H, W = psi.shape[-2:]
    st = A.stream_ptr()
    _, inv_scale = fft_scales(det, op.norm)
    nmeasured, mask_u8 = L.mask_info(exitwave_options, det)
    model = L._MODELS[exitwave_options.noise_model]
    poisson = exitwave_options.noise_model == "poisson"
    unmeasured = float(exitwave_options.unmeasured_pixels_scaling)
    costs = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
    chi0 = torch.empty((max(B, 1), pw, pw), dtype=torch.complex64, device=dev)
    patches0 = torch.empty_like(chi0)
    chunk = L.chunk_positions(S * D, det)
    acc = torch.zeros_like(psi) if recover_psi else None
    pacc = (torch.zeros((D, S, pw, pw), dtype=torch.complex64, device=dev)
            if recover_psi else None)
    for clo in range(lo, hi, chunk):
        chi_hi = min(hi, clo + chunk)
        n = chi_hi - clo
        blo = clo - lo
        sc = scan[clo:chi_hi]
        w_c = None if eigen_weights is None else eigen_weights[clo:chi_hi]
        unique = get_varying_probe(probe, eigen_probe, w_c)
        far, probes = op.fwd_return_intermediate_probes(probe=unique, scan=sc,
                                                        psi=psi)
        far = far.contiguous()
        dchunk = A.data_f32(data, clo, chi_hi)
        if poisson:
            inten = torch.empty((n, det, det), dtype=torch.float32,
                                device=dev)
            steps = torch.empty((n, S), dtype=torch.float32, device=dev)
            check(lib.tike_intensity(A.ptr(far), A.ptr(inten), n, S,
                                     det * det, st), "intensity")
            check(
                lib.tike_poisson_steps(
                    A.ptr(far), A.ptr(inten), A.ptr(dchunk),
                    A.ptr(mask_u8), A.ptr(steps), n, S, det,
                    float(exitwave_options.step_length_start),
                    float(exitwave_options.step_length_weight),
                    int(exitwave_options.step_length_usemodes ==
                        "dominant_mode"), st), "poisson step lengths")
        check(
            lib.tike_farplane_gradient(
                A.ptr(far), A.ptr(dchunk), A.ptr(mask_u8), None,
                A.ptr(costs[blo:blo + n]), n, S, det, model, 1, unmeasured,
                nmeasured, st), "farplane gradient")
        if poisson:
            check(lib.tike_scale_modes(A.ptr(far), A.ptr(steps),
                                       A.ptr(mask_u8), n * S, det, st),
                  "poisson step scaling")
        # diff = propagation.adj(farplane)[..., pad:end, pad:end]; a
        # multislice object has pad = 0
        diff = op.propagation.adj(far, overwrite=True)[:, 0].contiguous()
        # (rpie.py:444-472: the walk back through the slices happens only
        # when the object is recovered; otherwise `diff` stays the exit-wave
        # update of the LAST slice, and that is what the eigen weights see)
        for tt in range(D - 1, -1, -1) if recover_psi else ():
            # psi numerator: sum_s conj(probe_tt) diff scattered (1/S below)
            check(
                lib.tike_conv_adj(A.ptr(diff), A.ptr(sc),
                                  A.ptr(probes[tt].contiguous()), 1,
                                  A.ptr(acc[tt]), n, S, pw, pw, H, W, st),
                "object numerator")
            # probe numerator: sum_n conj(patch_n(psi_tt)) diff
            check(
                lib.tike_probe_grad(
                    A.ptr(diff), A.ptr(sc), A.ptr(psi[tt]),
                    A.ptr(patches0[blo:blo + n]) if tt == 0 else None,
                    A.ptr(pacc[tt]), n, S, pw, H, W, st),
                "probe numerator")
            if tt == 0:
                break
            diff = op.diffraction.propagation.adj(diff)
        if not recover_psi:
            patches0[blo:blo + n] = op.diffraction.patch.fwd(
                images=psi[0], positions=sc, patch_width=pw)
        chi0[blo:blo + n] = diff[:, 0]
    if recover_psi:
        if comm.collective:
            comm.Allreduce(acc, pacc)
        psi_num += acc / S
        probe_num[:, 0, 0] = pacc
    return costs[:B], chi0[:B], patches0[:B]
