"""Least-squares + gradient solver (Odstrcil 2018 LSQ-ML)
(reference src/tike/ptycho/solvers/lstsq.py:25-858).

One call = one epoch over this rank's share of every minibatch.  All
per-position x per-pixel work runs in the HIP kernels of ``tike_amd/csrc``
(fused forward, far-plane gradient, IFFT2+crop, object scatter, probe
gradient, step-size normal equations); the arrays left to torch are psi-,
probe- and (positions,)-sized.

Differences from the reference that do not change the mathematics:
  * the per-position ("unique") probe is never materialised: kernels
    synthesise ``w0*probe + sum_c w_c*eigen_c`` on the fly;
  * a minibatch is processed in chunks sized for the GPU, and only mode 0 of
    the exit-wave update ``chi`` survives a chunk (the step-size and
    eigen-probe passes need nothing else);
  * positions are sharded over ranks: every sum / mean over positions is
    completed by an RCCL all-reduce (``comm``), so P GPUs follow the 1-GPU
    iterates up to summation order.
"""
import logging
import os

import numpy as np
import torch

from ... import _arrays as A
from ... import linalg
from ... import opt
from ... import random as trandom
from ..._lib import check, lib
from ...operators.propagation import fft_scales
from ..position import gaussian_derivative_taps

logger = logging.getLogger(__name__)

_MODELS = {"gaussian": 0, "poisson": 1}


class _Workspace:
    """Device buffers reused across minibatches (keyed on the operator)."""

    def __init__(self):
        self.buffers = {}

    def get(self, name, shape, dtype, device):
        n = int(np.prod(shape))
        buf = self.buffers.get(name)
        if (buf is None or buf.numel() < n or buf.dtype != dtype
                or buf.device != device):
            buf = torch.empty(n, dtype=dtype, device=device)
            self.buffers[name] = buf
        return buf[:n].view(*shape)


def _workspace(op):
    ws = getattr(op, "_tike_amd_workspace", None)
    if ws is None:
        ws = op._tike_amd_workspace = _Workspace()
    return ws


POSITION_MAJOR_SIZES = (128, 256, 512)
NO_FARPLANE_SIZES = (256,)
SPLIT_FORWARD_SIZES = (256, 512)
"""Detector sizes served by the position-major forward kernel
(tike_ptycho_fwd_intensity) and the gradient-scaled inverse."""


import os as _os

ONE_LAUNCH_GRADIENT_SIZES = (256, 512)
"""Detector sizes whose forward column pass, gradient factor and inverse pass
1 are one launch (tike_fwd_grad_ifft2_pass1; 512^2 since round 5); A/B runs
and tests shorten it to fall back to the two launches."""
if _os.environ.get("TIKE_ONE_LAUNCH_512", "1") == "0":
    ONE_LAUNCH_GRADIENT_SIZES = (256,)

POISSON_FROM_HANDOFF = True
"""Per-mode poisson step lengths at 256^2 / 512^2 from the forward hand-off
(tike_poisson_steps_handoff) instead of from a stored far plane; tests set
this to False to compare the two pipelines."""

POISSON_STEPS_IN_PASS2 = _os.environ.get("TIKE_POISSON_LINEAR", "1") == "1"
"""Every pixel measured, 256^2: the second sweep of the per-mode poisson step
lengths and the gradient pass in one launch, the steps applied by pass 2
(tike_poisson_steps_grad_ifft2_pass1)."""

CHUNK_POSITIONS_OVERRIDE = (int(_os.environ["TIKE_CHUNK_POSITIONS"])
                            if _os.environ.get("TIKE_CHUNK_POSITIONS") else None)
"""Tests set this to force small kernel chunks (several per minibatch);
TIKE_CHUNK_POSITIONS does the same for A/B runs of the bench."""


GENERAL_FUSED = True
"""Shapes outside `fused_gradients` (probe window < detector, more than 8
modes, detector sizes with factors 3 / 5 / 7 ...) run the three general
launches of csrc/general.hip (tike_gen_*; gaussian model) instead of the
unfused round-1 kernels; tests set this to False to compare the two."""


def general_gradients(S, pw, det):
    """True where the shape-general fused launches serve (csrc/general.hip):
    the detector size has a mixed-radix plan and the S lines of a row fit
    LDS."""
    return bool(GENERAL_FUSED and lib.tike_gen_supported(S, pw, det))


def fused_gradients(S, pw, det):
    """True where tike_ifft2_pass2_gradients serves (probe window = detector,
    a v2-engine size, at most 8 modes -- 4 at 512^2)."""
    return (pw == det and det in POSITION_MAJOR_SIZES
            and S <= (4 if det == 512 else 8))


def chunk_positions(S, det, position_major=False):
    """Positions per kernel launch: enough workgroups to fill the chip several
    times over while bounding the far-plane workspace.  The position-major
    kernels run one workgroup per position (not per tile)."""
    if CHUNK_POSITIONS_OVERRIDE:
        return max(1, int(CHUNK_POSITIONS_OVERRIDE))
    tiles = max(2048, (1 << 28) // (det * det * 8))  # >= 2048 tiles or 256 MiB
    if position_major:
        return max(1024, tiles // max(S, 1))
    return max(64, tiles // max(S, 1))


def global_count(comm, op, lo, hi):
    """Number of positions of the minibatch [lo, hi) over all ranks, as a
    host float; minibatch sizes are static, so this is all-reduced once and
    cached on the operator."""
    cache = op.__dict__.setdefault("_tike_amd_counts", {})
    key = minibatch_key(comm, lo, hi)
    if key not in cache:
        cache[key] = comm.Allreduce_count(hi - lo)
    return cache[key]


def minibatch_key(comm, lo, hi):
    """Cache key of a per-minibatch collective.  The LOCAL bounds alone are
    not one: a rank whose shares of two minibatches are both empty sees the
    same (lo, hi) twice and would skip a collective the other ranks issue.
    The solvers name the minibatch they are in (`comm.minibatch`, the same
    index on every rank) before they touch it."""
    return (getattr(comm, "minibatch", None), lo, hi, comm.size)


def mask_info(exitwave_options, det=None):
    """(number of measured pixels, uint8 device mask or None when every pixel
    is measured).  Cached on the options object, keyed on the identity and
    shape of `measured_pixels` so that a caller who swaps the mask on a
    reused options object is seen.  `det`: the detector width the kernels
    will read the mask with (they read det*det bytes)."""
    mask = exitwave_options.measured_pixels
    if det is not None and tuple(mask.shape) != (det, det):
        raise ValueError(
            f"measured_pixels shape {tuple(mask.shape)} does not match the "
            f"detector ({det}, {det})")
    key = (id(mask), tuple(mask.shape))
    cached = exitwave_options.__dict__.get("_mask_info")
    if cached is None or cached[0] != key:
        m = mask if A.is_device(mask) else A.to_device(
            np.asarray(mask, dtype=bool))
        n = int(m.sum().item())
        info = (n, None if n == m.numel() else m.to(torch.uint8).contiguous())
        cached = (key, info)
        exitwave_options.__dict__["_mask_info"] = cached
    return cached[1]


def lstsq_grad(parameters, data, batches, comm, *, op, epoch):
    """Advance psi / probe / eigen probes by one epoch (lstsq.py:25-294)."""
    scan = parameters.scan
    psi = parameters.psi
    probe = parameters.probe
    algorithm_options = parameters.algorithm_options
    eigen_weights = parameters.eigen_weights
    eigen_probe = parameters.eigen_probe
    exitwave_options = parameters.exitwave_options
    object_options = parameters.object_options
    probe_options = parameters.probe_options
    position_options = parameters.position_options
    if exitwave_options.noise_model not in _MODELS:
        raise ValueError(
            f"unknown noise model {exitwave_options.noise_model!r}")
    recover_probe = (probe_options is not None
                     and epoch >= probe_options.update_start)
    recover_psi = object_options is not None
    num_batch = algorithm_options.num_batch
    compact = algorithm_options.batch_method == "compact"
    order = (range(num_batch) if compact else
             trandom.randomizer_np.permutation(num_batch))
    if hasattr(data, "hint"):  # patterns streamed from pinned host memory
        data.hint([_lo_hi(batches[b]) for b in order])

    dev = psi.device
    H, W = psi.shape[-2:]
    # 'compact': the epoch's object gradient, kept in the scatter kernels'
    # planar (re plane, im plane) float layout
    object_combined = (torch.zeros((2, H, W), dtype=torch.float32, device=dev)
                       if recover_psi and compact else None)
    probe_combined_update = torch.zeros_like(probe)
    # per minibatch: {sum step_o, sum step_p, beta_object, beta_probe, cost}
    steps = torch.zeros((num_batch, 5), dtype=torch.float32, device=dev)
    pmax = None
    if recover_psi:
        # max of the object preconditioner: constant during the epoch
        pmax = torch.amax(object_options.preconditioner.real).reshape(
            1).contiguous()
    beta_object, beta_probe = [], []
    position_terms = None
    if position_options is not None:
        # numerator / denominator of the shift estimate, all local positions
        position_terms = (torch.zeros_like(scan), torch.zeros_like(scan))
    # the packed minibatch tail (three launches, two small all-reduces) serves
    # every configuration with at most one eigen probe
    packed = PACKED_TAIL and (eigen_probe is None
                              or eigen_probe.shape[-4] == 1)
    eigen_norms = None
    if (packed and recover_probe and eigen_weights is not None
            and eigen_probe is not None):
        # sum over the positions of every minibatch of the eigen weights
        # squared (probe.py:417-424), all ranks: the weights of a minibatch
        # change only in its own update, so one table (and one all-reduce)
        # per epoch serves them all
        eigen_norms = torch.stack([
            torch.square(eigen_weights[_lo_hi(b)[0]:_lo_hi(b)[1], 1, 0]).sum()
            for b in batches
        ]).to(torch.float32).contiguous()
        if comm.collective:
            comm.Allreduce(eigen_norms)

    for batch_index in order:
        lo, hi = _lo_hi(batches[batch_index])
        comm.minibatch = int(batch_index)
        g = _get_nearplane_gradients(
            data, psi, scan, probe, eigen_probe, eigen_weights, lo, hi, comm,
            num_batch=num_batch, exitwave_options=exitwave_options, op=op,
            recover_psi=recover_psi, recover_probe=recover_probe,
            position_terms=position_terms)

        object_update_precond = None
        if recover_psi:
            object_update_precond = _precondition_object_update(
                g["object_acc"], object_options.preconditioner, pmax=pmax,
                combined=object_combined)
        if packed:
            bbeta_object, bbeta_probe = _packed_tail(
                g, psi, scan, probe, eigen_probe, eigen_weights,
                object_update_precond, lo, hi, comm, op=op,
                num_batch=num_batch, recover_psi=recover_psi,
                recover_probe=recover_probe,
                norm=None if eigen_norms is None else
                eigen_norms[batch_index:batch_index + 1],
                steps_row=steps[batch_index],
                probe_combined_update=probe_combined_update)
        else:
            stats = _step_stats(g, psi, scan, probe, eigen_probe,
                                object_update_precond, lo, hi, op=op)

            if recover_probe and eigen_weights is not None:
                eigen_probe, eigen_weights = _update_nearplane(
                    g, stats, probe, eigen_probe, eigen_weights, lo, hi, comm,
                    num_batch=num_batch)

            bbeta_object, bbeta_probe, _ = _solve_steps(
                stats, g["costs"], g["count"], comm, pw=probe.shape[-1],
                recover_psi=recover_psi, recover_probe=recover_probe,
                out=steps[batch_index])

        if recover_psi:
            if not compact:
                dpsi = bbeta_object * object_update_precond
                if object_options.use_adaptive_moment:
                    dpsi, object_options.v, object_options.m = opt.momentum(
                        g=dpsi, v=object_options.v, m=object_options.m,
                        vdecay=object_options.vdecay,
                        mdecay=object_options.mdecay)
                psi = psi + dpsi
            beta_object.append(bbeta_object)

        if recover_probe:
            if not packed:  # (the packed tail's last launch does it)
                # probe += beta * mpu; combined += beta * mpu / num_batch
                check(
                    lib.tike_probe_update(A.ptr(probe),
                                          A.ptr(probe_combined_update),
                                          A.ptr(g["m_probe_update"]),
                                          A.ptr(bbeta_probe), 1.0 / num_batch,
                                          probe.numel(), A.stream_ptr()),
                    "probe update")
            beta_probe.append(bbeta_probe)

    batch_cost = steps[:, 4]
    if position_options is not None:
        # the minibatches above all used the old positions (lstsq.py:209-220)
        scan = _update_position(scan, position_options, *position_terms, comm,
                                epoch=epoch)

    # one device->host scalar per epoch, as in the reference (lstsq.py:222)
    algorithm_options.costs.append([float(batch_cost.mean().item())])
    if (eigen_weights is not None and eigen_weights.shape[1] > 1
            and not np.isfinite(algorithm_options.costs[-1][0])):
        # probe.py:426-427 raises when a minibatch's eigen weights are all
        # zero; the check costs a host sync per minibatch there and is made
        # here once per epoch, on the symptom (a non-finite cost)
        nonzero = torch.stack([
            (eigen_weights[int(idx[0]):int(idx[0]) + len(idx), 1:, 0]
             != 0).sum(dim=0).to(torch.float32) if len(idx) else torch.zeros(
                 eigen_weights.shape[1] - 1, device=psi.device)
            for idx in batches
        ])
        if comm.collective:
            comm.Allreduce(nonzero)  # the reference tests the whole batch
        if bool((nonzero == 0).any().item()):
            raise ValueError("eigen_probe weights cannot all be zero?")

    if recover_psi and compact:
        object_update_precond = _precondition_object_update(
            object_combined, object_options.preconditioner, pmax=pmax)
        bo = torch.mean(torch.stack(beta_object))
        dpsi = bo * object_update_precond
        psi = psi + dpsi
        if object_options.use_adaptive_moment:
            dpsi, object_options.v, object_options.m = _momentum_checked(
                g=dpsi, v=object_options.v, m=object_options.m,
                mdecay=object_options.mdecay,
                errors=[float(x[0]) for x in algorithm_options.costs[-3:]],
                beta=bo, memory_length=3)
            weight = object_options.preconditioner
            weight = weight / (0.1 * weight.real.max() + weight)
            psi = psi + weight * dpsi

    if recover_probe and probe_options.use_adaptive_moment:
        bp = torch.mean(torch.stack(beta_probe))
        dprobe = probe_combined_update
        if probe_options.v is None:
            probe_options.v = torch.zeros((3, *dprobe.shape),
                                          dtype=dprobe.dtype,
                                          device=dprobe.device)
        if probe_options.m is None:
            probe_options.m = torch.zeros_like(dprobe)
        mode = 0  # ptychoshelves only applies momentum to the main probe
        (d, probe_options.v[..., mode, :, :],
         probe_options.m[..., mode, :, :]) = _momentum_checked(
             g=dprobe[..., mode, :, :], v=probe_options.v[..., mode, :, :],
             m=probe_options.m[..., mode, :, :], mdecay=probe_options.mdecay,
             errors=[float(x[0]) for x in algorithm_options.costs[-3:]],
             beta=bp, memory_length=3)
        probe[..., mode, :, :] = probe[..., mode, :, :] + d

    parameters.scan = scan
    parameters.psi = psi
    parameters.probe = probe
    parameters.eigen_weights = eigen_weights
    parameters.eigen_probe = eigen_probe
    return parameters


EIGEN_PATCH_RECOMPUTE = True
"""The eigen pixel update of the packed tail gathers the object patches from
psi (L2) instead of streaming the stored ones (HBM): 0.186 -> 0.160 ms per 1000
positions at 256^2.  (With two 16-byte tap loads per pixel the position sums
lost by it, 0.215 -> 0.221 ms; see EIGEN_SUMS_RECOMPUTE.)"""

EIGEN_SUMS_RECOMPUTE = os.environ.get("TIKE_EIGEN_SUMS_GATHER", "1") == "1"
"""The position sums gather too, through the two-positions-per-workgroup row
walk of tike_eigen_position_sums1 (one 16-byte tap load per pixel and
position, E_0 and the probe update loaded once per pair): 0.215 -> 0.176 ms
per 1000 positions at 256^2."""

STATS_PATCH_RECOMPUTE = os.environ.get("TIKE_STATS_GATHER", "0") == "1"
"""The step statistics gather O_n from the object with the two 16-byte tap
loads they already issue for the preconditioned update (same offsets)."""

PACKED_TAIL = True
"""Tests set this to False to run the staged tail (one entry per step of the
reference's _update_nearplane / _precondition_nearplane_gradients)."""


def _lo_hi(batch):
    """[lo, hi) of a contiguous minibatch index range."""
    lo = int(batch[0]) if len(batch) else 0
    return lo, lo + len(batch)


def _eigen_args(eigen_probe, weights):
    """(eigen ptr, weights tensor, C, Sm) for the on-the-fly varying probe."""
    if weights is None:
        return None, None, 0, 0
    C = Sm = 0
    if eigen_probe is not None:
        C, Sm = eigen_probe.shape[-4], eigen_probe.shape[-3]
    assert weights.shape[1] == C + 1, (weights.shape, C)
    return eigen_probe, weights, C, Sm


def _get_nearplane_gradients(data, psi, scan, probe, eigen_probe,
                             eigen_weights, lo, hi, comm, *, num_batch,
                             exitwave_options, op, recover_psi, recover_probe,
                             position_terms=None, need_chi0=True):
    """Object / probe gradients of one minibatch (lstsq.py:367-602).
    need_chi0=False (cgrad: no step statistics follow) skips the store of
    mode 0 of chi where the fused pass 2 would be its only producer."""
    dev = psi.device
    B = hi - lo
    S, pw = probe.shape[-3], probe.shape[-1]
    det = op.detector_shape
    H, W = psi.shape[-2:]
    ws = _workspace(op)
    st = A.stream_ptr()
    fwd_scale, inv_scale = fft_scales(det, op.norm)
    nmeasured, mask_u8 = mask_info(exitwave_options, det)

    # the weights this gradient uses; every consumer (forward, gradients, step
    # statistics) runs before _update_nearplane changes them in place
    w_old = None
    if eigen_weights is not None:
        w_old = eigen_weights[lo:hi]
    ep, _, C, Sm = _eigen_args(eigen_probe, w_old)

    # planar (real plane, imaginary plane) float32 accumulator of the object
    # gradient: the shape float atomics run fastest on; recombined below
    # ... and it shares ONE flat buffer with the probe gradient, so that the
    # two are all-reduced in place by a single collective (no packing copy)
    n_obj = 2 * H * W if recover_psi else 0
    n_prb = 2 * probe.numel() if recover_probe else 0
    grads = torch.zeros(n_obj + n_prb, dtype=torch.float32, device=dev)
    obj_acc = grads[:n_obj].view(2, H, W) if recover_psi else None
    m_probe_update = (torch.view_as_complex(
        grads[n_obj:].view(*probe.shape, 2)) if recover_probe else None)
    probe_sum = None  # handle of the early all-reduce of the probe gradient
    # ... started as soon as pass 2 has finished that slice; decided from what
    # every rank knows alike (never from this rank's share of the positions)
    early = bool(comm.collective and recover_psi and recover_probe)
    chi0 = None  # allocated below unless chi itself can be handed on
    patches = None
    pos_major = det in POSITION_MAJOR_SIZES
    # inverse pass 2 fused with both gradients (chi never stored): probe
    # window = detector, at most 8 modes (4 at 512^2)
    fused = fused_gradients(S, pw, det)
    # every other shape, gaussian model: the three shape-general launches
    # (zero padding, far plane and chi never stored); downstream it looks
    # like the fused route (patches and chi0 stored, 1/num_batch applied)
    # (detector sizes with position-major kernels -- 128, 256, 512 -- keep
    # those for pw < det or many modes: measured faster, c3pad 160 vs 88 k
    # patterns/s, c3m12 69 vs 41 k, profiles/r06_experiments.md)
    general = (not fused and (not pos_major or GENERAL_FUSED == "always")
               and exitwave_options.noise_model == "gaussian"
               and general_gradients(S, pw, det))
    if general:
        pos_major = False
    if ((recover_probe and eigen_weights is not None) or position_terms
            or fused or general):
        patches = ws.get("patches", (max(B, 1), pw, pw), torch.complex64, dev)
    if position_terms:
        taps, taps_r = gaussian_derivative_taps(sigma=0.333)
    costs = ws.get("costs", (max(B, 1),), torch.float32, dev)
    chunk = chunk_positions(S, det, pos_major or general)
    poisson = exitwave_options.noise_model == "poisson"
    inten = gscale = steps = None
    if pos_major or poisson:
        inten = ws.get("intensity", (min(chunk, max(B, 1)), det, det),
                       torch.float32, dev)
    if pos_major:
        gscale = ws.get("gscale", (min(chunk, max(B, 1)), det, det),
                        torch.float32, dev)
    # detector sizes with the far-plane-free pipeline (the per-mode poisson
    # steps of 'all_modes' need |F_s|^2 and keep the stored far plane)
    # (512^2: only together with the fused pass 2, the generic gradient +
    # inverse + crop kernel exists at 256^2 only)
    # (round 4: with the fused pass 2 the per-mode steps come from two more
    # column passes over the forward hand-off, tike_poisson_steps_handoff, and
    # the far plane is not kept either)
    all_modes = (poisson and exitwave_options.step_length_usemodes
                 != "dominant_mode")
    no_farplane = (pos_major
                   and (det in NO_FARPLANE_SIZES or (det == 512 and fused))
                   and not (all_modes and not (fused and POISSON_FROM_HANDOFF)))
    if poisson:
        # per-(position, mode) step lengths (exitwave.py:122-234)
        steps = ws.get("steps", (min(chunk, max(B, 1)), S), torch.float32,
                       dev)
        dominant = int(
            exitwave_options.step_length_usemodes == "dominant_mode")
        step_start = float(exitwave_options.step_length_start)
        step_weight = float(exitwave_options.step_length_weight)
    objproj = ws.get("objproj", (min(chunk, max(B, 1)), pw, pw),
                     torch.complex64, dev)
    unique = None
    if w_old is not None and Sm > 0 and not general:
        unique = ws.get("unique", (min(chunk, max(B, 1)), Sm, pw, pw),
                        torch.complex64, dev)
    # (general: the two hand-offs hold the pw rows of the probe window only)
    far = ws.get("far", (min(chunk, max(B, 1)), 1, S, pw if general else det,
                         det), torch.complex64, dev)
    # the inverse transform is out of place (far -> mid); chi is the cropped
    # result and aliases mid when the probe fills the detector
    mid = ws.get("mid", tuple(far.shape), torch.complex64, dev)
    chi_ws = mid
    if pw != det and not general:
        chi_ws = ws.get("chi", (min(chunk, max(B, 1)), 1, S, pw, pw),
                        torch.complex64, dev)

    # mode 0 of chi is read again after the whole minibatch (step sizes,
    # eigen probes).  When the minibatch is one chunk, chi is still intact
    # then and is handed on with a mode stride; otherwise mode 0 is packed.
    # 256^2 / 512^2 with the far plane kept: split forward, intermediate in `far`
    split_kept = (pos_major and fused and not no_farplane
                  and det in SPLIT_FORWARD_SIZES)
    single_chunk = B <= chunk and not fused and not general
    if not single_chunk:
        chi0 = ws.get("chi0", (max(B, 1), pw, pw), torch.complex64, dev)
    for clo in range(lo, hi, chunk):
        chi_hi = min(hi, clo + chunk)
        n = chi_hi - clo
        blo = clo - lo
        w_c = None if w_old is None else w_old[blo:blo + n]
        chi = chi_ws
        uq = None
        steps_in_pass2 = False
        if w_c is not None and Sm > 0 and not no_farplane and not general:
            # varying probe of the modes that own eigen probes, once per chunk
            # (the 256^2 kernels form it on the fly instead)
            uq = unique[:n]
            check(
                lib.tike_varying_probe(A.ptr(probe), A.ptr(ep), A.ptr(w_c), C,
                                       Sm, A.ptr(uq), n, S, pw, st),
                "varying probe")
        model = _MODELS[exitwave_options.noise_model]
        unmeasured = float(exitwave_options.unmeasured_pixels_scaling)
        # float32 view of the chunk for the kernels without a 16-bit loader
        # (the 256^2 gaussian hot path reads uint16 directly)
        dchunk = None
        if not (pos_major and no_farplane and not (poisson and dominant)):
            dchunk = A.data_f32(data, clo, chi_hi)
        if general:
            # K1 rows (patch x probe, zero padding made in LDS) -> K2 columns
            # (intensity, cost, gradient factor, inverse columns) -> K3 below
            check(
                lib.tike_gen_fwd_rows(
                    A.ptr(psi), A.ptr(scan[clo:chi_hi]), A.ptr(probe), 0, None,
                    A.ptr(ep), A.ptr(w_c), C, Sm, A.ptr(far),
                    A.ptr(patches[blo:blo + n]), n, S, pw, det, H, W, st),
                "general forward rows")
            check(
                lib.tike_gen_cols_gradient(
                    A.ptr(far), A.ptr(dchunk), A.ptr(mask_u8),
                    A.ptr(costs[blo:blo + n]), A.ptr(mid), n, S, pw, det,
                    fwd_scale, model, unmeasured, nmeasured, st),
                "general columns + gradient")
        elif pos_major and no_farplane:
            # the far-plane waves never reach memory: the forward kernel forms
            # them in registers for the intensity and leaves the input of its
            # column pass in `far`; the inverse kernel re-forms them from
            # there, applies the gradient factor and transforms back
            # (the gradient factor and the costs come out of the same launch;
            # the intensity itself is stored only for the poisson steps)
            check(
                lib.tike_fwd_pass1(
                    A.ptr(psi), A.ptr(scan[clo:chi_hi]), A.ptr(probe), 0,
                    None, A.ptr(ep), A.ptr(w_c), C, Sm, A.ptr(far),
                    A.ptr(patches[blo:blo + n]) if fused else None, n, S, pw,
                    det, H, W, st), "forward pass 1")
            # 256^2 without poisson step lengths: the column pass, the gradient
            # factor and the inverse's pass 1 are ONE launch (the factor never
            # goes through memory)
            one_launch = (fused and not poisson
                          and det in ONE_LAUNCH_GRADIENT_SIZES)
            # no gradient at unmeasured pixels (none of them, or the default
            # unmeasured_pixels_scaling = 1): the gradient is linear in the
            # step lengths, so their second sweep and the gradient pass are
            # one launch and pass 2 applies them
            steps_in_pass2 = (poisson and not dominant and fused
                              and det == 256
                              and (mask_u8 is None or unmeasured == 1.0)
                              and POISSON_STEPS_IN_PASS2)
            if steps_in_pass2:
                sums = ws.get("poisson_sums", (min(chunk, max(B, 1)), S, 2),
                              torch.float32, dev)
                check(
                    lib.tike_poisson_steps_grad_ifft2_pass1(
                        A.ptr(far), A.ptr(data[clo:chi_hi]),
                        int(data.dtype == torch.uint16), A.ptr(mask_u8),
                        A.ptr(costs[blo:blo + n]), A.ptr(steps), A.ptr(sums),
                        A.ptr(mid), n, S, det, fwd_scale, unmeasured,
                        nmeasured, step_start, step_weight, st),
                    "poisson step lengths + gradient + inverse pass 1")
            elif poisson and not dominant:
                # gradient factor, costs and the per-mode step lengths from
                # the hand-off: three reads of it, no far plane stored
                sums = ws.get("poisson_sums", (min(chunk, max(B, 1)), S, 2),
                              torch.float32, dev)
                check(
                    lib.tike_poisson_steps_handoff(
                        A.ptr(far), A.ptr(data[clo:chi_hi]),
                        int(data.dtype == torch.uint16), A.ptr(mask_u8),
                        A.ptr(gscale), A.ptr(costs[blo:blo + n]),
                        A.ptr(steps), A.ptr(sums), n, S, det, fwd_scale,
                        unmeasured, nmeasured, step_start, step_weight, st),
                    "forward pass 2 + poisson factor and step lengths")
            elif one_launch:
                check(
                    lib.tike_fwd_grad_ifft2_pass1(
                        A.ptr(far), A.ptr(data[clo:chi_hi]),
                        int(data.dtype == torch.uint16), A.ptr(mask_u8),
                        A.ptr(costs[blo:blo + n]), A.ptr(mid), n, S, det,
                        fwd_scale, model, unmeasured, nmeasured, st),
                    "column pass + gradient + inverse pass 1")
            else:
                check(
                    lib.tike_fwd_gradient_scale(
                        A.ptr(far), A.ptr(data[clo:chi_hi]),
                        int(data.dtype == torch.uint16), A.ptr(mask_u8),
                        A.ptr(gscale), A.ptr(inten) if poisson else None,
                        A.ptr(costs[blo:blo + n]), None, n, S, det, fwd_scale,
                        model, unmeasured, nmeasured, st),
                    "forward pass 2 + gradient scale")
            if poisson and dominant:  # the steps need no far-plane waves
                check(
                    lib.tike_poisson_steps(
                        None, A.ptr(inten), A.ptr(dchunk),
                        A.ptr(mask_u8), A.ptr(steps), n, S, det, step_start,
                        step_weight, 1, st), "poisson step lengths")
            if one_launch or steps_in_pass2:
                pass
            elif fused:
                check(
                    lib.tike_grad_ifft2_pass1(
                        A.ptr(far), A.ptr(gscale),
                        A.ptr(steps) if poisson else None,
                        A.ptr(mask_u8) if poisson else None, S, A.ptr(mid),
                        n * S, det, fwd_scale, st),
                    "gradient + inverse pass 1")
            else:
                check(
                    lib.tike_grad_ifft2_crop(
                        A.ptr(far), A.ptr(gscale),
                        A.ptr(steps) if poisson else None,
                        A.ptr(mask_u8) if poisson else None, S, A.ptr(mid),
                        A.ptr(chi), n * S, det, pw, fwd_scale, inv_scale, st),
                    "gradient + ifft2 + crop")
        elif pos_major and fused and det in SPLIT_FORWARD_SIZES:
            # the far plane is kept (512^2; per-mode poisson steps at 256^2):
            # forward pass 1 -> streamed column pass that stores the far-plane
            # waves (in `mid`) next to the gradient factor -> inverse pass 1
            # back into `far` -> pass 2 + gradients
            check(
                lib.tike_fwd_pass1(
                    A.ptr(psi), A.ptr(scan[clo:chi_hi]), A.ptr(probe), 0,
                    A.ptr(uq), None, A.ptr(w_c), C, Sm, A.ptr(far),
                    A.ptr(patches[blo:blo + n]), n, S, pw, det, H, W, st),
                "forward pass 1")
            check(
                lib.tike_fwd_gradient_scale(
                    A.ptr(far), A.ptr(dchunk), 0, A.ptr(mask_u8),
                    A.ptr(gscale), A.ptr(inten) if poisson else None,
                    A.ptr(costs[blo:blo + n]), A.ptr(mid), n, S, det,
                    fwd_scale, model, unmeasured, nmeasured, st),
                "forward pass 2 + gradient scale")
            if poisson:
                check(
                    lib.tike_poisson_steps(
                        A.ptr(mid), A.ptr(inten), A.ptr(dchunk),
                        A.ptr(mask_u8), A.ptr(steps), n, S, det, step_start,
                        step_weight, dominant, st), "poisson step lengths")
            check(
                lib.tike_ifft2_pass1_scaled(
                    A.ptr(mid), A.ptr(gscale),
                    A.ptr(steps) if poisson else None,
                    A.ptr(mask_u8) if poisson else None, S, A.ptr(far),
                    n * S, det, st), "scaled inverse pass 1")
        elif pos_major:
            # forward + intensity in one kernel; the gradient factor is a
            # per-pixel table applied while the inverse transform loads rows
            check(
                lib.tike_ptycho_fwd_intensity(
                    A.ptr(psi), A.ptr(scan[clo:chi_hi]), A.ptr(probe), 0,
                    A.ptr(uq), A.ptr(w_c), C, Sm, A.ptr(far), A.ptr(inten),
                    A.ptr(patches[blo:blo + n]) if fused else None, n, S, pw,
                    det, H, W, fwd_scale, st), "forward + intensity")
            check(
                lib.tike_gradient_scale(A.ptr(inten), A.ptr(dchunk),
                                        A.ptr(mask_u8), A.ptr(gscale),
                                        A.ptr(costs[blo:blo + n]), n, det,
                                        model, unmeasured, nmeasured, st),
                "gradient scale")
            if poisson:
                check(
                    lib.tike_poisson_steps(
                        A.ptr(far), A.ptr(inten), A.ptr(dchunk),
                        A.ptr(mask_u8), A.ptr(steps), n, S, det, step_start,
                        step_weight, dominant, st), "poisson step lengths")
            if fused:
                check(
                    lib.tike_ifft2_pass1_scaled(
                        A.ptr(far), A.ptr(gscale),
                        A.ptr(steps) if poisson else None,
                        A.ptr(mask_u8) if poisson else None, S, A.ptr(mid),
                        n * S, det, st), "scaled inverse pass 1")
            elif poisson:
                check(
                    lib.tike_ifft2_crop_scaled_modes(
                        A.ptr(far), A.ptr(gscale), A.ptr(steps),
                        A.ptr(mask_u8), S, A.ptr(mid), A.ptr(chi), n * S, det,
                        pw, inv_scale, st), "scaled ifft2 + crop (poisson)")
            else:
                check(
                    lib.tike_ifft2_crop_scaled(A.ptr(far), A.ptr(gscale), S,
                                               A.ptr(mid), A.ptr(chi), n * S,
                                               det, pw, inv_scale, st),
                    "scaled ifft2 + crop")
        else:
            op.fwd_device(probe, scan[clo:chi_hi], psi, eigen_probe, w_c,
                          out=far[:n])
            if poisson:
                check(
                    lib.tike_intensity(A.ptr(far), A.ptr(inten), n, S,
                                       det * det, st), "intensity")
                check(
                    lib.tike_poisson_steps(
                        A.ptr(far), A.ptr(inten), A.ptr(dchunk),
                        A.ptr(mask_u8), A.ptr(steps), n, S, det, step_start,
                        step_weight, dominant, st), "poisson step lengths")
            check(
                lib.tike_farplane_gradient(
                    A.ptr(far), A.ptr(dchunk), A.ptr(mask_u8), None,
                    A.ptr(costs[blo:blo + n]), n, S, det, model, 1, unmeasured,
                    nmeasured, st), "farplane gradient")
            if poisson:
                check(
                    lib.tike_scale_modes(A.ptr(far), A.ptr(steps),
                                         A.ptr(mask_u8), n * S, det, st),
                    "poisson step scaling")
            check(
                lib.tike_ifft2_crop(A.ptr(far), A.ptr(mid), A.ptr(chi), n * S,
                                    det, pw, inv_scale, st), "ifft2 + crop")
        if general:
            check(
                lib.tike_gen_inv_rows_gradients(
                    A.ptr(mid), A.ptr(patches[blo:blo + n]), A.ptr(probe), 0,
                    None, A.ptr(ep), A.ptr(w_c), C, Sm,
                    A.ptr(objproj) if recover_psi else None,
                    A.ptr(chi0[blo:blo + n]) if need_chi0 else None,
                    A.ptr(m_probe_update), 1.0 / num_batch, n, S, pw, det,
                    inv_scale, st), "general inverse rows + gradients")
        elif fused:
            # inverse column pass + both gradients + mode 0 of chi, one
            # pixel-major kernel (chi itself never exists in memory)
            p2 = (A.ptr(far if split_kept else mid),
                  A.ptr(patches[blo:blo + n]), A.ptr(probe), A.ptr(ep),
                  A.ptr(w_c), C, Sm, A.ptr(objproj) if recover_psi else None,
                  A.ptr(chi0[blo:blo + n]) if need_chi0 else None,
                  A.ptr(m_probe_update), 1.0 / num_batch, n, S, det, inv_scale)
            if steps_in_pass2:
                check(lib.tike_ifft2_pass2_gradients_scaled(*p2, A.ptr(steps),
                                                            st),
                      "inverse pass 2 + gradients (x poisson steps)")
            else:
                check(lib.tike_ifft2_pass2_gradients(*p2, st),
                      "inverse pass 2 + gradients")
        else:
            # one pass over chi: probe gradient, object projection, patches
            check(
                lib.tike_lstsq_gradients(
                    A.ptr(chi), A.ptr(scan[clo:chi_hi]), A.ptr(psi),
                    A.ptr(probe), A.ptr(ep), A.ptr(w_c), C, Sm,
                    None,  # on the fly: L2-resident
                    None if patches is None else A.ptr(patches[blo:blo + n]),
                    A.ptr(m_probe_update),
                    A.ptr(objproj) if recover_psi else None, n, S, pw, H, W,
                    st), "probe gradient + object projection")
        if position_terms:
            chi_m, chi_modes = ((chi0[blo:blo + n], 1) if fused or general
                                else (chi, S))
            check(
                lib.tike_position_sums(
                    A.ptr(patches[blo:blo + n]), A.ptr(chi_m), chi_modes,
                    A.ptr(probe),
                    A.ptr(ep), A.ptr(w_c), C, Sm, taps.ctypes.data, taps_r,
                    A.ptr(position_terms[0][clo:chi_hi]),
                    A.ptr(position_terms[1][clo:chi_hi]), n, S, pw, st),
                "position shift sums")
        if chi_hi == hi and early:
            # the probe gradient is complete: its slice of the flat buffer
            # travels while the last object scatter runs
            probe_sum = comm.Allreduce_start(grads[n_obj:])
        if recover_psi:
            check(
                lib.tike_scatter_patches(A.ptr(objproj),
                                         A.ptr(scan[clo:chi_hi]),
                                         A.ptr(obj_acc), n, pw, H, W, st),
                "object scatter")
        if not single_chunk and not fused and not general:
            chi0[blo:blo + n] = chi[:n, 0, 0]

    # complete the sums over positions across ranks.  Which collectives are
    # issued, and in which order, depends on rank-invariant facts only
    # (`early`): a rank whose share of this minibatch is EMPTY never entered
    # the chunk loop and starts the probe slice here, so that every rank
    # issues the same two all-reduces in the same order
    if comm.collective and grads.numel():
        if early:
            if probe_sum is None:
                probe_sum = comm.Allreduce_start(grads[n_obj:])
            comm.Allreduce(grads[:n_obj])
            probe_sum.wait()
        else:
            comm.Allreduce(grads)
    count = global_count(comm, op, lo, hi)
    if recover_probe and not fused and not general:
        m_probe_update = m_probe_update / num_batch  # (fused: in the kernel)
    return dict(chi0=chi_ws if single_chunk else chi0[:B],
                chi_modes=S if single_chunk else 1, w_old=w_old,
                patches=None if patches is None
                else patches[:B], object_acc=obj_acc,
                m_probe_update=m_probe_update, costs=costs[:B], count=count,
                local_count=B)


def object_upd_sum(g):
    """The object gradient of a minibatch as a (1, H, W) complex array
    (`object_upd_sum` of lstsq.py:514-520) from the planar accumulator."""
    acc = g["object_acc"]
    return None if acc is None else torch.complex(acc[0], acc[1])[None]


def _precondition_object_update(object_acc, psi_update_denominator,
                                alpha=0.05, *, pmax=None, combined=None):
    """g / sqrt(((1-a) P)^2 + (a max P)^2) (lstsq.py:605-616) straight from
    the planar (2, H, W) scatter accumulator; `combined` (planar) += g."""
    H, W = object_acc.shape[-2:]
    if pmax is None:
        pmax = torch.amax(psi_update_denominator.real).reshape(1).contiguous()
    out = torch.empty((1, H, W), dtype=torch.complex64,
                      device=object_acc.device)
    check(
        lib.tike_object_update_precond(A.ptr(object_acc),
                                       A.ptr(psi_update_denominator),
                                       A.ptr(pmax), alpha, None, A.ptr(out),
                                       A.ptr(combined), H * W, A.stream_ptr()),
        "preconditioned object update")
    return out


def _step_stats(g, psi, scan, probe, eigen_probe, object_update_precond, lo,
                hi, *, op):
    """Per-position sums of the normal equations (HIP; lstsq.py:652-694)."""
    B = hi - lo
    dev = psi.device
    S, pw = probe.shape[-3], probe.shape[-1]
    stats = _workspace(op).get("stats", (max(B, 1), 8), torch.float32, dev)
    ep, w_old, C, Sm = _eigen_args(eigen_probe, g["w_old"])
    # the eigen-probe update starts with the projection of every position's
    # residual onto the first eigen probe: same operands as this pass
    eigen0 = eproj = None
    if (ep is not None and C >= 1 and Sm >= 1 and g["patches"] is not None
            and g["m_probe_update"] is not None):
        eigen0 = ep[0, 0, 0]
        eproj = _workspace(op).get("eigen_proj", (max(B, 1),), torch.float32,
                                   dev)
    g["eigen_proj"] = None if eproj is None else eproj[:B]
    check(
        lib.tike_lstsq_step_stats(
            A.ptr(g["chi0"]), A.ptr(scan[lo:hi]), A.ptr(psi),
            A.ptr(object_update_precond), A.ptr(probe), A.ptr(ep),
            A.ptr(w_old), C, Sm, None, A.ptr(g["m_probe_update"]),
            None if (STATS_PATCH_RECOMPUTE and psi.shape[0] == 1
                     and object_update_precond is not None) else
            A.ptr(g["patches"]), A.ptr(stats), B,
            S, g["chi_modes"], pw, psi.shape[-2], psi.shape[-1],
            A.ptr(eigen0), A.ptr(eproj), A.stream_ptr()),
        "step-size statistics")
    return stats[:B]


def _packed_tail(g, psi, scan, probe, eigen_probe, eigen_weights,
                 object_update_precond, lo, hi, comm, *, op, num_batch,
                 recover_psi, recover_probe, norm, steps_row,
                 probe_combined_update):
    """Everything between the gradients of a minibatch and the next forward
    pass (lstsq.py:136-205: step statistics, `_update_nearplane` with at most
    one eigen probe, the 2x2 step lengths, `probe += beta * m_probe_update`)
    in five launches (six kernels) and, when ranks share the minibatch, two small
    all-reduces: { sum A1, sum A4, sum cost ; eigen update } and { sum step_o,
    sum step_p, eigen weight denominator }.  Returns (beta_object,
    beta_probe) as 0-d device tensors; eigen probe and weights are updated in
    place; steps_row (5,) receives tike_lstsq_tail_finish's `steps`."""
    B = hi - lo
    dev = psi.device
    st = A.stream_ptr()
    S, pw = probe.shape[-3], probe.shape[-1]
    P = pw * pw
    count = float(g["count"])
    ws = _workspace(op)
    eig = recover_probe and eigen_weights is not None
    C = Sm = 0
    if eig and eigen_probe is not None:
        C, Sm = eigen_probe.shape[-4], eigen_probe.shape[-3]
        assert eigen_weights.shape[-2] == C + 1 and C == 1
    mpu = g["m_probe_update"]
    one = (eig and C == 1 and Sm >= 1 and g["patches"] is not None
           and mpu is not None and norm is not None)
    # (an eigen probe that this tail cannot update must never be skipped
    # silently: lstsq_grad sends such configurations to the staged entries)
    assert one or C == 0, "packed tail: eigen probe without its operands"
    # one zeroed buffer: [sums3 (3), -, update (2 P) | nacc (3), - | tail3 (3), -]
    small = ws.get("tail_small", (4 + 2 * P + 8,), torch.float32, dev)
    small.zero_()
    sums3 = small[:3]
    update = small[4:4 + 2 * P]
    nacc = small[4 + 2 * P:4 + 2 * P + 4]
    tail3 = small[4 + 2 * P + 4:4 + 2 * P + 7]
    stats = _step_stats(g, psi, scan, probe, eigen_probe,
                        object_update_precond, lo, hi, op=op)
    eps_total = float(np.float32(np.float32(1e-9) / P) * P)
    E = w_rows = sums5 = None
    row = 0
    if eig:
        w_rows = eigen_weights[lo:hi]  # (B, C+1, S) rows of this minibatch
        row = w_rows.shape[-2] * w_rows.shape[-1]
    # the eigen passes recompute O_n from the object (it is the array the
    # stored patches were gathered from until psi is updated, after this tail)
    gpsi = A.ptr(psi[0]) if EIGEN_PATCH_RECOMPUTE and psi.shape[0] == 1 else None
    if one:
        E = eigen_probe[0, 0, 0]  # (pw, pw) view, contiguous
        check(
            lib.tike_eigen_pixel_update1(
                A.ptr(g["patches"]), A.ptr(g["chi0"]), A.ptr(mpu[0, 0, 0]),
                A.ptr(E), A.ptr(g["eigen_proj"]), w_rows[:, 1, 0].data_ptr(),
                row, A.ptr(norm), A.ptr(update), B, pw, g["chi_modes"],
                A.ptr(stats), A.ptr(g["costs"]), eps_total, A.ptr(sums3),
                gpsi, A.ptr(scan[lo:hi]), psi.shape[-2], psi.shape[-1], st),
            "eigen pixel update")
    else:
        check(
            lib.tike_lstsq_step_sums(A.ptr(stats), A.ptr(g["costs"]), B,
                                     eps_total, A.ptr(sums3), st),
            "step-size sums")
    if comm.collective:
        comm.Allreduce(small[:4 + 2 * P] if one else small[:4])
    beta_eigen = min(0.1, 1.0 / num_batch)
    check(
        lib.tike_lstsq_tail_mid(A.ptr(E), A.ptr(update), P, A.ptr(nacc),
                                beta_eigen, A.ptr(stats), B, eps_total,
                                A.ptr(sums3), count, int(recover_psi),
                                int(recover_probe), A.ptr(tail3), st),
        "step solve")
    if one:
        sums5 = ws.get("eigen_sums5", (max(B, 1), 5), torch.float32, dev)
        check(
            lib.tike_eigen_position_sums1(
                A.ptr(g["patches"]), A.ptr(g["chi0"]), A.ptr(mpu[0, 0, 0]),
                A.ptr(E), A.ptr(sums5), tail3[2:].data_ptr(), B, pw,
                g["chi_modes"], gpsi if EIGEN_SUMS_RECOMPUTE else None,
                A.ptr(scan[lo:hi]), psi.shape[-2], psi.shape[-1], st),
            "eigen position sums")
    if comm.collective:
        comm.Allreduce(tail3)
    check(
        lib.tike_lstsq_tail_finish(
            A.ptr(tail3), A.ptr(sums3), count, A.ptr(steps_row),
            A.ptr(probe) if recover_probe else None,
            A.ptr(probe_combined_update) if recover_probe else None,
            A.ptr(mpu), 1.0 / num_batch, probe.numel(),
            None if w_rows is None else w_rows.data_ptr(), row, S, 0,
            A.ptr(stats), A.ptr(sums5), B, P, st), "minibatch tail")
    return (steps_row[2] if recover_psi else None,
            steps_row[3] if recover_probe else None)


def _solve_steps(stats, costs, count, comm, *, pw, recover_psi, recover_probe,
                 out=None):
    """2x2 least-squares step sizes, averaged over the minibatch
    (lstsq.py:641-718), and the minibatch cost.  Means over positions are
    global (all ranks).  Returns 0-d device tensors (beta_object, beta_probe,
    cost); `out`: a (5,) float32 row to keep them in."""
    dev = stats.device
    B = stats.shape[0]
    st = A.stream_ptr()
    eps_total = float(np.float32(np.float32(1e-9) / (pw * pw)) * (pw * pw))
    if out is None:
        out = torch.zeros(5, dtype=torch.float32, device=dev)
    sums = torch.empty(3, dtype=torch.float32, device=dev)
    check(
        lib.tike_lstsq_step_sums(A.ptr(stats), A.ptr(costs), B, eps_total,
                                 A.ptr(sums), st), "step-size sums")
    if comm.collective:
        comm.Allreduce(sums)
    check(
        lib.tike_lstsq_step_solve(A.ptr(stats), B, eps_total, A.ptr(sums),
                                  float(count), int(recover_psi),
                                  int(recover_probe), A.ptr(out), st),
        "step-size solve")
    if comm.collective:
        comm.Allreduce(out[:2])
        out[2:4] = out[:2] / count
    return (out[2] if recover_psi else None,
            out[3] if recover_probe else None, out[4])


def _update_nearplane(g, stats, probe, eigen_probe, eigen_weights, lo, hi,
                      comm, *, num_batch):
    """Eigen-probe ("OPR") update for mode 0 (lstsq.py:297-364, 721-761;
    probe.py:362-476).  Means over positions are global."""
    m = 0
    dev = probe.device
    count = float(g["count"])
    B = hi - lo
    st = A.stream_ptr()
    Cw, S = eigen_weights.shape[-2] - 1, eigen_weights.shape[-1]
    row = (Cw + 1) * S
    w_rows = eigen_weights[lo:hi]  # (B, C+1, S) rows of this minibatch
    norms = torch.empty(max(Cw, 1), dtype=torch.float32, device=dev)
    # (lstsq.py:721-738) weights of the shared probe, and sum w_c^2
    check(
        lib.tike_eigen_weights0(A.ptr(w_rows), A.ptr(stats), B, Cw, S, m,
                                A.ptr(norms), st), "eigen weights (shared)")
    if Cw < 1 or eigen_probe is None:
        return eigen_probe, eigen_weights
    if m >= eigen_probe.shape[-3]:
        return eigen_probe, eigen_weights
    assert eigen_weights.shape[-2] == eigen_probe.shape[-4] + 1
    if comm.collective:
        comm.Allreduce(norms)
    chi0, patches = g["chi0"], g["patches"]
    mpu0 = g["m_probe_update"][0, 0, m]
    C, Sm, pw = eigen_probe.shape[-4], eigen_probe.shape[-3], probe.shape[-1]
    P = pw * pw
    beta = min(0.1, 1.0 / num_batch)
    sums = torch.empty((max(B, 1), 5), dtype=torch.float32, device=dev)
    coefs = torch.zeros((max(B, 1), C), dtype=torch.complex64, device=dev)
    pm = torch.empty(max(B, 1), dtype=torch.float32, device=dev)
    small = torch.empty(2, dtype=torch.float32, device=dev)  # dsum, esum
    nwork = torch.empty(4, dtype=torch.float32, device=dev)
    ep = eigen_probe  # (1, C, Sm, pw, pw), updated in place, read by kernels

    def position_sums(c):
        check(
            lib.tike_eigen_position_sums(A.ptr(patches), A.ptr(chi0),
                                         A.ptr(mpu0), A.ptr(ep), A.ptr(coefs),
                                         C, Sm, c, A.ptr(sums), B, pw,
                                         g["chi_modes"], st),
            "eigen position sums")

    for c in range(1, C + 1):
        w_c = w_rows[:, c, m]  # strided view: element n at n * row
        # a batch whose weights are all zero divides by zero here; the
        # reference raises ValueError after a host sync (probe.py:426) --
        # lstsq_grad raises the same error at the end of the epoch
        if c == 1 and g.get("eigen_proj") is not None:
            first, stride = g["eigen_proj"], 1  # from the step-statistics pass
        else:
            position_sums(c - 1)
            first, stride = sums, 5
        check(
            lib.tike_eigen_proj_mean(A.ptr(first), stride, w_c.data_ptr(), row,
                                     norms[c - 1:].data_ptr(), P, B,
                                     A.ptr(pm), st), "eigen projection mean")
        update = torch.zeros((pw, pw), dtype=torch.complex64, device=dev)
        check(
            lib.tike_eigen_pixel_update(A.ptr(patches), A.ptr(chi0),
                                        A.ptr(mpu0), A.ptr(ep), A.ptr(coefs),
                                        C, Sm, c - 1, A.ptr(pm),
                                        A.ptr(update), B, pw, g["chi_modes"],
                                        st),
            "eigen pixel update")
        if comm.collective:
            comm.Allreduce(update)
        E = ep[0, c - 1, m]  # (pw, pw) view, contiguous
        check(
            lib.tike_eigen_normalise(A.ptr(E), A.ptr(update), count, beta, P,
                                     small[1:].data_ptr(), A.ptr(nwork), st),
            "eigen probe normalisation")
        # new weights for the updated eigen probe (and the projection of the
        # residual onto it, removed before the next eigen probe)
        position_sums(c - 1)
        check(lib.tike_eigen_dsum(A.ptr(sums), B, P, A.ptr(small), st),
              "eigen weight denominator")
        if comm.collective:
            comm.Allreduce(small[:1])
        more = c + 1 < eigen_weights.shape[-2]
        check(
            lib.tike_eigen_weights(A.ptr(sums), B, P, A.ptr(small), count,
                                   w_c.data_ptr(), row,
                                   coefs[:, c - 1:].data_ptr() if more else
                                   None, C, small[1:].data_ptr(), st),
            "eigen weights")
    return eigen_probe, eigen_weights


def _trimmed_mean(rows, proportion=0.05):
    """Column means without the `proportion` smallest and largest entries of
    each column (scipy.stats.trim_mean)."""
    count = rows.shape[0]
    cut = int(proportion * count)
    return torch.sort(rows, dim=0).values[cut:count - cut].mean(dim=0)


def _update_position(scan, position_options, numerator, denominator, comm,
                     *, alpha=0.05, epoch=0):
    """Move every position by its damped least-squares shift estimate
    numerator / ((1 - alpha) denominator + alpha max(denominator)), clipped to
    `update_magnitude_limit`, with the common drift (5 % trimmed mean)
    removed and, if asked, passed through ADAM (lstsq.py:764-806).  The
    damping maximum and the trimmed mean run over the positions of ALL
    ranks."""
    po = position_options
    if epoch < po.update_start:
        return scan
    damping = alpha * torch.clamp(comm.Allreduce_max(denominator.max()),
                                  min=1e-6)
    shift = numerator / ((1 - alpha) * denominator + damping)
    limit = po.update_magnitude_limit
    if limit > 0:
        shift = shift.clamp(-limit, limit)
    shift -= _trimmed_mean(comm.Allgather_rows(shift))
    if po.use_adaptive_moment:
        moments = dict(vdecay=po.vdecay, mdecay=po.mdecay)
        shift, po.v, po.m = opt.adam(shift, po.v, po.m, **moments)
    return scan - shift


def _cost_is_falling(errors):
    """The smaller of the last two epoch costs lies below the larger of the
    two before the last one."""
    return len(errors) >= 3 and max(errors[-3:-1]) > min(errors[-2:])


def _momentum_checked(g, v, m, mdecay, errors, beta=1.0, memory_length=3):
    """Momentum that has to be earned (lstsq.py:809-858).  `v` keeps the last
    `memory_length` update directions, unit length times `beta`, newest last;
    `m` is the running sum.  Momentum is granted only if the cost trends
    downwards (`_cost_is_falling`) AND the newest direction has a positive
    overlap with every remembered one; the running sum is then damped by half
    the rate at which the log-overlap falls off into the past (a line fit
    through it, anchored at 0 for the newest) before `g` joins it.  Otherwise
    the step is zero and the sum is halved.  Returns (step, v, m)."""
    memory = (g.new_zeros((memory_length, *g.shape)) if v is None else v)
    memory = torch.cat((memory[1:], (g / linalg.norm(g) * beta)[None]))
    total = torch.zeros_like(g) if m is None else m
    overlap = None
    if _cost_is_falling(errors):
        overlap = (memory[:-1] * memory[-1].conj()).real.sum(
            dim=(-2, -1)).flatten().cpu().numpy()
    if overlap is None or not (overlap > 0).all():
        return torch.zeros_like(g), memory, total / 2
    slope, _ = opt.fit_line_least_squares(
        x=np.arange(overlap.size + 1),
        y=np.concatenate(([0.0], np.log(overlap))))
    total = (1 - 0.5 * max(-slope, 0)) * total + g
    return mdecay * total, memory, total
