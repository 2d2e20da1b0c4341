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
from types import SimpleNamespace

import numpy as np
import torch

from ... import _arrays as A
from ... import _tuning
from ... import linalg
from ... import opt
from ... import random as trandom
from ..._lib import check, lib
from ...operators.propagation import fft_scales
from ..position import gaussian_derivative_taps
from ._plan import GradientPlan

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


ONE_LAUNCH_GRADIENT_SIZES = (256, 512)
"""Detector sizes whose forward column pass, gradient factor and inverse pass
1 are one launch (tike_fwd_grad_ifft2_pass1; 512^2 since round 5); A/B runs
and tests shorten it to fall back to the two launches."""
if not _tuning.one_launch_512:
    ONE_LAUNCH_GRADIENT_SIZES = (256,)

POISSON_FROM_HANDOFF = True
"""Per-mode poisson step lengths at 256^2 / 512^2 from the forward hand-off
(tike_poisson_steps_handoff) instead of from a stored far plane; tests set
this to False to compare the two pipelines."""

POISSON_STEPS_IN_PASS2 = _tuning.poisson_steps_in_pass2
"""Every pixel measured, 256^2: the second sweep of the per-mode poisson step
lengths and the gradient pass in one launch, the steps applied by pass 2
(tike_poisson_steps_grad_ifft2_pass1)."""

CHUNK_POSITIONS_OVERRIDE = _tuning.chunk_positions
"""Tests set this to force small kernel chunks (several per minibatch);
TIKE_CHUNK_POSITIONS does the same for A/B runs of the bench."""


GENERAL_FUSED = True
"""Shapes outside `fused_gradients` (probe window < detector, more than 8
modes, detector sizes with factors 3 / 5 / 7 ...) run the three general
launches of csrc/general.hip (tike_gen_*; gaussian model) instead of the
unfused round-1 kernels; tests set this to False to compare the two."""


GENERAL_MIN_DETECTOR = 1 << 30
"""Detector sizes below this (and without a prime-factor decomposition) keep
the unfused kernels.  Late in round 6 that is every size: with 384, 768 and
1024 -- the sizes where the LDS line engine's three launches were 10 ... 17 %
ahead -- on the prime-factor kernels, what is left to it loses to the unfused
kernels on the mixed-radix transforms or ties with them (300^2 x 1 ... 8 modes
-43 ... -6 %, 400^2 +-0, 500^2 x 4 +7 %, 600^2 x 2 -12 %, 720^2 x 4 -15 %, 1000^2
x 1 ... 4 -3 ... -25 %; with an eigen probe -14 ... +7 %:
profiles/r06_experiments.md section 17).  The launches stay in the tree and
in the tests (`GENERAL_FUSED = "always"`); 256 restores round 6's first rule."""

PFA_ROUTE = True
"""Detector sizes 3 x 2^k, 5 x 2^k and 7 x 2^k (96 ... 3584), and 1024 / 2048, without position-major
kernels take the prime-factor launches of csrc/pfa.hip (tike_pfa_*): the
power-of-two register engine on p x p sub-tiles; tests set this to False to
compare with the LDS line engine of csrc/general.hip."""


MODE_GROUPS = True
"""More modes than one launch of the inverse's second pass holds in registers
(9 ... 32 at 128^2 / 256^2, 5 ... 16 at 512^2; probe window = detector,
gaussian model) run the far-plane-free kernels with that pass in groups of
modes (tike_ifft2_pass2_gradients_modes); False: the position-major kernels
with a stored far plane, as until round 6."""


def mode_groups(S, pw, det, eigen_modes=0):
    """((first mode, count), ...) for `tike_ifft2_pass2_gradients_modes`, or
    () where the shape is not served in groups: every group is what one launch
    holds in registers (<= 8 modes, <= 4 at 512^2), balanced, at least two
    modes each; the eigen probes belong to the first."""
    cap = 4 if det == 512 else 8
    if not (MODE_GROUPS and pw == det and det in POSITION_MAJOR_SIZES
            and cap < S <= 4 * cap):
        return ()
    k = -(-S // cap)
    base, extra = divmod(S, k)
    sizes = [base + (i < extra) for i in range(k)]
    if min(sizes) < 2 or eigen_modes > sizes[0]:
        return ()
    firsts = [sum(sizes[:i]) for i in range(k)]
    return tuple(zip(firsts, sizes))


PFA_SUBTILES_IN_LDS = True
"""Prime-factor sizes with 128^2 sub-tiles (384, 640, 896): gather and forward
sub-tile transform in one launch, the sub-tile inside LDS
(tike_pfa_fwd_subtiles); False: tike_pfa_fwd_gather + tike_pfa_fft2."""


def pfa_gradients(S, pw, det):
    """True where the prime-factor launches serve (csrc/pfa.hip)."""
    return bool(PFA_ROUTE and lib.tike_pfa_supported(S, pw, det))


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
    # >= 2048 tiles, or 8 GiB of far plane (256 MiB until late round 6: on a
    # 288 GB part a whole minibatch per launch is affordable and measured
    # faster wherever the old bound split one -- 100^2 x 4 modes 567 -> 602 k
    # patterns/s, 100^2 x 8 320 -> 334 k, 200^2 x 8 90.2 -> 92.0 k, 300^2 x 8
    # 36.4 -> 37.1 k; an 838 + 162 split is the worst case)
    tiles = max(2048, (1 << 33) // (det * det * 8))
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

EIGEN_SUMS_RECOMPUTE = _tuning.eigen_sums_gather
"""The position sums gather too, through the two-positions-per-workgroup row
walk of tike_eigen_position_sums1 (one 16-byte tap load per pixel and
position, E_0 and the probe update loaded once per pair): 0.215 -> 0.176 ms
per 1000 positions at 256^2."""

STATS_PATCH_RECOMPUTE = _tuning.stats_gather
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


def _once(make):
    """`make()` evaluated at the first call, the same value afterwards."""
    box = []

    def get():
        if not box:
            box.append(make())
        return box[0]

    return get


def _get_nearplane_gradients(data, psi, scan, probe, eigen_probe,
                             eigen_weights, lo, hi, comm, *, num_batch,
                             exitwave_options, op, recover_psi, recover_probe,
                             position_terms=None, need_chi0=True):
    """Object / probe gradients of one minibatch (lstsq.py:367-602).  Which
    kernels a chunk runs is the GradientPlan of the shape (`_plan.py`, cached
    on the operator like the reference's FFT plan, cache.py:32-46); this
    function walks the chunks and completes the sums over the ranks.
    need_chi0=False (cgrad: no step statistics follow) skips the store of
    mode 0 of chi where the fused pass 2 would be its only producer."""
    dev = psi.device
    B = hi - lo
    S, pw = probe.shape[-3], probe.shape[-1]
    det = op.detector_shape
    H, W = psi.shape[-2:]
    if exitwave_options.noise_model not in _MODELS:
        raise ValueError(
            f"unknown noise model {exitwave_options.noise_model!r}")
    nmeasured, mask_u8 = mask_info(exitwave_options, det)
    fwd_scale, inv_scale = fft_scales(det, op.norm)

    # the weights this gradient uses; every consumer (forward, gradients, step
    # statistics) runs before _update_nearplane changes them in place
    w_old = None if eigen_weights is None else eigen_weights[lo:hi]
    ep, _, C, Sm = _eigen_args(eigen_probe, w_old)
    plan = GradientPlan.for_(op, S, pw, det, exitwave_options, mask_u8,
                             eigen_modes=Sm, num_eigen=C)

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
    buf = plan.buffers(
        _workspace(op), B, dev, varying=Sm if w_old is not None else 0,
        want_patches=bool((recover_probe and eigen_weights is not None)
                          or position_terms))
    if position_terms:
        taps, taps_r = gaussian_derivative_taps(sigma=0.333)
    # what every chunk shares
    c = SimpleNamespace(
        op=op, data=data, psi=psi, probe=probe, ep=ep, eigen_probe=eigen_probe,
        C=C, Sm=Sm, H=H, W=W, st=A.stream_ptr(), fwd_scale=fwd_scale,
        inv_scale=inv_scale, nmeasured=nmeasured, mask_u8=mask_u8,
        unmeasured=float(exitwave_options.unmeasured_pixels_scaling),
        step_start=float(exitwave_options.step_length_start),
        step_weight=float(exitwave_options.step_length_weight), buf=buf,
        m_probe_update=m_probe_update, num_batch=num_batch,
        recover_psi=recover_psi, need_chi0=need_chi0)
    chunk = plan.chunk
    for clo in range(lo, hi, chunk):
        chi_hi = min(hi, clo + chunk)
        n, blo = chi_hi - clo, clo - lo
        rows = slice(blo, blo + n)
        k = SimpleNamespace(
            n=n, scan=scan[clo:chi_hi], data=data[clo:chi_hi],
            w=None if w_old is None else w_old[rows], uq=None,
            costs=buf.costs[rows],
            patches=None if buf.patches is None else buf.patches[rows],
            chi0=None if buf.chi0 is None else buf.chi0[rows])
        # float32 view of the chunk for the kernels without a 16-bit loader
        # (the 256^2 gaussian hot path reads uint16 directly), made on demand
        k.data_f32 = _once(lambda a=clo, b=chi_hi: A.data_f32(data, a, b))
        if k.w is not None and Sm > 0 and buf.unique is not None and (
                not plan.no_farplane):
            # varying probe of the modes that own eigen probes, once per chunk
            # (the 256^2 and the shape-general kernels form it on the fly)
            k.uq = buf.unique[:n]
            check(
                lib.tike_varying_probe(A.ptr(probe), A.ptr(ep), A.ptr(k.w), C,
                                       Sm, A.ptr(k.uq), n, S, pw, c.st),
                "varying probe")
        plan.forward(c, k)
        plan.gradients(c, k)
        stored_chi0 = plan.fused or plan.general
        if position_terms:
            chi_m, chi_modes = ((k.chi0, 1) if stored_chi0 else (buf.chi, S))
            check(
                lib.tike_position_sums(
                    A.ptr(k.patches), A.ptr(chi_m), chi_modes, A.ptr(probe),
                    A.ptr(ep), A.ptr(k.w), C, Sm, taps.ctypes.data, taps_r,
                    A.ptr(position_terms[0][clo:chi_hi]),
                    A.ptr(position_terms[1][clo:chi_hi]), n, S, pw, c.st),
                "position shift sums")
        if chi_hi == hi and early:
            # the probe gradient is complete: its slice of the flat buffer
            # travels while the last object scatter runs
            probe_sum = comm.Allreduce_start(grads[n_obj:])
        if recover_psi:
            check(
                lib.tike_scatter_patches(A.ptr(buf.objproj), A.ptr(k.scan),
                                         A.ptr(obj_acc), n, pw, H, W, c.st),
                "object scatter")
        if not buf.single_chunk and not stored_chi0:
            buf.chi0[rows] = buf.chi[:n, 0, 0]

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
    if recover_probe and not (plan.fused or plan.general):
        m_probe_update = m_probe_update / num_batch  # (else: in the kernel)
    return dict(chi0=buf.chi if buf.single_chunk else buf.chi0[:B],
                chi_modes=S if buf.single_chunk else 1, w_old=w_old,
                patches=None if buf.patches is None else buf.patches[:B],
                object_acc=obj_acc, m_probe_update=m_probe_update,
                costs=buf.costs[:B], count=count, local_count=B)


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
