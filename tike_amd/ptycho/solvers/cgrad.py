"""Conjugate-gradient ptychography solver.

The reference snapshot ships no ptychography ``cgrad``; it is
composed here, as BASELINE's configs ask, from the reference's own pieces:
``tike.opt.conjugate_gradient`` (opt.py:312-380: Dai-Yuan direction,
backtracking line search), the gaussian cost ``Ptycho.cost`` (ptycho.py:193-204)
and the gradient ``Ptycho.adj(gaussian_grad(...))`` (objective.py:31-44),
following the multi-GPU pattern of lamino/solvers/cgrad.py:58-92: the cost and
the gradient are summed over ranks, every rank then takes the same step.
"""
import logging

import numpy as np
import torch

from ... import _tuning
from ... import _arrays as A
from ... import opt
from ..._lib import check, lib
from ...operators.propagation import fft_scales
from ..exitwave import ExitWaveOptions
from .lstsq import (SPLIT_FORWARD_SIZES, _get_nearplane_gradients, _workspace,
                    chunk_positions, fused_gradients, global_count,
                    minibatch_key)


logger = logging.getLogger(__name__)

_GAUSSIAN = {}


def _gaussian_options(det):
    """Every pixel measured, gaussian noise: the cost cgrad minimises."""
    if det not in _GAUSSIAN:
        _GAUSSIAN[det] = ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool))
    return _GAUSSIAN[det]


class _CostPlan:
    """Everything a line-search probe (cost only) of one minibatch needs that
    does not change between probes: workspaces, chunk bounds, raw pointers of
    the scan / data / cost slices.  A probe is then two C-ABI calls per chunk
    and one reduction -- at BASELINE configs[0] (256 positions of 128^2) the
    Python between the launches was most of the epoch."""

    def __init__(self, op, data, scan, lo, hi, S, pw, H, W, dev):
        det = op.detector_shape
        N = hi - lo
        ws = _workspace(op)
        self.split = det in SPLIT_FORWARD_SIZES
        chunk = chunk_positions(S, det, self.split)
        self.far = ws.get("far", (min(chunk, max(N, 1)), 1, S, det, det),
                          torch.complex64, dev)
        # detector sizes p x 2^k (round 6): the prime-factor launches of the
        # gradient pipeline with nothing but the costs stored -- sub-tile
        # transforms + the p x p combine, no far plane in natural order
        from . import lstsq as L
        self.pfa = (not self.split and L.PFA_ROUTE and L.GENERAL_FUSED
                    and L.pfa_gradients(S, pw, det))
        self.pfa_lds = bool(self.pfa and L.PFA_SUBTILES_IN_LDS
                            and lib.tike_pfa_fwd_subtiles_supported(S, pw, det))
        if self.pfa_lds:
            self.aux = ws.get("pfa_probe", (S, det, det), torch.complex64, dev)
        elif self.pfa:
            self.aux = ws.get("mid", tuple(self.far.shape), torch.complex64,
                              dev)
        self.costs = ws.get("costs", (max(N, 1),), torch.float32, dev)[:N]
        self.fwd_scale = fft_scales(det, op.norm)[0]
        self.dims = (S, pw, det, H, W)
        self.u16 = int(data.dtype == torch.uint16)
        # resident data that the cost kernel of this size reads as it is
        direct = isinstance(data, torch.Tensor) and (
            self.split or data.dtype == torch.float32)
        self.chunks = []
        for clo in range(lo, hi, chunk):
            chi = min(hi, clo + chunk)
            self.chunks.append(
                (clo, chi, scan[clo:chi], data[clo:chi] if direct else None,
                 self.costs[clo - lo:chi - lo]))

    def run(self, data, psi, probe):
        S, pw, det, H, W = self.dims
        st = A.stream_ptr()
        ppsi, pprobe, pfar = A.ptr(psi), A.ptr(probe), A.ptr(self.far)
        for clo, chi, sc, d, cost in self.chunks:
            n = chi - clo
            if d is None:  # streamed from the host, or 16-bit counts at 128^2
                d = data[clo:chi] if self.split else A.data_f32(data, clo, chi)
            if self.split:
                check(
                    lib.tike_fwd_pass1(ppsi, sc.data_ptr(), pprobe, 0, None,
                                       None, None, 0, 0, pfar, None, n, S, pw,
                                       det, H, W, st), "cgrad forward pass 1")
                check(
                    lib.tike_fwd_gradient_scale(
                        pfar, d.data_ptr(), self.u16, None, None, None,
                        cost.data_ptr(), None, n, S, det, self.fwd_scale, 0,
                        1.0, det * det, st), "cgrad forward pass 2 + cost")
            elif self.pfa:
                if self.pfa_lds:
                    check(
                        lib.tike_pfa_fwd_subtiles(
                            ppsi, sc.data_ptr(), pprobe, None, None, 0, 0,
                            A.ptr(self.aux), pfar, None, n, S, pw, det, H, W,
                            st), "cgrad forward (sub-tiles in LDS)")
                else:
                    check(
                        lib.tike_pfa_fwd_gather(
                            ppsi, sc.data_ptr(), pprobe, 0, None, None, None,
                            0, 0, A.ptr(self.aux), None, n, S, pw, det, H, W,
                            st), "cgrad forward (prime-factor gather)")
                    check(lib.tike_pfa_fft2(A.ptr(self.aux), pfar, n * S, det,
                                            0, st), "cgrad sub-tile transforms")
                check(
                    lib.tike_pfa_combine_gradient(
                        pfar, d.data_ptr(), None, cost.data_ptr(), n, S, det,
                        self.fwd_scale, 0, 1.0, det * det, 0, st),
                    "cgrad cost (p x p combine)")
            else:
                check(
                    lib.tike_ptycho_fwd(ppsi, sc.data_ptr(), pprobe, 0, None,
                                        None, 0, 0, pfar, n, S, pw, det, H, W,
                                        self.fwd_scale, 0, st), "cgrad forward")
                check(
                    lib.tike_farplane_gradient(pfar, d.data_ptr(), None, None,
                                               cost.data_ptr(), n, S, det, 0,
                                               0, 1.0, det * det, st),
                    "cgrad cost")
        return self.costs

    def supports_gradients(self):
        """The far-plane-free sizes and 128^2 (float32 data), probe window =
        detector: the gradient of a chunk is ONE C-ABI call
        (tike_lstsq_chunk_gradients)."""
        S, pw, det, _, _ = self.dims
        return ((self.split or (det == 128 and not self.u16)) and pw == det
                and fused_gradients(S, pw, det)
                and all(c[3] is not None for c in self.chunks))

    def gradients(self, op, comm, psi, probe, want_psi, want_probe):
        """(costs, -d cost / d psi or None, -d cost / d probe or None), the
        sums over the positions of all ranks -- what _get_nearplane_gradients
        returns for this case, without its per-call set-up."""
        S, pw, det, H, W = self.dims
        dev = psi.device
        ws = _workspace(op)
        n_max = self.far.shape[0]
        mid = ws.get("mid", tuple(self.far.shape), torch.complex64, dev)
        # (128^2 keeps the far plane: factor table + intensity table)
        gscale = ws.get("gscale", ((2 if det == 128 else 1) * n_max, det, det),
                        torch.float32, dev)
        N = self.costs.shape[0]
        patches = ws.get("patches", (max(N, 1), pw, pw), torch.complex64, dev)
        objproj = ws.get("objproj", (n_max, pw, pw), torch.complex64, dev)
        n_obj = 2 * H * W if want_psi else 0
        n_prb = 2 * probe.numel() if want_probe else 0
        grads = torch.zeros(n_obj + n_prb, dtype=torch.float32, device=dev)
        acc = grads[:n_obj].view(2, H, W) if want_psi else None
        mpu = (torch.view_as_complex(grads[n_obj:].view(*probe.shape, 2))
               if want_probe else None)
        _, inv_scale = fft_scales(det, op.norm)
        st = A.stream_ptr()
        lo = self.chunks[0][0]
        for clo, chi, sc, d, cost in self.chunks:
            n = chi - clo
            check(
                lib.tike_lstsq_chunk_gradients(
                    A.ptr(psi), sc.data_ptr(), A.ptr(probe), None, None, 0, 0,
                    d.data_ptr(), self.u16, None, 0, 1.0, det * det,
                    A.ptr(self.far), A.ptr(mid), A.ptr(gscale),
                    A.ptr(patches[clo - lo:chi - lo]), cost.data_ptr(),
                    A.ptr(objproj) if want_psi else None, None, A.ptr(mpu), 1.0,
                    A.ptr(acc), n, S, det, H, W, self.fwd_scale, inv_scale,
                    st), "cgrad gradients")
        if comm.collective and grads.numel():
            comm.Allreduce(grads)
        return self.costs, acc, mpu


def _cost_and_grad(op, comm, data, psi, scan, probe, lo, hi, *, want_psi,
                   want_probe, want_grad, read_cost=True, plan=None):
    """Global gaussian cost (mean over all positions and pixels) of the
    minibatch [lo, hi) and, optionally, d cost / d psi and d cost / d probe
    (unnormalised adjoints).

    The gradient is the one lstsq_grad forms (the same kernels, whatever the
    detector size); a line-search probe (cost only) at 256^2 / 512^2 is the
    split forward of that pipeline with nothing but the costs stored."""
    dev = psi.device
    N = hi - lo
    S, pw = probe.shape[-3], probe.shape[-1]
    det = op.detector_shape
    H, W = psi.shape[-2:]
    if want_grad and plan is not None and plan.supports_gradients():
        costs, acc, mpu = plan.gradients(op, comm, psi, probe, want_psi,
                                         want_probe)
        gpsi = -torch.complex(acc[0], acc[1])[None] if want_psi else None
        gprobe = -mpu if want_probe else None
    elif want_grad:
        g = _get_nearplane_gradients(
            data, psi, scan, probe, None, None, lo, hi, comm, num_batch=1,
            exitwave_options=_gaussian_options(det), op=op,
            recover_psi=want_psi, recover_probe=want_probe, need_chi0=False)
        costs = g["costs"]
        gpsi = gprobe = None
        if want_psi:
            gpsi = -torch.complex(g["object_acc"][0], g["object_acc"][1])[None]
        if want_probe:
            gprobe = -g["m_probe_update"]
    else:
        gpsi = gprobe = None
        plan = plan or _CostPlan(op, data, scan, lo, hi, S, pw, H, W, dev)
        costs = plan.run(data, psi, probe)
    total = costs.sum(dtype=torch.float64)  # device scalar (this rank)
    if read_cost:
        return _finish_cost(total, comm, op, lo, hi), gpsi, gprobe
    return total, gpsi, gprobe


def _finish_cost(total, comm, op, lo, hi):
    """Mean cost over the positions of ALL ranks, on the host: one (all-)
    reduction and one read-back."""
    if comm.collective:
        total = comm.Allreduce_scalars([total], total.device)[0]
    return float(total.item()) / global_count(comm, op, lo, hi)


DEVICE_LINE_SEARCH = True
"""Tests set this to False to run opt.conjugate_gradient with the host-side
line search (one read-back per trial) everywhere."""

LINE_SEARCH_SLOTS = (8, 4)
"""Step lengths enqueued ahead per line search to begin with: first CG
iteration of a call (it starts from `step_length`), later iterations (they
start from the length accepted last).  A reconstruction then learns what its
searches need (`_SlotPolicy`)."""

MAX_SLOTS = 30  # tike_cgrad_line_search's limit: step_length / 2^29

LINEAR_LINE_SEARCH = True
"""The far plane is linear in the variable a search moves along, so ONE forward
pass of the direction and one pass over two hand-offs give the costs of 16
step lengths at once (tike_cgrad_line_search_linear): same candidates, same
acceptance rule, results equal to the trial-by-trial search up to float32
rounding, for about half the work (c2: 1.1 ms of trials per search -> 0.6 ms).
False: the trial-by-trial device search (tike_cgrad_line_search)."""

LINEAR_STEPS = 16  # TK_LS_STEPS x TK_LS_PASSES of csrc/ptycho.hip


class _SlotPolicy:
    """How many trial step lengths to enqueue ahead, per variable (object,
    probe) and kind of search (first of a call, later ones), learnt from the
    trials the previous calls of this reconstruction needed: one more than the
    largest number seen; a count that has been two or more too generous for
    eight calls in a row shrinks by one.  A skipped slot costs about 20 us, a
    search that runs out of slots costs the whole CG call again -- so a
    problem whose steps shrink below step_length / 2^7 pays for that once,
    not in every call of every epoch (round-3 advisor finding).  Counts change
    rarely, which keeps the captured launch sequences (`_CgGraph`) valid."""

    # calls of a variable that skip the all-at-once search after it has found
    # none of its 16 step lengths acceptable (doubled per repeated failure)
    LINEAR_PAUSE = 16

    def __init__(self):
        self.slots = {v: list(LINE_SEARCH_SLOTS) for v in (0, 1)}
        self.generous = {v: [0, 0] for v in (0, 1)}
        self.linear_skip = {0: 0, 1: 0}
        self.linear_pause = {0: self.LINEAR_PAUSE, 1: self.LINEAR_PAUSE}

    def linear_allowed(self, variable):
        """The all-at-once search keeps no steps below step / 2^15: a problem
        that needs them would run it, throw it away and repeat the whole CG
        call trial by trial -- in every call (round-4 advisor finding).  After
        a failure the variable goes straight to the trial-by-trial search for
        a while, then the all-at-once search gets another try."""
        if self.linear_skip[variable] > 0:
            self.linear_skip[variable] -= 1
            return False
        return True

    def linear_result(self, variable, ok):
        if ok:
            self.linear_pause[variable] = self.LINEAR_PAUSE
        else:
            self.linear_skip[variable] = self.linear_pause[variable]
            self.linear_pause[variable] = min(1024,
                                              2 * self.linear_pause[variable])

    def get(self, variable):
        return tuple(self.slots[variable])

    def learn(self, variable, trials_per_search):
        """trials_per_search: trials each search of a successful call made."""
        seen = (trials_per_search[0], max(trials_per_search[1:], default=1))
        for kind in (0, 1):
            want = max(min(MAX_SLOTS, int(seen[kind]) + 1),
                       LINE_SEARCH_SLOTS[kind])
            have = self.slots[variable][kind]
            if want > have:
                self.slots[variable][kind] = want
                self.generous[variable][kind] = 0
            elif want <= have - 2:
                self.generous[variable][kind] += 1
                if self.generous[variable][kind] >= 8:
                    self.slots[variable][kind] = have - 1
                    self.generous[variable][kind] = 0
            else:
                self.generous[variable][kind] = 0

    def widen(self, variable):
        """After a search ran out of slots: every slot the entry allows."""
        self.slots[variable] = [MAX_SLOTS, MAX_SLOTS]
        self.generous[variable] = [0, 0]


def _slot_policy(op):
    policy = getattr(op, "_cgrad_slot_policy", None)
    if policy is None:
        policy = op._cgrad_slot_policy = _SlotPolicy()
    return policy


def _cg_enqueue(plan, op, comm, x, other, variable, num_iter, step_init,
                count, data, scan, lo, hi, slots, bufs, linear=False):
    """Enqueue one conjugate-gradient call -- opt.conjugate_gradient
    (opt.py:312-380: Dai-Yuan directions, backtracking line search) for the
    object (variable 0) or the probe (variable 1) with every line search
    decided on the device (tike_cgrad_line_search): per iteration the gradient
    pass, the direction (tike_cgrad_direction) and up to `slots` cost-only
    trials, no host round trip, nothing but launches (so the whole call can be
    captured as a graph).  step_init: device double[5] {0, step_length, 0, 0,
    0}; bufs: two iterates' worth of scratch.  Returns (the last iterate,
    device double[5 + num_iter]: the search state { fx, step, done, trials,
    failures } followed by the running total of trials after every search)."""
    dev = x.device
    S, pw, det, H, W = plan.dims
    N = hi - lo
    # carried from search to search on the device (the step accepted last is
    # the first one tried next, opt.py:366-371)
    out = torch.zeros(5 + num_iter, dtype=torch.float64, device=dev)
    state = out[:5]
    state.copy_(step_init)
    skip = torch.zeros(1, dtype=torch.int32, device=dev)
    scan_ptr = scan[lo:hi].data_ptr()
    data_ptr = data[lo:hi].data_ptr()
    st_ptr = A.stream_ptr()
    # gradient and direction of the previous iteration, and the four sums of
    # tike_cgrad_direction: one entry per iteration instead of a dozen torch
    # launches (negation, two reductions, the Dai-Yuan update, the cost at x)
    gradient, d = torch.empty_like(x), torch.empty_like(x)
    sums = torch.empty(4, dtype=torch.float64, device=dev)
    if linear:
        ws = _workspace(op)
        far_b = ws.get("far_b", tuple(plan.far.shape), torch.complex64, dev)
        costs_k = ws.get("costs_k", ((LINEAR_STEPS + 1) * max(N, 1) + 1,),
                         torch.float32, dev)
        # one chunk: the gradient pass leaves the forward hand-off of x (at
        # 128^2 its far plane) in plan.far, which the search reads as it is
        a_valid = int(len(plan.chunks) == 1)
        row_sums = torch.zeros(LINEAR_STEPS + 1, dtype=torch.float64,
                               device=dev)
    for i in range(num_iter):
        a, b = (x, other) if variable == 0 else (other, x)  # psi, probe
        costs, acc, mpu = plan.gradients(op, comm, a, b, variable == 0,
                                         variable == 1)
        check(
            lib.tike_cgrad_direction(
                A.ptr(acc) if variable == 0 else None,
                None if variable == 0 else A.ptr(mpu), A.ptr(gradient),
                A.ptr(d), x.numel(), int(i == 0), A.ptr(costs),
                costs.numel(), count, A.ptr(state), A.ptr(sums), st_ptr),
            "cgrad direction")
        xs = bufs[i % 2]
        if linear:
            # one rank: the whole search in one call; several: the cost sums of
            # each pass are all-reduced between the pass and its decision
            for stage in ((1, 2, 3, 4) if comm.collective else (0,)):
                check(
                    lib.tike_cgrad_line_search_linear(
                        variable, A.ptr(x), A.ptr(d), A.ptr(xs), A.ptr(other),
                        scan_ptr, data_ptr, plan.u16, A.ptr(plan.far), a_valid,
                        A.ptr(far_b), A.ptr(costs_k), N, plan.far.shape[0], S,
                        det, H, W, plan.fwd_scale, count, A.ptr(state), stage,
                        A.ptr(row_sums), st_ptr),
                    "cgrad line search (all steps at once)")
                if stage in (1, 3):
                    comm.Allreduce_f64(row_sums)
        else:
            check(
                lib.tike_cgrad_line_search(
                    variable, A.ptr(x), A.ptr(d), A.ptr(xs), A.ptr(other),
                    scan_ptr, data_ptr, plan.u16, A.ptr(plan.far),
                    A.ptr(plan.costs), N, plan.far.shape[0], S, det, H, W,
                    plan.fwd_scale, count, A.ptr(state), A.ptr(skip),
                    slots[0 if i == 0 else 1], st_ptr),
                "cgrad line search")
        out[5 + i].copy_(state[3])
        x = xs
    return x, out


def _cg_result(x, out):
    """Read a call's state back (the ONE host synchronisation of a call):
    (x, mean cost, trials made by every search), or None when a search ran
    out of slots."""
    final = out.cpu().numpy()
    if final[4] != 0:
        return None
    return x, float(final[0]), np.diff(final[5:], prepend=0.0)


def _step_init(step_length, dev):
    return torch.from_numpy(
        np.array([0.0, float(step_length), 0.0, 0.0, 0.0])).to(dev)


def _cg_device(plan, op, comm, psi, probe, variable, num_iter, step_length,
               count, data, scan, lo, hi, slots=LINE_SEARCH_SLOTS,
               linear=False):
    """`_cg_enqueue` launched eagerly + its read-back.  Returns (x, mean
    cost, trials made by every search), or None when a search ran out of its
    slots -- the caller then repeats the call with more slots or with the
    host-side search, which has no limit."""
    x = psi if variable == 0 else probe
    other = probe if variable == 0 else psi
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    return _cg_result(*_cg_enqueue(
        plan, op, comm, x, other, variable, num_iter,
        _step_init(step_length, x.device), count, data, scan, lo, hi, slots,
        bufs, linear=linear))


USE_GRAPHS = _tuning.cgrad_graphs
"""A conjugate-gradient call is a fixed sequence of ~10 launches per trial
slot whose only data-dependent control flow lives on the device (the `skip`
word), so it can be captured once and replayed as a HIP graph from its second
occurrence on (`_CgGraph`).  MEASURED SLOWER on ROCm 7.2 / gfx950 and therefore
OFF unless TIKE_CGRAD_GRAPHS=1: BASELINE configs[0] (256 positions, every
launch shorter than its own issue) 73.4 k patterns/s launched one by one,
35.7 k replayed -- a replay of ~270 nodes costs ~6 us per node on the GPU
side, more than eager launches that the host issues ahead; c2 67.8 k vs 66.6 k
(profiles/r04_experiments.md).  Kept as an option and under test."""

MAX_GRAPHS = 64


class _CgGraph:
    """The captured launch sequence of one `_cg_enqueue` call.  Inputs are
    copied into static buffers, the graph is replayed, the result is cloned
    out (the buffers belong to the graph)."""

    def __init__(self, enqueue, x, other, step_length):
        self.x = torch.empty_like(x)
        self.other = torch.empty_like(other)
        self.bufs = [torch.empty_like(x), torch.empty_like(x)]
        self.init = _step_init(step_length, x.device)
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph):
            self.result, self.out = enqueue(self.x, self.other, self.init,
                                            self.bufs)

    def __call__(self, x, other):
        self.x.copy_(x)
        self.other.copy_(other)
        self.graph.replay()
        return self.result.clone(), self.out


def _every_rank(comm, op, lo, hi, mine):
    """True when `mine` holds on every rank (asked once per minibatch and
    reconstruction: minibatch sizes and data placement are static)."""
    cache = op.__dict__.setdefault("_tike_amd_every_rank", {})
    key = minibatch_key(comm, lo, hi)
    if key not in cache:
        cache[key] = comm.Allreduce_count(int(bool(mine))) == comm.size
    return cache[key]


def _cg_on_device(plan, op, comm, psi, probe, variable, o, count, data, scan,
                  lo, hi):
    """One CG call on the device with the slot counts this reconstruction has
    learnt, replayed from a graph once the same call has been seen before; a
    call whose search runs out of slots is repeated once with every slot the
    entry allows before the host-side search takes over.  Returns (x, cost)
    or None."""
    policy = _slot_policy(op)
    x = psi if variable == 0 else probe
    other = probe if variable == 0 else psi
    if LINEAR_LINE_SEARCH and not USE_GRAPHS:
        # (the policy is fed by results every rank sees alike -- the staged
        # search decides on all-reduced sums -- so the ranks stay in step)
        r = None
        if policy.linear_allowed(variable):
            r = _cg_device(plan, op, comm, psi, probe, variable, o.cg_iter,
                           o.step_length, count, data, scan, lo, hi,
                           linear=True)
            policy.linear_result(variable, r is not None)
        if r is not None:
            return r[0], r[1]
        if comm.collective:
            return None  # (the host-side search sums its costs over the ranks)
        # a search found none of its 16 step lengths acceptable (steps below
        # step / 2^15): the trial-by-trial search below reaches 2^-29
        policy.widen(variable)
    graphs = getattr(op, "_cgrad_graphs", None)
    if graphs is None:
        graphs = op._cgrad_graphs = {}
    for attempt in range(2):
        slots = policy.get(variable)

        def enqueue(x_, other_, init, bufs):
            return _cg_enqueue(plan, op, comm, x_, other_, variable,
                               o.cg_iter, init, count, data, scan, lo, hi,
                               slots, bufs)

        key = (variable, lo, hi, slots, tuple(x.shape), tuple(other.shape),
               o.cg_iter, float(o.step_length), float(count),
               plan.far.data_ptr(), plan.costs.data_ptr(), scan.data_ptr(),
               data.data_ptr())
        seen = graphs.get(key) if USE_GRAPHS else "eager"
        if seen is False:  # second occurrence: capture
            try:
                seen = graphs[key] = _CgGraph(enqueue, x, other, o.step_length)
            except Exception as e:  # noqa: BLE001 -- capture is an optimisation
                logger.warning("cgrad: graph capture failed (%s); this call "
                               "stays eager", e)
                seen = graphs[key] = "eager"
        if seen is None or seen == "eager":
            # first occurrence: eager (it also warms every lazily created
            # table and workspace a capture must not allocate)
            if seen is None:
                if len(graphs) >= MAX_GRAPHS:
                    graphs.clear()
                graphs[key] = False
            r = _cg_device(plan, op, comm, psi, probe, variable, o.cg_iter,
                           o.step_length, count, data, scan, lo, hi,
                           slots=slots)
        else:
            r = _cg_result(*seen(x, other))
        if r is not None:
            policy.learn(variable, r[2])
            return r[0], r[1]
        if slots == (MAX_SLOTS, MAX_SLOTS):
            break
        policy.widen(variable)
    return None


class _Evaluator:
    """cost / gradient callbacks of one conjugate-gradient call.  The gradient
    pass forms the cost of its argument as well: it is kept ON THE DEVICE and
    read back only if the line search asks for the cost of that very array
    (opt.line_search does, for its starting point) -- which then costs neither
    a forward pass nor, for the other gradient evaluations, a host
    synchronisation."""

    def __init__(self, run, finish):
        self._run, self._finish = run, finish
        self._x = self._total = None

    def cost(self, x):
        if x is self._x:
            return self._finish(self._total)
        return self._finish(self._run(x, False)[0])

    def grad(self, x):
        total, g = self._run(x, True)
        self._x, self._total = x, total
        return [g]


def cgrad(parameters, data, batches, comm, *, op, epoch):
    """One epoch: for every minibatch, `cg_iter` CG iterations on psi and
    then (when probe recovery is on) on the probe."""
    o = parameters.algorithm_options
    if parameters.eigen_probe is not None or parameters.eigen_weights is not None:
        raise NotImplementedError("cgrad does not support eigen probes")
    recover_psi = parameters.object_options is not None
    recover_probe = (parameters.probe_options is not None
                     and epoch >= parameters.probe_options.update_start)
    psi, probe, scan = parameters.psi, parameters.probe, parameters.scan
    batch_cost = []
    for batch_index, b in enumerate(batches):
        lo = int(b[0]) if len(b) else 0
        hi = lo + len(b)
        comm.minibatch = batch_index
        d, s = data, scan
        cost = None
        finish = lambda total: _finish_cost(total, comm, op, lo, hi)
        plan = _CostPlan(op, d, s, lo, hi, probe.shape[-3], probe.shape[-1],
                         psi.shape[-2], psi.shape[-1], psi.device)
        # line searches decided on the device: one rank, HBM-resident data,
        # the far-plane-free sizes and 128^2
        # (several ranks: the all-at-once search, whose cost sums are
        # all-reduced between its cost passes and its decisions -- every rank
        # must take the same route, so every rank must hold positions)
        on_device = (DEVICE_LINE_SEARCH and hi > lo
                     and isinstance(d, torch.Tensor)
                     and plan.supports_gradients())
        if comm.collective:
            on_device = (LINEAR_LINE_SEARCH and not USE_GRAPHS
                         and _every_rank(comm, op, lo, hi, on_device))
        count = global_count(comm, op, lo, hi)
        done_psi = done_probe = False
        if recover_psi and on_device:
            r = _cg_on_device(plan, op, comm, psi, probe, 0, o, count, d, s,
                              lo, hi)
            if r is not None:
                psi, cost = r
                done_psi = True
        if recover_psi and not done_psi:
            def run(x, want_grad):
                r = _cost_and_grad(op, comm, d, x, s, probe, lo, hi,
                                   want_psi=True, want_probe=False,
                                   want_grad=want_grad, read_cost=False,
                                   plan=plan)
                return r[0], r[1]
            ev = _Evaluator(run, finish)
            psi, cost = opt.conjugate_gradient(
                torch, x=psi, cost_function=ev.cost, grad=ev.grad,
                dir_multi=lambda x: x[0], num_iter=o.cg_iter,
                step_length=o.step_length)
        if recover_probe and on_device:
            r = _cg_on_device(plan, op, comm, psi, probe, 1, o, count, d, s,
                              lo, hi)
            if r is not None:
                probe, cost = r
                done_probe = True
        if recover_probe and not done_probe:
            def run(x, want_grad):
                r = _cost_and_grad(op, comm, d, psi, s, x, lo, hi,
                                   want_psi=False, want_probe=True,
                                   want_grad=want_grad, read_cost=False,
                                   plan=plan)
                return r[0], r[2]
            ev = _Evaluator(run, finish)
            probe, cost = opt.conjugate_gradient(
                torch, x=probe, cost_function=ev.cost, grad=ev.grad,
                dir_multi=lambda x: x[0], num_iter=o.cg_iter,
                step_length=o.step_length)
        if cost is None:
            cost = _cost_and_grad(op, comm, d, psi, s, probe, lo, hi, want_psi=False,
                                  want_probe=False, want_grad=False)[0]
        batch_cost.append(cost)
    if any(isinstance(c, torch.Tensor) for c in batch_cost):
        # device-side searches leave the cost on the device: one read-back
        batch_cost = torch.stack([
            c.to(torch.float64) if isinstance(c, torch.Tensor) else
            torch.tensor(float(c), dtype=torch.float64, device=psi.device)
            for c in batch_cost
        ]).cpu().numpy()
    o.costs.append([float(np.mean(batch_cost))])
    parameters.psi, parameters.probe = psi, probe
    return parameters
