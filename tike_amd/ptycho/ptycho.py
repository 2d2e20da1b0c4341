"""Ptychography reconstruction driver
(reference src/tike/ptycho/ptycho.py:95-972).

``reconstruct`` / ``Reconstruction`` / ``simulate`` keep the reference's
signatures.  Execution model (MI355X-first, differs from the reference):

* one process per GPU.  ``reconstruct(..., num_gpu=N)`` called from a plain
  process starts N child ranks itself (``_spawn.py``) and returns their
  result; inside a ``torch.distributed`` job (torchrun) the world size is the
  number of GPUs and a ``num_gpu`` that disagrees with it is an error;
* the whole dataset is HBM-resident (no pinned-host chunk streaming,
  reference communicators/stream.py:285-404);
* scan positions -- not object stripes -- are sharded across ranks and the
  object/probe gradients are all-reduced every minibatch (RCCL), so there is
  no probe averaging or stripe-edge blending after an epoch
  (reference ptycho.py:474-502).
"""
import copy
import logging
import threading
import time
import typing
import warnings

import numpy as np
import torch

from .. import _arrays as A
from .. import cluster
from .. import precision
from ..communicators import Comm
from ..operators import Ptycho
from . import _spawn, solvers
from .object import (positivity_constraint, remove_object_ambiguity,
                     smoothness_constraint)
from .position import (affine_position_regularization,
                       check_allowed_positions,
                       estimate_global_transformation_ransac, ransac_subsets)
from .probe import (apply_median_filter_abs_probe, constrain_center_peak,
                    constrain_probe_sparsity, constrain_variable_probe,
                    finite_probe_support, get_varying_probe, orthogonalize_eig,
                    power as probe_power,
                    rescale_probe_using_fixed_intensity_photons)
from .solvers.lstsq import chunk_positions, mask_info
from .._lib import check, lib

logger = logging.getLogger(__name__)


def _intensity_chunks(operator, psi, scan, probe, eigen_probe=None,
                      eigen_weights=None):
    """Yield (lo, hi, intensity (n, det, det)) over chunks of positions."""
    N = scan.shape[0]
    S = probe.shape[-3]
    det = operator.detector_shape
    chunk = chunk_positions(S, det)
    for lo in range(0, N, chunk):
        hi = min(N, lo + chunk)
        w = None if eigen_weights is None else eigen_weights[lo:hi]
        if psi.shape[0] > 1:  # several slices: the Multislice composition
            far = operator.fwd(probe=get_varying_probe(probe, eigen_probe, w),
                               scan=scan[lo:hi], psi=psi).contiguous()
        else:
            far = operator.fwd_device(probe, scan[lo:hi], psi, eigen_probe, w)
        inten = torch.empty((hi - lo, det, det), dtype=torch.float32,
                            device=psi.device)
        check(
            lib.tike_intensity(A.ptr(far), A.ptr(inten), hi - lo, S,
                               det * det, A.stream_ptr()), "intensity")
        yield lo, hi, inten


def simulate(detector_shape, probe, scan, psi, fly=1, eigen_probe=None,
             eigen_weights=None, **kwargs):
    """Real-valued detector counts of simulated ptychography data
    (ptycho.py:128-179).  Returns (FRAME, det, det) float32 on the host."""
    check_allowed_positions(scan, psi, probe.shape)
    det = int(detector_shape)
    with Ptycho(det, probe.shape[-1], nz=psi.shape[-2], n=psi.shape[-1],
                **kwargs) as operator:

        def on_device(array, dtype):
            return None if array is None else operator.asarray(array,
                                                               dtype=dtype)

        real, cplx = precision.floating, precision.cfloating
        frames = [
            inten for _, _, inten in _intensity_chunks(
                operator, on_device(psi, cplx), on_device(scan, real),
                on_device(probe, cplx), on_device(eigen_probe, cplx),
                on_device(eigen_weights, real))
        ]
        counts = torch.cat(frames)
        if fly > 1:  # `fly` consecutive positions expose one frame
            counts = counts.reshape(-1, fly, det, det).sum(dim=1)
        return operator.asnumpy(counts)


def reconstruct(data, parameters, num_gpu=None, use_mpi=False, **kwargs):
    """Solve the ptychography problem (ptycho.py:182-262).

    data (FRAME, WIDE, HIGH): measured intensities, FFT-shifted so that the
    diffraction peak is at the corners.  Returns the updated
    ``PtychoParameters`` (host arrays), which can be passed back in to resume.
    Extra keyword arguments go to `Reconstruction` (e.g. ``data_on_host``).

    num_gpu: an int N or a tuple of device numbers, as in the reference.
    Left out, the call uses this process's GPU -- or, inside a
    ``torch.distributed`` job, all of the job's ranks.  N > 1 from a plain
    process starts N child processes (one rank per GPU, RCCL) that shard the
    positions, and returns their result; a value that contradicts a running
    process group raises ValueError (`_spawn.resolve`).
    """
    if use_mpi:
        raise NotImplementedError(
            "multi-node MPI is out of scope; launch one process per GPU "
            "with torchrun instead")
    how, where = _spawn.resolve(num_gpu)
    if how == "spawn":
        return _spawn.reconstruct_spawned(data, parameters, where, **kwargs)
    with Reconstruction(data, parameters,
                        num_gpu if where is None else (where,),
                        use_mpi, **kwargs) as context:
        context.iterate(parameters.algorithm_options.num_iter)
        result = context.get_result()
    return result


def reconstruct_multigrid(data, parameters, num_gpu=None, use_mpi=False,
                          num_levels=3, interp=None):
    """Coarse-to-fine reconstruction (ptycho.py:975-1047): the real-space
    parameters are downsampled by 2^(num_levels-1) and the diffraction
    patterns cropped in Fourier space, every level runs
    `algorithm_options.num_iter` epochs and hands its upsampled result to the
    next."""
    data = A.to_host(data)
    coarsest = 2**(num_levels - 1)
    if data.shape[-1] / coarsest < 64:
        warnings.warn("Cropping diffraction patterns to less than 64 pixels "
                      "wide is not recommended because the full doughnut"
                      " may be visible.")
    current = parameters.resample(1.0 / coarsest, interp)
    for shrink in (2**k for k in range(num_levels - 1, -1, -1)):
        patterns = (data if shrink == 1 else solvers.crop_fourier_space(
            data, data.shape[-1] // shrink))
        result = reconstruct(patterns, current, num_gpu=num_gpu,
                             use_mpi=use_mpi)
        if shrink > 1:
            current = result.resample(2.0, interp)
    return result


def _check_batches(order, batches, n_total):
    """Injected batches must be contiguous ascending index ranges that tile
    [0, n) -- the solvers address a batch as [b[0], b[0] + len(b))."""
    if (order.ndim != 1 or len(order) != n_total or not np.array_equal(
            np.sort(order), np.arange(n_total))):
        raise ValueError("`order` must be a permutation of the positions")
    start = 0
    for b in batches:
        if len(b) and not np.array_equal(b, np.arange(start, start + len(b))):
            raise ValueError(
                "`batches` must be contiguous ascending index ranges into "
                "`order` that tile [0, N) (e.g. np.array_split(np.arange(N), "
                "num_batch)); reference-style index batches are expressed "
                "through `order`")
        start += len(b)
    if start != n_total:
        raise ValueError("`batches` must cover every position exactly once")


def _clip_magnitude(x, a_max):
    magnitude = x.abs()
    return torch.where(magnitude > a_max, a_max * x / magnitude, x)


def _warn_invalid_data():
    warnings.warn(
        "Diffraction patterns contain invalid data. "
        "All data should be non-negative and finite.", UserWarning)


def _check_data_shape(data, parameters):
    """The diffraction patterns against the forward model's shapes
    (reference ptycho.py:303-331: same conditions, same messages)."""
    frames = tuple(int(n) for n in data.shape)
    window = tuple(int(n) for n in parameters.probe.shape[-2:])
    mask = tuple(parameters.exitwave_options.measured_pixels.shape)
    masked = parameters.algorithm_options.name != "cgrad"  # cgrad: no mask
    rules = (
        (len(frames) == 3 and min(frames) >= 1 and frames[1] == frames[2],
         f"data shape {data.shape} is incorrect. "
         "It should be (N, W, H), "
         "where N >= 1 is the number of square diffraction patterns."),
        (frames[:1] == tuple(parameters.scan.shape[:1]),
         f"data shape {data.shape} and scan shape {parameters.scan.shape} "
         "are incompatible. They should have the same leading dimension."),
        (all(w <= d for w, d in zip(window, frames[-2:])),
         f"probe shape {parameters.probe.shape} "
         f"and data shape {data.shape} are incompatible. "
         "The probe width/height must be <= the data width/height ."),
        (not masked or mask == frames[-2:],
         f"exitwave_options.measured_pixels shape {mask} "
         f"does not match the diffraction patterns {frames[-2:]}"),
    )
    for holds, complaint in rules:
        if not holds:
            raise ValueError(complaint)


def spatially_sorted(scan_host, order, batches):
    """`order` with the positions of every minibatch re-listed leaf by leaf of
    a k-d tree (cluster.spatial_order): neighbours in space become neighbours
    in memory; batch membership, and therefore every sum over a batch, is
    unchanged."""
    order = np.array(order, copy=True)
    for b in batches:
        if len(b) > 1:
            idx = order[b]
            order[b] = idx[cluster.spatial_order(scan_host[idx])]
    return order


def rank_share(order, batches, size, rank):
    """Rank `rank`'s share of a job of `size` ranks: every global minibatch is
    split evenly (contiguously) over the ranks.  Returns (rows of the global
    arrays this rank holds, in its local order; its minibatches as contiguous
    ranges of that local order)."""
    local, local_batches, start = [], [], 0
    for b in batches:
        share = np.array_split(b, size)[rank]
        local.append(order[share])
        local_batches.append(np.arange(start, start + len(share)))
        start += len(share)
    return (np.concatenate(local) if local else np.zeros(0, dtype=np.int64),
            local_batches)


class Reconstruction():
    """Context manager keeping data and parameters on the GPU between
    ``iterate`` calls (ptycho.py:265-653).

    Extra keyword arguments (build-specific):
      presharded: ``data`` / ``parameters.scan`` / ``eigen_weights`` are
        already this rank's shard (weak-scaling benchmarks); batches are then
        contiguous splits of the local arrays.
      order, batches: inject a precomputed position order and batch split
        (parity tests replay the reference's clustering this way).
      spatial_sort: list the positions of every minibatch along a Z-order
        curve (default) so that the grouped scatter kernels find neighbours
        next to each other; results change only by summation order.
      local_data: this rank's rows of the dataset as a device tensor in its
        local order (internal: how `reconstruct(num_gpu=N)` hands every child
        only its own rows); `data` is then a shape/dtype stand-in.
      data_on_host: keep the diffraction patterns in pinned host memory and
        stream them to the GPU chunk by chunk (datasets larger than HBM; what
        the reference always does, communicators/stream.py:285-404) instead
        of holding them in HBM.  Results are identical.
    """

    def __init__(self, data, parameters, num_gpu=None, use_mpi=False, *,
                 presharded=False, order=None, batches=None,
                 spatial_sort=True, data_on_host=False, local_data=None):
        # a context lives in ONE process with ONE GPU: `num_gpu` must agree
        # with the process group it runs in (`reconstruct` is the entry that
        # starts ranks by itself)
        how, device = _spawn.resolve(num_gpu)
        if how == "spawn":
            raise ValueError(
                f"Reconstruction(num_gpu={num_gpu!r}): a Reconstruction "
                "context drives the GPU of its own process.  Call "
                f"tike_amd.ptycho.reconstruct(..., num_gpu={num_gpu!r}), "
                "which starts one rank per GPU itself, or launch the script "
                "with `python -m torch.distributed.run --nproc-per-node "
                f"{len(_spawn.requested_devices(num_gpu))}`")
        if device is not None:
            torch.cuda.set_device(device)  # ptycho.py:344-345
        _check_data_shape(data, parameters)
        name = parameters.algorithm_options.name
        if not hasattr(solvers, name):
            raise NotImplementedError(
                f"solver {name!r} is not available in tike_amd "
                "(available: lstsq_grad, rpie, cgrad)")
        if parameters.psi.shape[0] > 1 and name != "rpie":
            raise NotImplementedError(
                "multislice objects (psi.shape[0] > 1) are reconstructed by "
                "rpie only, as in the reference")
        if use_mpi:
            raise NotImplementedError(
                "multi-node MPI is out of scope; launch one process per GPU "
                "with torchrun instead")
        A.require_gpu()
        self._data_in = data
        self._parameters_in = parameters
        self._presharded = presharded
        self._data_on_host = bool(data_on_host)
        # this rank's patterns, already in its local order and on the device
        # (`reconstruct(num_gpu=N)` uploads every rank's rows block by block:
        # no process ever holds rows that are not its own); `data` then only
        # carries the shape and dtype of the whole dataset
        self._local_data = local_data
        self._order_in = order
        self._batches_in = batches
        self._spatial_sort = spatial_sort
        self.operator = Ptycho(
            probe_shape=parameters.probe.shape[-1],
            detector_shape=data.shape[-1],
            nz=parameters.psi.shape[-2],
            n=parameters.psi.shape[-1],
            norm=parameters.exitwave_options.propagation_normalization,
            probe_wavelength=getattr(parameters.probe_options,
                                     "probe_wavelength", float("nan")),
            probe_FOV_lengths=getattr(parameters.probe_options,
                                      "probe_FOV_lengths",
                                      (float("nan"), float("nan"))),
            multislice_propagation_distance=getattr(
                parameters.object_options, "multislice_propagation_distance",
                1e-9),
        )
        self.comm = Comm()
        self._pending_fits = []  # futures of deferred affine position fits
        self._free_snaps = []  # pinned host buffers of finished fits
        self._fit_pool = None
        self._fit_lock = threading.Lock()
        self._initial_scan_host = None

    # ------------------------------------------------------------- set-up
    def _shard(self, n_total):
        """Global order and this rank's local order / batches."""
        p = self._parameters_in
        o = p.algorithm_options
        scan_host = A.to_host(p.scan)
        if (self._order_in is None) != (self._batches_in is None):
            raise ValueError("`order` and `batches` must be given together")
        if self._order_in is not None:
            order = np.asarray(self._order_in)
            batches = [np.asarray(b) for b in self._batches_in]
            _check_batches(order, batches, n_total)
        else:
            order, batches = cluster.batches_contiguous(
                scan_host, o.batch_method, o.num_batch)
            if not self._presharded:
                # one clustering for the whole job: rank 0's (the generators
                # are synchronised too, see Comm.sync_random)
                order, batches = self.comm.broadcast_object((order, batches))
        # the arrangement the reference's clustering leaves the positions in
        # (concatenated batches): the RANSAC subsets of the affine position
        # fit are INDICES into it, and a fit whose rough 4-point models drop
        # some positions depends on which positions an index names
        self.cluster_order = np.array(order, copy=True)
        if self._spatial_sort:
            order = spatially_sorted(scan_host, order, batches)
        if self._presharded or self.comm.size == 1:
            return order, order, batches
        local, local_batches = rank_share(order, batches, self.comm.size,
                                          self.comm.rank)
        return order, local, local_batches

    def __enter__(self):
        self.operator.__enter__()
        self.comm.__enter__()
        data = self._data_in
        if self._local_data is not None:
            return self._enter_with_local_data()
        host = A.to_host(data) if not A.is_device(data) else None
        # "non-negative and finite" (ptycho.py:392-397) is checked where the
        # patterns end up: on the GPU for resident float data (one pass at HBM
        # rate instead of 0.34 s of host time per 2.6 GB), on the host for
        # integer counts and for patterns that stay in host memory
        on_host = host is not None and (self._data_on_host
                                        or host.dtype.kind in "iu")
        if on_host and (not np.all(np.isfinite(host)) or np.any(host < 0)):
            _warn_invalid_data()
        # every rank draws the same minibatch permutation / RANSAC subsets
        # (also when the caller sharded the data itself)
        self.comm.sync_random()
        self.order, self.local_order, self.batches = self._shard(
            data.shape[0])
        # HBM-resident data in batch-contiguous order: float32, or uint16 when
        # it arrived as <= 16-bit integers (ptycho.py:383-390)
        if self._data_on_host:
            from ..communicators.stream import PinnedData
            rows = (A.to_host(data) if host is None else host)[self.local_order]
            rows = (np.clip(rows, 0, None).astype(np.uint16)
                    if A.is_small_integer(rows.dtype) else
                    rows.astype(np.float32, copy=False))
            self.data = PinnedData(rows)
        else:
            self.data = A.data_to_device(data if A.is_device(data) else host,
                                         order=self.local_order)
            if (host is not None and not on_host
                    and self.data.dtype == torch.float32
                    and A.has_invalid_counts(self.data)):
                _warn_invalid_data()
        return self._finish_enter()

    def _finish_enter(self):
        self.parameters = solvers.PtychoParameters.split(
            self.local_order,
            x=self._host_parameters()).copy_to_device()
        self.parameters.algorithm_options = copy.deepcopy(
            self._parameters_in.algorithm_options)
        if (self.parameters.probe_options is not None and
                self.parameters.probe_options.init_rescale_from_measurements):
            self.parameters = _rescale_probe(self.operator, self.comm,
                                             self.data, self.parameters)
        return self

    def _enter_with_local_data(self):
        """`__enter__` when the patterns of this rank are on the device
        already (see `local_data`)."""
        self.comm.sync_random()
        self.order, self.local_order, self.batches = self._shard(
            self._data_in.shape[0])
        self.data = self._local_data
        if self._data_on_host:
            from ..communicators.stream import PinnedData
            self.data = PinnedData(np.asarray(self._local_data))
        if self.data.shape[0] != len(self.local_order):
            raise ValueError(
                f"local_data holds {self.data.shape[0]} patterns, this rank's "
                f"share is {len(self.local_order)}")
        if (isinstance(self.data, torch.Tensor)
                and self.data.dtype == torch.float32
                and A.has_invalid_counts(self.data)):
            _warn_invalid_data()
        return self._finish_enter()

    def _host_parameters(self):
        p = self._parameters_in
        h = lambda x: None if x is None else A.to_host(x)
        q = copy.copy(p)
        q.probe, q.psi, q.scan = h(p.probe), h(p.psi), h(p.scan)
        q.eigen_probe, q.eigen_weights = h(p.eigen_probe), h(p.eigen_weights)
        return q

    # -------------------------------------------------------------- epochs
    def iterate(self, num_iter: int) -> None:
        """Advance the reconstruction by num_iter epochs (ptycho.py:431-564)."""
        o = self.parameters.algorithm_options
        start = time.perf_counter()
        for _ in range(num_iter):
            if np.sum(o.times) > o.time_limit:
                logger.info("Maximum reconstruction time exceeded.")
                break
            total_epochs = len(o.times)
            logger.info(f"{o.name} epoch {total_epochs:,d}")
            self.parameters = _apply_probe_constraints(self.parameters,
                                                       epoch=total_epochs)
            # cgrad reads no preconditioner while it iterates; the object's is
            # needed only by the ambiguity rescale that follows some epochs
            lean = o.name == "cgrad"
            self.parameters = solvers.update_preconditioners(
                comm=self.comm, parameters=self.parameters,
                operator=self.operator, probe=not lean,
                psi=not lean or (o.rescale_method == "mean_of_abs_object" and
                                 (len(o.costs) + 1) % o.rescale_period == 0))
            self.parameters = getattr(solvers, o.name)(
                self.parameters, self.data, self.batches, self.comm,
                op=self.operator, epoch=total_epochs)
            self.parameters = _apply_object_constraints(self.parameters)
            self._apply_position_constraints()
            o.times.append(time.perf_counter() - start)
            start = time.perf_counter()
            logger.info("%10s cost is %+1.3e",
                        self.parameters.exitwave_options.noise_model,
                        np.mean(o.costs[-1]))

    def _apply_position_constraints(self):
        """Affine regularisation of the updated positions (ptycho.py:521-524,
        854-866).  The fit sees ALL positions of the job (every rank runs the
        same host fit with its synchronised generator), so the result does
        not depend on the number of ranks.

        With `use_position_regularization` off the fit changes nothing on the
        device (it only updates `position_options.transform`), and its ~5 ms of
        host work per epoch would leave the GPU idle: the positions are copied
        to pinned host memory asynchronously, the random subsets are drawn NOW
        (the generator is consumed in the reference's order) and the fit runs
        on a worker thread, in order, while the next epoch's kernels are being
        enqueued; results are joined when they are asked for (`get_result`,
        `__exit__`, the next regularised fit).  Between `iterate` and
        `get_result`, `parameters.position_options.transform` may therefore
        still be the previous epoch's: read it from `get_result()`."""
        p = self.parameters
        po = p.position_options
        if po is None:
            return
        local_fit = not self.comm.collective or self._presharded
        if not po.use_position_regularization and local_fit:
            # (pinned allocations cost milliseconds: the buffers are reused)
            with self._fit_lock:
                snap = self._free_snaps.pop() if self._free_snaps else None
            if snap is None:
                snap = torch.empty(tuple(p.scan.shape), dtype=p.scan.dtype,
                                   pin_memory=True)
            snap.copy_(p.scan, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            if self._fit_pool is None:
                import concurrent.futures
                self._fit_pool = concurrent.futures.ThreadPoolExecutor(1)
                # rows of this rank's arrays in the reference's arrangement
                self._fit_rows = np.argsort(self.local_order)[
                    self.cluster_order] if not self._presharded else slice(None)
                self._initial_scan_host = A.to_host(
                    po.initial_scan)[self._fit_rows]
            self._pending_fits.append(self._fit_pool.submit(
                self._fit_job, po, snap, done,
                ransac_subsets(p.scan.shape[0])))
            # finished fits leave the queue as we go (their exceptions
            # surface here); never more than two epochs' fits outstanding
            while self._pending_fits and (self._pending_fits[0].done()
                                          or len(self._pending_fits) > 2):
                self._pending_fits.pop(0).result()
            return
        self._resolve_fits()
        pos0 = pos1 = None
        if not self._presharded:
            # all positions of the job, in the reference's arrangement
            pos0 = self._gather_positions(po.initial_scan)[self.cluster_order]
            pos1 = self._gather_positions(p.scan)[self.cluster_order]
        p.scan, p.position_options = affine_position_regularization(
            updated=p.scan, position_options=po,
            positions0=pos0, positions1=pos1)

    def _fit_job(self, po, snap, done, subsets):
        """One deferred affine fit (worker thread; jobs run one at a time, in
        the order they were submitted)."""
        done.synchronize()
        origin = A.to_host(po.origin)
        po.transform, _ = estimate_global_transformation_ransac(
            positions0=self._initial_scan_host - origin,
            positions1=snap.numpy()[self._fit_rows] - origin,
            transform=po.transform,
            max_error=32, subsets=subsets)
        with self._fit_lock:
            self._free_snaps.append(snap)

    def _resolve_fits(self):
        """Join the deferred affine fits."""
        while self._pending_fits:
            self._pending_fits.pop(0).result()

    # ------------------------------------------------------------- results
    def _gather_positions(self, local):
        """Assemble a per-position array over all ranks in the input order."""
        if local is None:
            return None
        local = A.to_host(local)
        if not self.comm.collective or self._presharded:
            full_order, parts = self.local_order, local
        else:
            import torch.distributed as dist
            gathered = [None] * self.comm.size
            dist.all_gather_object(gathered, (self.local_order, local))
            full_order = np.concatenate([g[0] for g in gathered])
            parts = np.concatenate([g[1] for g in gathered], axis=0)
        if not np.array_equal(np.sort(full_order), np.arange(len(parts))):
            raise RuntimeError(
                "the ranks' position shards do not partition the scan "
                "(duplicated or dropped positions)")
        out = np.zeros_like(parts)
        out[full_order] = parts
        return out

    def get_scan(self):
        return self._gather_positions(self.parameters.scan)

    def get_result(self):
        """Current parameter estimates on the host (ptycho.py:573-597)."""
        if self.parameters.position_options is not None:
            self._resolve_fits()
        p = self.parameters.copy_to_host()
        p.scan = self._gather_positions(self.parameters.scan)
        p.eigen_weights = self._gather_positions(self.parameters.eigen_weights)
        po = self.parameters.position_options
        if po is not None:
            g = self._gather_positions
            p.position_options = po._like(g(po.initial_scan),
                                          g(po.confidence), g(po._momentum))
        return p

    def get_convergence(self):
        o = self.parameters.algorithm_options
        return o.costs, o.times

    def get_psi(self):
        return A.to_host(self.parameters.psi)

    def get_probe(self):
        p = self.parameters
        return (A.to_host(p.probe),
                None if p.eigen_probe is None else A.to_host(p.eigen_probe),
                self._gather_positions(p.eigen_weights))

    def append_new_data(self, new_data, new_scan):
        raise NotImplementedError(
            "Adding data on-the-fly is disabled until further notice.")

    def __exit__(self, type, value, traceback):
        # join the deferred fits on every path; on the error path their own
        # failures are logged instead of masking the exception in flight
        while self._pending_fits:
            fit = self._pending_fits.pop(0)
            try:
                fit.result()
            except Exception:  # noqa: BLE001
                if type is None:
                    raise
                logger.exception("deferred affine position fit failed")
        if self._fit_pool is not None:
            self._fit_pool.shutdown(wait=True)
            self._fit_pool = None
        self.comm.__exit__(type, value, traceback)
        self.operator.__exit__(type, value, traceback)
        self.data = None
        torch.cuda.empty_cache()


def _penalised(probe, mask):
    """probe - mask * conj(mask * probe): the reference's support penalties
    (ptycho.py:733-752)."""
    return probe - mask * torch.conj(mask * probe)


def _probe_steps(options, probe):
    """(enabled, transform) for every probe constraint, in the order the
    reference applies them (ptycho.py:731-778)."""
    yield options.probe_support > 0, lambda p: _penalised(
        p, finite_probe_support(p, p=options.probe_support,
                                radius=options.probe_support_radius,
                                degree=options.probe_support_degree))
    ramp = torch.linspace(0, 1, probe.shape[-3], dtype=torch.float32,
                          device=probe.device)[..., None, None]
    yield options.additional_probe_penalty > 0, lambda p: _penalised(
        p, options.additional_probe_penalty * ramp)
    yield options.median_filter_abs_probe, lambda p: (
        apply_median_filter_abs_probe(
            p, med_filt_px=options.median_filter_abs_probe_px))
    yield options.force_centered_intensity, constrain_center_peak
    yield options.force_sparsity < 1, lambda p: constrain_probe_sparsity(
        p, f=options.force_sparsity)


def _apply_probe_constraints(parameters, *, epoch):
    """End-of-epoch treatment of the probe (ptycho.py:723-808): the enabled
    constraints of `_probe_steps`, orthogonalisation (or just the mode
    powers), the periodic photon rescale, and the eigen-probe constraint."""
    po = parameters.probe_options
    if po is None:
        return parameters
    updating = po.recover_probe(epoch)
    if updating:
        probe = parameters.probe
        for enabled, transform in _probe_steps(po, probe):
            if enabled:
                probe = transform(probe)
        if po.force_orthogonality:
            probe, mode_power = orthogonalize_eig(probe)
            probe = probe.contiguous()
        else:
            mode_power = probe_power(probe)
        parameters.probe = probe
        po.power.append(mode_power)  # device tensor; host in copy_to_host()
    o = parameters.algorithm_options
    if (o.rescale_method == "constant_probe_photons"
            and len(o.costs) % o.rescale_period == 0):
        parameters.probe = rescale_probe_using_fixed_intensity_photons(
            parameters.probe, Nphotons=po.probe_photons,
            probe_power_fraction=None)
    if updating and parameters.eigen_probe is not None:
        parameters.eigen_probe, parameters.eigen_weights = (
            x.contiguous() for x in constrain_variable_probe(
                parameters.eigen_probe, parameters.eigen_weights))
    return parameters


def _apply_object_constraints(parameters):
    """End-of-epoch treatment of the object (ptycho.py:811-854): positivity,
    smoothness, unit magnitude clip -- each only when its option is set --
    and the periodic removal of the object / probe scale ambiguity."""
    oo = parameters.object_options
    if oo is None:
        return parameters
    steps = (
        (oo.positivity_constraint,
         lambda x: positivity_constraint(x, r=oo.positivity_constraint)),
        (oo.smoothness_constraint,
         lambda x: smoothness_constraint(x, a=oo.smoothness_constraint)),
        (oo.clip_magnitude, lambda x: _clip_magnitude(x, a_max=1.0)),
    )
    for enabled, transform in steps:
        if enabled:
            parameters.psi = transform(parameters.psi)
    o = parameters.algorithm_options
    due = len(o.costs) % o.rescale_period == 0
    if (due and o.name != "dm" and o.rescale_method == "mean_of_abs_object"
            and oo.preconditioner is not None):
        parameters.psi, parameters.probe = remove_object_ambiguity(
            parameters.psi, parameters.probe, oo.preconditioner)
    return parameters


def _rescale_probe(operator, comm, data, parameters):
    """probe *= sqrt(sum(data) / sum(intensity)) over measured pixels and all
    ranks (ptycho.py:873-972)."""
    nmeasured, mask_u8 = mask_info(parameters.exitwave_options,
                                   operator.detector_shape)
    sums = torch.zeros(2, dtype=torch.float64, device=parameters.psi.device)
    for lo, hi, inten in _intensity_chunks(operator, parameters.psi,
                                           parameters.scan, parameters.probe):
        d = A.data_f32(data, lo, hi)
        if mask_u8 is not None:
            m = mask_u8.bool()
            sums[0] += d[:, m].sum(dtype=torch.float64)
            sums[1] += inten[:, m].sum(dtype=torch.float64)
        else:
            sums[0] += d.sum(dtype=torch.float64)
            sums[1] += inten.sum(dtype=torch.float64)
    tot = comm.Allreduce_scalars([sums[0], sums[1]], sums.device)
    rescale = (torch.sqrt(tot[0]) / torch.sqrt(tot[1])).to(torch.float32)
    logger.info("Probe rescaled by %f", float(rescale))
    parameters.probe = parameters.probe * rescale
    po = parameters.probe_options
    if np.isnan(po.probe_photons):
        po.probe_photons = float(torch.sum(
            torch.square(parameters.probe.abs())).item())
    return parameters
