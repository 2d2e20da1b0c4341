"""`num_gpu` of the reference's one-call interface, MI355X-style.

The reference's ``reconstruct(data, parameters, num_gpu=N)`` starts N worker
THREADS, one per GPU, inside the calling process (reference
src/tike/ptycho/ptycho.py:182-187,371-381; communicators/pool.py).  Here a GPU
belongs to a process, so the same call starts N fresh child processes
(``spawn`` -- new interpreters, never a re-exec of the caller), one rank per
GPU with RCCL between them, hands them the problem through shared memory,
and returns rank 0's gathered result to the caller: a reference user's
script keeps working and really uses N GPUs.

Inside an already initialised ``torch.distributed`` job (torchrun) nothing is
started: ``num_gpu`` must then be the world size (or be left out).
"""
import logging
import os
import pickle
import queue
import traceback
import warnings

import numpy as np
import torch

logger = logging.getLogger(__name__)


def requested_devices(num_gpu):
    """`num_gpu` -> tuple of device indices, or None when it was left out.

    int N -> (0, ..., N-1); a tuple names the devices (ptycho.py:205-208)."""
    if num_gpu is None:
        return None
    if isinstance(num_gpu, (tuple, list)):
        devices = tuple(int(d) for d in num_gpu)
    else:
        devices = tuple(range(int(num_gpu)))
    if len(devices) < 1 or any(d < 0 for d in devices):
        raise ValueError(f"num_gpu={num_gpu!r}: expected a positive number of "
                         "GPUs or a tuple of device numbers")
    return devices


def world_size():
    import torch.distributed as dist
    return (dist.get_world_size()
            if dist.is_available() and dist.is_initialized() else 1)


def resolve(num_gpu):
    """What a call with this `num_gpu` has to do in THIS process.

    Returns ("here", device or None): run in this process (device: the one a
    tuple named), or ("spawn", devices): start len(devices) ranks.  Raises
    ValueError when `num_gpu` contradicts the running process group -- a
    request for GPUs is honoured or refused, never dropped."""
    devices = requested_devices(num_gpu)
    world = world_size()
    if devices is None:
        return "here", None
    if world > 1:
        if len(devices) != world:
            raise ValueError(
                f"num_gpu={num_gpu!r} asks for {len(devices)} GPU(s) but this "
                f"process is rank of a torch.distributed job of {world}: "
                "pass num_gpu equal to the world size or leave it out")
        return "here", None
    if len(devices) == 1:
        return "here", (devices[0] if isinstance(num_gpu,
                                                  (tuple, list)) else None)
    available = torch.cuda.device_count()
    if available < 1:
        raise RuntimeError("tike_amd needs a GPU: none is visible")
    if (len(devices) > available
            and os.environ.get("TIKE_AMD_OVERSUBSCRIBE") != "1"):
        # the reference's rule: "If the number of GPUs is less than the
        # requested number, only workers for the available GPUs are allocated"
        warnings.warn(
            f"num_gpu={num_gpu!r}: only {available} GPU(s) visible; using "
            f"{available} (set TIKE_AMD_OVERSUBSCRIBE=1 to let ranks share "
            "a GPU over gloo, test boxes only)", UserWarning)
        devices = devices[:available]
        if len(devices) == 1:
            return "here", devices[0]
    return "spawn", tuple(d % available for d in devices)


def _share(array):
    """NumPy array -> (shared-memory torch tensor, dtype to view it back as)."""
    array = np.ascontiguousarray(array)
    view = array.dtype
    if array.dtype == np.uint16:  # torch's uint16 support is partial
        array = array.view(np.int16)
    return torch.from_numpy(array).share_memory_(), view


UPLOAD_BLOCK_BYTES = 256 << 20


def _as_resident(block):
    """A block of patterns in the form they are kept in HBM: uint16 when they
    arrived as <= 16-bit integers (ptycho.py:383-390), float32 otherwise --
    the conversion of `_arrays.data_to_device`."""
    from .. import _arrays as A
    if A.is_small_integer(block.dtype):
        return np.ascontiguousarray(np.clip(block, 0, None).astype(np.uint16))
    return np.ascontiguousarray(block, dtype=np.float32)


def _receive_rows(inbox, n_local, frame, small, on_host):
    """This rank's patterns, block by block from the parent: into a device
    tensor in the rank's local order (or a host array with data_on_host)."""
    dtype = np.uint16 if small else np.float32
    if on_host:
        rows = np.empty((n_local,) + frame, dtype=dtype)
    else:
        rows = torch.empty((n_local,) + frame,
                           dtype=torch.uint16 if small else torch.float32,
                           device="cuda")
    while True:
        message = inbox.get()
        if message[0] == "done":
            break
        _, lo, shared = message
        block = shared.numpy().view(dtype)
        if on_host:
            rows[lo:lo + len(block)] = block
        else:
            t = torch.from_numpy(block.view(np.int16) if small else block)
            dst = rows[lo:lo + len(block)]
            (dst.view(torch.int16) if small else dst).copy_(t)
            torch.cuda.synchronize()
        del shared, block
        inbox.task_done()
    inbox.task_done()
    return rows


def _rank_main(rank, devices, store_path, backend, inbox, meta, blob, results):
    """One rank of a spawned reconstruction (runs in a fresh interpreter)."""
    import torch.distributed as dist
    try:
        torch.cuda.set_device(devices[rank])
        extra = {}
        if backend == "nccl":  # bind the communicator to this rank's GPU
            extra["device_id"] = torch.device("cuda", devices[rank])
        # rendezvous through a file the parent owns: no port to lose a race for
        dist.init_process_group(backend, init_method=f"file://{store_path}",
                                rank=rank, world_size=len(devices), **extra)
        try:
            import tike_amd.random
            from tike_amd.ptycho.ptycho import Reconstruction
            parameters, kwargs, legacy, rng = pickle.loads(blob)
            # the caller's generator states (rank 0 hands its own to the other
            # ranks, Comm.sync_random): the job draws what the caller's
            # process would have drawn
            np.random.set_state(legacy)
            tike_amd.random.randomizer_np = rng
            shape, dtype, small, n_local = meta
            rows = _receive_rows(inbox, n_local[rank], tuple(shape[1:]), small,
                                 bool(kwargs.get("data_on_host")))
            # the whole dataset's shape and dtype, none of its memory
            stand_in = np.broadcast_to(np.zeros((), dtype=dtype), shape)
            with Reconstruction(stand_in, parameters, local_data=rows,
                                **kwargs) as context:
                context.iterate(parameters.algorithm_options.num_iter)
                result = context.get_result()
            if rank == 0:
                results.put(("ok", pickle.dumps(
                    (result, np.random.get_state(),
                     tike_amd.random.randomizer_np))))
        finally:
            dist.destroy_process_group()
    except BaseException:  # noqa: BLE001 -- reported to the parent, then exit
        results.put(("error", rank, traceback.format_exc()))
        raise


def _plan_shares(parameters, n_total, size, kwargs):
    """The clustering of the whole job, computed ONCE (here, in the caller's
    process -- it draws from the caller's generators exactly as an in-process
    call would), and the rows of every rank in the rank's local order."""
    from .. import _arrays as A
    from .. import cluster
    from .ptycho import _check_batches, rank_share, spatially_sorted
    scan = A.to_host(parameters.scan)
    o = parameters.algorithm_options
    order, batches = kwargs.get("order"), kwargs.get("batches")
    if (order is None) != (batches is None):
        raise ValueError("`order` and `batches` must be given together")
    if order is None:
        order, batches = cluster.batches_contiguous(scan, o.batch_method,
                                                    o.num_batch)
    order = np.asarray(order)
    batches = [np.asarray(b) for b in batches]
    _check_batches(order, batches, n_total)
    listed = (spatially_sorted(scan, order, batches)
              if kwargs.get("spatial_sort", True) else order)
    rows = [rank_share(listed, batches, size, r)[0] for r in range(size)]
    return order, batches, rows


def _check_shm(need):
    """Blocks travel through /dev/shm (torch shared-memory tensors): fail with
    a clear message instead of a bus error when it is too small."""
    try:
        import shutil
        free = shutil.disk_usage("/dev/shm").free
    except OSError:
        return
    if free < need:
        raise RuntimeError(
            f"reconstruct(num_gpu=N) needs {need >> 20} MiB of /dev/shm to "
            f"hand the patterns to its ranks, {free >> 20} MiB are free "
            "(containers: raise --shm-size)")


def reconstruct_spawned(data, parameters, devices, **kwargs):
    """Run `reconstruct` on len(devices) child ranks; return rank 0's result.

    The parent clusters once, then streams to every rank ONLY that rank's rows,
    in blocks of UPLOAD_BLOCK_BYTES through shared memory (each block is
    uploaded and released before the next one is cut): no process holds a
    second copy of the dataset, and nothing depends on a probed TCP port."""
    import tempfile
    import torch.multiprocessing as mp
    import tike_amd.random
    from .. import _arrays as A
    if kwargs.get("presharded"):
        raise ValueError("presharded=True means the caller runs the ranks "
                         "itself; reconstruct(num_gpu=N) shards for them")
    distinct = len(set(devices)) == len(devices)
    backend = "nccl" if distinct else "gloo"
    if not distinct:
        logger.warning("ranks share GPUs %s: collectives over gloo", devices)
    size = len(devices)
    host = A.to_host(data)
    order, batches, rows = _plan_shares(parameters, host.shape[0], size,
                                        kwargs)
    kwargs = dict(kwargs, order=order, batches=batches)
    small = A.is_small_integer(host.dtype)
    frame_bytes = int(np.prod(host.shape[1:])) * (2 if small else 4)
    # one block PER RANK is in shared memory at a time (a round of the upload
    # loop cuts `size` blocks before it waits for any of them): the blocks
    # shrink with the number of ranks so that a round never holds more than
    # two default blocks, and /dev/shm is checked for exactly that
    step = max(1, 2 * UPLOAD_BLOCK_BYTES // (max(frame_bytes, 1) * max(size, 2)))
    _check_shm(size * min(step, max(len(r) for r in rows)) * frame_bytes)
    # (drawn AFTER the clustering: the ranks continue the caller's sequences)
    blob = pickle.dumps((parameters, kwargs, np.random.get_state(),
                         tike_amd.random.randomizer_np))
    meta = (tuple(host.shape), host.dtype, small, [len(r) for r in rows])
    ctx = mp.get_context("spawn")
    results = ctx.Queue()
    inboxes = [ctx.JoinableQueue() for _ in range(size)]
    # a directory only this user can enter: nobody else can plant the store
    # file between its naming here and its creation by FileStore
    store_dir = tempfile.mkdtemp(prefix="tike_amd_rendezvous_")
    store_path = os.path.join(store_dir, "store")
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: v for k, v in env.items() if saved[k] is None})
    try:
        procs = [
            ctx.Process(target=_rank_main,
                        args=(r, devices, store_path, backend, inboxes[r],
                              meta, blob, results), daemon=True)
            for r in range(size)
        ]
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)

    def alive_or_raise():
        dead = [(r, p.exitcode) for r, p in enumerate(procs)
                if p.exitcode not in (None, 0)]
        if dead:
            raise RuntimeError(
                f"reconstruct(num_gpu={size}): rank(s) {dead} exited without "
                "a result")

    def wait_for(inbox):
        """inbox.join() that notices a rank dying (it would never return)."""
        import threading
        done = threading.Event()
        t = threading.Thread(target=lambda: (inbox.join(), done.set()),
                             daemon=True)
        t.start()
        while not done.wait(0.5):
            alive_or_raise()

    message = None
    try:
        # every rank's rows, a block at a time, round robin over the ranks so
        # that their uploads overlap; a block is released once it is uploaded
        cursor = [0] * size
        warned = False
        while any(cursor[r] < len(rows[r]) for r in range(size)):
            for r in range(size):
                lo = cursor[r]
                if lo >= len(rows[r]):
                    continue
                raw = host[rows[r][lo:lo + step]]
                # the check a one-rank call makes on the host (reference
                # ptycho.py:392-397) BEFORE negative counts are clipped away
                if not warned and (not np.all(np.isfinite(raw))
                                   or np.any(raw < 0)):
                    from .ptycho import _warn_invalid_data
                    _warn_invalid_data()
                    warned = True
                block = _as_resident(raw)
                shared, _ = _share(block)
                inboxes[r].put(("rows", lo, shared))
                cursor[r] = lo + len(block)
                del block, shared
            for r in range(size):
                wait_for(inboxes[r])
        for r in range(size):
            inboxes[r].put(("done",))
        while message is None:
            try:
                message = results.get(timeout=1.0)
            except queue.Empty:  # is everybody still alive?
                alive_or_raise()
                if all(p.exitcode == 0 for p in procs):
                    raise RuntimeError(
                        "reconstruct: every rank exited but rank 0 sent no "
                        "result")
        if message[0] == "error":
            raise RuntimeError(
                f"reconstruct(num_gpu={size}): rank {message[1]} "
                f"failed:\n{message[2]}")
    except BaseException:
        # a rank that failed has said why: prefer its traceback
        try:
            late = results.get(timeout=2.0)
            if late[0] == "error":
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                raise RuntimeError(
                    f"reconstruct(num_gpu={size}): rank {late[1]} failed:\n"
                    f"{late[2]}") from None
        except queue.Empty:
            pass
        for p in procs:
            if p.is_alive():
                p.terminate()
        raise
    finally:
        for p in procs:
            p.join(60)
        import shutil
        shutil.rmtree(store_dir, ignore_errors=True)
    result, legacy, rng = pickle.loads(message[1])
    # the generators advance as they would have in an in-process call
    np.random.set_state(legacy)
    tike_amd.random.randomizer_np = rng
    return result
