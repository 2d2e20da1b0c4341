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
import socket
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _share(array):
    """NumPy array -> (shared-memory torch tensor, dtype to view it back as)."""
    array = np.ascontiguousarray(array)
    view = array.dtype
    if array.dtype == np.uint16:  # torch's uint16 support is partial
        array = array.view(np.int16)
    return torch.from_numpy(array).share_memory_(), view


def _rank_main(rank, devices, port, backend, shared, view, blob, results):
    """One rank of a spawned reconstruction (runs in a fresh interpreter)."""
    import torch.distributed as dist
    try:
        torch.cuda.set_device(devices[rank])
        extra = {}
        if backend == "nccl":  # bind the communicator to this rank's GPU
            extra["device_id"] = torch.device("cuda", devices[rank])
        dist.init_process_group(backend,
                                init_method=f"tcp://127.0.0.1:{port}",
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
            data = shared.numpy().view(view)
            with Reconstruction(data, parameters, **kwargs) as context:
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


def reconstruct_spawned(data, parameters, devices, **kwargs):
    """Run `reconstruct` on len(devices) child ranks; return rank 0's result."""
    import torch.multiprocessing as mp
    import tike_amd.random
    from .. import _arrays as A
    distinct = len(set(devices)) == len(devices)
    backend = "nccl" if distinct else "gloo"
    if not distinct:
        logger.warning("ranks share GPUs %s: collectives over gloo", devices)
    shared, view = _share(A.to_host(data))
    blob = pickle.dumps((parameters, kwargs, np.random.get_state(),
                         tike_amd.random.randomizer_np))
    ctx = mp.get_context("spawn")
    results = ctx.Queue()
    port = _free_port()
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: v for k, v in env.items() if saved[k] is None})
    try:
        procs = [
            ctx.Process(target=_rank_main,
                        args=(r, devices, port, backend, shared, view, blob,
                              results), daemon=True)
            for r in range(len(devices))
        ]
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
    message = None
    try:
        while message is None:
            try:
                message = results.get(timeout=1.0)
            except queue.Empty:  # is everybody still alive?
                dead = [(r, p.exitcode) for r, p in enumerate(procs)
                        if p.exitcode not in (None, 0)]
                if dead:
                    raise RuntimeError(
                        f"reconstruct(num_gpu={len(devices)}): rank(s) "
                        f"{dead} exited without a result")
                if all(p.exitcode == 0 for p in procs):
                    raise RuntimeError(
                        "reconstruct: every rank exited but rank 0 sent no "
                        "result")
        if message[0] == "error":
            raise RuntimeError(
                f"reconstruct(num_gpu={len(devices)}): rank {message[1]} "
                f"failed:\n{message[2]}")
    except BaseException:
        for p in procs:
            if p.is_alive():
                p.terminate()
        raise
    finally:
        for p in procs:
            p.join(60)
    result, legacy, rng = pickle.loads(message[1])
    # the generators advance as they would have in an in-process call
    np.random.set_state(legacy)
    tike_amd.random.randomizer_np = rng
    return result
