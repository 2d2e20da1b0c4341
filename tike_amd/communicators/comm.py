"""Collectives for the one-process-per-GPU design.

The reference runs one Python thread per GPU and implements its
"collectives" as serial peer copies + adds onto one device
(src/tike/communicators/pool.py:300-395, comm.py:96-136).  Here every rank is
its own process bound to one GPU (torchrun) and the only collective the hot
path needs is a sum all-reduce, executed by RCCL over xGMI through
``torch.distributed`` (backend "nccl"; "gloo" on CPU for the tests).
With a single rank every call is the identity.
"""
import os

import torch
import torch.distributed as dist


class _Done:
    """Handle of a collective that needs no waiting."""

    def wait(self):
        return True


class Comm:
    """All-reduce helper bound to a process group (or to a single rank)."""

    def __init__(self, group=None):
        self.group = group
        self.enabled = dist.is_available() and dist.is_initialized()
        self.size = dist.get_world_size(group) if self.enabled else 1
        self.rank = dist.get_rank(group) if self.enabled else 0
        # index of the minibatch the solver is working on (the same on every
        # rank): part of the key of per-minibatch cached collectives, see
        # solvers/lstsq.py `minibatch_key`
        self.minibatch = None
        # collectives are skipped for a single rank; TIKE_FORCE_COLLECTIVES=1
        # issues them anyway (used to exercise RCCL on a one-GPU test box)
        self.collective = self.size > 1 or (
            self.enabled and os.environ.get("TIKE_FORCE_COLLECTIVES") == "1")
        # TIKE_COMM_BACKEND=cabi: the float32 sum all-reduces (the gradient
        # buffers of every minibatch) go through the library's own RCCL entry
        # (tike_comm_allreduce_sum, include/tike_amd.h) instead of
        # torch.distributed; the process group only hands out the unique id.
        # With `cabi` every sum -- `Allreduce_start` included, which then
        # completes at once -- is issued on the caller's current stream, so
        # the two RCCL communicators of the process are ordered by that
        # stream.  (Without it, `Allreduce_start` is torch's async_op
        # all-reduce on the process group's own stream.)
        self._cabi = None
        if self.collective and os.environ.get("TIKE_COMM_BACKEND") == "cabi":
            self._cabi = self._create_cabi()

    def _create_cabi(self):
        import ctypes
        from .. import _lib
        ident = [None]
        if self.rank == 0:
            buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
            _lib.check(_lib.lib.tike_comm_unique_id(buf), "tike_comm_unique_id")
            ident[0] = buf.raw
        dist.broadcast_object_list(ident, src=0, group=self.group)
        handle = ctypes.c_void_p()
        _lib.check(
            _lib.lib.tike_comm_create(ident[0], self.size, self.rank,
                                      ctypes.byref(handle)),
            "tike_comm_create")
        return handle

    def _allreduce_f32(self, flat):
        if self._cabi is not None and flat.is_cuda:
            from .. import _lib
            _lib.check(
                _lib.lib.tike_comm_allreduce_sum(
                    self._cabi, flat.data_ptr(), flat.numel(), 0,
                    torch.cuda.current_stream(flat.device).cuda_stream),
                "tike_comm_allreduce_sum")
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def __enter__(self):
        return self

    def __exit__(self, type, value, traceback):
        self.close()

    def close(self):
        """Destroy the library's own RCCL communicator (if any)."""
        if self._cabi is not None:
            from .. import _lib
            handle, self._cabi = self._cabi, None
            torch.cuda.synchronize()
            _lib.check(_lib.lib.tike_comm_destroy(handle), "tike_comm_destroy")

    def __del__(self):
        # Never tear a communicator down from the garbage collector: that can
        # run mid-epoch on another thread, at interpreter shutdown, or while
        # other ranks are still inside a collective.  Use `with Comm() as c:`
        # or call close(); a leaked handle is reported, not destroyed.
        if getattr(self, "_cabi", None) is not None:
            import warnings
            warnings.warn("Comm was not closed: its RCCL communicator is "
                          "leaked (use it as a context manager)",
                          ResourceWarning, stacklevel=2)

    def Allreduce(self, *tensors):
        """Sum the tensors across ranks IN PLACE; complex tensors are reduced
        as interleaved float32.  Several float32 / complex64 tensors are
        packed into one flat buffer so that they cost one collective; tensors
        of any other dtype (float64 sums) are reduced on their own, in their
        own precision."""
        if not self.collective:
            return tensors if len(tensors) != 1 else tensors[0]
        views = [
            torch.view_as_real(t) if t.is_complex() else t for t in tensors
        ]
        if (len(views) == 1 and views[0].is_contiguous()
                and views[0].dtype == torch.float32):
            self._allreduce_f32(views[0])
        elif len(views) == 1 and views[0].is_contiguous():
            dist.all_reduce(views[0], op=dist.ReduceOp.SUM, group=self.group)
        else:
            f32 = [v for v in views if v.dtype == torch.float32]
            for v in views:
                if v.dtype != torch.float32:  # never downcast
                    w = v.contiguous()
                    dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group)
                    if w is not v:
                        v.copy_(w)
            if f32:
                flat = torch.cat([v.reshape(-1) for v in f32])
                self._allreduce_f32(flat)
                off = 0
                for v in f32:
                    n = v.numel()
                    v.copy_(flat[off:off + n].reshape(v.shape))
                    off += n
        return tensors if len(tensors) != 1 else tensors[0]

    def Allreduce_start(self, flat):
        """Begin the in-place sum of a contiguous float32 tensor across ranks
        and return a handle whose ``wait()`` makes the caller's stream wait
        for it.  The collective runs on the process group's own stream behind
        everything enqueued so far, so kernels launched between this call and
        ``wait()`` overlap it (the probe-gradient slice travels while the
        object scatter runs).  Collectives started here and plain `Allreduce`
        calls execute in issue order on every rank.  With the library's own
        communicator (TIKE_COMM_BACKEND=cabi), which lives on the caller's
        stream, the sum is carried out at once."""
        if not self.collective:
            return _Done()
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        if self._cabi is not None and flat.is_cuda:
            self._allreduce_f32(flat)
            return _Done()
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group,
                               async_op=True)

    def Allreduce_scalars(self, values, device):
        """Sum a short list of 0-d DEVICE tensors across ranks -> float64
        tensor on `device` (one tiny collective, no host synchronisation).
        Python numbers are accepted but cost a blocking host-to-device copy:
        keep them out of per-minibatch code (see `Allreduce_count`)."""
        t = torch.stack([
            v.detach().to(device=device, dtype=torch.float64).reshape(())
            if isinstance(v, torch.Tensor) else torch.tensor(
                float(v), dtype=torch.float64, device=device) for v in values
        ])
        if self.collective:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def Allreduce_f64(self, t):
        """Sum a float64 DEVICE tensor across ranks in place (a handful of
        scalars that stay on the device: the cost sums between a line search's
        cost pass and its decision)."""
        if self.collective:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def Allreduce_count(self, n: int) -> float:
        """Sum of a host integer over ranks, returned as a host float
        (used once per run for the global minibatch sizes)."""
        if not self.collective:
            return float(n)
        t = torch.tensor([float(n)], dtype=torch.float64)
        if dist.get_backend(self.group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item())

    def Allreduce_max(self, t):
        """Maximum of a 0-d device tensor over ranks (stays on the device)."""
        if self.collective:
            t = t.clone()
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return t

    def Allgather_rows(self, t):
        """Concatenation along axis 0 of a per-rank tensor whose leading
        length may differ between ranks (once per epoch: small arrays)."""
        if not self.collective:
            return t
        n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        sizes = [torch.zeros_like(n) for _ in range(self.size)]
        dist.all_gather(sizes, n, group=self.group)
        sizes = [int(x.item()) for x in sizes]
        pad = torch.zeros((max(sizes), *t.shape[1:]), dtype=t.dtype,
                          device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(self.size)]
        dist.all_gather(parts, pad, group=self.group)
        return torch.cat([p[:k] for p, k in zip(parts, sizes)], dim=0)

    def sync_random(self):
        """Give every rank the generator states of rank 0 (``np.random`` and
        ``tike_amd.random.randomizer_np``) -- this OVERWRITES the
        process-global ``np.random`` state of the other ranks, on purpose.
        The multi-rank solver relies on
        every rank drawing the same clustering seeds, minibatch permutation
        and RANSAC subsets; ranks started with different seeds (or with
        OS-seeded generators) would otherwise split the job differently."""
        if not self.collective:
            return
        import numpy as np
        from .. import random as trandom
        state = [None]
        if self.rank == 0:
            state[0] = (np.random.get_state(),
                        trandom.randomizer_np.bit_generator.state)
        dist.broadcast_object_list(state, src=0, group=self.group)
        legacy, generator = state[0]
        np.random.set_state(legacy)
        rng = np.random.default_rng()
        if type(rng.bit_generator).__name__ != generator["bit_generator"]:
            rng = np.random.Generator(
                getattr(np.random, generator["bit_generator"])())
        rng.bit_generator.state = generator
        trandom.randomizer_np = rng

    def broadcast_object(self, obj):
        """Rank 0's picklable object on every rank."""
        if not self.collective:
            return obj
        box = [obj if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=self.group)
        return box[0]

    def barrier(self):
        if self.collective:
            dist.barrier(group=self.group)
