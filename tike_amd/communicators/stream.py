"""Diffraction patterns that stay in pinned host memory.

The reference keeps `data` in pinned host memory and copies 64-position
chunks to the GPU on two CUDA streams while the previous chunk is worked on
(src/tike/communicators/stream.py:285-404 `stream_and_modify2`, called from
ptycho/solvers/lstsq.py:367-602).  Here the data is HBM-resident by default
(288 GB per GPU hold every BASELINE configuration); `PinnedData` is the
out-of-core alternative for datasets that do not fit: it looks like the
resident tensor to the solvers -- `data[lo:hi]` returns a device tensor -- and
hides the PCIe copy of the NEXT chunk behind the kernels of the current one
(one copy stream, two device slots, HIP events both ways; no host
synchronisation).
"""
import numpy as np
import torch


class PinnedData:
    """Rows of a (N, det, det) array in pinned host memory, sliced onto the
    GPU on demand.  Only contiguous row slices are supported (what the solvers
    take); the chunk after the one just handed out is prefetched."""

    def __init__(self, host, device=None):
        if isinstance(host, torch.Tensor):
            host = host.detach().cpu().numpy()
        host = np.ascontiguousarray(host)
        if host.dtype == np.uint16:
            # torch has no copy kernels for uint16: move it as int16 bits
            self._host = torch.from_numpy(host.view(np.int16))
            self.dtype = torch.uint16
        elif host.dtype == np.float32:
            self._host = torch.from_numpy(host)
            self.dtype = torch.float32
        else:
            raise TypeError(f"PinnedData holds uint16 or float32, not {host.dtype}")
        self._host = self._host.pin_memory()
        self.device = torch.device(
            "cuda", torch.cuda.current_device()) if device is None else device
        self.shape = tuple(self._host.shape)
        self.ndim = self._host.ndim
        self._copy = torch.cuda.Stream(self.device)
        self._slots = [None, None]
        # events (one per stream that read it) after which a slot may be rewritten
        self._free = [None, None]
        self._readers = [set(), set()]  # streams handed each slot's chunk
        self._ready = None  # (lo, hi, slot, event) of the prefetched chunk
        self._last = None  # (lo, hi, slot, copy-done event, stream) handed out last
        self.copies = 0  # host-to-device copies issued
        self.hits = 0  # chunks that were already on their way when asked for
        self._ranges = []  # row ranges the solver will walk next, in order
        self._nmax = 0  # rows of the largest chunk asked for so far

    def hint(self, ranges):
        """Tell the prefetcher which row ranges (minibatches) come next and in
        which order: the solvers visit the minibatches of an epoch in a random
        permutation, so "the rows behind this chunk" is the wrong guess at
        every minibatch boundary (10 000 positions in 10 minibatches: 42 hits
        in 130 copies, every miss an exposed 4.6 ms copy plus a wasted one)."""
        self._ranges = [(int(a), int(b)) for a, b in ranges if b > a]
        if self._ranges and self._last is not None and self._ready is None:
            # the first chunk of the epoch, while the set-up work of the
            # epoch (preconditioners, constraints) runs
            n = max(self._last[1] - self._last[0], self._nmax)
            a, b = self._ranges[0]
            if (a, min(b, a + n)) != self._last[:2]:
                other = 1 - self._last[2]
                self._release(other, torch.cuda.current_stream(self.device))
                self._ready = (a, min(b, a + n), other,
                               self._issue(a, min(b, a + n), other))

    def _predict(self, lo, hi):
        """The chunk that will be asked for after rows [lo, hi) (a full
        chunk: the one just handed out may have been the short tail of its
        minibatch)."""
        n = max(hi - lo, self._nmax)
        for i, (a, b) in enumerate(self._ranges):
            if a <= lo and hi <= b:
                if hi < b:
                    return hi, min(b, hi + n)
                if i + 1 < len(self._ranges):
                    a2, b2 = self._ranges[i + 1]
                    return a2, min(b2, a2 + n)
                return None  # the epoch ends here
        nlo, nhi = hi, min(self.shape[0], hi + n)  # no hint: the rows behind
        return (nlo, nhi) if nhi > nlo else None

    def _release(self, slot, cur):
        """Everything queued so far has finished with `slot`: it may be
        rewritten behind the events recorded here."""
        released = []
        for stream in self._readers[slot] | {cur}:
            event = torch.cuda.Event()
            event.record(stream)
            released.append(event)
        self._free[slot] = released
        self._readers[slot] = set()

    def __len__(self):
        return self.shape[0]

    def _slot(self, s, n):
        buf = self._slots[s]
        if buf is None or buf.shape[0] < n:
            if buf is not None:
                torch.cuda.synchronize(self.device)  # rare: a larger chunk
            buf = torch.empty((n, *self.shape[1:]), dtype=self._host.dtype,
                              device=self.device)
            buf.record_stream(self._copy)
            self._slots[s] = buf
            self._free[s] = None
        return buf

    def _issue(self, lo, hi, s):
        """Start copying rows [lo, hi) into slot s on the copy stream."""
        buf = self._slot(s, hi - lo)
        for event in self._free[s] or ():  # every stream that read this slot
            self._copy.wait_event(event)
        with torch.cuda.stream(self._copy):
            buf[:hi - lo].copy_(self._host[lo:hi], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy)
        self.copies += 1
        return done

    def __getitem__(self, key):
        if not isinstance(key, slice) or key.step not in (None, 1):
            raise TypeError("PinnedData supports contiguous row slices only")
        lo, hi, _ = key.indices(self.shape[0])
        hi = max(hi, lo)
        n = hi - lo
        if n == 0:
            return torch.empty((0, *self.shape[1:]), dtype=self.dtype,
                               device=self.device)
        self._nmax = max(self._nmax, n)
        cur = torch.cuda.current_stream(self.device)
        if self._last is not None and self._last[:2] == (lo, hi):
            s = self._last[2]  # asked for twice in one chunk iteration
            if self._last[3] is not None and cur != self._last[4]:
                cur.wait_event(self._last[3])  # a consumer on another stream
                self._readers[s].add(cur)  # ... whose reads the slot outlives
            return self._view(s, n)
        if self._ready is not None and self._ready[:2] == (lo, hi):
            _, _, s, done = self._ready
            self.hits += 1
        else:
            # not predicted: into the slot that was NOT handed out last (its
            # kernels may still be queued); a stale prefetch there is dropped
            s = 0 if self._last is None else 1 - self._last[2]
            done = self._issue(lo, hi, s)
        self._ready = None
        cur.wait_event(done)
        # everything queued so far has finished with the other slot
        other = 1 - s
        self._release(other, cur)
        self._readers[s] = {cur}
        self._last = (lo, hi, s, done, cur)
        # the solvers walk a minibatch chunk by chunk and the minibatches in
        # the order they announced (`hint`): fetch the next chunk now (without
        # a hint: the rows behind this one; a miss is copied on demand).
        # cgrad re-walks the chunks of a minibatch for every line-search probe:
        # with data_on_host and several chunks per minibatch the minibatch
        # crosses PCIe once per probe -- size the chunks (or the minibatches)
        # so that a minibatch is one chunk when that matters.
        nxt = self._predict(lo, hi)
        if nxt is not None:
            self._ready = (*nxt, other, self._issue(*nxt, other))
        return self._view(s, n)

    def _view(self, s, n):
        buf = self._slots[s][:n]
        return buf.view(torch.uint16) if self.dtype == torch.uint16 else buf
