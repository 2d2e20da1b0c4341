"""Device-array plumbing: torch tensors are the device arrays of this package.

Operators accept either host ``numpy.ndarray`` (copied to the GPU and back;
what the reference's tests do through ``op.xp.asarray``) or torch CUDA tensors
(zero copy; what the solvers use) and return the same kind.
"""
import numpy as np
import torch

_NP2T = {
    np.dtype("complex64"): torch.complex64,
    np.dtype("float32"): torch.float32,
    np.dtype("float64"): torch.float64,
    np.dtype("complex128"): torch.complex128,
    np.dtype("int32"): torch.int32,
    np.dtype("int64"): torch.int64,
    np.dtype("uint8"): torch.uint8,
    np.dtype("bool"): torch.bool,
    np.dtype("uint16"): torch.int32,  # no uint16 arithmetic in torch
}


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError(
            "tike_amd runs on an AMD GPU (ROCm) only; no GPU is visible and "
            "there is no CPU fallback.")


def current_device():
    require_gpu()
    from . import _lib
    if _lib.DETERMINISTIC:
        _lib.ensure_deterministic()
    return torch.device("cuda", torch.cuda.current_device())


def torch_dtype(dtype):
    if isinstance(dtype, torch.dtype):
        return dtype
    return _NP2T[np.dtype(dtype)]


def is_device(x):
    return isinstance(x, torch.Tensor)


def to_device(x, dtype=None, device=None):
    """Return a contiguous torch tensor on the GPU (copying host arrays)."""
    if device is None:
        device = current_device()
    elif not isinstance(device, torch.device):
        device = torch.device("cuda", int(device))
    if isinstance(x, torch.Tensor):
        t = x.to(device=device)
    else:
        a = np.asarray(x)
        if a.dtype == np.uint16:
            a = a.astype(np.int32)
        t = torch.from_numpy(np.ascontiguousarray(a)).to(device)
    if dtype is not None:
        t = t.to(torch_dtype(dtype))
    return t.contiguous()


def is_small_integer(dtype):
    """Detector counts that arrive as <= 16-bit integers stay 16-bit in HBM
    (reference ptycho.py:383-390 keeps `itemsize <= 2` data as it is)."""
    if isinstance(dtype, torch.dtype):
        return dtype in (torch.uint8, torch.int8, torch.int16, torch.uint16)
    dtype = np.dtype(dtype)
    return dtype.kind in "iu" and dtype.itemsize <= 2


def data_to_device(x, order=None, device=None):
    """The diffraction patterns in HBM, rows permuted by `order`: uint16 when
    they arrived as <= 16-bit integers, float32 otherwise.  (Detector counts
    are non-negative: negative int8 / int16 values are clipped to 0 on the way
    to uint16, and float16 is widened to float32 -- the reference keeps any
    `itemsize <= 2` array as it is, ptycho.py:383-390.)"""
    if device is None:
        device = current_device()
    small = is_small_integer(x.dtype)
    if isinstance(x, torch.Tensor):
        t = x.to(device)
        if order is not None:
            idx = torch.as_tensor(order, device=t.device)
            if t.dtype == torch.uint16:  # no index kernels for uint16
                t = t.view(torch.int16).index_select(0, idx).view(torch.uint16)
            else:
                t = t.index_select(0, idx)
        if small:
            if t.dtype != torch.uint16:
                t = t.to(torch.int32).clamp_(min=0).to(torch.uint16)
            return t.contiguous()
        return t.to(torch.float32).contiguous()
    a = np.asarray(x)
    if small:
        if order is not None:
            a = a[order]
        a = np.ascontiguousarray(np.clip(a, 0, None).astype(np.uint16))
        return torch.from_numpy(a.view(np.int16)).to(device).view(torch.uint16)
    # float patterns: uploaded block by block into their final, permuted place
    # -- only the rows `order` names ever cross PCIe (a rank's share of a
    # multi-GPU job), and HBM never holds more than the result + one block
    if order is None:
        order = np.arange(a.shape[0])
    order = np.asarray(order, dtype=np.int64)
    out = torch.empty((len(order),) + tuple(a.shape[1:]), dtype=torch.float32,
                      device=device)
    if len(order) == 0:
        return out
    where = np.full(a.shape[0], -1, dtype=np.int64)  # source row -> place
    where[order] = np.arange(len(order))
    if np.count_nonzero(where >= 0) != len(order):
        # a row named twice: no inverse map; gather on the host block-wise
        for lo in range(0, len(order), _upload_block_rows(a)):
            hi = min(len(order), lo + _upload_block_rows(a))
            out[lo:hi] = torch.from_numpy(np.ascontiguousarray(
                a[order[lo:hi]], dtype=np.float32)).to(device)
        return out
    step = _upload_block_rows(a)
    for lo in range(0, a.shape[0], step):
        hi = min(a.shape[0], lo + step)
        place = where[lo:hi]
        wanted = place >= 0
        if not wanted.any():
            continue
        # (a block this job wants in full goes up as it lies: no host gather)
        rows = a[lo:hi] if wanted.all() else a[lo:hi][wanted]
        blk = torch.from_numpy(np.ascontiguousarray(rows, dtype=np.float32)).to(
            device)
        out.index_copy_(0, torch.from_numpy(place[wanted]).to(device), blk)
    return out


def _upload_block_rows(a, block_bytes=256 << 20):
    """Rows of `a` per upload block (about 256 MiB of float32)."""
    row = 4 * int(np.prod(a.shape[1:], dtype=np.int64)) or 4
    return max(1, block_bytes // row)


def has_invalid_counts(t, block_bytes=256 << 20):
    """True when the resident float patterns hold a negative or non-finite
    value (ptycho.py:392-397), tested block by block: the temporaries are a
    block's, not the dataset's; one read-back at the end."""
    if t.numel() == 0:
        return False
    step = max(1, block_bytes // max(1, 4 * t[0].numel()))
    bad = torch.zeros((), dtype=torch.bool, device=t.device)
    for lo in range(0, t.shape[0], step):
        blk = t[lo:lo + step]
        bad |= (~torch.isfinite(blk)).any() | (blk < 0).any()
    return bool(bad)


def data_f32(data, lo, hi):
    """Rows [lo, hi) of the resident data as float32 (a copy when the data is
    kept as uint16; kernels without a 16-bit loader take this)."""
    d = data[lo:hi]
    return d if d.dtype == torch.float32 else d.to(torch.float32)


def to_host(x):
    if isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def like_input(t, reference_input):
    """Return `t` as the same kind (host/device) as `reference_input`."""
    if isinstance(reference_input, torch.Tensor):
        return t
    return to_host(t)


def ptr(t):
    """Raw device pointer of a C-contiguous tensor (None passes through)."""
    if t is None:
        return None
    assert t.is_contiguous(), "C-ABI arguments must be C-contiguous"
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
