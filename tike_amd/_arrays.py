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
