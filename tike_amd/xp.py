"""Array-module namespace exposed as ``Operator.xp``.

The reference exposes ``Operator.xp = cupy`` and its tests use a handful of
module functions through it (``asarray``, ``testing.assert_allclose``,
``zeros_like``, ``linalg.norm``...; tests/operators/util.py:42-54).  Device
arrays here are torch CUDA tensors.
"""
import types

import numpy as np
import torch

from . import _arrays

ndarray = torch.Tensor
float32 = np.float32
complex64 = np.complex64


def asarray(x, dtype=None, device=None):
    return _arrays.to_device(x, dtype=dtype, device=device)


array = asarray


def asnumpy(x):
    return _arrays.to_host(x)


def _shape_dtype(a, shape, dtype):
    shape = tuple(a.shape) if shape is None else tuple(shape)
    dtype = a.dtype if dtype is None else _arrays.torch_dtype(dtype)
    return shape, dtype


def zeros(shape, dtype=np.float32):
    return torch.zeros(tuple(np.atleast_1d(shape)),
                       dtype=_arrays.torch_dtype(dtype),
                       device=_arrays.current_device())


def empty(shape, dtype=np.float32):
    return torch.empty(tuple(np.atleast_1d(shape)),
                       dtype=_arrays.torch_dtype(dtype),
                       device=_arrays.current_device())


def zeros_like(a, dtype=None, shape=None):
    shape, dtype = _shape_dtype(a, shape, dtype)
    return torch.zeros(shape, dtype=dtype, device=a.device)


def empty_like(a, dtype=None, shape=None):
    shape, dtype = _shape_dtype(a, shape, dtype)
    return torch.empty(shape, dtype=dtype, device=a.device)


def get_array_module(*args):
    import sys
    return sys.modules[__name__]


def _assert_allclose(actual, desired, rtol=1e-7, atol=0, **kw):
    np.testing.assert_allclose(_arrays.to_host(actual), _arrays.to_host(desired),
                               rtol=rtol, atol=atol, **kw)


testing = types.SimpleNamespace(assert_allclose=_assert_allclose)
linalg = types.SimpleNamespace(norm=lambda x: torch.linalg.norm(x))


def sum(x, *a, **k):  # noqa: A001
    return torch.sum(x, *a, **k)


def sqrt(x):
    return torch.sqrt(x)
