"""NumPy-backed stand-in for the ``cupy`` API surface the reference imports.

FIXTURE GENERATION ONLY.  The reference (``/root/reference/src/tike``) is
``from .cupy import *`` everywhere and has no CPU backend; CuPy is not
installed in the build container.  This module lets the reference's *own*
Python run on the CPU so that ``make_fixtures.py`` can record golden
input/output vectors.  Arrays are ``numpy.ndarray`` subclasses; every array
function forwards to NumPy/SciPy; ``RawModule`` dispatches the reference's
``convolution.cu`` (compiled for the host by ``emu.cpp``, included by path,
never copied).  Nothing here ships with the product or travels to the GPU box.
"""
import contextlib
import ctypes
import os
import sys
import types

import numpy as _np


class ndarray(_np.ndarray):
    """np.ndarray with .get()/.set() and 0-d results instead of scalars."""

    def get(self, *a, **k):
        return _np.asarray(self).copy()

    def set(self, x, *a, **k):
        self[...] = _np.asarray(x)

    def __array_wrap__(self, arr, context=None, return_scalar=False):
        return _np.asarray(arr).view(ndarray)

    def __array_finalize__(self, obj):
        pass

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        # CuPy resolves NumPy *scalars* (np.float64(...)) weakly, like Python
        # scalars; NumPy >= 2 (NEP 50) would promote complex64 arrays to
        # complex128.  Demote them to Python scalars before dispatch.
        ins = tuple(
            x.item() if isinstance(x, _np.generic) else
            (x.view(_np.ndarray) if isinstance(x, ndarray) else x)
            for x in inputs)
        if out is not None:
            kwargs['out'] = tuple(
                o.view(_np.ndarray) if isinstance(o, ndarray) else o
                for o in out)
        res = getattr(ufunc, method)(*ins, **kwargs)
        if out is not None:
            return out[0] if len(out) == 1 else out
        return _wrap(res)


def _wrap(x):
    if isinstance(x, _np.ndarray):
        return x.view(ndarray)
    if isinstance(x, (_np.generic,)):
        return _np.asarray(x).view(ndarray)
    if isinstance(x, tuple):
        return tuple(_wrap(i) for i in x)
    if isinstance(x, list):
        return [_wrap(i) for i in x]
    return x


def _fix_kwargs(kw):
    if 'axis' in kw and isinstance(kw['axis'], list):
        kw['axis'] = tuple(kw['axis'])
    return kw


def _forward(fn):

    def f(*a, **kw):
        return _wrap(fn(*a, **_fix_kwargs(kw)))

    f.__name__ = getattr(fn, '__name__', 'f')
    return f


def asarray(x, dtype=None, **kw):
    kw.pop('blocking', None)
    return _np.asarray(x, dtype=dtype, **kw).view(ndarray)


def array(x, dtype=None, **kw):
    return _np.array(x, dtype=dtype, **kw).view(ndarray)


def asnumpy(x, *a, **kw):
    return _np.asarray(x).view(_np.ndarray)


def get_array_module(*args):
    return sys.modules[__name__]


def fuse(*a, **kw):
    if len(a) == 1 and callable(a[0]) and not kw:
        return a[0]
    return lambda f: f


def percentile(a, q, axis=None, keepdims=False, **kw):
    if isinstance(axis, list):
        axis = tuple(axis)
    squeeze_q = isinstance(q, (list, tuple)) and len(q) == 1
    out = _np.percentile(a, q, axis=axis, keepdims=keepdims, **kw)
    if squeeze_q and keepdims:
        out = out[0]  # CuPy drops the leading q axis here
    return _wrap(out)


def zeros_like(a, dtype=None, shape=None, **kw):
    return _np.zeros(a.shape if shape is None else shape,
                     dtype=a.dtype if dtype is None else dtype).view(ndarray)


def empty_like(a, dtype=None, shape=None, **kw):
    return zeros_like(a, dtype=dtype, shape=shape)


def ones_like(a, dtype=None, shape=None, **kw):
    return _np.ones(a.shape if shape is None else shape,
                    dtype=a.dtype if dtype is None else dtype).view(ndarray)


def full_like(a, fill_value, dtype=None, shape=None, **kw):
    return _np.full(a.shape if shape is None else shape, fill_value,
                    dtype=a.dtype if dtype is None else dtype).view(ndarray)


def __getattr__(name):
    if hasattr(_np, name):
        obj = getattr(_np, name)
        if callable(obj) and not isinstance(obj, type):
            return _forward(obj)
        return obj
    raise AttributeError(name)


# dtypes / constants used as cp.<name>
float32, float64, complex64, complex128 = (_np.float32, _np.float64,
                                           _np.complex64, _np.complex128)
single, csingle, double, cdouble = _np.single, _np.csingle, _np.double, _np.cdouble
intc, int32, int64, uint16, bool_ = _np.intc, _np.int32, _np.int64, _np.uint16, _np.bool_
newaxis, pi, inf, nan = _np.newaxis, _np.pi, _np.inf, _np.nan

# ---- sub-modules ----------------------------------------------------------


class _Sub(types.ModuleType):

    def __init__(self, name, target):
        super().__init__(name)
        self._t = target

    def __getattr__(self, name):
        obj = getattr(self._t, name)
        if callable(obj) and not isinstance(obj, type):
            return _forward(obj)
        return obj


linalg = _Sub('cupy.linalg', _np.linalg)
fft = _Sub('cupy.fft', _np.fft)
testing = _Sub('cupy.testing', _np.testing)


class _Random(types.ModuleType):

    def __getattr__(self, name):
        obj = getattr(_np.random, name)
        if name == 'random':
            def random(size=None, dtype=_np.float64):
                return _wrap(_np.random.random(size).astype(dtype))
            return random
        if callable(obj):
            return _forward(obj)
        return obj


random = _Random('cupy.random')

# ---- cuda ----------------------------------------------------------------


class _Device:

    def __init__(self, device=None):
        self.id = 0 if device is None else int(device)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def synchronize(self):
        pass

    def use(self):
        pass


class _Stream:

    def __init__(self, *a, **k):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def wait_event(self, e):
        pass

    def record(self, e=None):
        return e

    def synchronize(self):
        pass


class _Event:

    def __init__(self, *a, **k):
        pass

    def synchronize(self):
        pass


class OutOfMemoryError(MemoryError):
    pass


cuda = types.ModuleType('cupy.cuda')
cuda.Device = _Device
cuda.Stream = _Stream
cuda.Event = _Event
cuda.get_current_stream = lambda: _Stream()
cuda.runtime = types.ModuleType('cupy.cuda.runtime')
cuda.runtime.getDeviceCount = lambda: 1
cuda.runtime.getDevice = lambda: 0
cuda.runtime.deviceSynchronize = lambda: None
cuda.memory = types.ModuleType('cupy.cuda.memory')
cuda.memory.OutOfMemoryError = OutOfMemoryError
cuda.cufft = types.ModuleType('cupy.cuda.cufft')
cuda.cufft.Plan1d = object
cuda.cufft.PlanNd = object
cuda.profiler = types.ModuleType('cupy.cuda.profiler')
cuda.profiler.start = lambda: None
cuda.profiler.stop = lambda: None
for _m in (cuda, cuda.runtime, cuda.memory, cuda.cufft, cuda.profiler, linalg,
           fft, testing, random):
    sys.modules[_m.__name__] = _m


class _Pool:

    def free_all_blocks(self):
        pass


def get_default_memory_pool():
    return _Pool()


def get_default_pinned_memory_pool():
    return _Pool()


# ---- RawModule: the reference's convolution.cu compiled for the host ------

_EMU = None


def _emu():
    global _EMU
    if _EMU is None:
        path = os.environ.get('TIKE_REF_EMU_LIB')
        if not path or not os.path.isfile(path):
            raise RuntimeError('TIKE_REF_EMU_LIB must point at libemu.so '
                               '(built by make_fixtures.py from emu.cpp)')
        _EMU = ctypes.CDLL(path)
    return _EMU


class _Kernel:
    attributes = {'max_threads_per_block': 1024}

    def __init__(self, name):
        self.name = name

    def __call__(self, grids, blocks, args):
        if '<float2,float2,float>' not in self.name.replace(' ', ''):
            raise NotImplementedError(self.name)
        lib = _emu()
        bx = int(blocks[0])
        images, patches, positions = args[:3]
        for x in (images, patches, positions):
            assert x.flags.c_contiguous, 'emu needs C-contiguous arrays'
        assert images.dtype == _np.complex64 and patches.dtype == _np.complex64
        assert positions.dtype == _np.float32
        ints = [ctypes.c_int(int(v)) for v in args[3:]]
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        if self.name.startswith('fwd_patch'):
            lib.fwd_patch_c64(ptr(images), ptr(patches), ptr(positions), *ints,
                              ctypes.c_int(bx))
        elif self.name.startswith('adj_patch'):
            lib.adj_patch_c64(ptr(images), ptr(patches), ptr(positions), *ints,
                              ctypes.c_int(bx))
        else:
            raise NotImplementedError(self.name)


class RawModule:

    def __init__(self, code=None, name_expressions=None, options=None, **kw):
        pass

    def get_function(self, name):
        return _Kernel(name)
