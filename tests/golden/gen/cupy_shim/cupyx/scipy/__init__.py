"""SciPy forwarders for ``cupyx.scipy`` (fixture generation only)."""
import contextlib
import sys
import types

import scipy.fft as _sfft
import scipy.ndimage as _ndi
import scipy.stats as _stats

from cupy import _wrap


def _mk(name, target, extra=None):
    mod = types.ModuleType(name)

    def __getattr__(attr):
        obj = getattr(target, attr)
        if callable(obj):
            def f(*a, **kw):
                kw.pop('overwrite_x', None)
                return _wrap(obj(*a, **kw))
            return f
        return obj

    mod.__getattr__ = __getattr__
    for k, v in (extra or {}).items():
        setattr(mod, k, v)
    sys.modules[name] = mod
    return mod


def _fftn(a, *args, overwrite_x=False, **kw):
    return _wrap(_sfft.fftn(a, *args, **kw).astype(a.dtype, copy=False))


def _ifftn(a, *args, overwrite_x=False, **kw):
    return _wrap(_sfft.ifftn(a, *args, **kw).astype(a.dtype, copy=False))


fft = _mk('cupyx.scipy.fft', _sfft, dict(
    fftn=_fftn,
    ifftn=_ifftn,
    get_fft_plan=lambda *a, **k: contextlib.nullcontext(),
))
ndimage = _mk('cupyx.scipy.ndimage', _ndi)
stats = _mk('cupyx.scipy.stats', _stats)
