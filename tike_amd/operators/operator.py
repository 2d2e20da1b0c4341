"""Operator base class (reference src/tike/operators/cupy/operator.py:12-57)."""
from abc import ABC

from .. import _arrays
from .. import xp as _xp


class Operator(ABC):
    """Context manager providing the forward and adjoint of a linear map."""

    xp = _xp
    """Array module of the device arrays this operator accepts and returns."""

    @classmethod
    def asarray(cls, *args, device=None, **kwargs):
        return _arrays.to_device(*args, device=device, **kwargs)

    @classmethod
    def asnumpy(cls, *args, **kwargs):
        return _arrays.to_host(*args, **kwargs)

    def __enter__(self):
        return self

    def __exit__(self, type, value, traceback):
        pass

    def fwd(self, **kwargs):
        raise NotImplementedError("The forward operator was not implemented!")

    def adj(self, **kwargs):
        raise NotImplementedError("The adjoint operator was not implemented!")
