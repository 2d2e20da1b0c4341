"""Operators of the ptychography hot path (mirror of ``tike.operators``).

All device work goes through the C ABI of ``include/tike_amd.h``
(``tike_amd/csrc/libtike_amd.so``, hand-written HIP for gfx950).
"""
from .convolution import Convolution
from .fresnelspectprop import FresnelSpectProp
from .multislice import Multislice, SingleSlice
from .objective import (gaussian, gaussian_each_pattern, gaussian_grad,
                        poisson, poisson_each_pattern, poisson_grad)
from .operator import Operator
from .patch import Patch
from .propagation import Propagation, ZeroPropagation
from .ptycho import Ptycho

__all__ = [
    "Convolution", "FresnelSpectProp", "Multislice", "SingleSlice", "Operator", "Patch",
    "Propagation", "ZeroPropagation", "Ptycho", "gaussian",
    "gaussian_each_pattern", "gaussian_grad", "poisson",
    "poisson_each_pattern", "poisson_grad",
]
