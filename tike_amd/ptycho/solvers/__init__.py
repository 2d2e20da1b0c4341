"""Solver implementations (mirror of ``tike.ptycho.solvers``)."""
from ._preconditioner import update_preconditioners
from .cgrad import cgrad
from .lstsq import lstsq_grad
from .rpie import rpie
from .options import (CgradOptions, IterativeOptions, LstsqOptions,
                      crop_fourier_space, pad_fourier_space, _resize_fft,
                      _resize_spline,
                      PtychoParameters, RpieOptions)

__all__ = [
    "cgrad", "CgradOptions", "IterativeOptions", "lstsq_grad", "LstsqOptions",
    "PtychoParameters", "rpie", "RpieOptions", "update_preconditioners",
]
