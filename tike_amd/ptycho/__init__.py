"""Ptychography solvers and helpers (mirror of ``tike.ptycho``)."""
from .exitwave import ExitWaveOptions
from .fresnel import MW_probe, single_probe
from .object import (ObjectOptions, get_absorbtion_image, get_padded_object,
                     positivity_constraint,
                     remove_object_ambiguity, smoothness_constraint)
from .position import (AffineTransform, PositionOptions,
                       affine_position_regularization, check_allowed_positions)
from .probe import (ProbeOptions, add_modes_cartesian_hermite,
                    add_modes_random_phase, adjust_probe_power,
                    apply_median_filter_abs_probe, constrain_center_peak,
                    constrain_probe_sparsity, constrain_variable_probe,
                    gaussian, get_varying_probe, init_varying_probe,
                    orthogonalize_eig, simulate_varying_weights)
from .ptycho import (Reconstruction, reconstruct, reconstruct_multigrid,
                     simulate)
from .solvers import (CgradOptions, LstsqOptions, PtychoParameters,
                      RpieOptions, cgrad, lstsq_grad, rpie,
                      update_preconditioners)
from . import probe, object, position, exitwave, solvers, io, learn, fresnel  # noqa: F401,A004

__all__ = [
    "CgradOptions", "ExitWaveOptions", "LstsqOptions", "ObjectOptions",
    "AffineTransform", "PositionOptions", "affine_position_regularization",
    "ProbeOptions", "PtychoParameters", "Reconstruction",
    "RpieOptions", "cgrad", "check_allowed_positions", "lstsq_grad",
    "reconstruct", "reconstruct_multigrid", "rpie", "simulate",
    "update_preconditioners",
]
