"""Scan-position helpers (reference src/tike/ptycho/position.py).

Only ``check_allowed_positions`` is on the accelerated path; position
correction (``PositionOptions``) is listed as "next" in DESIGN.md.
"""
from __future__ import annotations

import dataclasses

import numpy as np

from .. import _arrays as A


def check_allowed_positions(scan, psi, probe_shape):
    """Raise ValueError unless 1 <= floor(scan) <= psi.shape - probe - 1
    (position.py:600-628)."""
    scan = A.to_host(scan)
    int_scan = scan // 1
    min_corner = np.min(int_scan, axis=-2)
    max_corner = np.max(int_scan, axis=-2)
    valid_min_corner = (1, 1)
    valid_max_corner = (psi.shape[-2] - probe_shape[-2] - 1,
                        psi.shape[-1] - probe_shape[-1] - 1)
    if (min_corner[0] < valid_min_corner[0]
            or min_corner[1] < valid_min_corner[1]
            or max_corner[0] > valid_max_corner[0]
            or max_corner[1] > valid_max_corner[1]):
        raise ValueError(
            "Scan positions must be >= 1 and "
            "scan positions + 1 + probe.shape must be <= psi.shape. "
            "psi may be too small or the scan positions may be scaled wrong. "
            f"The span of scan is {min_corner} to {max_corner}, and "
            f"the shape of psi is {psi.shape}.")


@dataclasses.dataclass
class PositionOptions:
    """Placeholder with the reference's leading fields (position.py:330-377).

    Position correction is not accelerated yet: solvers raise
    NotImplementedError when ``PtychoParameters.position_options`` is set.
    """

    initial_scan: np.ndarray
    use_adaptive_moment: bool = False
    vdecay: float = 0.999
    mdecay: float = 0.9
    use_position_regularization: bool = False
    update_magnitude_limit: float = 0
    update_start: int = 0
