"""Object (psi) options and constraints
(reference src/tike/ptycho/object.py)."""
from __future__ import annotations

import copy
import dataclasses
import typing

import numpy as np
import torch

from .. import _arrays as A
from .. import linalg, precision


def _to_dev(x):
    if x is None:
        return None
    h = A.to_host(x) if not A.is_device(x) else None
    if A.is_device(x):
        return x
    return A.to_device(h, np.complex64 if np.iscomplexobj(h) else np.float32)


@dataclasses.dataclass
class ObjectOptions:
    """Settings and state of the object update; same fields and defaults as
    the reference (object.py:25-81)."""

    convergence_tolerance: float = 0
    update_mnorm: typing.List[float] = dataclasses.field(init=False,
                                                         default_factory=list)
    positivity_constraint: float = 0
    smoothness_constraint: float = 0
    use_adaptive_moment: bool = False
    vdecay: float = 0.999
    mdecay: float = 0.9
    v: typing.Any = dataclasses.field(init=False, default=None)
    m: typing.Any = dataclasses.field(init=False, default=None)
    preconditioner: typing.Any = dataclasses.field(init=False, default=None)
    clip_magnitude: bool = False
    multislice_propagation_distance: float = 1.0e-9

    def _copy(self, f):
        """The settings as they are; the arrays the solver keeps here (ADAM
        moments, preconditioner) passed through `f`."""
        twin = dataclasses.replace(self)  # every constructor field
        twin.update_mnorm = list(self.update_mnorm)
        for name in ("v", "m", "preconditioner"):
            setattr(twin, name, f(getattr(self, name)))
        return twin

    def resample(self, factor: float, interp=None) -> "ObjectOptions":
        """Settings for a grid rescaled by `factor`; the momentum and the
        preconditioner restart (object.py:138-152)."""
        return self._copy(lambda x: None)

    def copy_to_device(self) -> "ObjectOptions":
        return self._copy(_to_dev)

    def copy_to_host(self) -> "ObjectOptions":
        return self._copy(lambda x: None if x is None else A.to_host(x))


def positivity_constraint(x, r):
    """Move the object the fraction r of the way to its own magnitude (which
    drains the phase: a "positive" object), r in [0, 1] (object.py:207-223)."""
    if r > 1:
        raise ValueError(
            f"Positivity constraint must be in the range [0, 1] not {r}.")
    return x if r <= 0 else torch.lerp(x, x.abs().to(x.dtype), float(r))


def smoothness_constraint(x, a):
    """3x3 averaging kernel with centre 1-8a, 'nearest' edges
    (object.py:226-253)."""
    if 0 <= a and a < 1.0 / 8.0:
        w = torch.full((1, 1, 3, 3), a, dtype=torch.float32, device=x.device)
        w[..., 1, 1] = 1.0 - 8.0 * a

        def conv(p):
            p = torch.nn.functional.pad(p[:, None], (1, 1, 1, 1),
                                        mode="replicate")
            return torch.nn.functional.conv2d(p, w)[:, 0]

        return torch.complex(conv(x.real), conv(x.imag))
    raise ValueError(
        f"Smoothness constraint must be in range [0, 1/8) not {a}.")


def get_padded_object(scan, probe, extra: int = 0):
    """(psi, scan): the smallest object of 0.5 + 0j that holds every patch
    with a one-pixel border plus `extra`, and the positions moved into it
    (object.py:256-274)."""
    corner = np.floor(scan)
    first = corner.min(axis=-2)
    pixels = corner.max(axis=-2) - first + (probe.shape[-1] + 2 + 2 * extra)
    psi = np.full(pixels.astype(precision.integer), 0.5 + 0j,
                  dtype=precision.cfloating)
    return psi, scan + (1 + extra) - first


def remove_object_ambiguity(psi, probe, preconditioner):
    """Normalise the object / probe scaling ambiguity (object.py:324-335)."""
    W = preconditioner.real
    W = W / linalg.mnorm(W)
    object_norm = 2 * torch.sqrt(torch.mean(torch.square(psi.abs()) * W))
    return psi / object_norm, probe * object_norm


def get_absorbtion_image(data, scan, *, rescale=1.0, method="cubic"):
    """Approximate scanning-transmission image: the total counts of every
    diffraction pattern interpolated (scipy.interpolate.griddata) onto the
    unit grid spanned by the rescaled scan positions (object.py:281-322)."""
    import scipy.interpolate
    data, scan = np.asarray(A.to_host(data)), np.asarray(A.to_host(scan))
    rescaled = scan * rescale
    axes = [np.arange(np.floor(rescaled[:, k].min()),
                      np.ceil(rescaled[:, k].max())) for k in (0, 1)]
    coord0, coord1 = np.meshgrid(*axes, indexing="ij")
    values = np.sum(np.abs(data.astype(np.float64))**2, axis=(-2, -1))
    image = scipy.interpolate.griddata(
        points=rescaled, values=values,
        xi=(coord0.ravel(), coord1.ravel()), method=method,
        fill_value=np.amax(values))
    return image.reshape(coord0.shape)
