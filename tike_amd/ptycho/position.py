"""Scan-position helpers and position correction
(reference src/tike/ptycho/position.py).

On the accelerated path: ``check_allowed_positions``, ``PositionOptions``
(state of the position correction inside ``lstsq_grad``), the affine
regularisation of updated positions (host-side, a few hundred flops per epoch)
and the Gaussian-derivative taps handed to ``tike_position_sums``.
"""
from __future__ import annotations

import dataclasses

import numpy as np
import torch

from .. import _arrays as A
from .. import precision
from .. import random as trandom


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


# ------------------------------------------------------------------ transform
@dataclasses.dataclass
class AffineTransform:
    """A 2-D affine transformation scale @ shear @ rotate + translate
    (position.py:137-249)."""

    scale0: float = 1.0
    scale1: float = 1.0
    shear1: float = 0.0
    angle: float = 0.0
    t0: float = 0.0
    t1: float = 0.0

    def resample(self, factor: float) -> "AffineTransform":
        return AffineTransform(self.scale0, self.scale1, self.shear1,
                               self.angle, self.t0 * factor, self.t1 * factor)

    @classmethod
    def frombuffer(cls, buffer) -> "AffineTransform":
        return AffineTransform(*buffer)

    def asbuffer(self) -> np.ndarray:
        return np.array(self.astuple())

    @classmethod
    def fromarray(cls, T) -> "AffineTransform":
        """Decompose a 2x2 (or 3x2) matrix, Graphics Gems 2 section 7.1
        (position.py:166-193)."""
        T = np.asarray(T)
        R = T[:2, :2].copy()
        scale0 = np.linalg.norm(R[0])
        if scale0 <= 0:
            return AffineTransform()
        R[0] /= scale0
        shear1 = R[0] @ R[1]
        R[1] -= shear1 * R[0]
        scale1 = np.linalg.norm(R[1])
        if scale1 <= 0:
            return AffineTransform()
        R[1] /= scale1
        shear1 /= scale1
        angle = np.arccos(R[0, 0])
        return AffineTransform(
            scale0=float(scale0), scale1=float(scale1), shear1=float(shear1),
            angle=float(angle),
            t0=float(T[2, 0] if T.shape[0] > 2 else 0),
            t1=float(T[2, 1] if T.shape[0] > 2 else 0))

    def asarray(self, xp=np) -> np.ndarray:
        """2x2 matrix scale @ shear @ rotate (position.py:195-220)."""
        cosx, sinx = np.cos(self.angle), np.sin(self.angle)
        f = precision.floating
        return (np.array([[self.scale0, 0.0], [0.0, self.scale1]], dtype=f)
                @ np.array([[1.0, 0.0], [self.shear1, 1.0]], dtype=f)
                @ np.array([[+cosx, -sinx], [+sinx, +cosx]], dtype=f))

    def asarray3(self, xp=np) -> np.ndarray:
        T = np.empty((3, 2), dtype=precision.floating)
        T[2] = (self.t0, self.t1)
        T[:2, :2] = self.asarray()
        return T

    def astuple(self) -> tuple:
        return (self.scale0, self.scale1, self.shear1, self.angle, self.t0,
                self.t1)

    def __call__(self, x, gpu=False, shift=True):
        if A.is_device(x):
            M = torch.as_tensor(self.asarray(), device=x.device, dtype=x.dtype)
            r = x @ M
            if shift:
                r = r + torch.tensor((self.t0, self.t1), device=x.device,
                                     dtype=x.dtype)
            return r
        r = x @ self.asarray()
        if shift:
            r = r + np.array((self.t0, self.t1))
        return r


def estimate_global_transformation(positions0, positions1, weights=None,
                                   transform=None):
    """Weighted least squares for the global affine transformation
    (position.py:252-270); host arrays."""
    a = np.pad(positions0, ((0, 0), (0, 1)), constant_values=1)
    b = positions1
    try:
        if weights is not None:
            w = np.sqrt(weights[..., None])
            a, b = a * w, b * w
        aT = a.conj().swapaxes(-1, -2)
        result = AffineTransform.fromarray(np.linalg.inv(aT @ a) @ aT @ b)
    except np.linalg.LinAlgError:
        # singular when the positions are colinear
        result = AffineTransform()
    return result, np.linalg.norm(result(positions0) - positions1)


def ransac_subsets(n, min_sample=4, max_iter=20):
    """The random subsets of one RANSAC fit over n positions
    (position.py:296-300), from the library's generator."""
    return trandom.randomizer_np.choice(a=n, size=(max_iter, min_sample),
                                        replace=True)


def estimate_global_transformation_ransac(positions0, positions1, weights=None,
                                          transform=None, min_sample=4,
                                          max_error=32, min_consensus=0.75,
                                          max_iter=20, subsets=None):
    """RANSAC estimate of the global affine transformation
    (position.py:273-327); the subsets come from ``tike_amd.random.
    randomizer_np`` exactly as the reference draws them (`subsets`: drawn
    earlier by `ransac_subsets`, for a fit that is carried out later)."""
    transform = AffineTransform() if transform is None else transform
    best_fitness = np.inf
    if subsets is None:
        subsets = ransac_subsets(len(positions0), min_sample, max_iter)
    for subset in subsets:
        candidate, _ = estimate_global_transformation(
            positions0[subset], positions1[subset], weights, transform)
        error = np.linalg.norm(candidate(positions0) - positions1, axis=-1)
        inliers = error <= max_error
        if np.sum(inliers) / len(inliers) >= min_consensus:
            candidate, fitness = estimate_global_transformation(
                positions0[inliers], positions1[inliers], weights, candidate)
            if fitness < best_fitness:
                best_fitness = fitness
                transform = candidate
    return transform, best_fitness


def gaussian_derivative_taps(sigma=0.333, truncate=6.0):
    """Taps t[d], d = -r..r, of ``gaussian_gradient`` (position.py:779-810):
    g[i] = sum_d t[d] x[i + d], i.e. scipy's ``gaussian_filter1d(-x, sigma,
    order=1)``: a normalised Gaussian times -d / sigma^2, reversed."""
    r = int(truncate * float(sigma) + 0.5)
    d = np.arange(-r, r + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * d**2)
    phi = phi / phi.sum()
    return (-(d / (sigma * sigma)) * phi).astype(np.float32), r


def gaussian_gradient(x, sigma=0.333):
    """First-order Gaussian derivatives of `x` along its last two axes
    (position.py:779-810): ``gaussian_filter1d(-x, sigma, order=1,
    mode='nearest', truncate=6)`` per axis.  The solver applies the same taps
    inside ``tike_position_sums``; this is the array-level form."""
    t, was_numpy = (x, False) if isinstance(x, torch.Tensor) else (
        torch.from_numpy(np.ascontiguousarray(x)), True)
    taps, r = gaussian_derivative_taps(sigma)
    taps = torch.from_numpy(taps).to(t.device)
    out = []
    for axis in (-2, -1):
        n = t.shape[axis]
        g = torch.zeros_like(t)
        for d in range(-r, r + 1):
            idx = torch.clamp(torch.arange(n, device=t.device) + d, 0, n - 1)
            g = g + taps[d + r] * t.index_select(axis % t.ndim, idx)
        out.append(g.cpu().numpy() if was_numpy else g)
    return tuple(out)


# -------------------------------------------------------------------- options
@dataclasses.dataclass
class PositionOptions:
    """Data and settings of the position correction (position.py:330-598)."""

    initial_scan: np.ndarray
    use_adaptive_moment: bool = False
    vdecay: float = 0.999
    mdecay: float = 0.9
    use_position_regularization: bool = False
    update_magnitude_limit: float = 0
    transform: AffineTransform = dataclasses.field(
        default_factory=AffineTransform)
    origin: np.ndarray = dataclasses.field(
        default_factory=lambda: np.zeros(2))
    confidence: np.ndarray = None
    update_start: int = 0
    _momentum: np.ndarray = dataclasses.field(init=False, default=None)

    def __post_init__(self):
        if A.is_device(self.initial_scan):
            self.initial_scan = self.initial_scan.to(torch.float32)
            n = tuple(self.initial_scan.shape)
            if self.confidence is None:
                self.confidence = torch.ones_like(self.initial_scan)
            if self.use_adaptive_moment:
                self._momentum = torch.zeros((*n[:-1], 4), dtype=torch.float32,
                                             device=self.initial_scan.device)
            return
        self.initial_scan = np.asarray(self.initial_scan).astype(
            precision.floating)
        if self.confidence is None:
            self.confidence = np.ones(self.initial_scan.shape,
                                      dtype=precision.floating)
        if self.use_adaptive_moment:
            self._momentum = np.zeros((*self.initial_scan.shape[:-1], 4),
                                      dtype=precision.floating)

    def _like(self, initial_scan, confidence, momentum):
        new = PositionOptions(
            initial_scan, use_adaptive_moment=self.use_adaptive_moment,
            vdecay=self.vdecay, mdecay=self.mdecay,
            use_position_regularization=self.use_position_regularization,
            update_magnitude_limit=self.update_magnitude_limit,
            transform=self.transform, origin=self.origin,
            confidence=confidence, update_start=self.update_start)
        if self.use_adaptive_moment:
            new._momentum = momentum
        return new

    def split(self, indices) -> "PositionOptions":
        """Keep only the positions in `indices` (position.py:432-447)."""
        return self._like(
            self.initial_scan[..., indices, :],
            None if self.confidence is None else
            self.confidence[..., indices, :],
            None if self._momentum is None else
            self._momentum[..., indices, :])

    def insert(self, other, indices):
        self.initial_scan[..., indices, :] = other.initial_scan
        if self.confidence is not None:
            self.confidence[..., indices, :] = other.confidence
        if self.use_adaptive_moment:
            self._momentum[..., indices, :] = other._momentum
        return self

    @staticmethod
    def join(x, reorder):
        """Concatenate per-worker options and undo the ordering
        (position.py:458-489)."""
        if None in x:
            return None
        cat = lambda parts: np.concatenate(parts, axis=0)[reorder]
        return x[0]._like(
            cat([e.initial_scan for e in x]),
            None if x[0].confidence is None else
            cat([e.confidence for e in x]),
            None if x[0]._momentum is None else cat([e._momentum for e in x]))

    def copy_to_device(self):
        d = lambda v: None if v is None else A.to_device(v, np.float32)
        return self._like(d(self.initial_scan), d(self.confidence),
                          d(self._momentum))

    def copy_to_host(self):
        h = lambda v: None if v is None else A.to_host(v)
        return self._like(h(self.initial_scan), h(self.confidence),
                          h(self._momentum))

    def resample(self, factor: float) -> "PositionOptions":
        new = PositionOptions(
            self.initial_scan * factor,
            use_adaptive_moment=self.use_adaptive_moment, vdecay=self.vdecay,
            mdecay=self.mdecay,
            use_position_regularization=self.use_position_regularization,
            update_magnitude_limit=self.update_magnitude_limit,
            transform=self.transform.resample(factor),
            confidence=self.confidence, update_start=self.update_start,
            origin=self.origin * factor)
        return new  # momentum restarts at zero when the grid changes

    # second (v) and first (m) moments of ADAM, packed as the reference does
    @property
    def v(self):
        return self._momentum[..., 0:2]

    @v.setter
    def v(self, x):
        self._momentum[..., 0:2] = x

    @property
    def m(self):
        return self._momentum[..., 2:4]

    @m.setter
    def m(self, x):
        self._momentum[..., 2:4] = x


def affine_position_regularization(updated, position_options, max_error=32,
                                   *, positions0=None, positions1=None,
                                   relax=0.9):
    """Fit the global affine transformation between the initial and the
    updated positions and, if asked, pull the positions towards it
    (position.py:716-776).

    `updated` / `position_options.initial_scan` are this rank's positions;
    `positions0` / `positions1` are the host arrays the fit uses (all
    positions of the job; default: this rank's own).
    """
    origin = A.to_host(position_options.origin)
    if positions0 is None:
        positions0 = A.to_host(position_options.initial_scan)
        positions1 = A.to_host(updated)
    new_transform, _ = estimate_global_transformation_ransac(
        positions0=positions0 - origin, positions1=positions1 - origin,
        transform=position_options.transform, max_error=max_error)
    position_options.transform = new_transform
    if position_options.use_position_regularization:
        predicted = new_transform(position_options.initial_scan, shift=False)
        updated = updated * (1 - relax) + relax * predicted
        if A.is_device(updated):
            updated = updated.to(torch.float32)
        else:
            updated = updated.astype(precision.floating)
    return updated, position_options
