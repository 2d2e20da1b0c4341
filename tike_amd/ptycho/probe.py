"""Probe options, varying (eigen) probes and probe constraints
(reference src/tike/ptycho/probe.py).

Array functions are written with torch ops and run on whatever device their
inputs live on (CUDA tensors inside the solver; NumPy arrays are accepted and
returned as NumPy, computed on the host -- used by the CPU tests).
"""
from __future__ import annotations

import dataclasses
import typing

import numpy as np
import torch

from .. import _arrays as A
from .. import linalg, precision
from .. import random as trandom


def _t(x):
    """(tensor, was_numpy)"""
    if isinstance(x, torch.Tensor):
        return x, False
    return torch.from_numpy(np.ascontiguousarray(x)), True


def _back(t, was_numpy):
    return t.cpu().numpy() if was_numpy else t


def _to_dev(x):
    if x is None or A.is_device(x):
        return x
    h = np.asarray(x)
    return A.to_device(h, np.complex64 if np.iscomplexobj(h) else np.float32)


@dataclasses.dataclass
class ProbeOptions:
    """Settings and state of the probe update; same fields and defaults as the
    reference (probe.py:55-165)."""

    update_start: int = 0
    update_period: int = 1
    init_rescale_from_measurements: bool = True
    probe_photons: float = np.nan
    probe_wavelength: float = np.nan
    probe_FOV_lengths: typing.Tuple[float, float] = (np.nan, np.nan)
    force_orthogonality: bool = False
    force_centered_intensity: bool = False
    force_sparsity: float = 0.0
    use_adaptive_moment: bool = False
    vdecay: float = 0.999
    mdecay: float = 0.9
    v: typing.Any = dataclasses.field(init=False, default=None)
    m: typing.Any = dataclasses.field(init=False, default=None)
    probe_support: float = 0.0
    probe_support_radius: float = 0.5 * 0.7
    probe_support_degree: float = 2.5
    additional_probe_penalty: float = 0.0
    median_filter_abs_probe: bool = False
    median_filter_abs_probe_px: typing.Tuple[float, float] = (1.0, 1.0)
    preconditioner: typing.Any = dataclasses.field(init=False, default=None)
    power: typing.List[typing.List[float]] = dataclasses.field(
        init=False, default_factory=list)

    def recover_probe(self, epoch: int) -> bool:
        return (epoch >= self.update_start) and (epoch % self.update_period
                                                 == 0)

    def _copy(self, f):
        o = ProbeOptions(**{
            k.name: getattr(self, k.name)
            for k in dataclasses.fields(self) if k.init
        })
        o.power = self.power
        o.v, o.m = f(self.v), f(self.m)
        o.preconditioner = f(self.preconditioner)
        return o

    def resample(self, factor: float, interp=None) -> "ProbeOptions":
        """Settings for a grid rescaled by `factor`; the momentum and the
        preconditioner restart (probe.py:246-270)."""
        return self._copy(lambda x: None)

    def copy_to_device(self) -> "ProbeOptions":
        return self._copy(_to_dev)

    def copy_to_host(self) -> "ProbeOptions":
        o = self._copy(lambda x: None if x is None else A.to_host(x))
        # the per-epoch mode powers are appended as device tensors (no host
        # round trip per epoch); they become host arrays here
        o.power = [A.to_host(p) if A.is_device(p) else p for p in self.power]
        return o


def get_varying_probe(shared_probe, eigen_probe=None, weights=None):
    """The probe of every position: weights[:, 0] * shared probe plus, for the
    modes that own eigen probes, sum_c weights[:, c + 1] * eigen_probe[c]
    (probe.py:272-303).  shared_probe (..., 1, 1, S, H, W); eigen_probe
    (..., 1, C, S', H, W) with S' <= S; weights (..., N, C + 1, S) ->
    (..., N, 1, S, H, W).  Without weights: a copy of the shared probe.

    The solver never materialises this (the HIP kernels synthesise it on the
    fly); kept for the API and for tests.
    """
    shared, was = _t(shared_probe)
    if weights is None:
        return _back(shared.clone(), was)
    w = _t(weights)[0][..., None, None]  # (..., N, C + 1, S, 1, 1)
    unique = w[..., :1, :, :, :] * shared
    if eigen_probe is not None:
        eigen = _t(eigen_probe)[0]
        varying = eigen.shape[-3]
        unique[..., :varying, :, :] += torch.sum(
            w[..., 1:, :varying, :, :] * eigen, dim=-4, keepdim=True)
    return _back(unique, was)


def _clip_outliers(weights, percentile=0.95, slack=1.5):
    """Limit |weights| to `slack` times their `percentile` over the positions
    (axis -3; linear interpolation between order statistics, what
    np.percentile / cp.percentile compute), keeping the signs.  A sort, not
    torch.quantile, which is two orders of magnitude slower on ROCm here."""
    magnitude = weights.abs()
    ranked = torch.sort(magnitude, dim=-3).values
    at = percentile * (magnitude.shape[-3] - 1)
    below = int(np.floor(at))
    above = min(below + 1, magnitude.shape[-3] - 1)
    blend = float(at - below)
    limit = slack * torch.lerp(ranked[..., below:below + 1, :, :],
                               ranked[..., above:above + 1, :, :], blend)
    return torch.minimum(magnitude, limit) * torch.sign(weights)


def constrain_variable_probe(variable_probe, weights):
    """Keep the eigen-probe decomposition well posed (probe.py:306-359): unit
    mean-square eigen probes (their scale moves into the weights),
    Gram-Schmidt orthogonal across the eigen index, ordered by the power of
    their weights, and weight outliers clipped (`_clip_outliers`)."""
    eigen, was = _t(variable_probe)
    w, _ = _t(weights)
    varying = eigen.shape[-3]
    scale = linalg.mnorm(eigen, axis=(-2, -1), keepdims=True)
    w[..., 1:, :varying] *= scale[..., 0, 0]
    eigen = linalg.orthogonalize_gs(eigen / scale, axis=(-2, -1), N=-4)
    strength = torch.square(w[..., 1:, :varying]).sum(dim=-3)  # (..., C, S')
    for mode in range(varying):
        ranking = torch.argsort(strength[..., mode].flatten(),
                                descending=True)
        w[..., 1:, mode] = w[..., 1 + ranking, mode]
        eigen[..., :, mode, :, :] = eigen[..., ranking, mode, :, :]
    return _back(eigen, was), _back(_clip_outliers(w), was)


def orthogonalize_eig(x):
    """Orthogonalise probe modes via the eigenvectors of the pairwise dot
    products; sort by power (probe.py:726-769)."""
    xt, was = _t(x)
    nmodes = xt.shape[-3]
    flat = xt.reshape(*xt.shape[:-2], -1)
    # A[i, j] = sum conj(x_i) x_j by broadcast-multiply-reduce: S is tiny and
    # rocBLAS picks a 128x128 macro-tile GEMM for this (14 ms per call).
    A_ = (flat.conj()[..., :, None, :] * flat[..., None, :, :]).sum(-1)
    # The S x S eigen-problem is solved on the host with LAPACK: eigenvectors
    # are defined up to a phase, and the phase convention must be the same on
    # every rank and reproducible against the CPU oracle (one tiny D2H per
    # epoch).
    val, vectors = np.linalg.eigh(A_.detach().cpu().numpy(), UPLO="U")
    vectors = torch.from_numpy(vectors).to(device=xt.device, dtype=xt.dtype)
    result = (vectors.swapaxes(-1, -2)[..., :, :, None] *
              flat[..., None, :, :]).sum(-2).reshape(xt.shape)
    power = torch.square(linalg.norm(result, axis=(-2, -1))).flatten()
    order = torch.argsort(power, stable=True).flip(0)
    result = result[..., order, :, :]
    power = power[order]
    return _back(result, was), _back(power, was)


def power(probe):
    """Power of each probe mode (probe.py:772-781)."""
    p, was = _t(probe)
    return _back(torch.square(linalg.norm(p, axis=(-2, -1))).flatten(), was)


def gaussian(size, rin=0.8, rout=1.0):
    """Flat-top probe amplitude on a size x size grid (probe.py:784-814): 1
    inside the radius rin, 0 outside rout, a linear ramp between them; radii
    are measured from the grid centre to pixel centres, in units of the
    half-diagonal scaled by sqrt(2)/2 (rout is one pixel wider)."""
    offset = (np.arange(size) + 0.5) - size / 2
    radius = np.sqrt(np.add.outer(offset**2, offset**2))
    reach = np.sqrt(2) * 0.5 * radius.max()
    inner, outer = rin * reach, rout * reach + 1.0
    ramp = (outer - radius) / (outer - inner)
    return np.select([radius < inner, (radius > inner) & (radius < outer)],
                     [1.0, ramp], 0.0).astype(precision.floating)


def adjust_probe_power(probe, power=None):
    """Give the modes the relative powers `power` (default 1, 1/2, ... 1/M) in
    units of the first mode's norm; rescales `probe` in place and returns it
    (probe.py:479-497)."""
    modes = probe.shape[-3]
    share = 1.0 / np.arange(1, modes + 1) if power is None else power
    amplitude = np.sqrt(np.sum(np.abs(probe)**2, axis=(-2, -1), keepdims=True))
    gain = share[..., None, None] * amplitude[..., :1, :, :]
    np.multiply(probe, gain / amplitude, out=probe)
    return probe


def add_modes_random_phase(probe, nmodes):
    """A probe of `nmodes` modes: the given ones, then copies of mode 0 under
    random linear phase ramps -- tilts of up to half a period across the
    window in x and y, two legacy-generator draws per new mode (Odstrcil et
    al., Opt. Express 24, 8360; probe.py:500-531)."""
    given = probe.shape[-3]
    width = probe.shape[-1]
    coordinate = (np.arange(width) + 0.5) / width - 0.5
    modes = [probe[..., m, :, :] for m in range(min(given, nmodes))]
    for _ in range(nmodes - given):
        tilt_x, tilt_y = np.random.rand(2) - 0.5
        ramp_x = np.exp(-2j * np.pi * tilt_x * coordinate)
        ramp_y = np.exp(-2j * np.pi * tilt_y * coordinate)
        modes.append(probe[..., 0, :, :] * ramp_x * ramp_y[:, None])
    return np.stack(modes, axis=-3).astype(probe.dtype, copy=False)


def _jittered_weights(shape, varying):
    """Eigen-probe weights before anything is known: 1 for the shared probe
    (index 0 of axis -2), a zero-mean 1e-6 jitter over the positions (axis -3)
    for the eigen probes of the first `varying` modes, 0 for the rest.  The
    whole block is drawn from the legacy generator, used or not."""
    jitter = 1e-6 * np.random.rand(*shape).astype(precision.floating)
    jitter -= jitter.mean(axis=-3, keepdims=True)
    weights = np.zeros_like(jitter)
    weights[..., 1:, :varying] = jitter[..., 1:, :varying]
    weights[..., 0, :] = 1.0
    return weights


def init_varying_probe(scan, shared_probe, num_eigen_probes,
                       probes_with_modes=1):
    """(eigen_probe, weights) to start an orthogonal-probe-relaxation run
    (probe.py:660-723).  weights (..., N, num_eigen_probes, S) float32, see
    `_jittered_weights`; eigen_probe (..., 1, num_eigen_probes - 1,
    probes_with_modes, H, W): unit-mean-square complex noise from
    `tike_amd.random.randomizer_np`, drawn after the weights (None when there
    is nothing besides the shared probe)."""
    *lead, _, modes, height, width = shared_probe.shape
    varying = max(probes_with_modes, 0)
    if varying > modes:
        raise ValueError(f"probes_with_modes ({varying}) cannot be more than "
                         f"the number of probes ({modes})!")
    eigen_probe = weights = None
    if num_eigen_probes >= 1:
        weights = _jittered_weights(
            (*scan.shape[:-1], num_eigen_probes, modes), varying)
    if num_eigen_probes >= 2:
        noise = trandom.numpy_complex(*lead, num_eigen_probes - 1, varying,
                                      height, width)
        eigen_probe = noise / linalg.mnorm(noise, axis=(-2, -1),
                                           keepdims=True)
    return eigen_probe, weights


def finite_probe_support(probe, *, radius=0.5, degree=5.0, p=1.0):
    """Penalty p * (1 - exp(-(r / radius)^(2 degree))) on the probe grid, r
    measured in window widths from the centre (probe.py:919-964): 0 in the
    middle, p far outside `radius`.  0.0 when p <= 0."""
    if p <= 0:
        return 0.0
    width = probe.shape[-1]
    axis = (torch.arange(width, device=probe.device, dtype=torch.float32) +
            0.5) / width - 0.5
    r2 = (torch.square(axis)[:, None] + torch.square(axis)[None, :]) / (
        radius * radius)
    return (p * -torch.expm1(-r2**degree)).to(torch.float32)


def rescale_probe_using_fixed_intensity_photons(probe, Nphotons,
                                                probe_power_fraction=None):
    """Scale every shared mode so that the modes hold Nphotons in total, split
    as `probe_power_fraction` (default: as they are) (probe.py:967-993)."""
    energy = probe.abs().square().sum(dim=(-2, -1), keepdim=True)
    share = (energy / energy.sum() if probe_power_fraction is None else
             torch.as_tensor(probe_power_fraction,
                             device=probe.device)[..., None, None])
    return probe * torch.sqrt(share * Nphotons / energy)


# ---------------------------------------------------------- probe constraints
def _gaussian_taps(sigma, truncate, device):
    """Normalised taps of scipy.ndimage's gaussian_filter1d (radius =
    int(truncate * sigma + 0.5))."""
    radius = int(truncate * float(sigma) + 0.5)
    x = torch.arange(-radius, radius + 1, dtype=torch.float64, device=device)
    w = torch.exp(-0.5 * x * x / float(sigma)**2)
    return (w / w.sum()), radius


def _smooth_intensity(intensity, sigma, *, wrap, truncate):
    """Separable Gaussian blur of a real (H, W) image, in float64 like
    scipy.ndimage.gaussian_filter on a float32 input accumulates; borders:
    zeros (mode='constant') or periodic (mode='wrap')."""
    img = intensity.to(torch.float64)
    for axis, s in ((0, sigma[0]), (1, sigma[1])):
        w, radius = _gaussian_taps(s, truncate, img.device)
        rows = img if axis == 1 else img.T
        if wrap:
            n = rows.shape[1]
            idx = torch.arange(-radius, n + radius, device=img.device) % n
            padded = rows[:, idx]
        else:
            padded = torch.nn.functional.pad(rows, (radius, radius))
        out = torch.nn.functional.conv1d(padded[:, None, :], w[None, None, :])[:, 0]
        img = out if axis == 1 else out.T
    return img.to(intensity.dtype)


def constrain_center_peak(probe):
    """Move the peak of the smoothed combined intensity towards the centre of
    the probe grid, one pixel per call (probe.py:817-856): blur with sigma =
    half / 3 (zero borders, 6 sigma), centre of mass, integer step of at most
    one pixel along each axis, vacated pixels zero."""
    p, was = _t(probe)
    h, w = p.shape[-2:]
    half = (h // 2, w // 2)
    stack = p.reshape(-1, h, w)
    intensity = _smooth_intensity(
        torch.sum(stack.abs()**2, dim=0), (half[0] / 3, half[1] / 3),
        wrap=False, truncate=6.0)
    total = intensity.sum()
    cy = torch.round((intensity.sum(1) * torch.arange(
        h, device=p.device, dtype=intensity.dtype)).sum() / total)
    cx = torch.round((intensity.sum(0) * torch.arange(
        w, device=p.device, dtype=intensity.dtype)).sum() / total)
    sy = int(min(1, max(-1, half[0] - float(cy))))
    sx = int(min(1, max(-1, half[1] - float(cx))))
    shifted = torch.zeros_like(stack)
    src_y = slice(max(0, -sy), h - max(0, sy))
    dst_y = slice(max(0, sy), h - max(0, -sy))
    src_x = slice(max(0, -sx), w - max(0, sx))
    dst_x = slice(max(0, sx), w - max(0, -sx))
    shifted[:, dst_y, dst_x] = stack[:, src_y, src_x]
    return _back(shifted.reshape(p.shape).contiguous(), was)


def apply_median_filter_abs_probe(probe, med_filt_px):
    """Median-filter the amplitude of every shared mode, keep the phase
    (probe.py:859-893).  Window (a, b) pixels (integers, as CuPy's
    median_filter takes its size), zero borders; for an even window the upper
    of the two middle values, as scipy / CuPy rank filters pick."""
    p, was = _t(probe)
    a, b = (max(1, int(v)) for v in med_filt_px)
    modes = p[0, 0]
    amp = modes.abs()
    S, h, w = amp.shape
    # window of pixel i: [i - a // 2, i - a // 2 + a)
    padded = torch.nn.functional.pad(amp, (b // 2, b - 1 - b // 2, a // 2,
                                           a - 1 - a // 2))
    windows = padded.unfold(1, a, 1).unfold(2, b, 1).reshape(S, h, w, a * b)
    filtered = torch.kthvalue(windows, (a * b) // 2 + 1, dim=-1).values
    out = p.clone()
    out[0, 0] = torch.polar(filtered, torch.angle(modes))
    return _back(out, was)


def constrain_probe_sparsity(probe, f):
    """Zero, in every mode, the fraction `f` of pixels where the smoothed
    combined intensity is smallest (probe.py:896-916; blur sigma = size / 8,
    periodic borders)."""
    if f == 0:
        return probe
    p, was = _t(probe)
    h, w = p.shape[-2:]
    intensity = _smooth_intensity(
        torch.sum(p.reshape(-1, h, w).abs()**2, dim=0), (h / 8, w / 8),
        wrap=True, truncate=4.0)
    k = int(f * h * w)
    out = p.clone()
    if k > 0:
        smallest = torch.topk(intensity.reshape(-1), k, largest=False).indices
        out.reshape(*p.shape[:-2], h * w)[..., smallest] = 0
    return _back(out, was)


def add_modes_cartesian_hermite(probe, nmodes):
    """More probe modes from one: the probe times 2-D Cartesian Hermite-like
    polynomials about its intensity centroid (Gaussian-windowed with its
    second moments), Gram-Schmidt orthonormalised in order (probe.py:534-644;
    Odstrcil et al., Opt. Express 2018).  Host arrays."""
    if nmodes < 1:
        raise ValueError(f"nmodes cannot be less than 1. It was {nmodes}.")
    if probe.ndim < 3:
        raise ValueError("probe is incorrect shape is should be "
                         f" (..., 1, W, H) not {probe.shape}.")
    probe = np.asarray(probe)
    M = int(np.ceil(np.sqrt(nmodes)))
    N = int(np.ceil(nmodes / M))
    off = probe.shape[-2] // 2 - 1
    X, Y = np.meshgrid(np.arange(probe.shape[-2]) - off,
                       np.arange(probe.shape[-1]) - off, indexing="xy")
    weight = np.abs(probe)**2
    total = weight.sum(axis=(-2, -1), keepdims=True)
    moment = lambda g: (g * weight).sum(axis=(-2, -1), keepdims=True) / total
    dx, dy = X - moment(X), Y - moment(Y)
    window = np.exp(-dx**2 / (2 * moment(dx**2)) - dy**2 / (2 * moment(dy**2)))
    norm = lambda a: np.sqrt(
        np.sum(np.abs(a)**2, axis=(-2, -1), keepdims=True))
    modes = []
    for count in range(nmodes):
        ny, mx = divmod(count, M)
        assert ny < N
        basis = dx**mx * dy**ny * probe
        if count:
            basis = basis * window
        basis = basis / norm(basis)
        for earlier in modes:
            basis = basis - earlier * np.sum(
                np.conj(earlier) * basis, axis=(-2, -1), keepdims=True)
        modes.append(basis / norm(basis))
    return np.concatenate(modes, axis=-3)


def simulate_varying_weights(scan, eigen_probe):
    """Sinusoidal eigen-probe weights along the scan, for simulations: unit
    amplitude, a random period of at most one scan length and a random phase
    per eigen probe and mode (probe.py:647-657; legacy generator, the periods
    drawn before the phases)."""
    count = scan.shape[1]
    period, phase = np.random.rand(2, *eigen_probe.shape[:-2])
    index = np.arange(count).reshape(count, 1, 1)
    return np.sin(2 * np.pi / (count * period) * index - 2 * np.pi * phase)
