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
    """weights[0]*probe + sum_c weights[c+1]*eigen[c] (probe.py:272-303).

    The solver never materialises this (the HIP kernels synthesise it on the
    fly); kept for the API and for tests.
    """
    sp, was = _t(shared_probe)
    if weights is None:
        return _back(sp.clone(), was)
    w, _ = _t(weights)
    unique = w[..., [0], :, None, None] * sp
    if eigen_probe is not None:
        ep, _ = _t(eigen_probe)
        m = ep.shape[-3]
        for c in range(ep.shape[-4]):
            unique[..., :m, :, :] += (w[..., [c + 1], :m, None, None] *
                                      ep[..., [c], :m, :, :])
    return _back(unique, was)


def constrain_variable_probe(variable_probe, weights):
    """Normalise, orthogonalise and sort the eigen probes; clip weight
    outliers at 1.5x the 95th percentile (probe.py:306-359)."""
    vp, was = _t(variable_probe)
    w, _ = _t(weights)
    vnorm = linalg.mnorm(vp, axis=(-2, -1), keepdims=True)
    vp = vp / vnorm
    pwm = vp.shape[-3]
    w[..., 1:, :pwm] *= vnorm[..., 0, 0]
    vp = linalg.orthogonalize_gs(vp, axis=(-2, -1), N=-4)
    power = linalg.norm(w[..., 1:, :pwm], keepdims=True, axis=-3)**2
    for i in range(pwm):
        order = torch.argsort(-power[..., i].flatten())
        w[..., 1:, i] = w[..., 1 + order, i]
        vp[..., :, i, :, :] = vp[..., order, i, :, :]
    aevol = w.abs()
    # 95th percentile over positions with linear interpolation (what
    # cp.percentile / np.percentile compute); a sort, not torch.quantile,
    # which is two orders of magnitude slower on ROCm for this shape.
    srt = torch.sort(aevol, dim=-3).values
    npos = aevol.shape[-3]
    pos = 0.95 * (npos - 1)
    lo, hi = int(np.floor(pos)), int(np.ceil(pos))
    frac = float(pos - lo)
    limit = 1.5 * (srt[..., lo:lo + 1, :, :] * (1.0 - frac) +
                   srt[..., hi:hi + 1, :, :] * frac)
    w = torch.minimum(aevol, limit) * torch.sign(w)
    return _back(vp, was), _back(w, was)


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
    """Flat-top radial probe amplitude (probe.py:784-814)."""
    r, c = np.mgrid[:size, :size] + 0.5
    rs = np.sqrt((r - size / 2)**2 + (c - size / 2)**2)
    rmax = np.sqrt(2) * 0.5 * rout * rs.max() + 1.0
    rmin = np.sqrt(2) * 0.5 * rin * rs.max()
    img = np.zeros((size, size), dtype=precision.floating)
    img[rs < rmin] = 1.0
    img[rs > rmax] = 0.0
    zone = np.logical_and(rs > rmin, rs < rmax)
    img[zone] = np.divide(rmax - rs[zone], rmax - rmin)
    return img


def adjust_probe_power(probe, power=None):
    """Rescale modes to relative power 1/N by default (probe.py:479-497)."""
    if power is None:
        power = 1.0 / np.arange(1, probe.shape[-3] + 1)
    power = power[..., None, None]
    norm = np.sqrt(np.sum(np.abs(probe)**2, axis=(-2, -1), keepdims=True))
    probe *= power * norm[..., 0:1, :, :] / norm
    return probe


def add_modes_random_phase(probe, nmodes):
    """Extra modes = first mode times random linear phase ramps
    (probe.py:500-531)."""
    all_modes = np.empty((*probe.shape[:-3], nmodes, *probe.shape[-2:]),
                         dtype=probe.dtype)
    pw = probe.shape[-1]
    for m in range(nmodes):
        if m < probe.shape[-3]:
            all_modes[..., m, :, :] = probe[..., m, :, :]
        else:
            shift = np.exp(-2j * np.pi * (np.random.rand(2, 1) - 0.5) *
                           ((np.arange(0, pw) + 0.5) / pw - 0.5))
            all_modes[..., m, :, :] = (probe[..., 0, :, :] * shift[0][None] *
                                       shift[1][:, None])
    return all_modes


def init_varying_probe(scan, shared_probe, num_eigen_probes,
                       probes_with_modes=1):
    """Initial eigen probes and weights (probe.py:660-723)."""
    probes_with_modes = max(probes_with_modes, 0)
    if probes_with_modes > shared_probe.shape[-3]:
        raise ValueError(
            f"probes_with_modes ({probes_with_modes}) cannot be more than "
            f"the number of probes ({shared_probe.shape[-3]})!")
    if num_eigen_probes < 1:
        return None, None
    weights = 1e-6 * np.random.rand(
        *scan.shape[:-1], num_eigen_probes,
        shared_probe.shape[-3]).astype(precision.floating)
    weights -= np.mean(weights, axis=-3, keepdims=True)
    weights[..., 0, :] = 1.0
    weights[..., 1:, probes_with_modes:] = 0
    if num_eigen_probes == 1:
        return None, weights
    eigen_probe = trandom.numpy_complex(*shared_probe.shape[:-4],
                                        num_eigen_probes - 1,
                                        probes_with_modes,
                                        *shared_probe.shape[-2:])
    eigen_probe /= linalg.mnorm(eigen_probe, axis=(-2, -1), keepdims=True)
    return eigen_probe, weights


def finite_probe_support(probe, *, radius=0.5, degree=5.0, p=1.0):
    """Supergaussian penalty mask (probe.py:919-964)."""
    if p <= 0:
        return 0.0
    N = probe.shape[-1]
    centers = torch.linspace(-0.5, 0.5, N + 1, device=probe.device)[:-1] + 0.5 / N
    i, j = torch.meshgrid(centers, centers, indexing="xy")
    mask = 1 - torch.exp(-(torch.square(i / radius) +
                           torch.square(j / radius))**degree)
    return (p * mask).to(torch.float32)


def rescale_probe_using_fixed_intensity_photons(probe, Nphotons,
                                                probe_power_fraction=None):
    """Rescale shared modes so their total intensity is Nphotons
    (probe.py:967-993)."""
    photons = torch.sum(probe.abs()**2, (-1, -2))
    if probe_power_fraction is None:
        probe_power_fraction = photons / torch.sum(photons)
    return probe * torch.sqrt(probe_power_fraction * Nphotons /
                              photons)[..., None, None]


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
    """Random sinusoidal eigen-probe weights along the scan: amplitude 1,
    random phase, period at most one scan (probe.py:647-657)."""
    N = scan.shape[1]
    x = np.arange(N)[..., :, None, None]
    period = N * np.random.rand(*eigen_probe.shape[:-2])
    phase = 2 * np.pi * np.random.rand(*eigen_probe.shape[:-2])
    return np.sin(2 * np.pi / period * x - phase)
