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
        return self._copy(lambda x: None if x is None else A.to_host(x))


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
