"""NumPy restatement of the reference's ptychography operators (CPU oracle).

TEST INFRASTRUCTURE ONLY: this module is the checker for the HIP path and the
``cpu_baseline`` of ``bench.py``.  The product package never imports it.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).  Parity of this restatement is pinned by
``tests/test_oracle_golden.py`` against
  * the reference's own golden vectors (``tests/golden/ptycho_setup.npz`` is a
    re-encoding of ``tests/data/ptycho_setup.pickle.lzma``; the two patch
    known-answer tests of ``tests/operators/test_patch.py:64-206``), and
  * fixtures produced by running the reference's own Python in the build
    container (``tests/golden/gen/make_fixtures.py``).

All arithmetic is float32 / complex64 as in ``src/tike/precision.py:4-11``.
"""
from __future__ import annotations

import os

import numpy as np
import scipy.fft

_WORKERS = int(os.environ.get("TIKE_ORACLE_WORKERS", "1"))


def set_workers(n: int) -> None:
    """Number of scipy.fft worker threads (used by the CPU baseline)."""
    global _WORKERS
    _WORKERS = max(1, int(n))


def get_workers() -> int:
    return _WORKERS


# --------------------------------------------------------------------------
# Patch  (src/tike/operators/cupy/patch.py:79-188, convolution.cu:79-165)
# --------------------------------------------------------------------------


def _patch_geometry(positions, H, W, pw):
    """Index/weight arrays shared by the forward gather and adjoint scatter.

    convolution.cu:101-134: sy,sx = floor(scan); weights
    ((1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx*fy); pixels whose leading tap is
    outside the image are skipped (:111,118); the three trailing taps are
    addressed *linearly* (ii+1, ii+W, ii+1+W; :38-47).
    """
    pos = np.asarray(positions, dtype=np.float32)
    sy = np.floor(pos[..., 0])
    sx = np.floor(pos[..., 1])
    fy = (pos[..., 0] - sy).astype(np.float32)
    fx = (pos[..., 1] - sx).astype(np.float32)
    one = np.float32(1.0)
    w = np.stack(
        [(one - fx) * (one - fy), fx * (one - fy), (one - fx) * fy, fx * fy],
        axis=-1,
    ).astype(np.float32)  # (..., N, 4)
    py = np.arange(pw)
    px = np.arange(pw)
    yy = sy.astype(np.int64)[..., None, None] + py[:, None]  # (..., N, pw, 1)
    xx = sx.astype(np.int64)[..., None, None] + px[None, :]  # (..., N, 1, pw)
    valid = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
    return yy, xx, valid, w


def patch_fwd(images, positions, patches=None, patch_width=0, nrepeat=1):
    """Bilinear gather of patches; patch.py:79-130 + convolution.cu:146-155.

    images (..., H, W) c64; positions (..., N, 2) f32 (y, x);
    patches (..., N*nrepeat, padded, padded) -- only the centred
    patch_width x patch_width window is written, padding is left untouched.
    """
    images = np.asarray(images)
    positions = np.asarray(positions, dtype=np.float32)
    pw = patch_width if patch_width else patches.shape[-1]
    lead = positions.shape[:-2]
    N = positions.shape[-2]
    if patches is None:
        patches = np.zeros((*lead, N * nrepeat, pw, pw), dtype=images.dtype)
    padded = patches.shape[-1]
    assert pw <= padded
    assert images.shape[:-2] == lead
    assert patches.shape[:-3] == lead
    assert patches.shape[-3] == N * nrepeat
    H, W = images.shape[-2:]
    pad = (padded - pw) // 2
    nimage = int(np.prod(lead)) if lead else 1
    img = images.reshape(nimage, H * W)
    flat_all = images.reshape(-1)
    pos = positions.reshape(nimage, N, 2)
    out = patches.reshape(nimage, N, nrepeat, padded, padded)
    total = flat_all.size
    for ti in range(nimage):
        yy, xx, valid, w = _patch_geometry(pos[ti], H, W, pw)
        ii = (yy + H * ti) * W + xx  # (N, pw, pw) linear index into all images
        val = np.zeros((N, pw, pw), dtype=images.dtype)
        for k, off in enumerate((0, 1, W, W + 1)):
            idx = ii + off
            wk = w[:, k][:, None, None]
            ok = valid & (idx < total) & (wk != 0)
            tap = np.where(ok, flat_all[np.clip(idx, 0, total - 1)], 0)
            val = val + (tap * wk).astype(images.dtype)
        cur = out[ti, :, :, pad:pad + pw, pad:pad + pw]
        out[ti, :, :, pad:pad + pw, pad:pad + pw] = np.where(
            valid[:, None], val[:, None], cur)
    return patches


def patch_adj(positions, patches, images=None, patch_width=0, height=0,
              width=0, nrepeat=1):
    """Scatter-add of patches; patch.py:132-188 + convolution.cu:51-66,157-165.

    patches (..., K, padded, padded) with (N*nrepeat) % K == 0 and K >= nrepeat
    (patch.py:155); patch index used for position ts, repeat r is
    ``r + (nrepeat*ts) % K`` (convolution.cu:138-139).
    """
    positions = np.asarray(positions, dtype=np.float32)
    patches = np.asarray(patches)
    pw = patches.shape[-1] if patch_width == 0 else patch_width
    padded = patches.shape[-1]
    lead = positions.shape[:-2]
    N = positions.shape[-2]
    if images is None:
        images = np.zeros((*lead, height, width), dtype=patches.dtype)
    H, W = images.shape[-2:]
    K = patches.shape[-3]
    assert (N * nrepeat) % K == 0 and K >= nrepeat
    pad = (padded - pw) // 2
    nimage = int(np.prod(lead)) if lead else 1
    flat = images.reshape(-1)
    total = flat.size
    pos = positions.reshape(nimage, N, 2)
    pat = patches.reshape(nimage, K, padded, padded)
    for ti in range(nimage):
        yy, xx, valid, w = _patch_geometry(pos[ti], H, W, pw)
        ii = (yy + H * ti) * W + xx
        for r in range(nrepeat):
            pidx = r + (nrepeat * np.arange(N)) % K
            src = pat[ti, pidx, pad:pad + pw, pad:pad + pw]  # (N, pw, pw)
            for k, off in enumerate((0, 1, W, W + 1)):
                idx = ii + off
                wk = w[:, k][:, None, None]
                ok = valid & (idx < total) & (wk != 0)
                contrib = (src * wk).astype(patches.dtype)
                np.add.at(flat, idx[ok], contrib[ok])
    return images


# --------------------------------------------------------------------------
# Convolution  (src/tike/operators/cupy/convolution.py:58-154)
# --------------------------------------------------------------------------


def convolution_fwd(psi, scan, probe, detector_shape=None):
    """patch(psi) zero-padded to the detector, times probe (:58-101).

    psi (..., H, W); scan (..., N, 2); probe (..., 1|N, S, pw, pw)
    -> nearplane (..., N, S, det, det).
    """
    pw = probe.shape[-1]
    det = pw if detector_shape is None else detector_shape
    pad = (det - pw) // 2
    S = probe.shape[-3]
    N = scan.shape[-2]
    patches = np.zeros((*scan.shape[:-2], N * S, det, det), dtype=psi.dtype)
    patches = patch_fwd(psi, scan, patches, patch_width=pw, nrepeat=S)
    patches = patches.reshape(*scan.shape[:-1], S, det, det)
    patches[..., pad:pad + pw, pad:pad + pw] *= probe
    return patches


def convolution_adj(nearplane, scan, probe, nz, n, psi=None):
    """conj(probe) * crop(nearplane) scattered into psi (:103-127)."""
    pw = probe.shape[-1]
    det = nearplane.shape[-1]
    pad = (det - pw) // 2
    nearplane = nearplane.copy()
    nearplane[..., pad:pad + pw, pad:pad + pw] *= probe.conj()
    if psi is None:
        psi = np.zeros((*scan.shape[:-2], nz, n), dtype=nearplane.dtype)
    S = nearplane.shape[-3]
    return patch_adj(
        positions=scan,
        patches=nearplane.reshape(*scan.shape[:-2], scan.shape[-2] * S, det,
                                  det),
        images=psi,
        patch_width=pw,
        nrepeat=S,
    )


def convolution_adj_probe(nearplane, scan, psi, probe_shape):
    """conj(patch(psi)) * crop(nearplane), no sum over positions (:129-154)."""
    pw = probe_shape
    det = nearplane.shape[-1]
    pad = (det - pw) // 2
    S = nearplane.shape[-3]
    N = scan.shape[-2]
    patches = np.zeros((*scan.shape[:-2], N * S, pw, pw), dtype=psi.dtype)
    patches = patch_fwd(psi, scan, patches, patch_width=pw, nrepeat=S)
    patches = patches.reshape(*scan.shape[:-1], S, pw, pw).conj()
    patches = patches * nearplane[..., pad:pad + pw, pad:pad + pw]
    return patches


# --------------------------------------------------------------------------
# Propagation  (src/tike/operators/cupy/propagation.py:43-73, cache.py:66-82)
# --------------------------------------------------------------------------


def propagation_fwd(nearplane, norm="ortho"):
    """Batched C2C FFT2 over the last two axes (propagation.py:43-57)."""
    return scipy.fft.fft2(
        nearplane, axes=(-2, -1), norm=norm,
        workers=_WORKERS).astype(np.complex64, copy=False)


def propagation_adj(farplane, norm="ortho"):
    """Batched C2C IFFT2 over the last two axes (propagation.py:59-73)."""
    return scipy.fft.ifft2(
        farplane, axes=(-2, -1), norm=norm,
        workers=_WORKERS).astype(np.complex64, copy=False)


# --------------------------------------------------------------------------
# FresnelSpectProp  (src/tike/operators/cupy/fresnelspectprop.py:52-137) and
# Multislice with several slices (multislice.py:69-194)
# --------------------------------------------------------------------------


def fresnel_spectrum_propagator(N, probe_FOV, distance, wavelength):
    """fresnelspectprop.py:115-137 (float64 grids, FFT-shifted, complex64)."""
    xgrid = (0.5 + np.linspace(-0.5 * N[1], 0.5 * N[1] - 1, num=N[1])) / N[1]
    ygrid = (0.5 + np.linspace(-0.5 * N[0], 0.5 * N[0] - 1, num=N[0])) / N[0]
    kx = 2 * np.pi * N[1] * xgrid / probe_FOV[1]
    ky = 2 * np.pi * N[0] * ygrid / probe_FOV[0]
    Kx, Ky = np.meshgrid(kx, ky, indexing="xy")
    prop = np.exp(1j * distance * np.sqrt((2 * np.pi / wavelength)**2 -
                                          Kx**2 - Ky**2))
    return np.fft.fftshift(prop).astype(np.complex64)


def fresnel_fwd(nearplane, propagator, norm="ortho"):
    """IFFT2(FFT2(x) * H)  (fresnelspectprop.py:52-82)."""
    return propagation_adj(propagation_fwd(nearplane, norm) * propagator, norm)


def fresnel_adj(farplane, propagator, norm="ortho"):
    """IFFT2(FFT2(x) * conj(H))  (fresnelspectprop.py:84-113)."""
    return propagation_adj(
        propagation_fwd(farplane, norm) * np.conj(propagator), norm)


def multislice_fwd(probe, scan, psi, propagator, norm="ortho"):
    """probe (N|1,S,pw,pw), psi (D,H,W) -> exit wave (N,S,pw,pw)
    (multislice.py:69-92; detector = probe shape)."""
    exitwave = convolution_fwd(psi[0], scan, probe)
    for s in range(1, len(psi)):
        exitwave = convolution_fwd(psi[s], scan,
                                   fresnel_fwd(exitwave, propagator, norm))
    return exitwave


def multislice_fwd_intermediate(probe, scan, psi, propagator, norm="ortho"):
    """(exit wave, probes incident on every slice (D,N,S,pw,pw))
    (multislice.py:97-141)."""
    N = scan.shape[-2]
    probes = np.zeros((psi.shape[0], N, *probe.shape[-3:]), dtype=probe.dtype)
    probes[0] = probe
    exitwave = None
    for t in range(len(psi)):
        exitwave = convolution_fwd(psi[t], scan, probes[t])
        if t == len(psi) - 1:
            break
        probes[t + 1] = fresnel_fwd(exitwave, propagator, norm)
    return exitwave, probes


def multislice_adj(nearplane, probe, scan, psi, propagator, norm="ortho"):
    """(psi_adj (D,H,W) / D, probe_adj (N,S,pw,pw))  (multislice.py:144-194)."""
    D = len(psi)
    probes = [None] * D
    probes[0] = probe
    for s in range(1, D):
        probes[s] = fresnel_fwd(convolution_fwd(psi[s - 1], scan, probes[s - 1]),
                                propagator, norm)
    psi_adj = np.zeros_like(psi)
    psi_adj[D - 1] = convolution_adj(nearplane, scan, probes[D - 1],
                                     psi.shape[-2], psi.shape[-1])
    probe_adj = convolution_adj_probe(nearplane, scan, psi[D - 1],
                                      probe.shape[-1])
    for s in range(D - 2, -1, -1):
        probe_adj = fresnel_adj(probe_adj, propagator, norm)
        psi_adj[s] = convolution_adj(probe_adj, scan, probes[s], psi.shape[-2],
                                     psi.shape[-1])
        probe_adj = convolution_adj_probe(probe_adj, scan, psi[s],
                                          probe.shape[-1])
    return psi_adj / D, probe_adj


# --------------------------------------------------------------------------
# Ptycho  (src/tike/operators/cupy/ptycho.py:114-204)
# --------------------------------------------------------------------------


def ptycho_fwd(probe, scan, psi, detector_shape, norm="ortho",
               propagator=None):
    """probe (N|1,1,S,pw,pw), scan (N,2), psi (D,H,W) -> (N,1,S,det,det).

    ptycho.py:114-129 -> multislice.py:69-92 -> convolution.py:58.  D > 1
    needs the Fresnel `propagator` (and det == pw).
    """
    assert psi.ndim == 3
    if psi.shape[0] == 1:
        near = convolution_fwd(psi[0], scan, probe[..., 0, :, :, :],
                               detector_shape)
    else:
        near = multislice_fwd(probe[..., 0, :, :, :], scan, psi, propagator,
                              norm)
    return propagation_fwd(near, norm)[..., None, :, :, :]


def ptycho_fwd_intermediate(probe, scan, psi, propagator, norm="ortho"):
    """ptycho.py:131-146: (farplane (N,1,S,pw,pw), probes (D,N,S,pw,pw))."""
    if psi.shape[0] == 1:
        p = probe[..., 0, :, :, :]
        N = scan.shape[-2]
        probes = np.broadcast_to(p, (N, *p.shape[-3:]))[None].copy()
        near = convolution_fwd(psi[0], scan, p)
    else:
        near, probes = multislice_fwd_intermediate(probe[..., 0, :, :, :], scan,
                                                   psi, propagator, norm)
    return propagation_fwd(near, norm)[..., None, :, :, :], probes


def ptycho_adj(farplane, probe, scan, psi, norm="ortho", propagator=None):
    """Returns (psi_adj (D,H,W), probe_adj (N,1,S,pw,pw)); ptycho.py:156-176,
    multislice.py:144-194 (psi_adj divided by the number of slices)."""
    assert psi.ndim == 3
    near = propagation_adj(farplane, norm)[..., 0, :, :, :]
    p = probe[..., 0, :, :, :]
    if psi.shape[0] > 1:
        psi_adj, probe_adj = multislice_adj(near, p, scan, psi, propagator,
                                            norm)
        return psi_adj, probe_adj[..., None, :, :, :]
    psi_adj = convolution_adj(near, scan, p, psi.shape[-2], psi.shape[-1])
    probe_adj = convolution_adj_probe(near, scan, psi[0], p.shape[-1])
    return psi_adj[None], probe_adj[..., None, :, :, :]


def intensity_from_farplane(farplane):
    """sum over axes 1..ndim-3 of |farplane|^2 (ptycho.py:18-23)."""
    return np.sum(
        (farplane * farplane.conj()).real,
        axis=tuple(range(1, farplane.ndim - 2)),
        dtype=np.float32,
    )


def ptycho_cost(data, psi, scan, probe, detector_shape, model="gaussian",
                norm="ortho"):
    """ptycho.py:193-204."""
    far = ptycho_fwd(probe, scan, psi, detector_shape, norm)
    return globals()[model](data, intensity_from_farplane(far))


# --------------------------------------------------------------------------
# Objective functions  (src/tike/operators/cupy/objective.py:11-124)
# --------------------------------------------------------------------------


def _gaussian_fuse(data, intensity):
    diff = np.sqrt(intensity) - np.sqrt(data)
    return diff * np.conj(diff)


def gaussian(data, intensity):
    return np.mean(_gaussian_fuse(data, intensity))


def gaussian_grad(data, farplane, intensity):
    return farplane * (1 - np.sqrt(data) / (np.sqrt(intensity) + 1e-9))[
        ..., np.newaxis, np.newaxis, :, :]


def gaussian_each_pattern(data, intensity):
    return np.mean(_gaussian_fuse(data, intensity), axis=(-2, -1))


def _poisson_fuse(data, intensity):
    return intensity - data * np.log(intensity + 1e-9)


def poisson(data, intensity):
    return np.mean(_poisson_fuse(data, intensity))


def poisson_grad(data, farplane, intensity):
    return farplane * (1 - data / (intensity + 1e-9))[..., np.newaxis,
                                                      np.newaxis, :, :]


def poisson_each_pattern(data, intensity):
    return np.mean(_poisson_fuse(data, intensity), axis=(-2, -1))


# --------------------------------------------------------------------------
# simulate  (src/tike/ptycho/ptycho.py:95-179)
# --------------------------------------------------------------------------


def simulate(detector_shape, probe, scan, psi, fly=1, eigen_probe=None,
             eigen_weights=None, norm="ortho"):
    """Per-mode forward, |.|^2 summed over modes and fly groups (:95-125)."""
    from .solvers import get_varying_probe
    scan = np.asarray(scan, dtype=np.float32)
    psi = np.asarray(psi, dtype=np.complex64)
    probe = np.asarray(probe, dtype=np.complex64)
    intensity = 0
    for m in range(probe.shape[-3]):
        far = ptycho_fwd(
            get_varying_probe(
                probe[..., [m], :, :],
                None if eigen_probe is None else eigen_probe[..., [m], :, :],
                None if eigen_weights is None else eigen_weights[..., [m]],
            ), scan, psi, detector_shape, norm)
        intensity = intensity + np.sum(
            np.square(np.abs(far)).reshape(scan.shape[-2] // fly, fly,
                                           detector_shape, detector_shape),
            axis=-3,
        )
    return intensity.real.astype(np.float32)
