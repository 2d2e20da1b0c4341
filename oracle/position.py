"""CPU restatement of the reference's position correction
(src/tike/ptycho/position.py, src/tike/ptycho/solvers/lstsq.py:545-579,
764-806, src/tike/opt.py:165-214).

TEST INFRASTRUCTURE ONLY -- never imported by the product (see
oracle/__init__.py).  Pinned by tests/test_oracle_golden.py against
reconstructions run by the reference itself
(tests/golden/lstsq_recon_positions_*.npz).
"""
import numpy as np

f32 = np.float32


def gaussian_derivative_taps(sigma=0.333, truncate=6.0):
    """Taps t[d], d = -r..r, with gaussian_gradient(x)[i] = sum_d t[d] x[i+d]
    (position.py:779-810: scipy's gaussian_filter1d(-x, order=1): a normalised
    Gaussian times -d/sigma^2, kernel reversed before the correlation)."""
    r = int(truncate * float(sigma) + 0.5)
    d = np.arange(-r, r + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * d**2)
    phi = phi / phi.sum()
    return -(d / (sigma * sigma)) * phi  # sign: the filter is applied to -x


def gaussian_gradient(x, sigma=0.333):
    """First-order Gaussian derivatives along axes -2 and -1, edge mode
    'nearest' (position.py:779-810)."""
    taps = gaussian_derivative_taps(sigma)
    r = len(taps) // 2
    out = []
    for axis in (-2, -1):
        n = x.shape[axis]
        idx = np.clip(np.arange(n)[:, None] + np.arange(-r, r + 1)[None, :], 0,
                      n - 1)  # (n, taps)
        g = np.take(x, idx, axis=axis)  # axis -> (n, taps)
        if axis == -2:
            g = np.moveaxis(g, -2, -1)  # (..., n, W, taps) -> taps last
        out.append((g * taps.astype(x.real.dtype)).sum(axis=-1).astype(
            x.dtype))
    return out[0], out[1]


def position_update_terms(patches, unique_probe, chi, m=0):
    """Numerator / denominator of the per-position shift estimate
    (lstsq.py:545-579): least squares of chi_m on the two Gaussian-derivative
    exit waves over the central half of the probe window."""
    grad_x, grad_y = gaussian_gradient(patches, sigma=0.333)
    crop = unique_probe.shape[-1] // 4
    c = slice(crop, -crop)
    P = unique_probe[..., m:m + 1, c, c]
    X = chi[..., m:m + 1, c, c]
    num = np.empty((patches.shape[0], 2), dtype=f32)
    den = np.empty((patches.shape[0], 2), dtype=f32)
    for k, g in enumerate((grad_x, grad_y)):
        gp = g[..., c, c] * P
        num[:, k] = np.sum(np.real(np.conj(gp) * X), axis=(-4, -3, -2, -1))
        den[:, k] = np.sum(np.abs(gp)**2, axis=(-4, -3, -2, -1))
    return num, den


def adam(g, v=None, m=None, vdecay=0.999, mdecay=0.9, eps=1e-8):
    """opt.py:165-214."""
    v = np.zeros_like(g.real) if v is None else v
    m = np.zeros_like(g) if m is None else m
    m = mdecay * m + (1 - mdecay) * g
    v = vdecay * v + (1 - vdecay) * (g * g.conj()).real
    m_ = m / (1 - mdecay)
    v_ = np.sqrt(v / (1 - vdecay))
    return m_ / (v_ + eps), v, m


def trim_mean(a, proportiontocut):
    """scipy.stats.trim_mean along axis 0."""
    n = a.shape[0]
    lo = int(proportiontocut * n)
    hi = n - lo
    s = np.sort(a, axis=0)
    return s[lo:hi].mean(axis=0)


def update_position(scan, pos, numerator, denominator, *, alpha=0.05, epoch=0):
    """lstsq.py:764-806."""
    if epoch < pos.get("update_start", 0):
        return scan
    step = numerator / ((1 - alpha) * denominator +
                        alpha * max(denominator.max(), 1e-6))
    limit = pos.get("update_magnitude_limit", 0)
    if limit > 0:
        step = np.clip(step, -limit, limit)
    step = step - trim_mean(step, 0.05)
    if pos.get("use_adaptive_moment", False):
        mom = pos["momentum"]
        step, mom[..., 0:2], mom[..., 2:4] = adam(
            step, mom[..., 0:2], mom[..., 2:4],
            vdecay=pos.get("vdecay", 0.999), mdecay=pos.get("mdecay", 0.9))
    return (scan - step).astype(f32)


# ---- affine regularisation (position.py:138-330, 716-776) ------------------
def transform_matrix(t):
    """AffineTransform.asarray: scale @ shear @ rotate (position.py:195-220)."""
    scale0, scale1, shear1, angle = t[0], t[1], t[2], t[3]
    c, s = np.cos(angle), np.sin(angle)
    return (np.array([[scale0, 0.0], [0.0, scale1]], dtype=f32)
            @ np.array([[1.0, 0.0], [shear1, 1.0]], dtype=f32)
            @ np.array([[c, -s], [s, c]], dtype=f32))


def transform_apply(t, x, shift=True):
    r = x @ transform_matrix(t)
    if shift:
        r = r + np.array((t[4], t[5]))
    return r


IDENTITY = (1.0, 1.0, 0.0, 0.0, 0.0, 0.0)


def transform_fromarray(T):
    """Graphics-Gems decomposition (position.py:166-193)."""
    R = T[:2, :2].copy()
    scale0 = np.linalg.norm(R[0])
    if scale0 <= 0:
        return IDENTITY
    R[0] /= scale0
    shear1 = R[0] @ R[1]
    R[1] -= shear1 * R[0]
    scale1 = np.linalg.norm(R[1])
    if scale1 <= 0:
        return IDENTITY
    R[1] /= scale1
    shear1 /= scale1
    angle = np.arccos(R[0, 0])
    return (float(scale0), float(scale1), float(shear1), float(angle),
            float(T[2, 0]) if T.shape[0] > 2 else 0.0,
            float(T[2, 1]) if T.shape[0] > 2 else 0.0)


def estimate_global_transformation(p0, p1):
    """Least squares [p0 1] @ T = p1 (position.py:252-270, linalg.py:33-58)."""
    a = np.pad(p0, ((0, 0), (0, 1)), constant_values=1)
    try:
        aT = a.conj().swapaxes(-1, -2)
        t = transform_fromarray(np.linalg.inv(aT @ a) @ aT @ p1)
    except np.linalg.LinAlgError:
        t = IDENTITY
    return t, np.linalg.norm(transform_apply(t, p0) - p1)


def estimate_global_transformation_ransac(p0, p1, transform, rng, *,
                                          min_sample=4, max_error=32,
                                          min_consensus=0.75, max_iter=20):
    """position.py:273-327."""
    best = np.inf
    for subset in rng.choice(a=len(p0), size=(max_iter, min_sample),
                             replace=True):
        cand, _ = estimate_global_transformation(p0[subset], p1[subset])
        err = np.linalg.norm(transform_apply(cand, p0) - p1, axis=-1)
        inl = err <= max_error
        if np.sum(inl) / len(inl) >= min_consensus:
            cand, fit = estimate_global_transformation(p0[inl], p1[inl])
            if fit < best:
                best, transform = fit, cand
    return transform, best


def affine_position_regularization(scan, pos, rng, max_error=32, relax=0.9):
    """position.py:716-776."""
    origin = pos.get("origin", np.zeros(2))
    pos["transform"], _ = estimate_global_transformation_ransac(
        pos["initial_scan"] - origin, scan - origin,
        pos.get("transform", IDENTITY), rng, max_error=max_error)
    if pos.get("use_position_regularization", False):
        predicted = transform_apply(pos["transform"], pos["initial_scan"],
                                    shift=False)
        scan = scan * (1 - relax) + relax * predicted
    return scan
