"""NumPy restatement of the reference's lstsq_grad / cgrad update loop.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  State is a plain
``dict`` of NumPy arrays so that this module depends on nothing in the
product package.  Paths in docstrings are relative to ``/root/reference``.

State keys
----------
psi (1,H,W) c64, probe (1,1,S,pw,pw) c64, scan (N,2) f32,
eigen_probe (1,C,S',pw,pw) c64 | None, eigen_weights (N,C+1,S) f32 | None,
psi_precond (1,H,W) c64, probe_precond (1,pw,pw) c64,
costs: list[list[float]], object_v/object_m/probe_v/probe_m: momentum buffers.
"""
from __future__ import annotations

import numpy as np

from . import operators as ops
from . import position as opos

# --------------------------------------------------------------------------
# tike.linalg (src/tike/linalg.py:12-32)
# --------------------------------------------------------------------------


def mnorm(x, axis=None, keepdims=False):
    return np.sqrt(np.mean((x * x.conj()).real, axis=axis, keepdims=keepdims))


def norm(x, axis=None, keepdims=False):
    return np.sqrt(np.sum((x * x.conj()).real, axis=axis, keepdims=keepdims))


def inner(x, y, axis=None, keepdims=False):
    return (x * y.conj()).sum(axis=axis, keepdims=keepdims)


def projection(a, b, axis=None):
    bh = b / inner(b, b, axis=axis, keepdims=True)
    return inner(a, b, axis=axis, keepdims=True) * bh


def orthogonalize_gs(x, axis=-1, N=None):
    """src/tike/linalg.py:77-116."""
    try:
        axis = tuple(a % x.ndim for a in axis)
    except TypeError:
        axis = (axis % x.ndim,)
    if N is None:
        N = x.ndim - 1
        while N in axis:
            N -= 1
    N = N % x.ndim
    if N in axis:
        raise ValueError("Cannot orthogonalize a single vector.")
    x = np.moveaxis(x, N, 0)
    u = x.copy()
    for i in range(1, len(x)):
        u[i:] -= projection(x[i:], u[i - 1:i], axis=axis)
    return np.moveaxis(u, 0, N)


# --------------------------------------------------------------------------
# tike.ptycho.probe (src/tike/ptycho/probe.py)
# --------------------------------------------------------------------------


def get_varying_probe(shared_probe, eigen_probe=None, weights=None):
    """probe.py:272-303."""
    if weights is not None:
        unique_probe = weights[..., [0], :, None, None] * shared_probe
        if eigen_probe is not None:
            m = eigen_probe.shape[-3]
            for c in range(eigen_probe.shape[-4]):
                unique_probe[..., :m, :, :] += (
                    weights[..., [c + 1], :m, None, None] *
                    eigen_probe[..., [c], :m, :, :])
        return unique_probe.astype(np.complex64, copy=False)
    return shared_probe.copy()


def constrain_variable_probe(variable_probe, weights):
    """probe.py:306-359 (CuPy percentile drops the q axis)."""
    vnorm = mnorm(variable_probe, axis=(-2, -1), keepdims=True)
    variable_probe = variable_probe / vnorm
    probes_with_modes = variable_probe.shape[-3]
    weights[..., 1:, :probes_with_modes] *= vnorm[..., 0, 0]
    variable_probe = orthogonalize_gs(variable_probe, axis=(-2, -1), N=-4)
    power = norm(weights[..., 1:, :probes_with_modes], keepdims=True,
                 axis=-3)**2
    for i in range(probes_with_modes):
        order = np.argsort(-power[..., i].flatten())
        weights[..., 1:, i] = weights[..., 1 + order, i]
        variable_probe[..., :, i, :, :] = variable_probe[..., order, i, :, :]
    aevol = np.abs(weights)
    limit = 1.5 * np.percentile(aevol, 95, axis=-3,
                                keepdims=True).astype(weights.dtype)
    weights = np.minimum(aevol, limit) * np.sign(weights)
    return variable_probe, weights


def orthogonalize_eig(x):
    """probe.py:726-769."""
    nmodes = x.shape[-3]
    A = np.empty((*x.shape[:-3], nmodes, nmodes), dtype=x.dtype)
    for i in range(nmodes):
        for j in range(i, nmodes):
            A[..., i, j] = np.sum(x[..., i, :, :].conj() * x[..., j, :, :],
                                  axis=(-1, -2))
    val, vectors = np.linalg.eigh(A, UPLO='U')
    result = (vectors.swapaxes(-1, -2) @ x.reshape(*x.shape[:-2], -1)).reshape(
        *x.shape)
    power = np.square(norm(result, axis=(-2, -1), keepdims=False)).flatten()
    order = np.argsort(power, axis=None, kind='stable')[::-1]
    result = result[..., order, :, :]
    power = power[order]
    return result, power


def probe_power(probe):
    """probe.py:772-781."""
    return np.square(norm(probe, axis=(-2, -1), keepdims=False)).flatten()


def gaussian_probe(size, rin=0.8, rout=1.0):
    """probe.py:784-814."""
    r, c = np.mgrid[:size, :size] + 0.5
    rs = np.sqrt((r - size / 2)**2 + (c - size / 2)**2)
    rmax = np.sqrt(2) * 0.5 * rout * rs.max() + 1.0
    rmin = np.sqrt(2) * 0.5 * rin * rs.max()
    img = np.zeros((size, size), dtype=np.float32)
    img[rs < rmin] = 1.0
    img[rs > rmax] = 0.0
    zone = np.logical_and(rs > rmin, rs < rmax)
    img[zone] = np.divide(rmax - rs[zone], rmax - rmin)
    return img


def update_eigen_probe(R, eigen_probe, weights, patches, diff, lo, hi, *,
                       beta=0.1, c=1, m=0):
    """probe.py:362-476."""
    norm_weights = norm(weights[lo:hi, c:c + 1, m:m + 1, None, None], axis=-5,
                        keepdims=True)**2
    if np.all(norm_weights == 0):
        raise ValueError("eigen_probe weights cannot all be zero?")
    proj = (np.real(R.conj() * eigen_probe[:, c - 1:c, m:m + 1, :, :]) +
            weights[lo:hi, c:c + 1, m:m + 1, None, None]) / norm_weights
    update = np.mean(R * np.mean(proj, axis=(-2, -1), keepdims=True), axis=-5,
                     keepdims=False)
    eigen_probe[:, c - 1:c, m:m + 1, :, :] += (
        beta * update / mnorm(update, axis=(-2, -1), keepdims=True))
    eigen_probe[:, c - 1:c, m:m + 1, :, :] /= mnorm(
        eigen_probe[:, c - 1:c, m:m + 1, :, :], axis=(-2, -1), keepdims=True)
    phi = patches * eigen_probe[:, c - 1:c, m:m + 1, :, :]
    n = np.mean(np.real(diff[:, :, m:m + 1, :, :] * phi.conj()), axis=(-1, -2),
                keepdims=False)
    d = np.mean(np.square(np.abs(phi)), axis=(-1, -2), keepdims=False)
    d_mean = np.mean(d, axis=-3, keepdims=False)
    weight_update = (n / (d + 0.1 * d_mean)).reshape(
        *weights[lo:hi, c:c + 1, m:m + 1].shape)
    weights[lo:hi, c:c + 1, m:m + 1] += weight_update
    return eigen_probe, weights


# --------------------------------------------------------------------------
# tike.ptycho.object (src/tike/ptycho/object.py:324-335)
# --------------------------------------------------------------------------


def remove_object_ambiguity(psi, probe, preconditioner):
    W = preconditioner.real
    W = W / mnorm(W)
    object_norm = 2 * np.sqrt(np.mean(np.square(np.abs(psi)) * W))
    psi = psi / object_norm
    probe = probe * object_norm
    return psi.astype(np.complex64), probe.astype(np.complex64)


# --------------------------------------------------------------------------
# preconditioners (src/tike/ptycho/solvers/_preconditioner.py:48-209)
# --------------------------------------------------------------------------


def psi_preconditioner(psi, probe, scan, propagator=None):
    """Sum_s |probe_s|^2 scattered at every position (K = 1 broadcast); for
    the slices behind the first, the probe propagated through the slices in
    front of it (_preconditioner.py:48-104)."""
    out = np.zeros(psi.shape, dtype=psi.dtype)
    probe_amp = np.sum(probe * probe.conj(), axis=-3)[:, 0]  # (1, pw, pw)
    out[0] = ops.patch_adj(patches=probe_amp, images=out[0], positions=scan)
    probe1 = probe[:, 0]
    for i in range(1, len(psi)):
        probe1 = ops.fresnel_fwd(ops.convolution_fwd(psi[i - 1], scan, probe1),
                                 propagator)
        probe_amp = np.sum(probe1 * probe1.conj(), axis=-3)
        out[i] = ops.patch_adj(patches=probe_amp, images=out[i],
                               positions=scan)
    return out


def probe_preconditioner(psi, probe, scan):
    """Sum_n |patch_n(psi)|^2 -> (D, pw, pw)."""
    pw = probe.shape[-1]
    out = np.zeros((psi.shape[0], pw, pw), dtype=probe.dtype)
    for i in range(len(psi)):
        patches = ops.patch_fwd(psi[i], scan, patch_width=pw)
        out[i] += np.sum(patches * patches.conj(), axis=0)
    return out


# --------------------------------------------------------------------------
# lstsq_grad (src/tike/ptycho/solvers/lstsq.py)
# --------------------------------------------------------------------------


def poisson_steplength_all_modes(xi, abs2_Psi, I_e, I_m, measured_pixels,
                                 step_length, weight_avg):
    """exitwave.py:122-184: two fixed-point sweeps of the per-mode optimal
    step, averaged with the previous value by `weight_avg`."""
    I_e = I_e[:, None, None, ...]
    I_m = I_m[:, None, None, ...]
    xi_abs_Psi2 = xi * abs2_Psi
    denom_final = np.sum((xi * xi_abs_Psi2)[..., measured_pixels], axis=-1)
    w = np.float32(weight_avg)
    for _ in range(2):
        xi_alpha_minus_one = xi * step_length - 1
        denom = abs2_Psi * np.square(xi_alpha_minus_one) + I_e - abs2_Psi
        numer = np.sum((xi_abs_Psi2 * (1 + (I_m * xi_alpha_minus_one) /
                                       denom))[..., measured_pixels], axis=-1)
        step_length = (step_length * (1 - w) +
                       (numer / denom_final)[..., None, None] * w)
    return step_length.astype(np.float32)


def poisson_steplength_dominant_mode(xi, I_e, I_m, measured_pixels,
                                     step_length, weight_avg):
    """exitwave.py:187-234: the same with the modes collapsed."""
    I_e = I_e[:, None, None, ...]
    I_m = I_m[:, None, None, ...]
    sum_denom = np.sum((np.square(xi) * I_e)[..., measured_pixels], axis=-1)
    w = np.float32(weight_avg)
    for _ in range(2):
        numer = xi * (I_e - I_m / (1 - step_length * xi))
        numer_over_denom = np.sum(numer[..., measured_pixels],
                                  axis=-1) / sum_denom
        step_length = ((1 - w) * step_length +
                       w * numer_over_denom[..., None, None])
    return step_length.astype(np.float32)


def precondition_object_update(object_upd_sum, psi_update_denominator,
                               alpha=0.05):
    """lstsq.py:605-616."""
    return (object_upd_sum / np.sqrt(
        np.square((1 - alpha) * psi_update_denominator) +
        np.square(alpha * np.amax(psi_update_denominator, axis=(-2, -1),
                                  keepdims=True)))).astype(np.complex64)


def get_nearplane_gradients(data, psi, scan, probe, eigen_probe,
                            eigen_weights, lo, hi, *, num_batch,
                            detector_shape, measured_pixels,
                            noise_model="gaussian",
                            unmeasured_pixels_scaling=1.0, norm="ortho",
                            recover_psi=True, recover_probe=True,
                            step_length_start=0.5,
                            step_length_usemodes="all_modes",
                            step_length_weight=0.5, recover_positions=False):
    """lstsq.py:367-602 for one minibatch [lo, hi) (chunking of 64 elided)."""
    pw = probe.shape[-1]
    pad = (detector_shape - pw) // 2
    end = pad + pw
    unique_probe = get_varying_probe(
        probe, eigen_probe,
        eigen_weights[lo:hi] if eigen_weights is not None else None)
    if unique_probe.shape[0] == 1 and hi - lo > 1:
        unique_probe = np.broadcast_to(
            unique_probe, (hi - lo, *unique_probe.shape[1:])).copy()
    farplane = ops.ptycho_fwd(unique_probe, scan[lo:hi], psi, detector_shape,
                              norm)
    intensity = np.sum(np.square(np.abs(farplane)),
                       axis=tuple(range(1, farplane.ndim - 2)))
    d = data[lo:hi].astype(np.float32)
    each = getattr(ops, f"{noise_model}_each_pattern")
    costs = each(d[:, measured_pixels][:, None, :],
                 intensity[:, measured_pixels][:, None, :])
    if noise_model == "poisson":
        # lstsq.py:454-489: gradient F * xi with a per-(position, mode) step
        with np.errstate(invalid="ignore", divide="ignore"):
            xi = (1 - d / (intensity + np.float32(1e-9)))[:, None, None, ...]
            grad_cost = farplane * xi
            step = np.full((farplane.shape[0], 1, farplane.shape[2], 1, 1),
                           np.float32(step_length_start), dtype=np.float32)
            if step_length_usemodes == "dominant_mode":
                step = poisson_steplength_dominant_mode(
                    xi, intensity, d, measured_pixels, step,
                    step_length_weight)
            else:
                step = poisson_steplength_all_modes(
                    xi, np.square(np.abs(farplane)), intensity, d,
                    measured_pixels, step, step_length_weight)
            farplane[..., measured_pixels] = (
                -step * grad_cost)[..., measured_pixels]
    elif noise_model == "gaussian":
        grad = ops.gaussian_grad(d, farplane, intensity)
        farplane[..., measured_pixels] = -grad[..., measured_pixels]
    else:
        raise ValueError(noise_model)
    unmeasured = np.logical_not(measured_pixels)
    farplane[..., unmeasured] *= np.float32(unmeasured_pixels_scaling - 1.0)
    farplane = ops.propagation_adj(farplane, norm)
    chi = np.ascontiguousarray(farplane[..., pad:end, pad:end])
    S = chi.shape[-3]
    object_upd_sum = None
    if recover_psi:
        object_upd_sum = np.zeros_like(psi)
        proj = np.conj(unique_probe) * chi
        object_upd_sum[0] = ops.patch_adj(
            patches=proj.reshape((hi - lo) * S, pw, pw),
            images=object_upd_sum[0],
            positions=scan[lo:hi],
            nrepeat=S,
        )
    patches = probe_update = m_probe_update = None
    if recover_probe or recover_positions:
        patches = ops.patch_fwd(psi[0], scan[lo:hi],
                                patch_width=pw)[..., None, None, :, :]
    if recover_probe:
        probe_update = np.conj(patches) * chi
        m_probe_update = np.sum(probe_update, axis=-5,
                                keepdims=True) / num_batch
    pos_num = pos_den = None
    if recover_positions:
        pos_num, pos_den = opos.position_update_terms(patches, unique_probe,
                                                      chi, m=0)
    return dict(chi=chi, unique_probe=unique_probe, probe_update=probe_update,
                object_upd_sum=object_upd_sum, m_probe_update=m_probe_update,
                costs=costs, patches=patches, intensity=intensity,
                position_numerator=pos_num, position_denominator=pos_den)


def precondition_nearplane_gradients(nearplane, scan, unique_probe, probe,
                                     object_upd_sum, m_probe_update,
                                     psi_update_denominator, patches, lo, hi,
                                     *, m=0, recover_psi=True,
                                     recover_probe=True):
    """lstsq.py:619-718.  Returns (object_update_precond, beta_o, beta_p)."""
    eps = np.float32(1e-9) / (nearplane.shape[-2] * nearplane.shape[-1])
    object_update_precond = None
    if recover_psi:
        object_update_precond = precondition_object_update(
            object_upd_sum, psi_update_denominator)
        object_update_proj = ops.patch_fwd(
            object_update_precond[0], scan[lo:hi],
            patch_width=nearplane.shape[-1])
        dOP = object_update_proj[..., None, None, :, :] * unique_probe[
            ..., m:m + 1, :, :]
        A1 = np.sum((dOP * dOP.conj()).real + eps, axis=(-2, -1))
        A1 += 0.5 * np.mean(A1, axis=-3)
    if recover_probe:
        dPO = m_probe_update[..., m:m + 1, :, :] * patches
        A4 = np.sum((dPO * dPO.conj()).real + eps, axis=(-2, -1))
        A4 += 0.5 * np.mean(A4, axis=-3)
    chi_m = nearplane[..., m:m + 1, :, :]
    if recover_psi and recover_probe:
        b1 = np.sum((dOP.conj() * chi_m).real, axis=(-2, -1))
        b2 = np.sum((dPO.conj() * chi_m).real, axis=(-2, -1))
        A2 = np.sum(dOP * dPO.conj(), axis=(-2, -1))
        A3 = A2.conj()
        determinant = A1 * A4 - A2 * A3
        x1 = -np.conj(A2 * b2 - A4 * b1) / determinant
        x2 = np.conj(A1 * b2 - A3 * b1) / determinant
    elif recover_psi:
        b1 = np.sum((dOP.conj() * chi_m).real, axis=(-2, -1))
        x1 = b1 / A1
    elif recover_probe:
        b2 = np.sum((dPO.conj() * chi_m).real, axis=(-2, -1))
        x2 = b2 / A4
    beta_object = beta_probe = None
    if recover_psi:
        step = 0.9 * np.maximum(0, x1[..., None, None].real)
        beta_object = np.mean(step, keepdims=False, axis=-5)[..., 0, 0, 0]
    if recover_probe:
        step = 0.9 * np.maximum(0, x2[..., None, None].real)
        beta_probe = np.mean(step, axis=-5, keepdims=False)
    return object_update_precond, beta_object, beta_probe


def get_coefs_intensity(weights, xi, P, O, lo, hi, *, m=0):
    """lstsq.py:721-738."""
    OP = O * P[:, :, m:m + 1, :, :]
    num = np.sum(np.real(np.conj(OP) * xi[:, :, m:m + 1, :, :]), axis=(-1, -2))
    den = np.sum(np.abs(OP)**2, axis=(-1, -2))
    weights[lo:hi, 0:1, m:m + 1] += 0.1 * num / den
    return weights


def update_nearplane(g, probe, eigen_probe, eigen_weights, lo, hi, *,
                     num_batch):
    """lstsq.py:297-364 (m = 0)."""
    m = 0
    if eigen_weights is not None:
        eigen_weights = get_coefs_intensity(eigen_weights, g["chi"], probe,
                                            g["patches"], lo, hi, m=m)
        if eigen_weights.shape[-2] > 1:
            R = (g["probe_update"][..., m:m + 1, :, :] -
                 g["m_probe_update"][..., m:m + 1, :, :])
        if eigen_probe is not None and m < eigen_probe.shape[-3]:
            assert eigen_weights.shape[-2] == eigen_probe.shape[-4] + 1
            for eigen_index in range(1, eigen_probe.shape[-4] + 1):
                eigen_probe, eigen_weights = update_eigen_probe(
                    R, eigen_probe, eigen_weights, g["patches"], g["chi"], lo,
                    hi, beta=min(0.1, 1.0 / num_batch), c=eigen_index, m=m)
                if eigen_index + 1 < eigen_weights.shape[-2]:
                    R = R - projection(
                        R, eigen_probe[:, eigen_index - 1:eigen_index,
                                       m:m + 1, :, :], axis=(-2, -1))
    return eigen_probe, eigen_weights


def fit_line_least_squares(y, x):
    """src/tike/opt.py:383-400."""
    x = np.asarray(x, dtype=float)
    y = np.asarray(y, dtype=float)
    count = len(x)
    sum_x, sum_y = np.sum(x), np.sum(y)
    slope = (count * np.sum(x * y) - sum_x * sum_y) / (count * np.sum(x * x) -
                                                       sum_x * sum_x)
    return slope, (sum_y - slope * sum_x) / count


def momentum_checked(g, v, m, mdecay, errors, beta=1.0, memory_length=3):
    """lstsq.py:809-858."""
    m = np.zeros_like(g) if m is None else m
    previous_g = (np.zeros((memory_length, *g.shape), dtype=g.dtype)
                  if v is None else v)
    previous_g = np.roll(previous_g, shift=-1, axis=0)
    previous_g[-1] = g / norm(g) * beta
    if (len(errors) > 2
            and max(errors[-3], errors[-2]) > min(errors[-2], errors[-1])):
        corr = inner(previous_g[:-1], previous_g[-1],
                     axis=(-2, -1)).real.flatten()
        if np.all(corr > 0):
            friction, _ = fit_line_least_squares(
                x=np.arange(len(corr) + 1), y=[0] + np.log(corr).tolist())
            friction = 0.5 * max(-friction, 0)
            m = (1 - friction) * m + g
            return mdecay * m, previous_g, m
    return np.zeros_like(g), previous_g, m / 2


def lstsq_grad(state, data, batches, *, epoch, detector_shape,
               batch_method="compact", measured_pixels=None,
               unmeasured_pixels_scaling=1.0, norm="ortho",
               noise_model="gaussian", step_length_start=0.5,
               step_length_usemodes="all_modes", step_length_weight=0.5,
               recover_psi=True, recover_probe=True, probe_update_start=0,
               object_adaptive_moment=False, object_mdecay=0.9,
               probe_adaptive_moment=False, probe_mdecay=0.9, rng=None):
    """One epoch of lstsq.py:25-294 (positions fixed)."""
    psi, probe, scan = state["psi"], state["probe"], state["scan"]
    eigen_probe = state.get("eigen_probe")
    eigen_weights = state.get("eigen_weights")
    num_batch = len(batches)
    if measured_pixels is None:
        measured_pixels = np.ones((detector_shape, detector_shape), dtype=bool)
    recover_probe = recover_probe and epoch >= probe_update_start
    if batch_method == "compact":
        order = range(num_batch)
    else:
        order = (rng or np.random.default_rng()).permutation(num_batch)
    object_combined_update = np.zeros_like(psi)
    probe_combined_update = np.zeros_like(probe)
    batch_cost = np.empty(num_batch, dtype=np.float32)
    beta_object, beta_probe = [], []
    position = state.get("position")
    pos_num = np.zeros_like(scan) if position is not None else None
    pos_den = np.zeros_like(scan) if position is not None else None
    for batch_index in order:
        lo = int(batches[batch_index][0])
        hi = lo + len(batches[batch_index])
        g = get_nearplane_gradients(
            data, psi, scan, probe, eigen_probe, eigen_weights, lo, hi,
            num_batch=num_batch, detector_shape=detector_shape,
            measured_pixels=measured_pixels,
            unmeasured_pixels_scaling=unmeasured_pixels_scaling, norm=norm,
            noise_model=noise_model, step_length_start=step_length_start,
            step_length_usemodes=step_length_usemodes,
            step_length_weight=step_length_weight,
            recover_psi=recover_psi, recover_probe=recover_probe,
            recover_positions=position is not None)
        if position is not None:
            pos_num[lo:hi] = g["position_numerator"]
            pos_den[lo:hi] = g["position_denominator"]
        if recover_probe:
            eigen_probe, eigen_weights = update_nearplane(
                g, probe, eigen_probe, eigen_weights, lo, hi,
                num_batch=num_batch)
        precond, bbeta_object, bbeta_probe = precondition_nearplane_gradients(
            g["chi"], scan, g["unique_probe"], probe, g["object_upd_sum"],
            g["m_probe_update"], state["psi_precond"], g["patches"], lo, hi,
            m=0, recover_psi=recover_psi, recover_probe=recover_probe)
        if recover_psi:
            if batch_method != "compact":
                dpsi = bbeta_object * precond
                if object_adaptive_moment:
                    mm = state.get("object_m")
                    mm = 0 if mm is None else mm
                    mm = object_mdecay * mm + (1 - object_mdecay) * dpsi
                    state["object_m"] = mm
                    dpsi = mm
                psi = (psi + dpsi).astype(np.complex64)
            else:
                object_combined_update += g["object_upd_sum"]
            beta_object.append(bbeta_object)
        if recover_probe:
            dprobe = bbeta_probe * g["m_probe_update"]
            probe_combined_update += dprobe / num_batch
            probe = (probe + dprobe).astype(np.complex64)
            beta_probe.append(bbeta_probe)
        batch_cost[batch_index] = np.mean(g["costs"])
    if position is not None:
        # lstsq.py:209-220 (the minibatches above used the old positions)
        state["scan"] = opos.update_position(scan, position, pos_num, pos_den,
                                             epoch=epoch)
    state["costs"].append([float(batch_cost.mean())])
    if recover_psi and batch_method == "compact":
        precond = precondition_object_update(object_combined_update,
                                             state["psi_precond"])
        bo = np.mean(np.stack(beta_object))
        dpsi = bo * precond
        psi = (psi + dpsi).astype(np.complex64)
        if object_adaptive_moment:
            dpsi, state["object_v"], state["object_m"] = momentum_checked(
                g=dpsi, v=state.get("object_v"), m=state.get("object_m"),
                mdecay=object_mdecay,
                errors=[float(x[0]) for x in state["costs"][-3:]], beta=bo,
                memory_length=3)
            weight = state["psi_precond"]
            weight = weight / (0.1 * weight.max() + weight)
            psi = (psi + weight * dpsi).astype(np.complex64)
    if recover_probe and probe_adaptive_moment:
        bp = np.mean(np.stack(beta_probe))
        dprobe = probe_combined_update
        if state.get("probe_v") is None:
            state["probe_v"] = np.zeros((3, *dprobe.shape), dtype=dprobe.dtype)
        if state.get("probe_m") is None:
            state["probe_m"] = np.zeros_like(dprobe)
        mode = 0
        d, state["probe_v"][..., mode, :, :], state["probe_m"][
            ..., mode, :, :] = momentum_checked(
                g=dprobe[..., mode, :, :], v=state["probe_v"][..., mode, :, :],
                m=state["probe_m"][..., mode, :, :], mdecay=probe_mdecay,
                errors=[float(x[0]) for x in state["costs"][-3:]], beta=bp,
                memory_length=3)
        probe[..., mode, :, :] = probe[..., mode, :, :] + d
    state["psi"], state["probe"] = psi, probe
    state["eigen_probe"], state["eigen_weights"] = eigen_probe, eigen_weights
    return state


# --------------------------------------------------------------------------
# rpie (src/tike/ptycho/solvers/rpie.py:26-612), as this snapshot has it:
# the probe numerator is re-zeroed by every minibatch (:349), the probe step is
# alpha * max(preconditioner) only (:271-280), position correction is
# commented out, and the object gradient is divided by the number of modes
# (:466).  Multislice objects go through fwd_return_intermediate_probes.
# --------------------------------------------------------------------------


def rpie_gradients(data, psi, scan, probe, eigen_probe, eigen_weights, lo, hi,
                   psi_num, *, detector_shape, measured_pixels,
                   propagator=None, noise_model="gaussian",
                   unmeasured_pixels_scaling=1.0, norm="ortho",
                   recover_psi=True, recover_probe=True,
                   step_length_start=0.5, step_length_usemodes="all_modes",
                   step_length_weight=0.5):
    """rpie.py:310-548 for one minibatch [lo, hi)."""
    pw = probe.shape[-1]
    pad = (detector_shape - pw) // 2
    end = pad + pw
    S = probe.shape[-3]
    unique_probe = get_varying_probe(
        probe, eigen_probe,
        eigen_weights[lo:hi] if eigen_weights is not None else None)
    if unique_probe.shape[0] == 1 and hi - lo > 1:
        unique_probe = np.broadcast_to(
            unique_probe, (hi - lo, *unique_probe.shape[1:])).copy()
    if psi.shape[0] == 1:
        farplane = ops.ptycho_fwd(unique_probe, scan[lo:hi], psi,
                                  detector_shape, norm)
        probes = unique_probe[None, :, 0]
    else:
        farplane, probes = ops.ptycho_fwd_intermediate(
            unique_probe, scan[lo:hi], psi, propagator, norm)
    intensity = np.sum(np.square(np.abs(farplane)),
                       axis=tuple(range(1, farplane.ndim - 2)))
    d = data[lo:hi].astype(np.float32)
    each = getattr(ops, f"{noise_model}_each_pattern")
    costs = each(d[:, measured_pixels][:, None, :],
                 intensity[:, measured_pixels][:, None, :])
    if noise_model == "poisson":
        with np.errstate(invalid="ignore", divide="ignore"):
            xi = (1 - d / intensity)[:, None, None, ...]
            grad_cost = farplane * xi
            step = np.full((farplane.shape[0], 1, farplane.shape[2], 1, 1),
                           np.float32(step_length_start), dtype=np.float32)
            if step_length_usemodes == "dominant_mode":
                step = poisson_steplength_dominant_mode(
                    xi, intensity, d, measured_pixels, step,
                    step_length_weight)
            else:
                step = poisson_steplength_all_modes(
                    xi, np.square(np.abs(farplane)), intensity, d,
                    measured_pixels, step, step_length_weight)
            farplane[..., measured_pixels] = (
                -step * grad_cost)[..., measured_pixels]
    else:
        grad = ops.gaussian_grad(d, farplane, intensity)
        farplane[..., measured_pixels] = -grad[..., measured_pixels]
    farplane[..., np.logical_not(measured_pixels)] *= np.float32(
        unmeasured_pixels_scaling - 1.0)
    diff = np.ascontiguousarray(
        ops.propagation_adj(farplane, norm)[..., pad:end, pad:end])
    N = hi - lo
    probe_num = np.zeros((psi.shape[0], *probe.shape), dtype=probe.dtype)
    if recover_psi:
        for tt in range(len(psi) - 1, -1, -1):
            grad_psi = (np.conj(probes[tt][:, None]) * diff / S).reshape(
                N * S, pw, pw)
            psi_num[tt] = ops.patch_adj(patches=grad_psi, images=psi_num[tt],
                                        positions=scan[lo:hi], nrepeat=S)
            patches = ops.patch_fwd(psi[tt], scan[lo:hi],
                                    patch_width=pw)[..., None, None, :, :]
            probe_num[tt] += np.sum(np.conj(patches) * diff, axis=-5,
                                    keepdims=True)
            if tt == 0:
                break
            diff = ops.fresnel_adj(diff, propagator, norm)
    patches = ops.patch_fwd(psi[0], scan[lo:hi],
                            patch_width=pw)[..., None, None, :, :]
    if recover_probe and eigen_weights is not None:
        m = 0
        OP = patches * probe[..., m:m + 1, :, :]
        num = np.sum(np.real(np.conj(OP) * diff[..., m:m + 1, :, :]),
                     axis=(-1, -2))
        den = np.sum(np.abs(OP)**2, axis=(-1, -2))
        eigen_weights[lo:hi, 0:1, m:m + 1] += 0.1 * (num / den)
    return costs, psi_num, probe_num, eigen_weights


def rpie_update(state, psi_num, probe_num, *, alpha, recover_psi,
                recover_probe, errors=None, object_adaptive_moment=False,
                probe_adaptive_moment=False, object_mdecay=0.9,
                probe_mdecay=0.9, vdecay=0.999):
    """rpie.py:217-307."""
    psi, probe = state["psi"], state["probe"]
    if recover_psi:
        P = state["psi_precond"]
        deno = (1 - alpha) * P + alpha * P.real.max(axis=(-2, -1),
                                                    keepdims=True)
        dpsi = psi_num
        psi = psi + dpsi / deno
        if object_adaptive_moment:
            if errors:
                dpsi, state["object_v"], state["object_m"] = momentum_checked(
                    g=dpsi, v=state.get("object_v"), m=state.get("object_m"),
                    mdecay=object_mdecay, errors=errors, memory_length=3)
            else:
                dpsi, state["object_v"], state["object_m"] = opos.adam(
                    dpsi, state.get("object_v"), state.get("object_m"),
                    vdecay=vdecay, mdecay=object_mdecay)
            psi = psi + dpsi / deno
    if recover_probe:
        dprobe = probe_num[0]
        deno = alpha * state["probe_precond"][0].real.max(axis=(-2, -1),
                                                          keepdims=True)
        probe = probe + dprobe / deno
        if probe_adaptive_moment:
            mode = 0
            if errors:
                (dprobe[0, 0, mode], state["probe_v"],
                 state["probe_m"]) = momentum_checked(
                     g=dprobe[0, 0, mode], v=state.get("probe_v"),
                     m=state.get("probe_m"), mdecay=probe_mdecay,
                     errors=errors, memory_length=3)
            else:
                (dprobe[0, 0, mode], state["probe_v"],
                 state["probe_m"]) = opos.adam(
                     dprobe[0, 0, mode], state.get("probe_v"),
                     state.get("probe_m"), vdecay=vdecay, mdecay=probe_mdecay)
            probe = probe + dprobe / deno
    state["psi"] = psi.astype(np.complex64)
    state["probe"] = probe.astype(np.complex64)
    return state


def rpie(state, data, batches, *, epoch, detector_shape, alpha=0.05,
         batch_method="compact", measured_pixels=None, rng=None,
         recover_psi=True, recover_probe=True, probe_update_start=0,
         object_adaptive_moment=False, probe_adaptive_moment=False,
         propagator=None, **kw):
    """One epoch of rpie.py:26-214."""
    num_batch = len(batches)
    if measured_pixels is None:
        measured_pixels = np.ones((detector_shape, detector_shape), dtype=bool)
    recover_probe = recover_probe and epoch >= probe_update_start
    if batch_method == "compact":
        order = range(num_batch)
    else:
        order = (rng or np.random.default_rng()).permutation(num_batch)
    psi_num = probe_num = None
    batch_cost = np.empty(num_batch, dtype=np.float32)
    upd = dict(alpha=alpha, recover_psi=recover_psi,
               recover_probe=recover_probe,
               object_adaptive_moment=object_adaptive_moment,
               probe_adaptive_moment=probe_adaptive_moment)
    for n in order:
        lo = int(batches[n][0])
        hi = lo + len(batches[n])
        if psi_num is None:
            psi_num = np.zeros_like(state["psi"])
        costs, psi_num, probe_num, state["eigen_weights"] = rpie_gradients(
            data, state["psi"], state["scan"], state["probe"],
            state.get("eigen_probe"), state.get("eigen_weights"), lo, hi,
            psi_num, detector_shape=detector_shape,
            measured_pixels=measured_pixels, propagator=propagator,
            recover_psi=recover_psi, recover_probe=recover_probe, **kw)
        batch_cost[n] = np.mean(costs)
        if batch_method != "compact":
            state = rpie_update(state, psi_num, probe_num, **upd)
            psi_num = probe_num = None
    state["costs"].append([float(batch_cost.mean())])
    if batch_method == "compact":
        state = rpie_update(
            state, psi_num, probe_num,
            errors=[float(x[0]) for x in state["costs"][-3:]], **upd)
    if state.get("eigen_weights") is not None:
        w = state["eigen_weights"]
        state["eigen_weights"] = (w / mnorm(w, axis=-3, keepdims=True)).astype(
            np.float32)
    return state


# --------------------------------------------------------------------------
# epoch driver (src/tike/ptycho/ptycho.py:431-564, 723-808, 873-972)
# --------------------------------------------------------------------------


def rescale_probe(state, data, detector_shape, measured_pixels=None,
                  norm="ortho", propagator=None):
    """ptycho.py:873-972: probe *= sqrt(sum(data) / sum(intensity))."""
    if measured_pixels is None:
        measured_pixels = np.ones((detector_shape, detector_shape), dtype=bool)
    far = ops.ptycho_fwd(state["probe"], state["scan"], state["psi"],
                         detector_shape, norm, propagator=propagator)
    intensity = ops.intensity_from_farplane(far)
    n0 = np.sum(data[:, measured_pixels], dtype=np.double)
    n1 = np.sum(intensity[:, measured_pixels], dtype=np.double)
    rescale = np.float32(np.sqrt(n0) / np.sqrt(n1))
    state["probe"] = (state["probe"] * rescale).astype(np.complex64)
    return state


def contiguous_batches(n, num_batch):
    """Equal contiguous index ranges (the build's 'compact'-style batches)."""
    return np.array_split(np.arange(n), num_batch)


def iterate(state, data, batches, num_iter, *, detector_shape,
            solver="lstsq_grad", force_orthogonality=False,
            rescale_period=10, **kw):
    """ptycho.py:431-564 for one worker: constraints -> preconditioners ->
    solver -> object constraints."""
    for _ in range(num_iter):
        epoch = len(state["costs"])
        if kw.get("recover_probe", True) and epoch >= kw.get(
                "probe_update_start", 0):
            if force_orthogonality:
                state["probe"], power = orthogonalize_eig(state["probe"])
            else:
                power = probe_power(state["probe"])
            state.setdefault("power", []).append(power)
            if state.get("eigen_probe") is not None:
                state["eigen_probe"], state[
                    "eigen_weights"] = constrain_variable_probe(
                        state["eigen_probe"], state["eigen_weights"])
        state["psi_precond"] = psi_preconditioner(
            state["psi"], state["probe"], state["scan"],
            propagator=kw.get("propagator"))
        state["probe_precond"] = probe_preconditioner(state["psi"],
                                                      state["probe"],
                                                      state["scan"])
        if solver == "lstsq_grad":
            state = lstsq_grad(state, data, batches, epoch=epoch,
                               detector_shape=detector_shape, **kw)
        elif solver == "cgrad":
            state = cgrad(state, data, batches, detector_shape=detector_shape,
                          **kw)
        elif solver == "rpie":
            state = rpie(state, data, batches, epoch=epoch,
                         detector_shape=detector_shape, **kw)
        else:
            raise ValueError(solver)
        if (state.get("psi_precond") is not None
                and len(state["costs"]) % rescale_period == 0):
            state["psi"], state["probe"] = remove_object_ambiguity(
                state["psi"], state["probe"], state["psi_precond"])
        if state.get("position") is not None:
            # _apply_position_constraints (ptycho.py:521-524, 854-866)
            state["scan"] = opos.affine_position_regularization(
                state["scan"], state["position"],
                kw.get("rng") or np.random.default_rng()).astype(np.float32)
    return state


# --------------------------------------------------------------------------
# cgrad: composed from tike.opt.conjugate_gradient (src/tike/opt.py:216-380)
# over Ptycho.cost / Ptycho.adj(gaussian_grad) -- absent from the reference
# as a ptychography solver; template lamino/solvers/cgrad.py:58-92
# --------------------------------------------------------------------------


def line_search(f, x, d, step_length=1.0, step_shrink=0.5, cost=None):
    """opt.py:216-278."""
    fx = f(x) if cost is None else cost
    while True:
        xsd = x + np.float32(step_length) * d
        fxsd = f(xsd)
        if fxsd <= fx:
            break
        step_length *= step_shrink
        if step_length < 1e-32:
            step_length, fxsd, xsd = 0, fx, x
            break
    return step_length, fxsd, xsd


def direction_dy(grad1, grad0=None, dir_=None):
    """opt.py:281-301 (Dai-Yuan)."""
    if dir_ is None:
        return -grad1
    return (-grad1 + dir_ * np.linalg.norm(grad1.ravel())**2 /
            (np.sum(dir_.conj() * (grad1 - grad0)) + 1e-32))


def conjugate_gradient(x, cost_function, grad, num_iter=1, step_length=1.0,
                       cost=None):
    """opt.py:312-380 with num_search == num_iter."""
    dir_ = grad0 = None
    for i in range(num_iter):
        grad1 = grad(x)
        dir_ = direction_dy(grad1) if i == 0 else direction_dy(
            grad1, grad0, dir_)
        grad0 = grad1
        step_length, cost, x = line_search(cost_function, x, dir_,
                                           step_length=step_length, cost=cost)
        x = x.astype(np.complex64)
    return x, cost


def cgrad(state, data, batches, *, detector_shape, cg_iter=4, step_length=1.0,
          recover_psi=True, recover_probe=False, norm="ortho", **_):
    """Nonlinear CG on the gaussian cost, object then (optionally) probe."""
    psi, probe, scan = state["psi"], state["probe"], state["scan"]
    batch_cost = []
    for b in batches:
        lo, hi = int(b[0]), int(b[0]) + len(b)
        d = data[lo:hi].astype(np.float32)
        s = scan[lo:hi]

        def cost_psi(p):
            return float(
                ops.ptycho_cost(d, p, s, probe, detector_shape, "gaussian",
                                norm))

        def grad_psi(p):
            far = ops.ptycho_fwd(probe, s, p, detector_shape, norm)
            inten = ops.intensity_from_farplane(far)
            g = ops.gaussian_grad(d, far, inten).astype(np.complex64)
            uprobe = np.broadcast_to(probe, (hi - lo, *probe.shape[1:]))
            return ops.ptycho_adj(g, uprobe, s, p, norm)[0]

        cost = None
        if recover_psi:
            psi, cost = conjugate_gradient(psi, cost_psi, grad_psi,
                                           num_iter=cg_iter,
                                           step_length=step_length)
        if recover_probe:

            def cost_probe(q):
                return float(
                    ops.ptycho_cost(d, psi, s, q, detector_shape, "gaussian",
                                    norm))

            def grad_probe(q):
                far = ops.ptycho_fwd(q, s, psi, detector_shape, norm)
                inten = ops.intensity_from_farplane(far)
                g = ops.gaussian_grad(d, far, inten).astype(np.complex64)
                uprobe = np.broadcast_to(q, (hi - lo, *q.shape[1:]))
                return np.sum(ops.ptycho_adj(g, uprobe, s, psi, norm)[1],
                              axis=0, keepdims=True)

            probe, cost = conjugate_gradient(probe, cost_probe, grad_probe,
                                             num_iter=cg_iter,
                                             step_length=step_length)
        batch_cost.append(cost if cost is not None else cost_psi(psi))
    state["costs"].append([float(np.mean(batch_cost))])
    state["psi"], state["probe"] = psi, probe
    return state
