"""CPU tests of the host-side logic around the hot path: option validation
(reference options.py:141-168, ptycho.py:306-323), batch clustering
invariants (reference tests/test_random.py), probe helpers against the
oracle and the reference's golden files, linalg / opt helpers
(reference tests/test_linalg.py, tests/test_opt.py)."""
import numpy as np
import pytest
import torch

import tike_amd.cluster as cluster
import tike_amd.linalg as linalg
import tike_amd.opt as opt
import tike_amd.ptycho as tp
from oracle import solvers as osol


def rc(rng, *shape):
    return (rng.random((*shape, 2), dtype=np.float32) - 0.5).view(
        np.complex64)[..., 0]


def test_check_allowed_positions():
    """reference tests/ptycho/test_ptycho.py:92-100."""
    psi = np.empty((1, 4, 9))
    probe = np.empty((8, 2, 2))
    scan = np.array([[1, 1], [1, 6.9], [1.1, 1], [1.9, 5.5]])
    tp.check_allowed_positions(scan, psi, probe.shape)
    for bad in np.array([[1, 7], [1, 0.9], [0.9, 1], [1, 0]]):
        with pytest.raises(ValueError):
            tp.check_allowed_positions(bad[None], psi, probe.shape)


def test_parameters_validation():
    good = dict(probe=np.ones((1, 1, 2, 8, 8), np.complex64),
                psi=np.ones((1, 32, 32), np.complex64),
                scan=np.full((5, 2), 4, np.float32))
    p = tp.PtychoParameters(**good)
    assert p.exitwave_options.measured_pixels.shape == (8, 8)
    with pytest.raises(ValueError):
        tp.PtychoParameters(**{**good, "scan": np.zeros((5, 3), np.float32)})
    with pytest.raises(ValueError):
        tp.PtychoParameters(**{**good,
                               "probe": np.ones((1, 2, 8, 8), np.complex64)})
    with pytest.raises(ValueError):
        tp.PtychoParameters(**{**good, "psi": np.ones((1, 8, 8),
                                                      np.complex64)})
    with pytest.raises(ValueError):  # positions outside the field of view
        tp.PtychoParameters(**{**good, "scan": np.full((5, 2), 30,
                                                       np.float32)})


def test_options_defaults_match_reference():
    o = tp.LstsqOptions()
    assert (o.name, o.num_batch, o.batch_method, o.rescale_period) == (
        "lstsq_grad", 1, "wobbly_center", 10)
    po = tp.ProbeOptions()
    assert po.init_rescale_from_measurements and not po.force_orthogonality
    assert po.recover_probe(3) and not tp.ProbeOptions(
        update_start=5).recover_probe(3)
    eo = tp.ExitWaveOptions(measured_pixels=np.ones((4, 4), bool))
    assert (eo.noise_model, eo.propagation_normalization,
            eo.unmeasured_pixels_scaling) == ("gaussian", "ortho", 1.0)


@pytest.mark.parametrize("method", ["wobbly_center", "compact", "contiguous"])
def test_cluster_partitions(method):
    """Every index in exactly one cluster, sizes differ by <= 1
    (reference tests/test_random.py invariants)."""
    rng = np.random.default_rng(0)
    np.random.seed(0)
    pop = rng.random((97, 2))
    order, batches = cluster.batches_contiguous(pop, method, 5)
    assert sorted(order.tolist()) == list(range(97))
    sizes = [len(b) for b in batches]
    assert sum(sizes) == 97 and max(sizes) - min(sizes) <= 1
    assert np.concatenate(batches).tolist() == list(range(97))


def test_gaussian_probe_bit_exact(golden):
    g = golden("ref_ptycho_gaussian.npz")
    np.testing.assert_array_equal(tp.gaussian(15, rin=0.8, rout=1.0),
                                  g["weights"])


def test_orthogonalize_eig_reference_mat(golden):
    """reference tests/ptycho/test_probe.py:138-158."""
    g = golden("ref_ortho.npz")
    out, power = tp.orthogonalize_eig(g["modes"])
    np.testing.assert_allclose(np.abs(out), np.abs(g["pr"]), rtol=1e-4)
    assert np.all(np.diff(power) <= 0)


def test_probe_helpers_match_oracle():
    rng = np.random.default_rng(1)
    N, C, S, pw = 12, 2, 3, 8
    probe, eigen = rc(rng, 1, 1, S, pw, pw), rc(rng, 1, C, 1, pw, pw)
    w = rng.standard_normal((N, C + 1, S)).astype(np.float32)
    np.testing.assert_allclose(tp.get_varying_probe(probe, eigen, w),
                               osol.get_varying_probe(probe, eigen, w),
                               rtol=1e-6)
    e1, w1 = tp.constrain_variable_probe(eigen.copy(), w.copy())
    e2, w2 = osol.constrain_variable_probe(eigen.copy(), w.copy())
    np.testing.assert_allclose(e1, e2, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(w1, w2, rtol=1e-4, atol=1e-6)
    psi = rc(rng, 1, 20, 20)
    pre = rng.random((1, 20, 20)).astype(np.complex64)
    a, b = tp.remove_object_ambiguity(torch.from_numpy(psi),
                                      torch.from_numpy(probe),
                                      torch.from_numpy(pre))
    c, d = osol.remove_object_ambiguity(psi, probe, pre)
    np.testing.assert_allclose(a.numpy(), c, rtol=1e-5)
    np.testing.assert_allclose(b.numpy(), d, rtol=1e-5)


@pytest.mark.parametrize("kind", ["numpy", "torch"])
def test_linalg(kind):
    """reference tests/test_linalg.py: norm, projection, lstsq, GS."""
    rng = np.random.default_rng(2)
    x, y = rc(rng, 4, 5, 6), rc(rng, 4, 5, 6)
    conv = (lambda a: a) if kind == "numpy" else torch.from_numpy
    back = (lambda a: a) if kind == "numpy" else (lambda a: a.numpy())
    np.testing.assert_allclose(back(linalg.norm(conv(x), axis=(-2, -1))),
                               np.sqrt((np.abs(x)**2).sum((-2, -1))),
                               rtol=1e-5)
    np.testing.assert_allclose(back(linalg.mnorm(conv(x))),
                               np.sqrt((np.abs(x)**2).mean()), rtol=1e-5)
    np.testing.assert_allclose(back(linalg.inner(conv(x), conv(y))),
                               (x * y.conj()).sum(), rtol=1e-4)
    p = back(linalg.projection(conv(x), conv(y), axis=(-2, -1)))
    resid = x - p
    np.testing.assert_allclose((resid * y.conj()).sum((-2, -1)), 0, atol=1e-5)
    u = back(linalg.orthogonalize_gs(conv(x), axis=(-2, -1)))
    for i in range(4):  # vectors run along the first axis not in `axis`
        for j in range(i):
            np.testing.assert_allclose((u[i] * u[j].conj()).sum(), 0,
                                       atol=1e-5)
    a = rng.standard_normal((3, 7, 2)).astype(np.float32)
    xs = rng.standard_normal((3, 2, 1)).astype(np.float32)
    np.testing.assert_allclose(back(linalg.lstsq(conv(a), conv(a @ xs))), xs,
                               rtol=1e-3, atol=1e-4)


def test_opt_helpers():
    """reference tests/test_opt.py: line fit; CG converges on a quadratic."""
    s, i = opt.fit_line_least_squares(y=[1, 3, 5, 7], x=[0, 1, 2, 3])
    np.testing.assert_allclose([s, i], [2, 1])
    A_ = np.array([[3.0, 1.0], [1.0, 2.0]])
    b = np.array([1.0, -1.0])
    cost = lambda x: float(0.5 * x @ A_ @ x - b @ x)
    x, c = opt.conjugate_gradient(np, np.zeros(2), cost,
                                  lambda x: [A_ @ x - b],
                                  dir_multi=lambda d: d[0], num_iter=20)
    np.testing.assert_allclose(x, np.linalg.solve(A_, b), atol=1e-3)
    m, _, m2 = opt.momentum(np.ones(3), None, None, mdecay=0.9)
    np.testing.assert_allclose(m, 0.1 * np.ones(3))


def test_unsupported_features_fail_loudly():
    params = tp.PtychoParameters(
        probe=np.ones((1, 1, 1, 8, 8), np.complex64),
        psi=np.ones((1, 32, 32), np.complex64),
        scan=np.full((4, 2), 4, np.float32),
        algorithm_options=tp.LstsqOptions())
    # multislice objects are reconstructed by rpie only, as in the reference
    deep = tp.PtychoParameters(
        probe=params.probe, psi=np.ones((2, 32, 32), np.complex64),
        scan=params.scan, algorithm_options=tp.LstsqOptions())
    with pytest.raises(NotImplementedError):
        tp.Reconstruction(np.ones((4, 8, 8), np.float32), deep)
    with pytest.raises(ValueError):
        tp.Reconstruction(np.ones((3, 8, 8), np.float32),
                          tp.PtychoParameters(
                              probe=params.probe, psi=params.psi,
                              scan=params.scan,
                              algorithm_options=tp.LstsqOptions()))


# ------------------------------------------------------- position correction
def test_gaussian_derivative_taps_match_scipy():
    """The taps handed to tike_position_sums reproduce the reference's
    gaussian_gradient (position.py:779-810 = scipy gaussian_filter1d)."""
    import scipy.ndimage
    from tike_amd.ptycho.position import gaussian_derivative_taps
    from oracle import position as opos
    taps, r = gaussian_derivative_taps(0.333)
    assert r == 2 and taps.shape == (5,)
    rng = np.random.default_rng(0)
    x = (rng.random((2, 9, 11)) + 1j * rng.random((2, 9, 11))).astype(
        np.complex64)
    ref = scipy.ndimage.gaussian_filter1d(-x, sigma=0.333, order=1, axis=-1,
                                          mode="nearest", truncate=6.0)
    pad = np.pad(x, ((0, 0), (0, 0), (r, r)), mode="edge")
    mine = sum(taps[d + r] * pad[..., r + d:r + d + x.shape[-1]]
               for d in range(-r, r + 1))
    np.testing.assert_allclose(mine, ref, atol=1e-6)
    gx, gy = opos.gaussian_gradient(x)
    np.testing.assert_allclose(gy, ref, atol=1e-6)


def test_affine_transform_roundtrip_and_fit():
    """AffineTransform decomposition / composition and the least-squares and
    RANSAC estimators (position.py:137-327) against the oracle."""
    import tike_amd.random as trandom
    from tike_amd.ptycho import position as pos
    from oracle import position as opos
    t = pos.AffineTransform(scale0=1.02, scale1=0.97, shear1=0.03, angle=0.05,
                            t0=1.5, t1=-2.0)
    back = pos.AffineTransform.fromarray(t.asarray3())
    np.testing.assert_allclose(back.astuple(), t.astuple(), atol=1e-5)
    np.testing.assert_allclose(t.asarray(), opos.transform_matrix(t.astuple()),
                               atol=1e-7)
    rng = np.random.default_rng(3)
    p0 = rng.random((50, 2)) * 100
    p1 = t(p0) + 0.01 * rng.standard_normal((50, 2))
    fit, err = pos.estimate_global_transformation(p0, p1)
    np.testing.assert_allclose(fit.astuple(), t.astuple(), atol=2e-2)
    ofit, oerr = opos.estimate_global_transformation(p0, p1)
    np.testing.assert_allclose(fit.astuple(), ofit, atol=1e-6)
    trandom.randomizer_np = np.random.default_rng(5)
    rfit, _ = pos.estimate_global_transformation_ransac(p0, p1)
    ofit, _ = opos.estimate_global_transformation_ransac(
        p0, p1, opos.IDENTITY, np.random.default_rng(5))
    np.testing.assert_allclose(rfit.astuple(), ofit, atol=1e-6)
    # a fit carried out later with subsets drawn earlier (the deferred fit of
    # Reconstruction._apply_position_constraints) consumes the generator exactly
    # as the immediate fit does, and gives the same transformation
    trandom.randomizer_np = np.random.default_rng(5)
    subsets = pos.ransac_subsets(len(p0))
    state_after = trandom.randomizer_np.bit_generator.state
    later, _ = pos.estimate_global_transformation_ransac(p0, p1,
                                                         subsets=subsets)
    assert trandom.randomizer_np.bit_generator.state == state_after
    np.testing.assert_allclose(later.astuple(), rfit.astuple(), atol=0)
    trandom.randomizer_np = np.random.default_rng(5)
    pos.estimate_global_transformation_ransac(p0, p1)
    assert trandom.randomizer_np.bit_generator.state == state_after


def test_position_options_split_join():
    """PositionOptions.split / join / insert keep per-position state aligned
    (reference tests/ptycho/test_position.py:24-60)."""
    from tike_amd.ptycho.position import PositionOptions
    rng = np.random.default_rng(0)
    scan = rng.random((20, 2)).astype(np.float32) * 30
    opts = PositionOptions(scan, use_adaptive_moment=True)
    opts._momentum[:] = rng.random((20, 4))
    order = rng.permutation(20)
    parts = [order[:7], order[7:]]
    split = [opts.split(b) for b in parts]
    joined = PositionOptions.join(split, reorder=np.argsort(order))
    np.testing.assert_array_equal(joined.initial_scan, opts.initial_scan)
    np.testing.assert_array_equal(joined._momentum, opts._momentum)
    np.testing.assert_array_equal(joined.confidence, opts.confidence)
    assert joined.v.shape == (20, 2) and joined.m.shape == (20, 2)


def test_resample_matches_reference(golden):
    """PtychoParameters.resample (Fourier-interpolated probe, spline-zoomed
    object, scaled positions, Fourier-cropped mask; options.py:170-196,332-409)
    against the reference's own coarse level, and crop/pad round trip."""
    from tike_amd.ptycho.solvers import (_resize_fft, crop_fourier_space,
                                         pad_fourier_space)
    g = golden("multigrid_fft.npz")
    params = tp.PtychoParameters(
        probe=g["probe0"].copy(), psi=g["psi0"].copy(), scan=g["scan"].copy(),
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2),
        probe_options=tp.ProbeOptions(use_adaptive_moment=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((32, 32), dtype=bool)))
    params.probe_options.m = np.ones(3)  # momentum must restart on a new grid
    coarse = params.resample(0.5)
    np.testing.assert_allclose(coarse.probe, g["coarse_probe"], atol=1e-7)
    np.testing.assert_allclose(coarse.psi, g["coarse_psi"], atol=1e-7)
    np.testing.assert_allclose(coarse.scan, g["coarse_scan"], atol=0)
    assert coarse.exitwave_options.measured_pixels.shape == (16, 16)
    assert coarse.probe_options.m is None
    rng = np.random.default_rng(0)
    x = rng.random((2, 8, 8)) + 1j * rng.random((2, 8, 8))
    np.testing.assert_allclose(crop_fourier_space(pad_fourier_space(x, 12), 8),
                               x)
    np.testing.assert_allclose(_resize_fft(x, 1), x)
    assert _resize_fft(x, 2.0).shape == (2, 16, 16)


def test_bench_gpus_flag_is_never_silently_ignored():
    """bench.py --gpus N: with WORLD_SIZE unset it must start N ranks or fail
    (here: no GPU -> non-zero exit, nothing printed on stdout); with a
    WORLD_SIZE that disagrees it must refuse."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "64"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and not r.stdout.strip()
    assert "refusing" in r.stderr
    r = subprocess.run([sys.executable, bench, "--gpus", "4"],
                       env=dict(env, WORLD_SIZE="2"), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_bench_roofline_kernel_has_a_byte_model():
    """bench.py's roofline object names the entry with the most time AMONG
    those with an algorithmic byte model: a composite entry (the device line
    search: many small launches under one name) must not take its place."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location(
        "bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    summary = {
        "tike_cgrad_line_search": dict(total_ms=9.0),
        "tike_lstsq_chunk_gradients": dict(total_ms=7.0),
        "tike_scatter_patches": dict(total_ms=1.0),
        "allreduce": dict(total_ms=50.0),
    }
    assert bench.dominant_entry(summary, 1000, 1, 256,
                                0) == "tike_lstsq_chunk_gradients"
    # nothing modelled: the largest tike_* entry, never a non-kernel row
    assert bench.dominant_entry(
        {"tike_cgrad_line_search": dict(total_ms=1.0),
         "allreduce": dict(total_ms=5.0)}, 10, 1, 128,
        0) == "tike_cgrad_line_search"
    # T + 2P + 8 per position + the shared probe
    n, S, det = 10, 8, 256
    T, P = 8 * S * det * det, 8 * det * det
    assert bench.algorithmic_bytes("tike_fwd_pass1", n, S, det, det,
                                   1) == n * (T + 2 * P + 8) + (S + 1) * P
