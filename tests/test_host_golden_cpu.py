"""Host-side helpers against the REFERENCE's own functions
(tests/golden/host_helpers.npz, written by tests/golden/gen/make_host_fixtures.py
which runs the reference in the build container): same inputs, same generator
seeds, same outputs.  These helpers are re-derived here, not transcribed --
the fixtures are what ties their observable behaviour to the reference."""
import os

import numpy as np
import pytest

from tike_amd import cluster

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    return np.load(os.path.join(GOLDEN, "host_helpers.npz"))


def _labels(groups, n):
    lab = np.full(n, -1, dtype=np.int64)
    for c, grp in enumerate(groups):
        lab[np.asarray(grp)] = c
    return lab


@pytest.mark.parametrize("case", range(7))
def test_cluster_labels_equal_the_reference(g, case):
    """`compact` (k-means++ seeds from the legacy generator, greedy fill, swap
    refinement, clusters sorted by size) and `wobbly_center` reproduce the
    reference's cluster membership point for point, and `compact` leaves the
    legacy generator where the reference leaves it."""
    pop = g[f"cluster_pop_{case}"]
    k = int(g[f"cluster_k_{case}"])
    np.random.seed(int(g[f"cluster_seed_{case}"]))
    got = _labels(cluster.compact(pop, k), len(pop))
    np.testing.assert_array_equal(got, g[f"cluster_compact_{case}"])
    assert np.random.random_sample() == float(g[f"cluster_compact_next_{case}"])
    got = _labels(cluster.wobbly_center(pop, k), len(pop))
    np.testing.assert_array_equal(got, g[f"cluster_wobbly_{case}"])
    # the randomly started variant (two fractions, same generator stream) and
    # the equal-count stripes
    np.random.seed(1000 + int(g[f"cluster_seed_{case}"]))
    got = _labels(cluster.wobbly_center_random_bootstrap(pop, k), len(pop))
    np.testing.assert_array_equal(got, g[f"cluster_bootstrap_{case}"])
    got = _labels(
        cluster.wobbly_center_random_bootstrap(pop, k, boot_fraction=0.5),
        len(pop))
    np.testing.assert_array_equal(got, g[f"cluster_bootstrap_half_{case}"])
    assert np.random.random_sample() == float(
        g[f"cluster_bootstrap_next_{case}"])
    got = _labels(cluster.stripes_equal_count(pop, k, dim=case % 2), len(pop))
    np.testing.assert_array_equal(got, g[f"cluster_stripes_{case}"])


# ---------------------------------------------------------------- tike.opt
def _lsq(g):
    A_, b_ = g["opt_A"], g["opt_b"]

    def cost(x):
        r = A_ @ x - b_
        return float(np.real(np.vdot(r, r)))

    def grad(x):
        return [A_.conj().T @ (A_ @ x - b_)]

    return cost, grad


@pytest.mark.parametrize("tag,kw", [
    ("full", dict(num_iter=5, step_length=1.0)),
    ("partial", dict(num_iter=6, step_length=0.5, num_search=2)),
    ("tiny", dict(num_iter=2, step_length=1e-3))])
def test_conjugate_gradient_iterates_equal_the_reference(g, tag, kw):
    from tike_amd import opt
    cost, grad = _lsq(g)
    x, c = opt.conjugate_gradient(
        np, np.zeros(6, np.complex64), cost, grad,
        update_multi=lambda x, s, d: x + s * d[0], **kw)
    # complex64 iterates: the Dai-Yuan factor is formed in another order
    np.testing.assert_allclose(x, g[f"opt_cg_x_{tag}"], rtol=2e-5, atol=1e-6)
    assert c == pytest.approx(float(g[f"opt_cg_cost_{tag}"]), rel=1e-5)


def test_line_search_and_directions_equal_the_reference(g):
    from tike_amd import opt
    cost, grad = _lsq(g)
    x0 = np.zeros(6, np.complex64)
    # uphill: accepted only once the step is too small to change the cost
    s, c, x = opt.line_search(cost, x0.copy(), [-grad(x0)[0]],
                              lambda x, s, d: x - s * d[0])
    np.testing.assert_array_equal([s, c], g["opt_ls_fail"])
    # no step length is ever accepted: the floor ends the search
    with pytest.warns(UserWarning, match="Line search failed"):
        s, c, x = opt.line_search(lambda x: 1.0 if x is x0 else 2.0, x0, None,
                                  lambda x, s, d: x + s, step_length=1e-30)
    assert (s, c) == (0, 1.0) and x is x0
    s, c, x = opt.line_search(cost, x0.copy(), [-grad(x0)[0]],
                              lambda x, s, d: x + s * d[0], step_length=4.0)
    np.testing.assert_array_equal([s, c], g["opt_ls_ok"])
    np.testing.assert_array_equal(x, g["opt_ls_ok_x"])
    d0 = opt.direction_dy(np, [g["opt_dy_g0"]])
    np.testing.assert_array_equal(d0[0], g["opt_dy_first"])
    d1 = opt.direction_dy(np, [g["opt_dy_g1"]], [g["opt_dy_g0"]], d0)
    np.testing.assert_allclose(d1[0], g["opt_dy_next"], rtol=1e-14)


def test_adam_momentum_line_fit_and_convergence_equal_the_reference(g):
    import torch
    from tike_amd import opt
    x = g["opt_adam_g"]
    d1, v1, m1 = opt.adam(x)
    np.testing.assert_array_equal(np.stack([d1, v1, m1]), g["opt_adam_1"])
    d2, v2, m2 = opt.adam(2 * x - 1, v1, m1, vdecay=0.99, mdecay=0.8)
    np.testing.assert_array_equal(np.stack([d2, v2, m2]), g["opt_adam_2"])
    dc, vc, mc = opt.adam(torch.from_numpy(g["opt_adam_gc"]))
    np.testing.assert_allclose(dc.numpy(), g["opt_adam_dc"], rtol=1e-6)
    np.testing.assert_allclose(vc.numpy(), g["opt_adam_vc"], rtol=1e-6)
    assert vc.dtype == torch.float32
    a = opt.momentum(x, None, None)
    b = opt.momentum(x + 1, None, a[2], mdecay=0.7)
    assert a[1] is None and a[0] is a[2]
    np.testing.assert_array_equal(np.stack([a[0], b[0]]), g["opt_momentum"])
    xs, ys = g["opt_fit_xy"]
    np.testing.assert_allclose(opt.fit_line_least_squares(y=ys, x=xs),
                               g["opt_fit"], rtol=1e-13)

    class O:
        pass

    got = []
    for costs, window in [([5, 4, 3, 2, 1, 0.5], 3), ([1, 2, 3, 4, 5, 6], 3),
                          ([3, 3, 3, 3], 2),
                          ([[3, 1], [2, 2], [1, 1], [2, 2.5]], 4), ([1, 1], 4),
                          ([5, 4, 3, 3.5, 4, 4.5, 5], 4),
                          ([5, 4, 3, 3.5, 4, 4.5, 5, 5.5], 4),
                          ([5, 4, 3, 3.5, 4, 4.5, 5, 5.5, 6], 4),
                          ([1, 2, 3], 0)]:
        o = O()
        o.costs, o.convergence_window = costs, window
        got.append(opt.is_converged(o))
    np.testing.assert_array_equal(got, g["opt_converged"])


# ------------------------------------------------------- probe initialisers
def test_probe_initialisers_equal_the_reference(g):
    import torch
    import tike_amd.ptycho as tp
    import tike_amd.random
    base = g["probe_base"]
    np.random.seed(31)
    modes = tp.add_modes_random_phase(base, 5)
    assert modes.dtype == np.complex64
    np.testing.assert_allclose(modes, g["probe_random_phase"], rtol=1e-6,
                               atol=1e-6)
    np.testing.assert_array_equal(tp.add_modes_random_phase(base, 1),
                                  g["probe_random_phase_fewer"])
    scaled = g["probe_random_phase"].copy()
    assert tp.adjust_probe_power(scaled) is scaled  # in place
    np.testing.assert_allclose(scaled, g["probe_adjust"], rtol=1e-6)
    np.testing.assert_allclose(
        tp.adjust_probe_power(base.copy(), power=np.array([1.0, 0.3])),
        g["probe_adjust_given"], rtol=1e-6)
    scan = g["probe_scan"]
    np.random.seed(32)
    tike_amd.random.randomizer_np = np.random.default_rng(33)
    ep, ew = tp.init_varying_probe(scan[0], base, 3, 1)
    assert ep.dtype == np.complex64 and ew.dtype == np.float32
    np.testing.assert_allclose(ep, g["probe_init_eigen"], rtol=1e-6)
    np.testing.assert_array_equal(ew, g["probe_init_weights"])
    ep1, ew1 = tp.init_varying_probe(scan[0], base, 1, 2)
    assert ep1 is None
    np.testing.assert_array_equal(ew1, g["probe_init_weights_1"])
    assert np.random.random_sample() == float(g["probe_init_after"])
    assert tp.init_varying_probe(scan[0], base, 0) == (None, None)
    with pytest.raises(ValueError, match="probes_with_modes"):
        tp.init_varying_probe(scan[0], base, 2, 3)
    np.random.seed(34)
    np.testing.assert_allclose(
        tp.simulate_varying_weights(scan, g["probe_init_eigen"]),
        g["probe_sim_weights"], rtol=1e-9, atol=1e-12)
    from tike_amd.ptycho.probe import (
        finite_probe_support, rescale_probe_using_fixed_intensity_photons)
    t = torch.from_numpy(base)
    np.testing.assert_allclose(
        finite_probe_support(t, radius=0.35, degree=2.5, p=0.7).numpy(),
        g["probe_support"], rtol=2e-5, atol=1e-7)
    assert finite_probe_support(t, p=0) == 0.0
    np.testing.assert_allclose(
        rescale_probe_using_fixed_intensity_photons(t, 1e4).numpy(),
        g["probe_photons"], rtol=1e-5)
    np.testing.assert_allclose(
        rescale_probe_using_fixed_intensity_photons(
            t, 1e4, np.array([0.8, 0.2], np.float32)).numpy(),
        g["probe_photons_split"], rtol=1e-5)
    np.testing.assert_array_equal(tp.gaussian(16, rin=0.6, rout=0.9),
                                  g["probe_gaussian_16"])
    np.testing.assert_array_equal(tp.gaussian(33), g["probe_gaussian_33"])


# ---------------------------------------------------- affine position model
def test_affine_transform_decomposition_equals_the_reference(g):
    from tike_amd.ptycho.position import AffineTransform
    for i in range(int(g["affine_cases"])):
        M = g[f"affine_in_{i}"]
        t = AffineTransform.fromarray(M.copy())
        tol = 1e-5 if M.dtype == np.float32 else 1e-12
        np.testing.assert_allclose(t.astuple(), g[f"affine_tuple_{i}"],
                                   rtol=tol, atol=tol, err_msg=str(i))
        got = t.asarray3()
        assert got.dtype == np.float32 and got.shape == (3, 2)
        np.testing.assert_allclose(got, g[f"affine_array_{i}"], rtol=2e-6,
                                   atol=1e-6)


def test_global_transformation_fits_equal_the_reference(g):
    import tike_amd.random
    from tike_amd.ptycho import position as P
    p0, p1 = g["affine_p0"], g["affine_p1"]
    t, res = P.estimate_global_transformation(p0, p1, None)
    np.testing.assert_allclose(t.astuple() + (res,), g["affine_fit"],
                               rtol=2e-4, atol=2e-4)
    t, res = P.estimate_global_transformation(p0, p1, g["affine_fit_weights"])
    np.testing.assert_allclose(t.astuple() + (res,), g["affine_fit_weighted"],
                               rtol=2e-4, atol=2e-4)
    t, res = P.estimate_global_transformation(
        np.stack([np.arange(5.0), np.arange(5.0)], 1), np.ones((5, 2)), None)
    np.testing.assert_allclose(t.astuple() + (res,), g["affine_fit_colinear"],
                               rtol=1e-12)
    tike_amd.random.randomizer_np = np.random.default_rng(35)
    t, fit = P.estimate_global_transformation_ransac(p0, p1)
    np.testing.assert_allclose(t.astuple() + (fit,), g["affine_ransac"],
                               rtol=2e-4, atol=2e-4)
    t, fit = P.estimate_global_transformation_ransac(p0, p1, max_error=1e-3)
    np.testing.assert_allclose(t.astuple() + (fit,), g["affine_ransac_none"])
    assert tike_amd.random.randomizer_np.random() == float(
        g["affine_ransac_next"])
