"""The reference's small unit tests of the host-side pieces on the path,
restated against tike_amd with the reference's shapes, constants and
assertions (tests/ptycho/test_probe.py:68-135, tests/ptycho/test_position.py:
22-135, tests/test_opt.py, tests/test_linalg.py, tests/ptycho/test_ptycho.py:
80-107) -- a maintainer who switches the import should find them all green.
NumPy arrays on the CPU; the same functions take device tensors inside the
solver."""
import numpy as np
import pytest
import torch

import tike_amd.linalg
import tike_amd.opt
import tike_amd.precision
import tike_amd.ptycho
import tike_amd.ptycho.probe
import tike_amd.random
from tike_amd.ptycho import AffineTransform, PositionOptions


# ------------------------------------------------ tests/ptycho/test_probe.py
@pytest.mark.parametrize("p,e,s,w,vary", [
    (0, 0, 1, 16, False),  # mono probe
    (0, 0, 7, 16, False),  # multi probe
    (31, 0, 1, 16, True),  # mono probe, varying
    (31, 0, 7, 16, True),  # multi probe, varying
    (31, 3, 1, 16, True),  # mono probe, varying, eigen probes
    (31, 3, 7, 16, True),  # multi probe, varying, eigen probes
])
def test_get_varying_probe_shapes(p, e, s, w, vary):
    unique = tike_amd.ptycho.probe.get_varying_probe(
        shared_probe=np.random.rand(1, 1, s, w, w),
        eigen_probe=np.random.rand(1, e, s, w, w) if e > 0 else None,
        weights=np.ones((p, e + 1, s)) if vary else None,
    )
    assert unique.shape == (p if vary else 1, 1, s, w, w)


@pytest.mark.parametrize("p,e,s,w,v", [
    (31, 0, 2, 16, 0), (31, 0, 2, 16, 1),  # no varying probe
    (31, 1, 2, 16, 1),  # one varying probe
    (31, 1, 3, 16, 3),  # many varying probes
    (31, 7, 3, 16, 3),  # ... with a basis of several eigen probes
    (31, 7, 3, 16, 1),
])
def test_init_varying_probe_shapes(p, e, s, w, v):
    eigen_probe, weights = tike_amd.ptycho.probe.init_varying_probe(
        scan=np.random.rand(p, 2),
        shared_probe=np.random.rand(1, 1, s, w, w),
        num_eigen_probes=e,
        probes_with_modes=v,
    )
    if e < 2:
        assert eigen_probe is None
    else:
        assert eigen_probe.shape == (1, e - 1, v, w, w)
    if e < 1:
        assert weights is None
    else:
        assert weights.shape == (p, e, s)


def test_probe_support():
    """Finite probe support penalty function is within expected bounds."""
    penalty = tike_amd.ptycho.probe.finite_probe_support(
        probe=torch.zeros((101, 101)),  # must be odd shaped for min to be 0
        radius=0.5 * 0.7,
        degree=2.5,  # must have degree >= 1 for upper bound to be p
        p=2.345,
    )
    assert np.around(float(penalty.min()), 3) == 0.000
    assert np.around(float(penalty.max()), 3) == 2.345


# ---------------------------------------------- tests/ptycho/test_position.py
def test_position_join(N=245, num_batch=11):
    scan = np.random.rand(N, 2)
    indices = np.arange(N)
    np.random.shuffle(indices)
    batches = np.array_split(indices, num_batch)
    reorder = np.argsort(np.concatenate(batches))
    opts = PositionOptions(scan, use_adaptive_moment=True)
    optsb = [opts.split(b) for b in batches]
    joined = PositionOptions.join(optsb, reorder=reorder)
    assert joined is not None
    np.testing.assert_array_equal(joined.initial_scan, opts.initial_scan)
    np.testing.assert_array_equal(joined._momentum, opts._momentum)


def test_affine_translate():
    T = AffineTransform(t0=11, t1=-5)
    positions1 = np.array([[0, 0], [0, 1], [1, 0], [-1, -1]])
    np.testing.assert_equal(T(positions1),
                            [[11, -5], [11, -4], [12, -5], [10, -6]])


def test_affine_scale():
    T = AffineTransform(scale0=11, scale1=0.5)
    positions1 = np.array([[0, 0], [0, 1], [1, 0], [-1, -1]])
    np.testing.assert_equal(T(positions1),
                            [[0, 0], [0, 0.5], [11, 0], [-11, -0.5]])


def test_affine_estimation_recovers_the_transform(N=213):
    """TestAffineEstimation.test_fit_linear / test_fit_weighted: the fitted
    transform maps the positions like the true one (the reference only plots;
    here the agreement is asserted)."""
    rng = np.random.default_rng(0)
    truth = [3.4567, 5.4321, 0.9876, 1.2345, 2.3456, -4.5678]
    T = AffineTransform(*truth)
    error = rng.normal(size=(N, 2), scale=0.1)
    positions0 = rng.random((N, 2)) - 0.5
    positions1 = T(positions0) + error
    weights = 1 / (1 + np.square(error).sum(axis=-1))
    linear = tike_amd.linalg.lstsq(
        a=np.pad(positions0, ((0, 0), (0, 1)), constant_values=1),
        b=positions1, weights=weights)
    result = AffineTransform.fromarray(linear)
    np.testing.assert_allclose(result(positions0), T(positions0), atol=0.05)
    fitted, _ = tike_amd.ptycho.position.estimate_global_transformation(
        positions0, positions1, weights)
    np.testing.assert_allclose(fitted(positions0), T(positions0), atol=0.05)
    np.testing.assert_allclose(fitted.astuple()[:3], truth[:3], rtol=0.05)


# ----------------------------------------------------------- tests/test_opt.py
class AlgorithmOptionsStub:

    def __init__(self, costs, window=5) -> None:
        self.costs = costs
        self.convergence_window = window


def test_is_converged():
    assert tike_amd.opt.is_converged(
        AlgorithmOptionsStub((np.arange(11) / 1234).tolist(), 5))
    assert tike_amd.opt.is_converged(
        AlgorithmOptionsStub((np.zeros(11) / 1234).tolist(), 5))
    assert not tike_amd.opt.is_converged(
        AlgorithmOptionsStub((-np.arange(11) / 1234).tolist(), 5))


def test_fit_line():
    result = np.around(
        tike_amd.opt.fit_line_least_squares(
            y=np.asarray([0, np.log(0.9573), np.log(0.8386)]),
            x=np.asarray([0, 1, 2]),
        ), 4)
    np.testing.assert_array_equal((-0.0880, 0.0148), result)


# -------------------------------------------------------- tests/test_linalg.py
def test_norm():
    a = tike_amd.random.numpy_complex(5)
    np.testing.assert_allclose(tike_amd.precision.floating(1.0),
                               np.linalg.norm(a / np.linalg.norm(a)),
                               rtol=1e-6)
    np.testing.assert_allclose(np.sqrt(tike_amd.linalg.inner(a, a)),
                               np.linalg.norm(a), rtol=1e-6)


def test_lstsq():
    a = tike_amd.random.numpy_complex(5, 1, 4, 3, 3)
    x = tike_amd.random.numpy_complex(5, 1, 4, 3, 1)
    w = np.random.random(size=(5, 1, 4, 3)).astype(tike_amd.precision.floating)
    b = a @ x
    x1 = tike_amd.linalg.lstsq(a, b, weights=w)
    np.testing.assert_allclose(x1, x, rtol=1e-2, atol=0)


def test_projection():
    a = tike_amd.random.numpy_complex(5)
    b = tike_amd.random.numpy_complex(5)
    pab = tike_amd.linalg.projection(a, b)
    pba = tike_amd.linalg.projection(b, a)
    assert abs(tike_amd.linalg.inner(a - pab, b)) < 1e-6
    assert abs(tike_amd.linalg.inner(a, b - pba)) < 1e-6


class TestOrthogonal:

    def setup_method(self):
        self.x = tike_amd.random.numpy_complex(1, 4, 3, 3)

    def test_gram_schmidt_single_vector(self):
        with pytest.raises(ValueError):
            tike_amd.linalg.orthogonalize_gs(self.x, axis=(0, 1, 2, 3))

    def test_gram_schmidt_single_axis(self):
        y = tike_amd.linalg.orthogonalize_gs(self.x)
        assert self.x.shape == y.shape

    def test_gram_schmidt_multi_axis(self):
        y = tike_amd.linalg.orthogonalize_gs(self.x, axis=(1, -1))
        assert self.x.shape == y.shape

    def test_gram_schmidt_orthogonal(self, axis=(-2, -1)):
        u = tike_amd.linalg.orthogonalize_gs(self.x, axis=axis)
        for i in range(4):
            for j in range(i + 1, 4):
                error = abs(tike_amd.linalg.inner(u[:, i:i + 1], u[:, j:j + 1],
                                                  axis=axis))
                assert np.all(error < 1e-6)


# --------------------------------------- tests/ptycho/test_ptycho.py:92-107
def test_check_allowed_positions():
    psi = np.empty((1, 4, 9))
    probe = np.empty((8, 2, 2))
    scan = np.array([[1, 1], [1, 6.9], [1.1, 1], [1.9, 5.5]])
    tike_amd.ptycho.check_allowed_positions(scan, psi, probe.shape)
    for scan in np.array([[1, 7], [1, 0.9], [0.9, 1], [1, 0]]):
        with pytest.raises(ValueError):
            tike_amd.ptycho.check_allowed_positions(scan, psi, probe.shape)
        with pytest.raises(ValueError, match="Scan positions must be >= 1"):
            tike_amd.ptycho.check_allowed_positions(scan[None], psi,
                                                    probe.shape)


def test_get_padded_object():
    probe = np.empty((8, 3, 4))
    scan = (np.random.rand(15, 2) * 100) - 50
    psi, scan = tike_amd.ptycho.get_padded_object(scan, probe)
    tike_amd.ptycho.check_allowed_positions(scan, psi,
                                            probe_shape=probe.shape)


# ------------------------------------------------------ tests/test_random.py
import scipy.stats  # noqa: E402

import tike_amd.cluster  # noqa: E402


@pytest.mark.parametrize("name", ["wobbly_center",
                                  "wobbly_center_random_bootstrap", "compact"])
class TestCluster:
    """The reference's ClusterTests (tests/test_random.py:12-139) for its
    three minibatch selectors: a normally distributed 3-D float64 population
    of 500 (the general, NumPy-expression path of tike_amd.cluster; scan
    positions -- float32, 2-D -- take the library's loops and are pinned to
    the reference's labels in test_host_golden_cpu.py)."""
    num_pop, num_cluster = 500, 10

    @pytest.fixture()
    def population(self):
        rng = tike_amd.random.randomizer_np
        population = np.concatenate([
            rng.normal(m, s, (self.num_pop, 1))
            for m, s in zip([-np.sqrt(2), np.pi, np.e], [0.5, 3, 7])
        ], axis=1)
        rng.shuffle(population, axis=0)
        return population

    def test_no_clusters(self, name, population):
        method = getattr(tike_amd.cluster, name)
        for count in (0, -1, 0xFFFFFF):
            with pytest.raises(ValueError):
                method(population, count)

    def test_one_cluster(self, name, population):
        samples = getattr(tike_amd.cluster, name)(population, 1)
        assert len(samples) == 1
        np.testing.assert_array_equal(samples[0].flatten(),
                                      np.arange(self.num_pop))

    def test_more_clusters_than_population(self, name, population):
        samples = getattr(tike_amd.cluster, name)(population, self.num_pop + 1)
        assert len(samples) == self.num_pop + 1
        assert len(samples[-1]) == 0

    def test_max_clusters(self, name, population):
        samples = getattr(tike_amd.cluster, name)(population, self.num_pop)
        assert len(samples) == self.num_pop
        np.testing.assert_array_equal(np.array(samples).flatten(),
                                      np.arange(self.num_pop))

    def test_complete_set(self, name, population):
        samples = getattr(tike_amd.cluster, name)(population, self.num_cluster)
        np.testing.assert_array_equal(np.sort(np.concatenate(samples)),
                                      np.arange(self.num_pop))

    def test_some_clusters_are_singleton(self, name, population):
        samples = getattr(tike_amd.cluster, name)(population, self.num_pop - 1)
        assert len(samples[0]) == 2
        assert len(samples[1]) == 1

    def test_clusters_sorted_by_size(self, name, population, num_cluster=7):
        samples = getattr(tike_amd.cluster, name)(population, num_cluster)
        small, remainder = divmod(self.num_pop, num_cluster)
        truth = [small + 1] * remainder + [small] * (num_cluster - remainder)
        assert [len(batch) for batch in samples] == truth

    def test_same_mean(self, name, population):
        """The heterogeneous selectors give samples whose means an ANOVA
        cannot tell apart, better than a random split does."""
        if name == "compact":
            pytest.skip("compact clusters are homogeneous by design")

        def p_value(indices):
            return scipy.stats.f_oneway(*[population[i] for i in indices])[1]

        p0 = p_value(tike_amd.cluster.wobbly_center(population,
                                                    self.num_cluster))
        np.random.seed(0)
        p1 = p_value(tike_amd.opt.batch_indicies(self.num_pop,
                                                 self.num_cluster))
        assert np.all(p0 > p1)
