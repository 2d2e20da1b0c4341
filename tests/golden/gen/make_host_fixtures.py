#!/usr/bin/env python3
"""Golden vectors for the host-side helpers (build container only):

    python tests/golden/gen/make_host_fixtures.py

`host_helpers.npz`: outputs of the REFERENCE's own host functions (run under
the NumPy-backed CuPy stand-in) on seeded inputs -- batch clustering
(`tike.cluster.compact`, `wobbly_center`), probe initialisers
(`tike.ptycho.probe`), the affine position model (`tike.ptycho.position`),
`tike.opt` and `tike.linalg` helpers.  Data only; tests/test_host_golden_cpu.py
replays the same inputs (and the same generator seeds) through tike_amd.
"""
import os
import sys

sys.dont_write_bytecode = True  # never write into /root/reference

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "cupy_shim"))
sys.path.insert(0, os.path.join(REF, "src"))
os.environ.setdefault("TIKE_REF_EMU_LIB", "")

import cupy as cp  # noqa: E402,F401  (the shim)
import tike.cluster as ref_cluster  # noqa: E402
import tike.linalg as ref_linalg  # noqa: E402
import tike.opt as ref_opt  # noqa: E402
import tike.ptycho.position as ref_position  # noqa: E402
import tike.ptycho.probe as ref_probe  # noqa: E402
import tike.random  # noqa: E402

out = {}


def labels_of(groups, n):
    lab = np.full(n, -1, dtype=np.int64)
    for c, g in enumerate(groups):
        lab[np.asarray(g)] = c
    return lab


# ---- clustering: raster scans with jitter (ties between equal distances are
# the interesting part), float32 as `scan` is, several sizes and seeds
cases = []
for i, (n, k, seed) in enumerate([(97, 5, 0), (256, 4, 1), (400, 10, 2),
                                  (61, 7, 3), (1000, 10, 4), (30, 30, 5),
                                  (12, 1, 6)]):
    rng = np.random.default_rng(100 + seed)
    side = int(np.ceil(np.sqrt(n)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:n]
    jitter = rng.random((n, 2)) if seed % 2 else np.zeros((n, 2))
    pop = (1 + 8.0 * ij + jitter).astype(np.float32)
    rng.shuffle(pop)
    out[f"cluster_pop_{i}"] = pop
    out[f"cluster_k_{i}"] = np.int64(k)
    out[f"cluster_seed_{i}"] = np.int64(seed)
    np.random.seed(seed)
    out[f"cluster_compact_{i}"] = labels_of(ref_cluster.compact(pop, k), n)
    out[f"cluster_compact_next_{i}"] = np.float64(np.random.random_sample())
    out[f"cluster_wobbly_{i}"] = labels_of(ref_cluster.wobbly_center(pop, k), n)
    state = np.random.get_state()  # (added later: leaves the draws below alone)
    np.random.seed(1000 + seed)
    out[f"cluster_bootstrap_{i}"] = labels_of(
        ref_cluster.wobbly_center_random_bootstrap(pop, k), n)
    out[f"cluster_bootstrap_half_{i}"] = labels_of(
        ref_cluster.wobbly_center_random_bootstrap(pop, k, boot_fraction=0.5), n)
    out[f"cluster_bootstrap_next_{i}"] = np.float64(np.random.random_sample())
    out[f"cluster_stripes_{i}"] = labels_of(
        ref_cluster.stripes_equal_count(pop, k, dim=i % 2), n)
    np.random.set_state(state)
out["cluster_cases"] = np.int64(i + 1)

# ---- tike.opt: a small complex least-squares problem |A x - b|^2
rng = np.random.default_rng(11)
A_ = (rng.standard_normal((12, 6)) + 1j * rng.standard_normal((12, 6))).astype(np.complex64)
b_ = (rng.standard_normal(12) + 1j * rng.standard_normal(12)).astype(np.complex64)
x0 = np.zeros(6, np.complex64)
out["opt_A"], out["opt_b"] = A_, b_


def cost_fn(x):
    r = A_ @ x - b_
    return float(np.real(np.vdot(r, r)))


def grad_fn(x):
    return [A_.conj().T @ (A_ @ x - b_)]


for tag, kw in dict(full=dict(num_iter=5, step_length=1.0),
                    partial=dict(num_iter=6, step_length=0.5, num_search=2),
                    tiny=dict(num_iter=2, step_length=1e-3)).items():
    x, c = ref_opt.conjugate_gradient(
        np, x0.copy(), cost_fn, grad_fn,
        update_multi=lambda x, s, d: x + s * d[0], **kw)
    out[f"opt_cg_x_{tag}"], out[f"opt_cg_cost_{tag}"] = x, np.float64(c)
import warnings  # noqa: E402
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    s_, c_, x_ = ref_opt.line_search(cost_fn, x0.copy(), [-grad_fn(x0)[0]],
                                     lambda x, s, d: x - s * d[0])  # uphill
out["opt_ls_fail"] = np.array([s_, c_])
s_, c_, x_ = ref_opt.line_search(cost_fn, x0.copy(), [-grad_fn(x0)[0]],
                                 lambda x, s, d: x + s * d[0], step_length=4.0)
out["opt_ls_ok"], out["opt_ls_ok_x"] = np.array([s_, c_]), x_
g1 = [rng.standard_normal(9) + 1j * rng.standard_normal(9)]
g0 = [rng.standard_normal(9) + 1j * rng.standard_normal(9)]
d0 = ref_opt.direction_dy(np, g0)
out["opt_dy_g1"], out["opt_dy_g0"] = g1[0], g0[0]
out["opt_dy_first"] = d0[0]
out["opt_dy_next"] = ref_opt.direction_dy(np, g1, g0, d0)[0]
g = (rng.standard_normal((7, 2))).astype(np.float32)
d1, v1, m1 = ref_opt.adam(g)
d2, v2, m2 = ref_opt.adam(2 * g - 1, v1, m1, vdecay=0.99, mdecay=0.8)
out["opt_adam_g"] = g
out["opt_adam_1"] = np.stack([d1, v1, m1])
out["opt_adam_2"] = np.stack([d2, v2, m2])
gc = (rng.standard_normal(5) + 1j * rng.standard_normal(5)).astype(np.complex64)
dc, vc, mc = ref_opt.adam(gc)
out["opt_adam_gc"], out["opt_adam_dc"], out["opt_adam_vc"] = gc, dc, vc
mm1 = ref_opt.momentum(g, None, None)
mm2 = ref_opt.momentum(g + 1, None, mm1[2], mdecay=0.7)
out["opt_momentum"] = np.stack([mm1[0], mm2[0]])
xs = np.array([0.0, 1, 2, 3, 4])
ys = np.array([2.0, 2.9, 4.2, 5.1, 5.8])
out["opt_fit_xy"] = np.stack([xs, ys])
out["opt_fit"] = np.array(ref_opt.fit_line_least_squares(y=ys, x=xs))


class _O:
    pass


conv = []
for costs, window in [([5, 4, 3, 2, 1, 0.5], 3), ([1, 2, 3, 4, 5, 6], 3),
                      ([3, 3, 3, 3], 2), ([[3, 1], [2, 2], [1, 1], [2, 2.5]], 4),
                      ([1, 1], 4), ([5, 4, 3, 3.5, 4, 4.5, 5], 4),
                      ([5, 4, 3, 3.5, 4, 4.5, 5, 5.5], 4),
                      ([5, 4, 3, 3.5, 4, 4.5, 5, 5.5, 6], 4), ([1, 2, 3], 0)]:
    o = _O()
    o.costs, o.convergence_window = costs, window
    conv.append(bool(ref_opt.is_converged(o)))
out["opt_converged"] = np.array(conv)

# ---- probe initialisers (legacy generator + tike.random.randomizer_np)
rng = np.random.default_rng(21)
base = (rng.standard_normal((1, 1, 2, 12, 12)) +
        1j * rng.standard_normal((1, 1, 2, 12, 12))).astype(np.complex64)
out["probe_base"] = base
np.random.seed(31)
out["probe_random_phase"] = ref_probe.add_modes_random_phase(base, 5)
out["probe_random_phase_fewer"] = ref_probe.add_modes_random_phase(base, 1)
out["probe_adjust"] = ref_probe.adjust_probe_power(
    out["probe_random_phase"].copy())
out["probe_adjust_given"] = ref_probe.adjust_probe_power(
    base.copy(), power=np.array([1.0, 0.3]))
scan_ = (rng.random((1, 9, 2)) * 20).astype(np.float32)
out["probe_scan"] = scan_
np.random.seed(32)
tike.random.randomizer_np = np.random.default_rng(33)
ep, ew = ref_probe.init_varying_probe(scan_[0], base, 3, 1)
out["probe_init_eigen"], out["probe_init_weights"] = ep, ew
ep1, ew1 = ref_probe.init_varying_probe(scan_[0], base, 1, 2)
assert ep1 is None
out["probe_init_weights_1"] = ew1
out["probe_init_after"] = np.float64(np.random.random_sample())
np.random.seed(34)
out["probe_sim_weights"] = ref_probe.simulate_varying_weights(scan_, ep)
out["probe_support"] = np.asarray(ref_probe.finite_probe_support(
    base, radius=0.35, degree=2.5, p=0.7))
out["probe_photons"] = np.asarray(
    ref_probe.rescale_probe_using_fixed_intensity_photons(base, 1e4))
out["probe_photons_split"] = np.asarray(
    ref_probe.rescale_probe_using_fixed_intensity_photons(
        base, 1e4, np.array([0.8, 0.2], np.float32)))
out["probe_gaussian_16"] = ref_probe.gaussian(16, rin=0.6, rout=0.9)
out["probe_gaussian_33"] = ref_probe.gaussian(33)

# ---- affine position model
mats = []
for k in range(6):
    M = rng.standard_normal((3, 2)) * (1, 1)
    M[:2] += np.eye(2) * (1.5 if k % 2 else -0.7)
    mats.append(M)
mats.append(np.array([[0.0, 0.0], [1.0, 2.0], [3.0, 4.0]]))
mats.append(np.array([[1.0, 2.0], [2.0, 4.0]]))
mats = [m.astype(np.float32) if i % 2 else m for i, m in enumerate(mats)]
for i, M in enumerate(mats):
    t = ref_position.AffineTransform.fromarray(M.copy())
    out[f"affine_in_{i}"] = M
    out[f"affine_tuple_{i}"] = np.array(t.astuple())
    out[f"affine_array_{i}"] = t.asarray3()
out["affine_cases"] = np.int64(len(mats))
p0 = (rng.random((60, 2)) * 100).astype(np.float32)
true = ref_position.AffineTransform(1.02, 0.97, 0.03, 0.02, 1.5, -2.0)
p1 = (true(p0) + rng.normal(0, 0.3, p0.shape)).astype(np.float32)
p1[::7] += 80  # outliers
out["affine_p0"], out["affine_p1"] = p0, p1
t, res = ref_position.estimate_global_transformation(p0, p1, None)
out["affine_fit"] = np.array(t.astuple() + (res,))
wts = rng.random(60).astype(np.float32)
t, res = ref_position.estimate_global_transformation(p0, p1, wts)
out["affine_fit_weights"], out["affine_fit_weighted"] = wts, np.array(
    t.astuple() + (res,))
t, res = ref_position.estimate_global_transformation(
    np.stack([np.arange(5.0), np.arange(5.0)], 1), np.ones((5, 2)), None)
out["affine_fit_colinear"] = np.array(t.astuple() + (res,))
tike.random.randomizer_np = np.random.default_rng(35)
t, fit = ref_position.estimate_global_transformation_ransac(p0, p1)
out["affine_ransac"] = np.array(t.astuple() + (fit,))
t, fit = ref_position.estimate_global_transformation_ransac(
    p0, p1, max_error=1e-3)
out["affine_ransac_none"] = np.array(t.astuple() + (fit,))
out["affine_ransac_next"] = np.float64(tike.random.randomizer_np.random())

np.savez_compressed(os.path.join(OUT, "host_helpers.npz"), **out)
print("wrote host_helpers.npz:", len(out), "arrays")
