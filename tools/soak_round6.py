"""Round 6's routes over more epochs than the parity tests run: the product
and the CPU oracle (pinned to the reference) on the same seeded problems,
cost histories side by side and the distance of the final iterates.

    gpurun -- python tools/soak_round6.py [epochs=8]

(test infrastructure: imports oracle/, like tests/ and bench's cpu leg)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402
from oracle import operators as oops  # noqa: E402
from oracle import solvers as osol  # noqa: E402
from test_solvers_gpu import _headline_problem  # noqa: E402
from tike_amd.ptycho.solvers._plan import GradientPlan  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) /
                 max(np.linalg.norm(np.asarray(b)), 1e-30))


def report(tag, got, state, route):
    a = np.array([np.ravel(c)[0] for c in got.algorithm_options.costs])
    b = np.array([np.ravel(c)[0] for c in state["costs"]])
    print(f"{tag}  [{route}]")
    print("  hip   : " + " ".join(f"{c:.5e}" for c in a))
    print("  oracle: " + " ".join(f"{c:.5e}" for c in b))
    print(f"  max |cost/oracle - 1| = {np.max(np.abs(a / b - 1)):.2e}   "
          f"psi {rel(got.psi, state['psi']):.2e}   "
          f"probe {rel(got.probe, state['probe']):.2e}", flush=True)


# ---- lstsq_grad: mode groups, prime-factor (64^2 and 128^2 sub-tiles), LDS
# line engine, unfused kernels on the mixed-radix transforms
routes = []
real = GradientPlan.gradients


def spy(self, c, k):
    routes.append(self.route + (" + mode groups" if self.groups else ""))
    return real(self, c, k)


GradientPlan.gradients = spy
for tag, det, S, N, nb, eigen in (
        ("256^2 x 12 modes", 256, 12, 24, 3, True),
        ("128^2 x 10 modes", 128, 10, 40, 4, True),
        ("192^2 x 3 modes", 192, 3, 40, 4, True),
        ("384^2 x 2 modes", 384, 2, 16, 2, True),
        ("300^2 x 2 modes", 300, 2, 16, 2, True),
        ("300^2 x 2 modes, LDS line engine asked for", 300, 2, 16, 2, True),
        ("100^2 x 3 modes", 100, 3, 48, 4, True)):
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=det + S, eigen=eigen)
    psi0 = np.full_like(psi_true, 0.5)
    batches = np.array_split(np.arange(N), nb)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        eigen_probe=ep.copy(), eigen_weights=ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=nb, num_iter=epochs,
                                          batch_method="compact"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    del routes[:]
    from tike_amd.ptycho.solvers import lstsq as L
    saved = L.GENERAL_MIN_DETECTOR
    if "asked for" in tag:  # (nobody's default route since late round 6)
        L.GENERAL_MIN_DETECTOR = 256
    try:
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=batches) as ctx:
            ctx.iterate(epochs)
            got = ctx.get_result()
    finally:
        L.GENERAL_MIN_DETECTOR = saved
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=ep.copy(), eigen_weights=ew.copy())
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, epochs, detector_shape=det,
                         batch_method="compact", force_orthogonality=True,
                         rng=np.random.default_rng(11))
    report(f"lstsq_grad {tag}, {N} positions, {nb} minibatches, eigen probe",
           got, state, sorted(set(routes))[0])
GradientPlan.gradients = real

# ---- rpie on a two-slice object: the fused chain at 128^2 and 512^2, both
# noise models
phys = dict(wavelength=1e-10, fov=(2e-6, 2e-6), distance=1e-6)
for det, S, N, model in ((128, 3, 24, "gaussian"), (512, 2, 8, "gaussian"),
                         (128, 2, 24, "poisson"), (256, 4, 12, "poisson")):
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=3 * det + S, eigen=False)
    if model == "poisson":
        data = np.round(data * (20000.0 / data.max())).astype(np.float32)
    psi0 = np.repeat(np.full_like(psi_true, 0.5), 2, axis=0)
    psi0[1:] = 1.0
    batches = np.array_split(np.arange(N), 2)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        algorithm_options=tp.RpieOptions(num_batch=2, num_iter=epochs,
                                         batch_method="compact", alpha=1.0),
        probe_options=tp.ProbeOptions(
            force_orthogonality=True, probe_wavelength=phys["wavelength"],
            probe_FOV_lengths=phys["fov"]),
        object_options=tp.ObjectOptions(
            multislice_propagation_distance=phys["distance"]),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool),
            noise_model=model))
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=batches) as ctx:
        ctx.iterate(epochs)
        got = ctx.get_result()
    propagator = oops.fresnel_spectrum_propagator(
        (det, det), phys["fov"], phys["distance"], phys["wavelength"])
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=None, eigen_weights=None)
    state = osol.rescale_probe(state, data, det, propagator=propagator)
    state = osol.iterate(state, data, batches, epochs, detector_shape=det,
                         solver="rpie", alpha=1.0, batch_method="compact",
                         force_orthogonality=True, propagator=propagator,
                         noise_model=model)
    report(f"rpie, two slices, {det}^2 x {S} modes, {model}", got, state,
           "fused multislice chain")
