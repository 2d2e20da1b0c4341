"""Does a rising cost over many epochs belong to the algorithm or to the HIP
path?  The product and the CPU oracle (pinned to the reference) run the same
seeded lstsq_grad problem for `epochs` epochs; both cost histories are
printed side by side.  `gpurun -- python tools/soak_vs_oracle.py [epochs]`
(test infrastructure: imports oracle/, like tests/ and bench's cpu leg)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402
from oracle import solvers as osol  # noqa: E402
from test_solvers_gpu import _headline_problem  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
SMALL = ((128, 4, 96, 6, True), (256, 4, 30, 3, True), (128, 2, 96, 6, False))
for det, S, N, num_batch, eigen in (
        SMALL if os.environ.get("SOAK_SMALL", "1") == "1" else ()):
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=det + S, eigen=eigen)
    psi0 = np.full_like(psi_true, 0.5)
    batches = np.array_split(np.arange(N), num_batch)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        eigen_probe=None if ep is None else ep.copy(),
        eigen_weights=None if ew is None else ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=num_batch,
                                          num_iter=epochs,
                                          batch_method="compact"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=batches) as ctx:
        ctx.iterate(epochs)
        got = ctx.get_result()
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=None if ep is None else ep.copy(),
                 eigen_weights=None if ew is None else ew.copy())
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, epochs, detector_shape=det,
                         batch_method="compact", force_orthogonality=True,
                         rng=np.random.default_rng(11))
    a = np.array([c[0] for c in got.algorithm_options.costs])
    b = np.array([np.ravel(c)[0] for c in state["costs"]])
    print(f"det {det} S {S} N {N} batches {num_batch} eigen {eigen}")
    print("  hip   : " + " ".join(f"{c:.4e}" for c in a))
    print("  oracle: " + " ".join(f"{c:.4e}" for c in b), flush=True)


def bench_c3(N, num_batch, S=8, det=256, eigen="init", jitter=True):
    """bench.py's c3 problem (its synthetic object / probe, the reference's
    init_varying_probe) at N positions: the product and the oracle."""
    import bench
    import tike_amd._arrays as A
    p = bench.synthetic(N, S, det, 0, N)
    np.random.seed(1234)
    tike_amd.random.randomizer_np = np.random.default_rng(4321)
    ep, ew = tp.init_varying_probe(p["scan"], p["probe"], num_eigen_probes=2,
                                   probes_with_modes=1)
    if eigen == "none":
        ep = ew = None
    elif eigen == "large":  # weights that matter, as tests/_headline_problem
        ew[:, 1, 0] = 0.05 * np.random.default_rng(5).standard_normal(
            N).astype(np.float32)
    data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
    psi0 = np.full_like(p["psi"], 0.5 + 0j)
    batches = np.array_split(np.arange(N), num_batch)
    cp = lambda x: None if x is None else x.copy()
    params = tp.PtychoParameters(
        probe=p["probe"].copy(), psi=psi0.copy(), scan=p["scan"].copy(),
        eigen_probe=cp(ep), eigen_weights=cp(ew),
        algorithm_options=tp.LstsqOptions(num_batch=num_batch,
                                          batch_method=bench.BATCH_RULE),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions())
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    with tp.Reconstruction(A.to_device(data, np.float32), params,
                           presharded=True, order=np.arange(N),
                           batches=batches) as ctx:
        ctx.iterate(epochs)
        got = ctx.get_result()
    state = dict(psi=psi0.copy(), probe=p["probe"].copy(),
                 scan=p["scan"].copy(), costs=[], eigen_probe=cp(ep),
                 eigen_weights=cp(ew))
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, epochs, detector_shape=det,
                         batch_method=bench.BATCH_RULE,
                         force_orthogonality=True,
                         rng=np.random.default_rng(11))
    a = np.array([c[0] for c in got.algorithm_options.costs])
    b = np.array([np.ravel(c)[0] for c in state["costs"]])
    print(f"bench c3 problem, N {N}, batches {num_batch}, update rule "
          f"{bench.BATCH_RULE}, eigen probe: {eigen}")
    print("  hip   : " + " ".join(f"{c:.4e}" for c in a))
    print("  oracle: " + " ".join(f"{c:.4e}" for c in b), flush=True)


if os.environ.get("SOAK_BENCH_C3", "1") == "1":
    for eigen in os.environ.get("SOAK_KINDS", "none,large,init").split(","):
        bench_c3(int(os.environ.get("SOAK_N", "160")), 10, eigen=eigen)
