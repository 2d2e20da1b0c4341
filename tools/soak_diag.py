"""Where do product and oracle part on bench.py's problem with eigen probes?
Relative difference of the epoch costs for variations of the problem.
`gpurun -- python tools/soak_diag.py`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402
from oracle import solvers as osol  # noqa: E402


def run(det, S, N, num_batch, presharded, epochs=3, orth=True, sort=True,
        weights="init"):
    p = bench.synthetic(N, S, det, 0, N)
    np.random.seed(1234)
    tike_amd.random.randomizer_np = np.random.default_rng(4321)
    ep, ew = tp.init_varying_probe(p["scan"], p["probe"], num_eigen_probes=2,
                                   probes_with_modes=1)
    if weights == "large":
        ew[:, 1, 0] = 0.05 * np.random.default_rng(5).standard_normal(
            N).astype(np.float32)
    data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
    psi0 = np.full_like(p["psi"], 0.5 + 0j)
    batches = np.array_split(np.arange(N), num_batch)
    params = tp.PtychoParameters(
        probe=p["probe"].copy(), psi=psi0.copy(), scan=p["scan"].copy(),
        eigen_probe=ep.copy(), eigen_weights=ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=num_batch,
                                          batch_method="wobbly_center"),
        probe_options=tp.ProbeOptions(force_orthogonality=orth),
        object_options=tp.ObjectOptions())
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    with tp.Reconstruction(A.to_device(data, np.float32), params,
                           presharded=presharded, order=np.arange(N),
                           batches=batches, spatial_sort=sort) as ctx:
        ctx.iterate(epochs)
        got = ctx.get_result()
    state = dict(psi=psi0.copy(), probe=p["probe"].copy(),
                 scan=p["scan"].copy(), costs=[], eigen_probe=ep.copy(),
                 eigen_weights=ew.copy())
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, epochs, detector_shape=det,
                         batch_method="wobbly_center",
                         force_orthogonality=orth,
                         rng=np.random.default_rng(11))
    a = np.array([c[0] for c in got.algorithm_options.costs])
    b = np.array([np.ravel(c)[0] for c in state["costs"]])
    rel = np.abs(a - b) / np.abs(b)
    dw = np.abs(got.eigen_weights - state["eigen_weights"]).max()
    print(f"det {det} S {S} N {N} batches {num_batch} presharded "
          f"{presharded} orth {orth} sort {sort} weights {weights}: "
          "cost rel diff " + " ".join(f"{r:.1e}" for r in rel)
          + f" | max |dw| {dw:.2e}", flush=True)


run(128, 8, 160, 10, True)
run(128, 8, 160, 10, False)
run(128, 4, 160, 10, True)
run(128, 8, 96, 6, True)
run(128, 8, 160, 10, True, orth=False)
run(128, 8, 160, 10, True, sort=False)
run(128, 8, 160, 10, True, weights="large")
run(128, 8, 160, 2, True)
