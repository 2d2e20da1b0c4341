"""End-to-end time of the one-call API on the headline problem
(`gpurun -- python tools/soak_api.py`): tike_amd.ptycho.reconstruct(data,
parameters) with host arrays in and host arrays out, as a tike user calls it
-- where does the time outside the epochs go?"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402

N, S, det, epochs = 10000, 8, 256, 10
p = bench.synthetic(N, S, det, 0, N)
np.random.seed(1234)
tike_amd.random.randomizer_np = np.random.default_rng(4321)
ep, ew = tp.init_varying_probe(p["scan"], p["probe"], num_eigen_probes=2,
                               probes_with_modes=1)
data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
torch.cuda.synchronize()


def params():
    return tp.PtychoParameters(
        probe=p["probe"].copy(), psi=np.full_like(p["psi"], 0.5 + 0j),
        scan=p["scan"].copy(), eigen_probe=ep.copy(), eigen_weights=ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=10, num_iter=epochs),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions())


for attempt in range(2):
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    result = tp.reconstruct(data, params())
    pr.disable()
    dt = time.perf_counter() - t0
    print(f"call {attempt}: reconstruct() of {N} positions, {epochs} epochs: "
          f"{dt:.2f} s wall; epochs themselves "
          f"{sum(result.algorithm_options.times):.2f} s", flush=True)
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
