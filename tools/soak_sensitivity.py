"""Is bench.py's c3 problem a stable iteration?  The product runs it twice,
the second time from an object guess perturbed by 1e-6 (relative): if the two
cost histories part as fast as product and oracle do (tools/soak_vs_oracle.py),
the iteration amplifies rounding differences on this data and a per-epoch
comparison with the oracle says nothing beyond the first epochs.
`gpurun -- python tools/soak_sensitivity.py [epochs] [N]`"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 160
S, det, num_batch = 8, 256, 10
p = bench.synthetic(N, S, det, 0, N)
np.random.seed(1234)
tike_amd.random.randomizer_np = np.random.default_rng(4321)
ep, ew = tp.init_varying_probe(p["scan"], p["probe"], num_eigen_probes=2,
                               probes_with_modes=1)
data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
batches = np.array_split(np.arange(N), num_batch)
for eps in (0.0, 1e-6, 0.0):
    psi0 = np.full_like(p["psi"], 0.5 + 0j) * np.complex64(1 + eps)
    params = tp.PtychoParameters(
        probe=p["probe"].copy(), psi=psi0, scan=p["scan"].copy(),
        eigen_probe=ep.copy(), eigen_weights=ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=num_batch,
                                          batch_method="compact"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions())
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    with tp.Reconstruction(A.to_device(data, np.float32), params,
                           presharded=True, order=np.arange(N),
                           batches=batches) as ctx:
        ctx.iterate(epochs)
        got = ctx.get_result()
    print(f"psi0 * (1 + {eps:g}): " + " ".join(
        f"{c[0]:.4e}" for c in got.algorithm_options.costs), flush=True)
