"""reconstruct(data, parameters, num_gpu=2) at the headline size from a plain
process (`gpurun -- python tools/soak_spawn.py`; the two ranks share the test
box's GPU over gloo, TIKE_AMD_OVERSUBSCRIBE=1): wall time of the call against
the epochs it ran, and the result against the one-rank call."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

if __name__ == "__main__":
    os.environ["TIKE_AMD_OVERSUBSCRIBE"] = "1"
    import bench
    import tike_amd.ptycho as tp
    import tike_amd.random
    N, S, det, epochs = 4000, 8, 256, 4
    p = bench.synthetic(N, S, det, 0, N)
    data = tp.simulate(det, p["probe"], p["scan"], p["psi"])

    def params():
        np.random.seed(7)
        tike_amd.random.randomizer_np = np.random.default_rng(8)
        return tp.PtychoParameters(
            probe=p["probe"].copy(), psi=np.full_like(p["psi"], 0.5 + 0j),
            scan=p["scan"].copy(),
            algorithm_options=tp.LstsqOptions(num_batch=4, num_iter=epochs),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())

    out = {}
    for num_gpu in (None, 2):
        t0 = time.perf_counter()
        out[num_gpu] = tp.reconstruct(data, params(), num_gpu=num_gpu)
        dt = time.perf_counter() - t0
        r = out[num_gpu]
        print(f"num_gpu={num_gpu}: {dt:.1f} s wall, epochs "
              f"{sum(r.algorithm_options.times):.2f} s, costs "
              + " ".join(f"{c[0]:.4e}" for c in r.algorithm_options.costs),
              flush=True)
    a, b = out[None], out[2]
    print("psi normwise difference one rank / two ranks:",
          float(np.linalg.norm(a.psi - b.psi) / np.linalg.norm(a.psi)))
