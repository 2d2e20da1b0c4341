"""cgrad at a prime-factor detector size: line-search probes through the
prime-factor launches (cost only) against the unfused forward operator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
import tike_amd.ptycho as tp
import bench
import tike_amd._arrays as A
from tike_amd.ptycho.solvers import lstsq as L
for det, S, N in ((384, 1, 2000), (192, 2, 4000), (640, 1, 800)):
    p = bench.synthetic(N, S, det, 0, N)
    data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
    for pfa in (True, False):
        L.PFA_ROUTE = pfa
        params = tp.PtychoParameters(
            probe=p["probe"].copy(), psi=np.full_like(p["psi"], 0.5 + 0j), scan=p["scan"],
            algorithm_options=tp.CgradOptions(num_batch=4, cg_iter=4),
            probe_options=tp.ProbeOptions(), object_options=tp.ObjectOptions())
        with tp.Reconstruction(A.to_device(data, np.float32), params, order=np.arange(N),
                               batches=np.array_split(np.arange(N), 4)) as ctx:
            ctx.iterate(2); torch.cuda.synchronize()
            t = time.perf_counter(); ctx.iterate(3); torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / 3
            c = ctx.get_result().algorithm_options.costs[-1]
        print(f"cgrad {det}^2 x {S}, {N} positions, prime-factor route {pfa}: {N / dt / 1e3:.1f} k patterns/s, cost {np.ravel(c)[0]:.6e}", flush=True)
    L.PFA_ROUTE = True
