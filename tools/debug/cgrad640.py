import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tike_amd.ptycho as tp
from test_solvers_gpu import _headline_problem
from tike_amd.ptycho.solvers import lstsq as L
det, S, N, nb = 640, 1, 5, 3
for seed in range(1000, 1012):
    scan, psi_true, probe0, _, _, data = _headline_problem(tp, det, S, N, seed=seed, eigen=False)
    data = np.round(data * (20000.0 / data.max())).astype(np.float32)
    out = []
    for pfa in (True, False):
        L.PFA_ROUTE = pfa
        params = tp.PtychoParameters(probe=probe0.copy(), psi=np.full_like(psi_true, 0.5), scan=scan.copy(),
            algorithm_options=tp.CgradOptions(num_batch=nb, num_iter=2, cg_iter=2),
            probe_options=tp.ProbeOptions(), object_options=tp.ObjectOptions())
        with tp.Reconstruction(data, params, order=np.arange(N), batches=np.array_split(np.arange(N), nb)) as ctx:
            ctx.iterate(2); r = ctx.get_result()
        out.append([float(np.ravel(c)[0]) for c in r.algorithm_options.costs])
    L.PFA_ROUTE = True
    print(seed, out[0], out[1], "rel", [abs(a / b - 1) for a, b in zip(*out)], flush=True)
