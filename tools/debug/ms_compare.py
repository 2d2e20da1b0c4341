"""debug: fused vs slice-by-slice multislice gradients of one minibatch"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, importlib
import tike_amd.ptycho as tp
from test_solvers_gpu import _headline_problem
R = importlib.import_module("tike_amd.ptycho.solvers.rpie")
det, S, N, depth = 256, 8, 10, 2
scan, psi_true, probe0, ep, ew, data = _headline_problem(tp, det, S, N, seed=13 * depth + S, eigen=False)
psi0 = np.repeat(np.full_like(psi_true, 0.5), depth, axis=0); psi0[1:] = 1.0
out = {}
for fused in (True, False):
    R.FUSED_MULTISLICE = fused
    params = tp.PtychoParameters(probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        algorithm_options=tp.RpieOptions(num_batch=1, num_iter=1, batch_method="compact", alpha=1.0),
        probe_options=tp.ProbeOptions(force_orthogonality=True, probe_wavelength=1e-10, probe_FOV_lengths=(2e-6, 2e-6)),
        object_options=tp.ObjectOptions(multislice_propagation_distance=1e-6),
        exitwave_options=tp.ExitWaveOptions(measured_pixels=np.ones((det, det), dtype=bool)))
    with tp.Reconstruction(data, params, order=np.arange(N), batches=[np.arange(N)]) as ctx:
        p = ctx.parameters
        psi_num = torch.zeros_like(p.psi)
        cost, probe_num = R._get_nearplane_gradients(ctx.data, p.psi, p.scan, p.probe, None, None, 0, N, ctx.comm,
            psi_num, op=ctx.operator, exitwave_options=p.exitwave_options, recover_psi=True, recover_probe=True)
        out[fused] = (float(cost), psi_num.cpu().numpy(), probe_num.cpu().numpy())
a, b = out[True], out[False]
print("cost", a[0], b[0])
for d in range(depth):
    print("slice", d, "psi_num rel", np.linalg.norm(a[1][d] - b[1][d]) / np.linalg.norm(b[1][d]),
          "probe_num rel", np.linalg.norm(a[2][d] - b[2][d]) / np.linalg.norm(b[2][d]),
          "norms", np.linalg.norm(a[1][d]), np.linalg.norm(b[1][d]), np.linalg.norm(a[2][d]), np.linalg.norm(b[2][d]))
