import sys, os; sys.path.insert(0, os.getcwd())
import numpy as np, torch, time
import bench, tike_amd.ptycho as tp, tike_amd._arrays as A, tike_amd.random
S, det, N = 8, 256, 4000
p = bench.synthetic(N, S, det, 0, N)
np.random.seed(1); tike_amd.random.randomizer_np = np.random.default_rng(4321)
ep, ew = tp.init_varying_probe(p["scan"], p["probe"], num_eigen_probes=2, probes_with_modes=1)
data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
rng = np.random.default_rng(0)
probe0 = (p["probe"] * (1 + 0.1 * (rng.standard_normal(p["probe"].shape) + 1j * rng.standard_normal(p["probe"].shape)))).astype(np.complex64)
params = tp.PtychoParameters(probe=probe0, psi=np.full_like(p["psi"], 0.5 + 0j), scan=p["scan"],
    eigen_probe=ep, eigen_weights=ew,
    algorithm_options=tp.LstsqOptions(num_batch=4, batch_method="compact", num_iter=12),
    probe_options=tp.ProbeOptions(force_orthogonality=True), object_options=tp.ObjectOptions(use_adaptive_moment=True))
t = time.time()
r = tp.reconstruct(data, params)
print("costs", np.array(r.algorithm_options.costs).ravel().round(6))
err = np.abs(r.psi[0, 200:-200, 200:-200]) - np.abs(p["psi"][0, 200:-200, 200:-200])
print("central |psi| rms error", float(np.sqrt((err**2).mean())), "time", time.time() - t)
