import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import tike_amd.ptycho as tp, tike_amd.random
from oracle import solvers as osol
g = np.load(os.path.join(os.path.dirname(__file__), "fuzz_bad.npz"), allow_pickle=True)
det = int(g["det"]); scan = g["scan"]; N = len(scan)
def rel(a, b): return float(np.linalg.norm(np.asarray(a) - b) / max(np.linalg.norm(b), 1e-30))
def run(tag, nb=int(g["nb"]), method=str(g["method"]), model=str(g["model"]), usemodes=str(g["usemodes"]),
        eigen=True, f32=False, epochs=2):
    data = g["data"].astype(np.float32) if f32 else g["data"]
    fdata = g["data"].astype(np.float32)
    ep = g["ep"] if eigen else None; ew = g["ew"] if eigen else None
    mask = g["mask"]
    batches = np.array_split(np.arange(N), nb)
    params = tp.PtychoParameters(probe=g["probe0"].copy(), psi=g["psi0"].copy(), scan=scan.copy(),
        eigen_probe=None if ep is None else ep.copy(), eigen_weights=None if ew is None else ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=nb, num_iter=epochs, batch_method=method),
        probe_options=tp.ProbeOptions(force_orthogonality=False), object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(measured_pixels=mask, noise_model=model, step_length_usemodes=usemodes))
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    with tp.Reconstruction(data, params, order=np.arange(N), batches=batches) as ctx:
        ctx.iterate(epochs); got = ctx.get_result()
    state = dict(psi=g["psi0"].copy(), probe=g["probe0"].copy(), scan=scan.copy(), costs=[],
                 eigen_probe=None if ep is None else ep.copy(), eigen_weights=None if ew is None else ew.copy())
    state = osol.rescale_probe(state, fdata, det, measured_pixels=mask)
    state = osol.iterate(state, fdata, batches, epochs, detector_shape=det, batch_method=method,
                         force_orthogonality=False, rng=np.random.default_rng(11), measured_pixels=mask,
                         noise_model=model, step_length_usemodes=usemodes)
    ca = np.array(got.algorithm_options.costs).ravel(); cb = np.array([np.ravel(c)[0] for c in state["costs"]])
    print(f"{tag:28s} cost {np.max(np.abs(ca / cb - 1)):.1e}  psi {rel(got.psi, state['psi']):.1e}  probe {rel(got.probe, state['probe']):.1e}", flush=True)
run("as found")
run("one epoch", epochs=1)
run("compact", method="compact")
run("one minibatch", nb=1)
run("two minibatches", nb=2)
run("gaussian", model="gaussian")
run("dominant_mode", usemodes="dominant_mode")
run("no eigen probe", eigen=False)
run("float32 counts", f32=True)
