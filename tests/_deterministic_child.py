"""Child process of test_deterministic_mode: runs the reference's
ReconstructTwice configuration (fixture lstsq_recon_compact) and a headline-shaped
problem (256^2, 8 modes + eigen probe, the fused kernels), then two rpie epochs
on a two-slice object (the fused slice stages), and prints one JSON
line with hashes of every result and the errors against the fixture.  The
parent runs it twice under TIKE_DETERMINISTIC=1 and once without."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402

import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402
from test_solvers_gpu import (_headline_problem,  # noqa: E402
                              _reconstruct_like_reference)
from util import relerr  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    out = {}
    g = np.load(os.path.join(HERE, "golden", "lstsq_recon_compact.npz"),
                allow_pickle=False)
    r1, r2 = _reconstruct_like_reference(tp, g, second=True)
    out["compact"] = digest(r1.psi, r1.probe, np.array(r1.algorithm_options.costs),
                            r2.psi, r2.probe, np.array(r2.algorithm_options.costs))
    out["compact_err2"] = [float(relerr(r2.psi, g["psi_2"])),
                           float(relerr(r2.probe, g["probe_2"]))]
    out["compact_cost2"] = float(np.max(np.abs(
        np.array(r2.algorithm_options.costs) / g["costs_2"] - 1)))
    det, S, N = 256, 8, 24
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=5, eigen=True)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=np.full_like(psi_true, 0.5), scan=scan.copy(),
        eigen_probe=ep.copy(), eigen_weights=ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2,
                                          batch_method="wobbly_center"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    tike_amd.random.randomizer_np = np.random.default_rng(3)
    os.environ.setdefault("TIKE_CHUNK_POSITIONS", "")
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=np.array_split(np.arange(N), 2)) as ctx:
        ctx.iterate(2)
        r = ctx.get_result()
    out["headline"] = digest(r.psi, r.probe, r.eigen_probe, r.eigen_weights,
                             np.array(r.algorithm_options.costs))
    out["headline_cost"] = [float(c[0]) for c in r.algorithm_options.costs]
    out["headline_psi_norm"] = float(np.linalg.norm(r.psi))
    # a two-slice object through rpie's fused slice stages (their probe
    # numerators: per-chunk partial sums under the switch)
    S, N = 3, 20
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=8, eigen=False)
    psi0 = np.repeat(np.full_like(psi_true, 0.5), 2, axis=0)
    psi0[1:] = 1.0
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0, scan=scan.copy(),
        algorithm_options=tp.RpieOptions(num_batch=2, num_iter=2,
                                         batch_method="compact", alpha=1.0),
        probe_options=tp.ProbeOptions(
            force_orthogonality=True, probe_wavelength=1e-10,
            probe_FOV_lengths=(2e-6, 2e-6)),
        object_options=tp.ObjectOptions(multislice_propagation_distance=1e-6),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=np.array_split(np.arange(N), 2)) as ctx:
        ctx.iterate(2)
        r = ctx.get_result()
    out["multislice"] = digest(r.psi, r.probe,
                               np.array(r.algorithm_options.costs))
    out["multislice_cost"] = [float(c[0]) for c in r.algorithm_options.costs]
    # (round 6) the conjugate-gradient solver -- its direction sums and the
    # all-steps-at-once line search -- at 256^2 (hand-off cost pass) and at
    # 128^2 (stored far planes), object then probe per minibatch
    for det_c in (256, 128):
        S, N = 1, 20
        scan, psi_true, probe0, _, _, data = _headline_problem(
            tp, det_c, S, N, seed=21 + det_c, eigen=False)
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=tp.CgradOptions(num_batch=2, num_iter=2,
                                              cg_iter=3),
            probe_options=tp.ProbeOptions(), object_options=tp.ObjectOptions())
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=np.array_split(np.arange(N), 2)) as ctx:
            ctx.iterate(2)
            r = ctx.get_result()
        out[f"cgrad{det_c}"] = digest(r.psi, r.probe,
                                      np.array(r.algorithm_options.costs))
        out[f"cgrad{det_c}_cost"] = [float(np.ravel(c)[0])
                                     for c in r.algorithm_options.costs]
    # ... and the Poisson model with per-mode step lengths (under the switch
    # the plan takes the stored-far-plane entries: no atomics)
    det, S, N = 256, 8, 16
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=31, eigen=False)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=np.full_like(psi_true, 0.5), scan=scan.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2,
                                          batch_method="wobbly_center"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool),
            noise_model="poisson"))
    tike_amd.random.randomizer_np = np.random.default_rng(4)
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=np.array_split(np.arange(N), 2)) as ctx:
        ctx.iterate(2)
        r = ctx.get_result()
    out["poisson"] = digest(r.psi, r.probe, np.array(r.algorithm_options.costs))
    out["poisson_cost"] = [float(c[0]) for c in r.algorithm_options.costs]
    # (round 6, late) the routes added this round: 10 modes at 128^2 (pass 2 in
    # two groups of modes), a 192^2 detector (prime-factor launches), a 100^2
    # one (unfused kernels on the mixed-radix transforms)
    for key, det_r, S_r, N_r in (("groups", 128, 10, 14), ("pfa", 192, 3, 12),
                                 ("offgrid", 100, 2, 12)):
        scan, psi_true, probe0, ep, ew, data = _headline_problem(
            tp, det_r, S_r, N_r, seed=41 + det_r, eigen=True)
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(), eigen_probe=ep.copy(), eigen_weights=ew.copy(),
            algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2,
                                              batch_method="wobbly_center"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions(),
            exitwave_options=tp.ExitWaveOptions(
                measured_pixels=np.ones((det_r, det_r), dtype=bool)))
        tike_amd.random.randomizer_np = np.random.default_rng(6)
        with tp.Reconstruction(data, params, order=np.arange(N_r),
                               batches=np.array_split(np.arange(N_r), 2)) as ctx:
            ctx.iterate(2)
            r = ctx.get_result()
        out[key] = digest(r.psi, r.probe, r.eigen_probe, r.eigen_weights,
                          np.array(r.algorithm_options.costs))
        out[key + "_cost"] = [float(c[0]) for c in r.algorithm_options.costs]
    # ... and a two-slice object at 128^2 under the Poisson model (the fused
    # chain with the last slice's far field stored)
    det_m, S, N = 128, 2, 14
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det_m, S, N, seed=9, eigen=False)
    data = np.round(data * (20000.0 / data.max())).astype(np.float32)
    psi0 = np.repeat(np.full_like(psi_true, 0.5), 2, axis=0)
    psi0[1:] = 1.0
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0, scan=scan.copy(),
        algorithm_options=tp.RpieOptions(num_batch=2, num_iter=2,
                                         batch_method="compact", alpha=1.0),
        probe_options=tp.ProbeOptions(
            force_orthogonality=True, probe_wavelength=1e-10,
            probe_FOV_lengths=(2e-6, 2e-6)),
        object_options=tp.ObjectOptions(multislice_propagation_distance=1e-6),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det_m, det_m), dtype=bool),
            noise_model="poisson"))
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=np.array_split(np.arange(N), 2)) as ctx:
        ctx.iterate(2)
        r = ctx.get_result()
    out["multislice128p"] = digest(r.psi, r.probe,
                                   np.array(r.algorithm_options.costs))
    out["multislice128p_cost"] = [float(c[0])
                                  for c in r.algorithm_options.costs]
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
