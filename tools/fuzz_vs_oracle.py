"""Random lstsq_grad problems against the CPU oracle (pinned to the reference):
detector 32 ... 256 of any factorisation (320 / 384 now and then), probe window
<= detector, 1 ... 10 modes, none / one / two eigen probes on 1 ... 3 modes, masks
with NaN counts, both noise models and both Poisson step rules, uint16 counts,
probe-only runs, 1 ... 3 minibatches under both update rules -- two epochs,
costs and final iterates.

    gpurun -- python tools/fuzz_vs_oracle.py [cases=60] [seed=0]

(test infrastructure: imports oracle/, like tests/ and bench's cpu leg)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402
from oracle import solvers as osol  # noqa: E402
from test_solvers_gpu import _headline_problem  # noqa: E402
from tike_amd.ptycho.solvers._plan import GradientPlan  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
SIZES = (32, 45, 64, 96, 100, 127, 128, 160, 192, 200, 224, 256)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - b) / max(np.linalg.norm(b), 1e-30))


routes = []
real = GradientPlan.gradients
GradientPlan.gradients = lambda self, c, k: (routes.append(
    self.route + ("+groups" if self.groups else "")), real(self, c, k))[1]
bad = illcond = 0
for case in range(cases):
    det = int(rng.choice(SIZES)) if rng.random() < 0.9 else int(rng.choice((320, 384)))
    pw = det if rng.random() < 0.6 else int(det - 2 * rng.integers(1, max(2, det // 6)))
    S = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 10])) if det <= 256 else int(rng.integers(1, 4))
    N = int(rng.integers(4, 10))
    eig = (0, 0) if rng.random() < 0.4 else (int(rng.integers(2, 4)),
                                             int(rng.integers(1, min(S, 3) + 1)))
    masked = bool(rng.random() < 0.3)
    model = "poisson" if rng.random() < 0.3 else "gaussian"
    usemodes = "dominant_mode" if rng.random() < 0.4 else "all_modes"
    u16 = bool(rng.random() < 0.25)
    recover_psi = bool(rng.random() < 0.85)
    nb = int(rng.choice((1, 2, 3)))
    method = "compact" if rng.random() < 0.6 else "wobbly_center"
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=3000 + case, eigen=eig if eig[0] else False, pw=pw)
    data = np.round(data * (20000.0 / data.max()))
    mask = (rng.random((det, det)) > 0.1) if masked else np.ones((det, det), bool)
    data = data.astype(np.uint16 if u16 else np.float32)
    fdata = data.astype(np.float32)
    if masked and not u16:
        data = data.copy()
        data[:, ~mask] = np.nan  # the reference never reads unmeasured counts
    tag = (f"det {det} pw {pw} S {S} N {N} eigen {eig} mask {int(masked)} {model}"
           f"{'/' + usemodes if model == 'poisson' else ''} u16 {int(u16)} psi "
           f"{int(recover_psi)} batches {nb} {method}")
    psi0 = np.full_like(psi_true, 0.5) if recover_psi else psi_true.copy()
    batches = np.array_split(np.arange(N), nb)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        eigen_probe=None if ep is None else ep.copy(),
        eigen_weights=None if ew is None else ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=nb, num_iter=2,
                                          batch_method=method),
        probe_options=tp.ProbeOptions(force_orthogonality=False),
        object_options=tp.ObjectOptions() if recover_psi else None,
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=mask, noise_model=model,
            step_length_usemodes=usemodes))
    del routes[:]
    try:
        tike_amd.random.randomizer_np = np.random.default_rng(11)
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=batches) as ctx:
            ctx.iterate(2)
            got = ctx.get_result()
        state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                     costs=[], eigen_probe=None if ep is None else ep.copy(),
                     eigen_weights=None if ew is None else ew.copy())
        state = osol.rescale_probe(state, fdata, det, measured_pixels=mask)
        state = osol.iterate(
            state, fdata, batches, 2, detector_shape=det, batch_method=method,
            force_orthogonality=False, rng=np.random.default_rng(11),
            measured_pixels=mask, noise_model=model,
            step_length_usemodes=usemodes, recover_psi=recover_psi)
        ca = np.array(got.algorithm_options.costs).ravel()
        cb = np.array([np.ravel(c)[0] for c in state["costs"]])
        dc = float(np.max(np.abs(ca / cb - 1)))
        dp, dq = rel(got.psi, state["psi"]), rel(got.probe, state["probe"])
        ok = dc < 1e-3 and dp < 1e-3 and dq < 2e-3
        note = ""
        if not ok:
            # is the PROBLEM ill-conditioned?  The oracle against itself from a
            # probe changed by 1e-6 (relative, random): a spread of the size of
            # the discrepancy means no float32 implementation can agree better
            prng = np.random.default_rng(1)
            st2 = dict(psi=psi0.copy(), scan=scan.copy(), costs=[],
                       probe=(probe0 * (1 + 1e-6 * prng.standard_normal(
                           probe0.shape))).astype(np.complex64),
                       eigen_probe=None if ep is None else ep.copy(),
                       eigen_weights=None if ew is None else ew.copy())
            st2 = osol.rescale_probe(st2, fdata, det, measured_pixels=mask)
            st2 = osol.iterate(
                st2, fdata, batches, 2, detector_shape=det, batch_method=method,
                force_orthogonality=False, rng=np.random.default_rng(11),
                measured_pixels=mask, noise_model=model,
                step_length_usemodes=usemodes, recover_psi=recover_psi)
            sp, sq = rel(st2["psi"], state["psi"]), rel(st2["probe"], state["probe"])
            if sp >= 0.3 * dp and sq >= 0.3 * dq:
                ok = True
                illcond += 1
                note = (f"  ILL-CONDITIONED: the oracle moves by psi {sp:.1e} probe "
                        f"{sq:.1e} under a 1e-6 change of the probe")
        print(f"{'ok ' if ok else 'BAD'} {tag}: {sorted(set(routes))}  cost "
              f"{dc:.1e} psi {dp:.1e} probe {dq:.1e}{note}", flush=True)
        bad += not ok
        if not ok:
            print("    hip    costs", " ".join(f"{c:.6e}" for c in ca))
            print("    oracle costs", " ".join(f"{c:.6e}" for c in cb), flush=True)
            if os.environ.get("FUZZ_DUMP"):
                np.savez(os.environ["FUZZ_DUMP"], scan=scan, psi0=psi0, probe0=probe0,
                         ep=ep, ew=ew, data=data, mask=mask, det=det, nb=nb,
                         method=method, model=model, usemodes=usemodes,
                         recover_psi=recover_psi)
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"ERR {tag}: {type(e).__name__}: {str(e)[:240]}", flush=True)
print(f"lstsq_grad vs oracle: {cases - bad} of {cases} agree ({illcond} of them on a "
      f"problem the oracle itself does not reproduce under a 1e-6 change of the probe)",
      flush=True)
GradientPlan.gradients = real

# ---- rpie: single-slice and multislice objects (2 ... 3 slices; the fused chain
# at 128^2 / 256^2, slice by slice elsewhere), both noise models
from oracle import operators as oops  # noqa: E402

rcases = max(4, cases // 3)
rbad = 0
for case in range(rcases):
    depth = int(rng.choice((1, 1, 2, 3)))
    det = int(rng.choice((64, 96, 128, 256) if depth > 1 else SIZES))
    S = int(rng.integers(1, 7))
    N = int(rng.integers(4, 9))
    model = "poisson" if rng.random() < 0.3 else "gaussian"
    alpha = float(rng.choice((0.05, 0.5, 1.0)))
    nb = int(rng.choice((1, 2)))
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=7000 + case, eigen=False)
    data = np.round(data * (20000.0 / data.max())).astype(np.float32)
    psi0 = np.repeat(np.full_like(psi_true, 0.5), depth, axis=0)
    psi0[1:] = 1.0
    phys = dict(wavelength=1e-10, fov=(2e-6, 2e-6), distance=1e-6)
    tag = f"rpie det {det} S {S} N {N} slices {depth} {model} alpha {alpha} batches {nb}"
    batches = np.array_split(np.arange(N), nb)
    ms = dict(probe_wavelength=phys["wavelength"],
              probe_FOV_lengths=phys["fov"]) if depth > 1 else {}
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        algorithm_options=tp.RpieOptions(num_batch=nb, num_iter=2,
                                         batch_method="compact", alpha=alpha),
        probe_options=tp.ProbeOptions(force_orthogonality=False, **ms),
        object_options=tp.ObjectOptions(
            **(dict(multislice_propagation_distance=phys["distance"])
               if depth > 1 else {})),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool), noise_model=model))
    try:
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=batches) as ctx:
            ctx.iterate(2)
            got = ctx.get_result()
        propagator = (oops.fresnel_spectrum_propagator(
            (det, det), phys["fov"], phys["distance"], phys["wavelength"])
            if depth > 1 else None)
        state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                     costs=[], eigen_probe=None, eigen_weights=None)
        kw = dict(propagator=propagator) if depth > 1 else {}
        state = osol.rescale_probe(state, data, det, **kw)
        state = osol.iterate(state, data, batches, 2, detector_shape=det,
                             solver="rpie", alpha=alpha, batch_method="compact",
                             force_orthogonality=False, noise_model=model, **kw)
        ca = np.array(got.algorithm_options.costs).ravel()
        cb = np.array([np.ravel(c)[0] for c in state["costs"]])
        dc = float(np.max(np.abs(ca / cb - 1)))
        dp, dq = rel(got.psi, state["psi"]), rel(got.probe, state["probe"])
        ok = dc < 1e-3 and dp < 1e-3 and dq < 2e-3
        print(f"{'ok ' if ok else 'BAD'} {tag}: cost {dc:.1e} psi {dp:.1e} probe "
              f"{dq:.1e}", flush=True)
        rbad += not ok
    except Exception as e:  # noqa: BLE001
        rbad += 1
        print(f"ERR {tag}: {type(e).__name__}: {str(e)[:240]}", flush=True)
print(f"rpie vs oracle: {rcases - rbad} of {rcases} agree", flush=True)

# ---- cgrad (the composition of SURVEY a17: Dai-Yuan directions, backtracking
# line search; object then probe per minibatch): every accepted / rejected step
# is a branch, so the FIRST call's cost must agree closely and the iterates
# after two calls to 2e-3 (the tests' bar), unless a branch flipped
ccases = max(4, cases // 4)
cbad = 0
for case in range(ccases):
    det = int(rng.choice(SIZES))
    pw = det if rng.random() < 0.7 else int(det - 2 * rng.integers(1, max(2, det // 6)))
    S = int(rng.integers(1, 4))
    N = int(rng.integers(4, 12))
    cg_iter = int(rng.integers(1, 4))
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=9000 + case, eigen=False, pw=pw)
    tag = f"cgrad det {det} pw {pw} S {S} N {N} cg_iter {cg_iter}"
    psi0 = np.full_like(psi_true, 0.5)
    batches = [np.arange(N)]
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        algorithm_options=tp.CgradOptions(num_batch=1, cg_iter=cg_iter,
                                          num_iter=1, batch_method="contiguous"),
        probe_options=tp.ProbeOptions(init_rescale_from_measurements=False),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    try:
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=batches) as ctx:
            ctx.iterate(2)
            got = ctx.get_result()
        state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                     costs=[])
        for _ in range(2):
            state = osol.cgrad(state, data, batches, detector_shape=det,
                               cg_iter=cg_iter, recover_probe=True)
        ca = np.array(got.algorithm_options.costs).ravel()
        cb = np.array([np.ravel(c)[0] for c in state["costs"]])
        d0 = abs(ca[0] / cb[0] - 1)
        dc = float(np.max(np.abs(ca / cb - 1)))
        dp, dq = rel(got.psi, state["psi"]), rel(got.probe, state["probe"])
        ok = d0 < 1e-4 and dc < 2e-3 and dp < 2e-3 and dq < 2e-3
        print(f"{'ok ' if ok else 'BAD'} {tag}: first cost {d0:.1e} costs {dc:.1e} "
              f"psi {dp:.1e} probe {dq:.1e}", flush=True)
        cbad += not ok
    except Exception as e:  # noqa: BLE001
        cbad += 1
        print(f"ERR {tag}: {type(e).__name__}: {str(e)[:240]}", flush=True)
print(f"cgrad vs oracle: {ccases - cbad} of {ccases} agree")
sys.exit(1 if bad or rbad or cbad else 0)
