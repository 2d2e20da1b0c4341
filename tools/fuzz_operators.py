"""Random shapes through the operators against the CPU oracle: `Ptycho.fwd`,
`Ptycho.adj` (detector 16 ... 640 of any factorisation, probe window <= detector,
1 ... 12 modes, shared or per-position probes, positions anywhere the
reference allows) and `Propagation.fwd` / `.adj` at random sizes and norms.

    gpurun -- python tools/fuzz_operators.py [cases=60] [seed=0]

(test infrastructure: imports oracle/, like tests/ and bench's cpu leg)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import tike_amd.operators as ops  # noqa: E402
from oracle import operators as oops  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


def rc(*shape):
    return (rng.standard_normal(shape) +
            1j * rng.standard_normal(shape)).astype(np.complex64)


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - b) / max(np.linalg.norm(b), 1e-30))


bad = 0
for case in range(cases):
    det = int(rng.choice((16, 31, 45, 64, 96, 100, 127, 128, 160, 192, 200, 224,
                          256, 300, 320, 384, 448, 512, 640)))
    pw = det if rng.random() < 0.6 else int(rng.integers(max(4, det // 2), det))
    S = int(rng.choice((1, 2, 3, 4, 5, 8, 9, 12)))
    if det > 320:
        S = min(S, 4)
    N = int(rng.integers(1, 9))
    shared = bool(rng.random() < 0.5)
    HW = pw + int(rng.integers(8, 60))
    scan = (rng.random((N, 2)) * (HW - pw - 3) + 1).astype(np.float32)
    if rng.random() < 0.3:
        scan[0] = np.floor(scan[0])
    probe = rc(1 if shared else N, 1, S, pw, pw)
    psi = rc(1, HW, HW)
    tag = f"det {det} pw {pw} S {S} N {N} shared {int(shared)} HW {HW}"
    try:
        with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
            far = op.fwd(probe=probe, scan=scan, psi=psi)
            want = oops.ptycho_fwd(np.broadcast_to(probe, (N, 1, S, pw, pw)),
                                   scan, psi, det)
            e_f = rel(far, want)
            g = rc(N, 1, S, det, det)
            psi_adj, probe_adj = op.adj(farplane=g, probe=probe, scan=scan,
                                        psi=psi)
            o_psi, o_probe = oops.ptycho_adj(
                g, np.broadcast_to(probe, (N, 1, S, pw, pw)), scan, psi)
            e_a, e_p = rel(psi_adj, o_psi), rel(probe_adj, o_probe)
        ok = max(e_f, e_a, e_p) < 2e-5
        print(f"{'ok ' if ok else 'BAD'} {tag}: fwd {e_f:.1e} psi_adj {e_a:.1e} "
              f"probe_adj {e_p:.1e}", flush=True)
        bad += not ok
    except Exception as e:  # noqa: BLE001
        bad += 1
        print(f"ERR {tag}: {type(e).__name__}: {str(e)[:200]}", flush=True)
print(f"Ptycho.fwd / .adj: {cases - bad} of {cases} agree", flush=True)

pbad = 0
for case in range(cases):
    n = int(rng.integers(2, 1100)) if rng.random() < 0.8 else int(rng.choice((1536, 2000, 2048)))
    nt = 1 if n > 600 else int(rng.integers(1, 5))
    norm = str(rng.choice(("ortho", "forward", "backward")))
    x = rc(nt, n, n)
    try:
        with ops.Propagation(detector_shape=n, norm=norm) as P:
            f = P.fwd(nearplane=x)
            a = P.adj(farplane=x)
        e1 = rel(f, np.fft.fft2(x.astype(np.complex128), norm=norm))
        e2 = rel(a, np.fft.ifft2(x.astype(np.complex128), norm=norm))
        ok = max(e1, e2) < 2e-5
        print(f"{'ok ' if ok else 'BAD'} Propagation n {n} tiles {nt} {norm}: "
              f"fwd {e1:.1e} adj {e2:.1e}", flush=True)
        pbad += not ok
    except Exception as e:  # noqa: BLE001
        pbad += 1
        print(f"ERR Propagation n {n} {norm}: {type(e).__name__}: {str(e)[:160]}",
              flush=True)
print(f"Propagation: {cases - pbad} of {cases} agree")
sys.exit(1 if bad or pbad else 0)
