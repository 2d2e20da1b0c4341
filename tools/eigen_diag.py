"""Where do product and oracle part on bench.py's c3 inputs with eigen probes?
(tools/oracle_sensitivity.py showed the ORACLE is stable
under a 1e-6 perturbation and under float64 -- epoch-2 costs agree to 1e-3 --
so the 12 % gap of profiles/r05_soak_vs_oracle.txt is a difference of the HIP
path.)  Runs the product with its A/B levers one at a time and prints the
epoch costs and the state differences after every epoch.

    gpurun -- python tools/eigen_diag.py [N=160] [epochs=3] [eigen=init]
(test infrastructure: imports oracle/)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402
import tike_amd.random  # noqa: E402
from oracle import solvers as osol  # noqa: E402
from oracle_sensitivity import c3_problem  # noqa: E402
from tike_amd.ptycho.solvers import lstsq as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
eigen = sys.argv[3] if len(sys.argv) > 3 else "init"
num_batch = int(os.environ.get("DIAG_BATCHES", "10"))
rule = os.environ.get("DIAG_RULE", bench.BATCH_RULE)
det = int(os.environ.get("DIAG_DET", "256"))
S = int(os.environ.get("DIAG_MODES", "8"))

p, ep, ew, data = c3_problem(N, det=det, S=S, eigen=eigen)
if os.environ.get("DIAG_TP_SIMULATE") == "1":  # the product's simulate
    data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
cp = lambda x: None if x is None else x.copy()
psi0 = np.full_like(p["psi"], 0.5 + 0j)
batches = np.array_split(np.arange(N), num_batch)


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def oracle_states():
    state = dict(psi=psi0.copy(), probe=p["probe"].copy(),
                 scan=p["scan"].copy(), costs=[], eigen_probe=cp(ep),
                 eigen_weights=cp(ew))
    state = osol.rescale_probe(state, data, det)
    out = []
    rng = np.random.default_rng(11)
    for _ in range(epochs):
        state = osol.iterate(state, data, batches, 1, detector_shape=det,
                             batch_method=rule, force_orthogonality=True,
                             rng=rng)
        out.append({k: cp(state[k]) for k in
                    ("psi", "probe", "eigen_probe", "eigen_weights")})
    return out, [float(np.ravel(c)[0]) for c in state["costs"]]


def product_states():
    params = tp.PtychoParameters(
        probe=p["probe"].copy(), psi=psi0.copy(), scan=p["scan"].copy(),
        eigen_probe=cp(ep), eigen_weights=cp(ew),
        algorithm_options=tp.LstsqOptions(num_batch=num_batch,
                                          batch_method=rule),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions())
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    out = []
    with tp.Reconstruction(A.to_device(data, np.float32), params,
                           presharded=True, order=np.arange(N),
                           batches=batches) as ctx:
        if os.environ.get("DIAG_AT_ONCE") == "1":  # one call for all epochs
            ctx.iterate(epochs)
        for _ in range(epochs):
            if os.environ.get("DIAG_AT_ONCE") != "1":
                ctx.iterate(1)
            got = ctx.get_result()
            out.append(dict(psi=got.psi.copy(), probe=got.probe.copy(),
                            eigen_probe=cp(got.eigen_probe),
                            eigen_weights=cp(got.eigen_weights)))
        costs = [c[0] for c in got.algorithm_options.costs]
    return out, costs


print(f"c3 problem N {N} {det}x{det} S {S} batches {num_batch} rule {rule} "
      f"eigen {eigen}", flush=True)
if os.environ.get("DIAG_PRODUCT_FIRST") == "1":
    _, costs = product_states()
    print("[product before the oracle] costs: " +
          " ".join(f"{c:.4e}" for c in costs), flush=True)
ostates, ocosts = oracle_states()
print("oracle costs: " + " ".join(f"{c:.4e}" for c in ocosts), flush=True)


def report(tag):
    states, costs = product_states()
    print(f"[{tag}] costs: " + " ".join(f"{c:.4e}" for c in costs))
    for e, (s, o) in enumerate(zip(states, ostates)):
        line = [f"  after epoch {e + 1}:"]
        for k in ("psi", "probe", "eigen_probe"):
            if o[k] is not None:
                line.append(f"{k} {rel(s[k], o[k]):.2e}")
        if o["eigen_weights"] is not None:
            w, wo = s["eigen_weights"], o["eigen_weights"]
            line.append(f"w[:,0,0] {rel(w[:, 0, 0], wo[:, 0, 0]):.2e}")
            line.append(f"w[:,0,1:] {rel(w[:, 0, 1:], wo[:, 0, 1:]):.2e}")
            if w.shape[1] > 1:
                line.append(f"w[:,1,0] {rel(w[:, 1, 0], wo[:, 1, 0]):.2e} "
                            f"(|w1| {np.abs(w[:, 1, 0]).mean():.2e} vs "
                            f"{np.abs(wo[:, 1, 0]).mean():.2e})")
        print(" ".join(line), flush=True)


report("default")
for name, setter in () if os.environ.get("DIAG_LEVERS", "1") != "1" else (
        ("PACKED_TAIL=False", lambda: setattr(L, "PACKED_TAIL", False)),
        ("EIGEN_PATCH_RECOMPUTE=False",
         lambda: setattr(L, "EIGEN_PATCH_RECOMPUTE", False)),
        ("EIGEN_SUMS_RECOMPUTE=False",
         lambda: setattr(L, "EIGEN_SUMS_RECOMPUTE", False)),
        ("CHUNK=8", lambda: setattr(L, "CHUNK_POSITIONS_OVERRIDE", 8)),
        ("POSITION_MAJOR_SIZES=() (general kernels)",
         lambda: setattr(L, "POSITION_MAJOR_SIZES", ())),
):
    saved = {k: getattr(L, k) for k in (
        "PACKED_TAIL", "EIGEN_PATCH_RECOMPUTE", "EIGEN_SUMS_RECOMPUTE",
        "CHUNK_POSITIONS_OVERRIDE", "POSITION_MAJOR_SIZES")}
    setter()
    try:
        report(name)
    except Exception as e:  # a lever some configuration refuses
        print(f"[{name}] failed: {e!r}")
    for k, v in saved.items():
        setattr(L, k, v)
