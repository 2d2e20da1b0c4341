"""Is the product-vs-oracle cost divergence on bench.py's c3 inputs (with
eigen probes; profiles/r05_soak_vs_oracle.txt) chaos of the algorithm or a
difference of the HIP path?  CPU only, no product code in the runs: the
ORACLE against itself --

  (i)   as is (float32 / complex64, the reference's precision);
  (ii)  from psi0 * (1 + 1e-6)  (a perturbation of one float32 ulp order);
  (iii) with every array promoted to float64 / complex128 (a second copy of
        the oracle modules whose `np.float32` / `np.complex64` resolve to the
        wide types).

  (iv)  `structured`: after one epoch, probe mode S-1 += eps e^(i theta) x
        mode 0, eps = 4e-6 -- a change of the probe of relative size 3.8e-6,
        what product and oracle differ by after one epoch (3.9e-6, measured:
        tools/eigen_diag.py) -- then the second epoch.

Result (profiles/r06_oracle_sensitivity.txt): (i)-(iii) agree to 1e-3 per
epoch -- UNSTRUCTURED rounding is not amplified -- but (iv) moves the
oracle's own epoch-2 cost by +6.5 ... +25 %: `orthogonalize_eig`
(probe.py:726-769) hands the modes to LAPACK's eigh, which fixes the phase of
every eigenvector by making its LAST component real, and for the dominant
mode of these inputs that component is 8e-7 of the vector.  The phase of
probe mode 0 -- relative to the eigen probe, which is not rotated with it --
is therefore decided by a quantity the size of float32 rounding, and two
float32 implementations (the reference on cuSOLVER and on LAPACK included)
part by several per cent of the cost in the second epoch on these inputs.

    python tools/oracle_sensitivity.py [N=160] [epochs=6] [eigen=init,large,none]
    python tools/oracle_sensitivity.py structured [N=160] [det=256] [modes=8]

Test infrastructure (imports oracle/, like tests/ and bench's cpu leg); the
problem generator is bench.synthetic + init_varying_probe (host code).  The
result is committed as profiles/r06_oracle_sensitivity.txt and asserted, at
a smaller size, by tests/test_oracle_sensitivity_cpu.py."""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


class _WideNumpy(types.ModuleType):
    """numpy with float32 -> float64 and complex64 -> complex128."""

    def __init__(self):
        super().__init__("numpy_wide")

    def __getattr__(self, name):
        if name == "float32":
            return np.float64
        if name == "complex64":
            return np.complex128
        return getattr(np, name)


def wide_oracle():
    """A second copy of oracle.{operators,position,solvers} computing in
    float64 / complex128 (the modules look `np` up in their globals)."""
    wide = _WideNumpy()
    pkg = types.ModuleType("oracle_wide")
    pkg.__path__ = [os.path.join(ROOT, "oracle")]
    sys.modules["oracle_wide"] = pkg
    mods = {}
    for name in ("operators", "position", "solvers"):
        spec = importlib.util.spec_from_file_location(
            f"oracle_wide.{name}", os.path.join(ROOT, "oracle", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        mod.np = wide
        mods[name] = mod
    return mods["solvers"]


def c3_problem(N, det=256, S=8, eigen="init"):
    """bench.py's c3 problem at N positions (host code only)."""
    import bench
    import tike_amd.ptycho.probe as tprobe
    import tike_amd.random
    from oracle import operators as oops
    p = bench.synthetic(N, S, det, 0, N)
    np.random.seed(1234)
    tike_amd.random.randomizer_np = np.random.default_rng(4321)
    ep, ew = tprobe.init_varying_probe(p["scan"], p["probe"],
                                       num_eigen_probes=2,
                                       probes_with_modes=1)
    if eigen == "none":
        ep = ew = None
    elif eigen == "large":
        ew[:, 1, 0] = 0.05 * np.random.default_rng(5).standard_normal(
            N).astype(np.float32)
    data = oops.simulate(det, p["probe"], p["scan"], p["psi"])
    return p, ep, ew, data


def run(osol, p, ep, ew, data, num_batch, epochs, rule, psi_scale=1.0,
        wide=False):
    det = data.shape[-1]
    N = len(p["scan"])
    cp = lambda x: None if x is None else x.copy()
    psi0 = np.full_like(p["psi"], 0.5 + 0j) * np.complex64(psi_scale)
    state = dict(psi=psi0, probe=p["probe"].copy(), scan=p["scan"].copy(),
                 costs=[], eigen_probe=cp(ep), eigen_weights=cp(ew))
    if wide:
        for k in ("psi", "probe", "eigen_probe"):
            if state[k] is not None:
                state[k] = state[k].astype(np.complex128)
        if state["eigen_weights"] is not None:
            state["eigen_weights"] = state["eigen_weights"].astype(np.float64)
        data = data.astype(np.float64)
    batches = np.array_split(np.arange(N), num_batch)
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, epochs, detector_shape=det,
                         batch_method=rule, force_orthogonality=True,
                         rng=np.random.default_rng(11))
    return np.array([float(np.ravel(c)[0]) for c in state["costs"]]), state


def first_epoch(osol, p, ep, ew, data, num_batch, rule, psi_scale=1.0):
    """(state after one epoch, the generator that continues the minibatch
    order, batches)."""
    det = data.shape[-1]
    N = len(p["scan"])
    cp = lambda x: None if x is None else x.copy()
    psi0 = np.full_like(p["psi"], 0.5 + 0j) * np.complex64(psi_scale)
    state = dict(psi=psi0, probe=p["probe"].copy(), scan=p["scan"].copy(),
                 costs=[], eigen_probe=cp(ep), eigen_weights=cp(ew))
    batches = np.array_split(np.arange(N), num_batch)
    state = osol.rescale_probe(state, data, det)
    rng = np.random.default_rng(11)
    state = osol.iterate(state, data, batches, 1, detector_shape=det,
                         batch_method=rule, force_orthogonality=True, rng=rng)
    return state, rng, batches


def second_epoch_cost(osol, state, rng, batches, data, rule, eps=0.0,
                      theta=0.0):
    """Cost of the second epoch from a copy of `state` whose LAST probe mode
    received eps e^(i theta) x mode 0; also the relative size of that change."""
    import copy
    s = copy.deepcopy(state)
    r = np.random.default_rng()
    r.bit_generator.state = copy.deepcopy(rng.bit_generator.state)
    probe = s["probe"]
    d = (eps * np.exp(1j * theta) * probe[0, 0, 0]).astype(np.complex64)
    probe[0, 0, -1] += d
    size = float(np.linalg.norm(d) / np.linalg.norm(probe))
    s = osol.iterate(s, data, batches, 1, detector_shape=data.shape[-1],
                     batch_method=rule, force_orthogonality=True, rng=r)
    return float(np.ravel(s["costs"][-1])[0]), size


def dominant_eigenvector(probe):
    """LAPACK's eigenvector of the largest eigenvalue of the modes' Gram
    matrix, as orthogonalize_eig (probe.py:726-769) obtains it."""
    S = probe.shape[-3]
    gram = np.array([[np.vdot(probe[0, 0, i], probe[0, 0, j])
                      for j in range(S)] for i in range(S)])
    return np.linalg.eigh(gram, UPLO="U")[1][:, -1]


def structured():
    import bench
    from oracle import solvers as osol
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    det = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    S = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    p, ep, ew, data = c3_problem(N, det=det, S=S, eigen="init")
    state, rng, batches = first_epoch(osol, p, ep, ew, data, 10,
                                      bench.BATCH_RULE)
    v = dominant_eigenvector(state["probe"])
    print(f"bench c3 problem, N {N}, {det}x{det}, {S} modes, eigen probe init:"
          f" second epoch of the ORACLE from its own state after epoch 1")
    print(f"  LAPACK's dominant eigenvector of the modes' Gram matrix: last "
          f"component {abs(v[-1]):.2e} (real: the phase convention), first "
          f"{abs(v[0]):.4f}")
    base, _ = second_epoch_cost(osol, state, rng, batches, data,
                                bench.BATCH_RULE)
    print(f"  as is                                   : cost {base:.4e}",
          flush=True)
    for eps, theta, name in ((4e-6, 0.0, "0"), (4e-6, np.pi / 2, "pi/2"),
                             (4e-6, np.pi, "pi"), (1e-6, np.pi, "pi")):
        c, size = second_epoch_cost(osol, state, rng, batches, data,
                                    bench.BATCH_RULE, eps, theta)
        print(f"  mode {S - 1} += {eps:.0e} e^(i {name:4s}) mode 0 (probe "
              f"changes by {size:.1e}): cost {c:.4e}  ({(c - base) / base:+.1%})",
              flush=True)
    s2, r2, _ = first_epoch(osol, p, ep, ew, data, 10, bench.BATCH_RULE,
                            psi_scale=1 + 1e-6)
    c, _ = second_epoch_cost(osol, s2, r2, batches, data, bench.BATCH_RULE)
    print(f"  both epochs from psi0 * (1 + 1e-6)       : cost {c:.4e}  "
          f"({(c - base) / base:+.1%})")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "structured":
        return structured()
    import bench
    from oracle import solvers as osol
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    kinds = (sys.argv[3] if len(sys.argv) > 3 else "init,large,none").split(",")
    det = int(os.environ.get("SENS_DET", "256"))
    S = int(os.environ.get("SENS_MODES", "8"))
    wsol = wide_oracle()
    for eigen in kinds:
        p, ep, ew, data = c3_problem(N, det=det, S=S, eigen=eigen)
        print(f"bench c3 problem, N {N}, {det}x{det}, {S} modes, 10 "
              f"minibatches, rule {bench.BATCH_RULE}, eigen probe: {eigen}",
              flush=True)
        rows = {}
        for tag, kw in (("oracle f32          ", dict()),
                        ("oracle f32, psi0*(1+1e-6)", dict(psi_scale=1 + 1e-6)),
                        ("oracle f64          ", dict(wide=True))):
            sol = wsol if kw.get("wide") else osol
            c, _ = run(sol, p, ep, ew, data, 10, epochs, bench.BATCH_RULE, **kw)
            rows[tag] = c
            print(f"  {tag:26s}: " + " ".join(f"{x:.4e}" for x in c),
                  flush=True)
        a, b, w = rows.values()
        print("  |f32 - perturbed| / f32   : " +
              " ".join(f"{abs(x - y) / x:.1e}" for x, y in zip(a, b)))
        print("  |f32 - f64| / f64         : " +
              " ".join(f"{abs(x - y) / y:.1e}" for x, y in zip(a, w)),
              flush=True)


if __name__ == "__main__":
    main()
