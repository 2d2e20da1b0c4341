"""Is the product-vs-oracle cost divergence on bench.py's c3 inputs (with
eigen probes; profiles/r05_soak_vs_oracle.txt) chaos of the algorithm or a
difference of the HIP path?  CPU only, no product code in the runs: the
ORACLE against itself --

  (i)   as is (float32 / complex64, the reference's precision);
  (ii)  from psi0 * (1 + 1e-6)  (a perturbation of one float32 ulp order);
  (iii) with every array promoted to float64 / complex128 (a second copy of
        the oracle modules whose `np.float32` / `np.complex64` resolve to the
        wide types).

If (i) and (ii) part by several per cent after the first epoch, the
iteration amplifies rounding and any two float32 implementations (the
reference on CuPy and on NumPy included) part the same way.

    python tools/oracle_sensitivity.py [N=160] [epochs=6] [eigen=init,large,none]

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


def main():
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
