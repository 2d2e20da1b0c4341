"""reconstruct(data, parameters, num_gpu=N) from a plain process at job size
(`gpurun -- python tools/soak_spawn.py --positions 80000 --gpus 2`; ranks that
outnumber the box's GPUs share them over gloo, TIKE_AMD_OVERSUBSCRIBE=1):
set-up seconds (wall time of the call minus its epochs), peak host memory of
the whole process tree against the size of the dataset, and -- at small sizes
(--check) -- the result against the one-rank call.

The patterns are synthetic noise of the right shape and dtype (simulating
80 000 x 256 x 256 takes longer than the run): what is measured is the
set-up -- clustering once in the parent, every rank receiving only its own
rows block by block -- not convergence."""
import argparse
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402


def tree_rss_bytes(pid):
    """Resident memory of a process and its descendants, shared-memory pages
    counted once (Pss)."""
    import psutil
    total = 0
    try:
        procs = [psutil.Process(pid)] + psutil.Process(pid).children(recursive=True)
    except psutil.Error:
        return 0
    for p in procs:
        try:
            total += p.memory_full_info().pss
        except psutil.Error:
            pass
    return total


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=4000)
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--modes", type=int, default=8)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--check", action="store_true",
                    help="simulate real patterns and compare with one rank")
    a = ap.parse_args()
    os.environ["TIKE_AMD_OVERSUBSCRIBE"] = "1"
    import bench
    import tike_amd.ptycho as tp
    import tike_amd.random
    N, S, det = a.positions, a.modes, a.det
    p = bench.synthetic(N, S, det, 0, N)
    if a.check:
        data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
    else:
        rng = np.random.default_rng(0)
        data = np.empty((N, det, det), dtype=np.float32)
        for lo in range(0, N, 4096):
            data[lo:lo + 4096] = rng.random((min(4096, N - lo), det, det),
                                            dtype=np.float32)
    nbytes = data.nbytes

    def params():
        np.random.seed(7)
        tike_amd.random.randomizer_np = np.random.default_rng(8)
        return tp.PtychoParameters(
            probe=p["probe"].copy(), psi=np.full_like(p["psi"], 0.5 + 0j),
            scan=p["scan"].copy(),
            algorithm_options=tp.LstsqOptions(
                num_batch=max(4, N // 1000), num_iter=a.epochs),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())

    peak = [0]
    stop = threading.Event()

    def watch():
        while not stop.wait(0.25):
            peak[0] = max(peak[0], tree_rss_bytes(os.getpid()))

    out = {}
    for num_gpu in ((None, a.gpus) if a.check else (a.gpus,)):
        peak[0] = 0
        stop.clear()
        t = threading.Thread(target=watch, daemon=True)
        t.start()
        t0 = time.perf_counter()
        out[num_gpu] = tp.reconstruct(data, params(), num_gpu=num_gpu)
        dt = time.perf_counter() - t0
        stop.set()
        t.join()
        r = out[num_gpu]
        epochs_s = sum(r.algorithm_options.times)
        print(f"num_gpu={num_gpu}: {N} positions {det}^2 x {S}, dataset "
              f"{nbytes / 2**30:.2f} GiB: call {dt:.1f} s = epochs "
              f"{epochs_s:.2f} s + set-up {dt - epochs_s:.1f} s; peak host "
              f"memory of the process tree {peak[0] / 2**30:.2f} GiB = "
              f"{peak[0] / nbytes:.2f} x the dataset", flush=True)
    if a.check:
        x, y = out[None], out[a.gpus]
        print("psi normwise difference one rank / several ranks:",
              float(np.linalg.norm(x.psi - y.psi) / np.linalg.norm(x.psi)))
