"""Does rpie + Poisson on the two-slice bench problem diverge on the fused chain only?"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import tike_amd._arrays as A
import tike_amd.ptycho as tp
R = importlib.import_module("tike_amd.ptycho.solvers.rpie")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
E = int(sys.argv[2]) if len(sys.argv) > 2 else 70
for fused in (True, False):
    R.FUSED_MULTISLICE = fused
    built = bench.epoch_problem("c3rpie2poisson", N, 1, 0, tp, A)
    ctx = built["ctx"]
    costs = []
    try:
        for e in range(E):
            try:
                ctx.iterate(1)
            except Exception as ex:
                print("fused", fused, "epoch", e, type(ex).__name__)
                break
            costs.append(float(np.ravel(ctx.parameters.algorithm_options.costs[-1])[0]))
    finally:
        ctx.__exit__(None, None, None)
    print("fused", fused, " ".join(f"{c:.4g}" for c in costs[::3]), flush=True)
R.FUSED_MULTISLICE = True
