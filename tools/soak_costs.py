"""Cost histories of the bench workloads over more epochs than the bench
times (`gpurun -- python tools/soak_costs.py [epochs]`): every solver /
model at its BASELINE shapes must keep lowering the cost and stay finite."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 12
bad = 0
for workload, positions in (("c1", 0), ("c2", 2000), ("c3", 2000),
                            ("c3poisson", 2000), ("c3rpie", 2000), ("c3rpie2", 2000),
                            ("c5", 1000)):
    built = bench.epoch_problem(workload, positions, 1, 0, tp, A)
    ctx = built["ctx"]
    try:
        ctx.iterate(epochs)
        costs = np.array([c[0] for c in
                          ctx.parameters.algorithm_options.costs])
    finally:
        ctx.__exit__(None, None, None)
    torch.cuda.empty_cache()
    ok = np.all(np.isfinite(costs)) and costs[-1] < costs[0]
    rises = int(np.sum(np.diff(costs) > 1e-6 * np.abs(costs[:-1])))
    bad += not ok
    print(f"{workload:10s} {'ok ' if ok else 'BAD'} first {costs[0]:.5e} last "
          f"{costs[-1]:.5e} rises {rises}/{len(costs) - 1}: "
          + " ".join(f"{c:.3e}" for c in costs), flush=True)
sys.exit(1 if bad else 0)
