"""Sustained rate and memory of the bench workloads over many epochs
(`gpurun -- python tools/soak_rate.py [epochs]`): patterns/s of the first and
of the last third of the run, allocator bytes before and after."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for workload in (sys.argv[2:] or ("c3", "c2", "c5", "c1", "c3poisson")):
    built = bench.epoch_problem(workload, 0, 1, 0, tp, A)
    ctx, N = built["ctx"], built["N"]
    try:
        ctx.iterate(3)
        torch.cuda.synchronize()
        mem0 = torch.cuda.memory_allocated()
        times = []
        for _ in range(epochs):
            t0 = time.perf_counter()
            ctx.iterate(1)
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        mem1 = torch.cuda.memory_allocated()
    finally:
        ctx.__exit__(None, None, None)
    third = max(epochs // 3, 1)
    first = N * third / sum(times[:third])
    last = N * third / sum(times[-third:])
    print(f"{workload:10s} first third {first:9.0f}  last third {last:9.0f} "
          f"patterns/s  slowest epoch {max(times) * 1e3:7.1f} ms  median "
          f"{sorted(times)[len(times) // 2] * 1e3:7.1f} ms  allocated "
          f"{mem0 / 2**20:.0f} -> {mem1 / 2**20:.0f} MiB", flush=True)
    del built, ctx
    torch.cuda.empty_cache()
