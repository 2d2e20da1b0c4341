"""Which torch (not tike_*) kernels run inside a c3 epoch, and from which
Python lines?  `gpurun -- python tools/torch_ops_c3.py`"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402

built = bench.epoch_problem(sys.argv[1] if len(sys.argv) > 1 else "c3", 0, 1,
                            0, tp, A)
ctx = built["ctx"]
ctx.iterate(2)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA],
             with_stack=True) as prof:
    ctx.iterate(1)
    torch.cuda.synchronize()
rows = prof.key_averages(group_by_stack_n=6)
rows = sorted(rows, key=lambda r: -r.device_time_total)
print(f"{'device us':>10} {'calls':>6}  op / stack")
for r in rows[:45]:
    if r.device_time_total <= 0:
        continue
    stack = [s for s in r.stack if "tike_amd" in s or "bench.py" in s][:3]
    print(f"{r.device_time_total:10.0f} {r.count:6d}  {r.key[:60]}")
    for s in stack:
        print(" " * 20 + s[-110:])
