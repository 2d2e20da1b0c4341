"""Two ranks sharing one GPU over gloo run cgrad: which line search did they
take, and what does an epoch cost against one rank?
python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29551 tools/cgrad_two_ranks.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

torch.cuda.set_device(0)
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo")
rank = dist.get_rank() if world > 1 else 0

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402
from tike_amd import _lib  # noqa: E402

calls = {"linear": 0, "trial": 0}
for name, key in (("tike_cgrad_line_search_linear", "linear"),
                  ("tike_cgrad_line_search", "trial")):
    real = getattr(_lib.lib, name)

    def counted(*a, _real=real, _key=key):
        calls[_key] += 1
        return _real(*a)

    setattr(_lib.lib, name, counted)
import tike_amd.ptycho.solvers.cgrad as C  # noqa: E402
C.lib = _lib.lib

built = bench.epoch_problem("c2", 2000, world, rank, tp, A)
ctx = built["ctx"]
ctx.iterate(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
ctx.iterate(3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
costs = [c[0] for c in ctx.parameters.algorithm_options.costs]
print(f"rank {rank} of {world}: epoch {dt * 1e3:.1f} ms for {built['N']} "
      f"positions per rank; all-at-once searches (stages) {calls['linear']}, "
      f"trial-by-trial {calls['trial']}; costs "
      + " ".join(f"{c:.3e}" for c in costs), flush=True)
ctx.__exit__(None, None, None)
if world > 1:
    dist.destroy_process_group()
