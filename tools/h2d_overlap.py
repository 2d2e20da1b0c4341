"""How long does the copy of a minibatch's patterns take while the kernels of
the previous minibatch run?  (`gpurun -- python tools/h2d_overlap.py`)
c3 with data_on_host=True; HIP events around every prefetch on the copy
stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402
from tike_amd.communicators.stream import PinnedData  # noqa: E402

spans = []
issue = PinnedData._issue


def timed(self, lo, hi, s):
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    for event in self._free[s] or ():
        self._copy.wait_event(event)
    a.record(self._copy)
    done = issue(self, lo, hi, s)
    b.record(self._copy)
    spans.append((a, b, (hi - lo) * self.shape[1] * self.shape[2] * 4))
    return done


PinnedData._issue = timed
built = bench.epoch_problem("c3", 0, 1, 0, tp, A, data_on_host=True)
ctx = built["ctx"]
ctx.iterate(2)
torch.cuda.synchronize()
del spans[:]
t0 = time.perf_counter()
ctx.iterate(3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
ms = [a.elapsed_time(b) for a, b, _ in spans]
nbytes = spans[0][2]
print(f"epoch {dt * 1e3:.1f} ms; {len(ms)} copies of {nbytes / 2**20:.0f} MiB: "
      f"mean {sum(ms) / len(ms):.2f} ms = {nbytes / (sum(ms) / len(ms)) / 1e6:.1f} "
      f"GB/s (min {min(ms):.2f}, max {max(ms):.2f} ms); hits "
      f"{ctx.data.hits} of {ctx.data.copies}")
ctx.__exit__(None, None, None)
