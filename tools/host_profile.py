"""cProfile of the host side of a bench workload's epochs
(`gpurun -- python tools/host_profile.py c1 [epochs]`)."""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402

workload = sys.argv[1] if len(sys.argv) > 1 else "c1"
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 50
built = bench.epoch_problem(workload, 0, 1, 0, tp, A)
ctx = built["ctx"]
ctx.iterate(10)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
ctx.iterate(epochs)
torch.cuda.synchronize()
pr.disable()
ctx.__exit__(None, None, None)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
