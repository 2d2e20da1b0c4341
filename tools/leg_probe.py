"""Round 4: how many calls does a forward-operator leg of bench.py need?
(`gpurun -- python tools/leg_probe.py`)  Prints patterns/s and the roofline
fraction of `forward_leg` for several (warm-up, timed) call counts, in a fresh
process and again after a c3 leg."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import tike_amd._arrays as A  # noqa: E402
import tike_amd.operators as ops  # noqa: E402
import tike_amd.ptycho as tp  # noqa: E402

CASES = ((2, 10), (2, 50), (8, 40), (20, 40), (8, 100))


def sweep(tag):
    for det, S, N in ((256, 1, 4096), (128, 1, 16384), (256, 8, 512)):
        for warm, iters in CASES:
            r = bench.forward_leg(ops, A, torch, det, S, N, iters=iters,
                                  warm=warm)
            print(f"{tag} fwd{det}x{S} warm {warm} timed {iters}: "
                  f"{r['value']:.0f} patterns/s, frac {r['frac']:.3f}",
                  flush=True)


sweep("fresh")
r = bench.epoch_leg("c3", tp, A, torch, 10000, epochs=3)
print(f"c3 {r['value']:.0f} patterns/s")
sweep("after c3")
