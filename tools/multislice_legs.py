"""Round 6: the two-slice rpie workloads of bench.py at 128^2 / 256^2 / 512^2
and under the Poisson model -- the fused chain (`rpie._gradients_multislice_fused`)
against the slice-by-slice composition of the general operators, and
`tike_slice_step` against its two launches.  Prints k patterns/s per leg.

    python tools/multislice_legs.py [workload ...]
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))


def main(names):
    import torch
    import bench
    import tike_amd._arrays as A
    import tike_amd.ptycho as tp
    R = importlib.import_module("tike_amd.ptycho.solvers.rpie")
    for w in names or ("c128rpie2", "c3rpie2", "c5rpie2", "c3rpie2poisson"):
        row = []
        for label, fused, step in (("slice by slice", False, True),
                                   ("fused, separate slice passes", True, False),
                                   ("fused", True, True)):
            R.FUSED_MULTISLICE, R.SLICE_STEP_FUSED = fused, step
            try:
                leg = bench.epoch_leg(w, tp, A, torch, epochs=2)
                row.append(f"{label}: {leg['value'] / 1e3:.1f}")
            finally:
                R.FUSED_MULTISLICE = R.SLICE_STEP_FUSED = True
            torch.cuda.empty_cache()
        print(f"{w}: " + " | ".join(row) + "  (k patterns/s)", flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
