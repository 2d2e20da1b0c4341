#!/usr/bin/env python3
"""Drop-in use of tike_amd where a script used tike: simulate a small
ptychography data set and reconstruct it with lstsq_grad.

    python examples/reconstruct_synthetic.py                 # this process's GPU
    python examples/reconstruct_synthetic.py --num-gpu 8     # starts 8 ranks itself
    python -m torch.distributed.run --nproc-per-node 8 examples/reconstruct_synthetic.py

The only change against the reference is the import line (`import tike.ptycho`
-> `import tike_amd.ptycho`); arrays are NumPy on the way in and out.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tike_amd.ptycho as tike_ptycho  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--positions", type=int, default=400)
    ap.add_argument("--width", type=int, default=128, help="probe = detector width")
    ap.add_argument("--modes", type=int, default=2)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--num-gpu", type=int, default=None)
    a = ap.parse_args()
    if "RANK" in os.environ:  # launched by torchrun: one rank per GPU
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl")

    rng = np.random.default_rng(0)
    side = int(np.ceil(np.sqrt(a.positions)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:a.positions]
    scan = (2 + 8.0 * ij + rng.random((a.positions, 2))).astype(np.float32)
    width = a.width
    extent = 8 * (side - 1) + width + 8
    psi = ((0.75 + 0.25 * rng.random((1, extent, extent))) * np.exp(
        1j * np.pi * (rng.random((1, extent, extent)) - 0.5))).astype(np.complex64)
    probe = (tike_ptycho.gaussian(width)[None, None, None] *
             np.exp(0.3j * np.pi * rng.random((1, 1, 1, width, width)))).astype(np.complex64)
    np.random.seed(1)
    probe = tike_ptycho.add_modes_random_phase(probe, a.modes)
    probe = tike_ptycho.adjust_probe_power(probe)
    data = tike_ptycho.simulate(width, probe, scan, psi)

    parameters = tike_ptycho.PtychoParameters(
        probe=probe, psi=np.full_like(psi, 0.5), scan=scan,
        algorithm_options=tike_ptycho.LstsqOptions(num_batch=4, num_iter=a.epochs),
        probe_options=tike_ptycho.ProbeOptions(force_orthogonality=True),
        object_options=tike_ptycho.ObjectOptions())
    result = tike_ptycho.reconstruct(data, parameters, num_gpu=a.num_gpu)
    costs = [float(np.mean(c)) for c in result.algorithm_options.costs]
    if os.environ.get("RANK", "0") == "0":
        print("cost per epoch:", " ".join(f"{c:.4g}" for c in costs))
        print(f"psi {result.psi.shape} {result.psi.dtype}, probe {result.probe.shape}")
    return 0 if costs[-1] < costs[0] else 1


if __name__ == "__main__":
    sys.exit(main())
