#!/usr/bin/env python3
"""Forward pass 1 -> column pass + gradient factor -> gradient + inverse pass 1
of a c3-shaped minibatch, run over sub-chunks of m positions that share ONE
small hand-off buffer: does the hand-off stay in the 256 MB Infinity Cache?
Usage: python tools/mall_pipeline_probe.py [--n 960] [--S 8]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tike_amd._arrays as A  # noqa: E402
from tike_amd._lib import check, lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=960)
    ap.add_argument("--S", type=int, default=8)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--streams", type=int, default=1)
    a = ap.parse_args()
    N, S, det = a.n, a.S, a.det
    pw = det
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(0)
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:N]
    scan_np = (1 + 8.0 * ij + rng.random((N, 2))).astype(np.float32)
    HW = int(np.ceil((8 * (side - 1) + pw + 4) / 32.0) * 32)
    from tike_amd import cluster
    scan = A.to_device(scan_np[cluster.spatial_order(scan_np)])
    c = lambda *s: torch.randn(*s, dtype=torch.complex64, device=dev)
    psi, probe = c(1, HW, HW), c(1, 1, S, pw, pw)
    work = c(N, 1, S, det, det)
    data = torch.rand(N, det, det, device=dev)
    costs = torch.empty(N, device=dev)
    patches = c(N, pw, pw)
    st = A.stream_ptr()
    p = A.ptr
    for m in (N, 480, 240, 120, 96, 64, 48, 32, 24, 16):
        K = a.streams
        scratches = [c(m, 1, S, det, det) for _ in range(K)]
        gscales = [torch.empty(m, det, det, device=dev) for _ in range(K)]
        streams = [torch.cuda.Stream() for _ in range(K)] if K > 1 else [None]

        def chain():
            if K > 1:
                main = torch.cuda.current_stream()
                for sx in streams:
                    sx.wait_stream(main)
            for i, lo in enumerate(range(0, N, m)):
                n = min(m, N - lo)
                scratch, gscale = scratches[i % K], gscales[i % K]
                st = (streams[i % K].cuda_stream if K > 1 else A.stream_ptr())
                check(lib.tike_fwd_pass1(p(psi), p(scan[lo:lo + n]), p(probe), 0,
                                         None, None, None, 0, 0, p(scratch),
                                         p(patches[lo:lo + n]), n, S, pw, det,
                                         HW, HW, st))
                check(lib.tike_fwd_gradient_scale(
                    p(scratch), p(data[lo:lo + n]), 0, None, p(gscale), None,
                    p(costs[lo:lo + n]), None, n, S, det, 1.0 / det, 0, 1.0,
                    det * det, st))
                check(lib.tike_grad_ifft2_pass1(
                    p(scratch), p(gscale), None, None, S, p(work[lo:lo + n]),
                    n * S, det, 1.0 / det, st))
            if K > 1:
                for sx in streams:
                    main.wait_stream(sx)

        for _ in range(2):
            chain()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        reps = 5
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            chain()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"m = {m:5d}  hand-off {m * 8 * S * det * det / 2**20:7.0f} MiB  "
              f"{ms:7.3f} ms per {N} positions  ({ms * 1000 / N:.3f} ms / 1000)")
        del scratches, gscales


if __name__ == "__main__":
    main()
