#!/usr/bin/env python3
"""Kernel micro-benchmarks (HIP events) for the FFT-bearing kernels.

python tools/kbench.py [--det 256] [--tiles 2048] [--reps 10]
"""
import argparse
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tike_amd._arrays as A  # noqa: E402
from tike_amd._lib import lib, check  # noqa: E402


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def timeit_gap(fn, reps, gap_s, pre=None):
    """Per-launch events with the GPU left idle for gap_s (or running `pre`)
    before every launch: exposes clock-ramp / cold-cache effects."""
    import time
    tot = 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        time.sleep(gap_s)
        if pre is not None:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
            enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--det", type=int, default=256)
    p.add_argument("--tiles", type=int, default=2048)
    p.add_argument("--reps", type=int, default=10)
    p.add_argument("--modes", type=int, default=1)
    p.add_argument("--hw", type=int, default=0, help="object side (0: small)")
    p.add_argument("--sort", action="store_true", help="row-major sort positions")
    p.add_argument("--c3", action="store_true", help="bench.py's synthetic problem")
    a = p.parse_args()
    n, T, S = a.det, a.tiles, a.modes
    dev = torch.device("cuda", 0)
    x = torch.randn(T, n, n, dtype=torch.complex64, device=dev)
    y = torch.empty_like(x)
    st = A.stream_ptr()
    tile_bytes = n * n * 8
    rows = []

    ms = timeit(lambda: check(lib.tike_fft2(x.data_ptr(), y.data_ptr(), T, n, 0,
                                            1.0 / n, st)), a.reps)
    rows.append(("fft2 out-of-place", ms, 2 * T * tile_bytes))
    ms = timeit(lambda: check(lib.tike_fft2(y.data_ptr(), y.data_ptr(), T, n, 1,
                                            1.0 / n, st)), a.reps)
    rows.append(("ifft2 in-place", ms, 2 * T * tile_bytes))
    ms = timeit(lambda: check(lib.tike_ifft2_crop(y.data_ptr(), y.data_ptr(),
                                                  y.data_ptr(), T, n, n,
                                                  1.0 / n, st)), a.reps)
    rows.append(("ifft2_crop in-place", ms, 2 * T * tile_bytes))
    ms = timeit(lambda: y.copy_(x), a.reps)
    rows.append(("torch copy (HBM ref)", ms, 2 * T * tile_bytes))

    N = T // S
    side = int(np.ceil(np.sqrt(N)))
    HW = a.hw or 8 * side + n + 8
    rng = np.random.default_rng(0)
    sc = 1 + rng.random((N, 2)) * (HW - n - 3)
    if a.sort:
        sc = sc[np.lexsort((sc[:, 1], sc[:, 0] // 64))]
    scan = torch.tensor(sc, dtype=torch.float32, device=dev)
    psi = torch.randn(1, HW, HW, dtype=torch.complex64, device=dev)
    probe = torch.randn(1, 1, S, n, n, dtype=torch.complex64, device=dev)
    if a.c3:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        pr = bench.synthetic(10000, S, n, 0, N)
        HW = pr["HW"]
        scan, psi, probe = (A.to_device(pr[k]) for k in ("scan", "psi", "probe"))
    far = y[:N * S]
    ms = timeit(lambda: check(lib.tike_ptycho_fwd(
        psi.data_ptr(), scan.data_ptr(), probe.data_ptr(), 0, None, None, 0, 0,
        far.data_ptr(), N, S, n, n, HW, HW, 1.0 / n, 0, st)), a.reps)
    rows.append((f"ptycho_fwd S={S}", ms, N * S * tile_bytes + N * tile_bytes))
    if n in (128, 256, 512):
        inten = torch.empty(N, n, n, dtype=torch.float32, device=dev)
        ms = timeit(lambda: check(lib.tike_ptycho_fwd_intensity(
            psi.data_ptr(), scan.data_ptr(), probe.data_ptr(), 0, None, None, 0,
            0, far.data_ptr(), inten.data_ptr(), None, N, S, n, n, HW, HW,
            1.0 / n, st)), a.reps)
        rows.append((f"fwd_intensity S={S}", ms,
                     N * S * tile_bytes + N * tile_bytes))
        uniq = torch.randn(N, 1, n, n, dtype=torch.complex64, device=dev)
        wts = torch.rand(N, 2, S, dtype=torch.float32, device=dev)
        ms = timeit(lambda: check(lib.tike_ptycho_fwd_intensity(
            psi.data_ptr(), scan.data_ptr(), probe.data_ptr(), 0,
            uniq.data_ptr(), wts.data_ptr(), 1, 1, far.data_ptr(),
            inten.data_ptr(), None, N, S, n, n, HW, HW, 1.0 / n, st)), a.reps)
        rows.append((f"fwd_intensity eigen S={S}", ms,
                     N * S * tile_bytes + 2 * N * tile_bytes))
        fwd_e = lambda: check(lib.tike_ptycho_fwd_intensity(
            psi.data_ptr(), scan.data_ptr(), probe.data_ptr(), 0,
            uniq.data_ptr(), wts.data_ptr(), 1, 1, far.data_ptr(),
            inten.data_ptr(), None, N, S, n, n, HW, HW, 1.0 / n, st))
        for gap in (0.0, 0.002, 0.02):
            ms = timeit_gap(fwd_e, a.reps, gap)
            rows.append((f"  .. after {gap*1e3:.0f} ms idle", ms,
                         N * S * tile_bytes + 2 * N * tile_bytes))
        small = torch.zeros(1 << 16, device=dev)
        def pre():
            for _ in range(40):
                small.add_(1.0)
        ms = timeit_gap(fwd_e, a.reps, 0.0, pre)
        rows.append(("  .. after 40 tiny kernels", ms,
                     N * S * tile_bytes + 2 * N * tile_bytes))
        big = torch.empty(1 << 29, device=dev)
        ms = timeit_gap(fwd_e, a.reps, 0.0, lambda: big.fill_(1.0))
        rows.append(("  .. after 2 GiB fill", ms,
                     N * S * tile_bytes + 2 * N * tile_bytes))
        g = torch.rand(N, n, n, dtype=torch.float32, device=dev)
        mid = torch.empty_like(far)
        ms = timeit(lambda: check(lib.tike_ifft2_crop_scaled(
            far.data_ptr(), g.data_ptr(), S, mid.data_ptr(), mid.data_ptr(),
            N * S, n, n, 1.0 / n, st)), a.reps)
        rows.append((f"ifft2_crop_scaled S={S}", ms, 2 * N * S * tile_bytes))
        if n == 256:
            scratch = torch.empty_like(far)
            ms = timeit(lambda: check(lib.tike_ptycho_fwd_intensity_only(
                psi.data_ptr(), scan.data_ptr(), probe.data_ptr(), 0,
                uniq.data_ptr(), wts.data_ptr(), 1, 1, scratch.data_ptr(),
                inten.data_ptr(), N, S, n, n, HW, HW, 1.0 / n, st)), a.reps)
            rows.append((f"fwd intensity-only S={S}", ms,
                         N * S * tile_bytes + 2 * N * tile_bytes))
            ms = timeit(lambda: check(lib.tike_grad_ifft2_crop(
                scratch.data_ptr(), g.data_ptr(), None, None, S,
                mid.data_ptr(), mid.data_ptr(), N * S, n, n, 1.0 / n, 1.0 / n,
                st)), a.reps)
            rows.append((f"grad_ifft2_crop S={S}", ms, 2 * N * S * tile_bytes))
        chi = torch.empty_like(far)
        ms = timeit(lambda: check(lib.tike_ifft2_crop_scaled(
            far.data_ptr(), g.data_ptr(), S, mid.data_ptr(), chi.data_ptr(),
            N * S, n, n, 1.0 / n, st)), a.reps)
        rows.append((f"ifft2_crop_scaled sep. chi", ms, 2 * N * S * tile_bytes))
    for name, ms, nbytes in rows:
        print(f"{name:28s} {ms:8.3f} ms  {T / ms / 1e3:8.3f} Mtile/s  "
              f"{nbytes / ms / 1e6:8.1f} GB/s (algorithmic)")


if __name__ == "__main__":
    main()
