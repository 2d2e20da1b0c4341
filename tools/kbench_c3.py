#!/usr/bin/env python3
"""Per-kernel timing of one c3-shaped lstsq_grad minibatch (random operands):
the kernels bench.py's default workload runs, one at a time, with the bytes
each must move.  Usage: python tools/kbench_c3.py [--n 1000] [--S 8] [--det 256]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tike_amd._arrays as A  # noqa: E402
from tike_amd._lib import check, lib  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000)
    ap.add_argument("--S", type=int, default=8)
    ap.add_argument("--det", type=int, default=256)
    ap.add_argument("--eigen", type=int, default=1)
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
    scan_np = scan_np[cluster.spatial_order(scan_np)]
    scan = A.to_device(scan_np)
    c = lambda *s: torch.randn(*s, dtype=torch.complex64, device=dev)
    psi, probe = c(1, HW, HW), c(1, 1, S, pw, pw)
    eig = c(1, a.eigen, 1, pw, pw) if a.eigen else None
    w = (1 + 0.1 * torch.randn(N, a.eigen + 1, S, device=dev)) if a.eigen else None
    C, Sm = (a.eigen, 1) if a.eigen else (0, 0)
    far, mid = c(N, 1, S, det, det), c(N, 1, S, det, det)
    data = torch.rand(N, det, det, device=dev)
    gscale = torch.empty(N, det, det, device=dev)
    costs = torch.empty(N, device=dev)
    patches, objproj, chi0 = c(N, pw, pw), c(N, pw, pw), c(N, pw, pw)
    mpu = torch.zeros(1, 1, S, pw, pw, dtype=torch.complex64, device=dev)
    acc = torch.zeros(2, HW, HW, device=dev)
    uq = c(N, 1, pw, pw) if a.eigen else None
    stats = torch.empty(N, 8, device=dev)
    st = A.stream_ptr()
    p = A.ptr
    T = 8 * S * det * det
    P = 8 * pw * pw
    D = 4 * det * det
    rows = []

    def run(name, fn, nbytes):
        ms = timeit(fn)
        rows.append((name, ms, nbytes))

    if a.eigen:
        run("varying_probe", lambda: check(lib.tike_varying_probe(
            p(probe), p(eig), p(w), C, Sm, p(uq), N, S, pw, st)), N * P)
    if det == 256:
        run("fwd_gradient_scale", lambda: check(lib.tike_ptycho_fwd_gradient_scale(
            p(psi), p(scan), p(probe), 0, p(uq), p(w), C, Sm, p(far), None,
            p(patches), p(data), None, p(gscale), p(costs), N, S, pw, det, HW,
            HW, 1.0 / det, 0, 1.0, det * det, st)), N * (T + 2 * P + 2 * D))
        run("fwd_pass1 (unique)", lambda: check(lib.tike_fwd_pass1(
            p(psi), p(scan), p(probe), 0, p(uq), None, p(w), C, Sm, p(far),
            p(patches), N, S, pw, det, HW, HW, st)), N * (T + 2 * P))
        run("fwd_pass1 (eigen on the fly)", lambda: check(lib.tike_fwd_pass1(
            p(psi), p(scan), p(probe), 0, None, p(eig), p(w), C, Sm, p(far),
            p(patches), N, S, pw, det, HW, HW, st)), N * (T + 2 * P))
        run("fwd_gradient_scale (split)", lambda: check(lib.tike_fwd_gradient_scale(
            p(far), p(data), 0, None, p(gscale), None, p(costs), None, N, S, det,
            1.0 / det, 0, 1.0, det * det, st)), N * (T + 2 * D))
        run("grad_ifft2_pass1", lambda: check(lib.tike_grad_ifft2_pass1(
            p(far), p(gscale), None, None, S, p(mid), N * S, det, 1.0 / det,
            st)), N * (2 * T + D))
        run("grad_ifft2_crop (old)", lambda: check(lib.tike_grad_ifft2_crop(
            p(far), p(gscale), None, None, S, p(mid), p(mid), N * S, det, pw,
            1.0 / det, 1.0 / det, st)), N * (2 * T + D))
    else:
        inten = torch.empty(N, det, det, device=dev)
        run("fwd_intensity", lambda: check(lib.tike_ptycho_fwd_intensity(
            p(psi), p(scan), p(probe), 0, p(uq), p(w), C, Sm, p(far), p(inten),
            p(patches), N, S, pw, det, HW, HW, 1.0 / det, st)),
            N * (T + 2 * P + D))
        run("ifft2_pass1_scaled", lambda: check(lib.tike_ifft2_pass1_scaled(
            p(far), p(gscale), None, None, S, p(mid), N * S, det, st)),
            N * (2 * T + D))
    run("pass2_gradients", lambda: check(lib.tike_ifft2_pass2_gradients(
        p(mid), p(patches), p(probe), p(eig), p(w), C, Sm, p(objproj),
        p(chi0), p(mpu), 1.0, N, S, det, 1.0 / det, st)), N * (T + 3 * P))
    run("pass2_gradients (no eigen)", lambda: check(lib.tike_ifft2_pass2_gradients(
        p(mid), p(patches), p(probe), None, None, 0, 0, p(objproj),
        p(chi0), p(mpu), 1.0, N, S, det, 1.0 / det, st)), N * (T + 3 * P))
    run("pass2_gradients (probe only)", lambda: check(lib.tike_ifft2_pass2_gradients(
        p(mid), p(patches), None, None, None, 0, 0, None, None, p(mpu),
        1.0, N, S, det, 1.0 / det, st)), N * (T + P))
    run("lstsq_gradients (old)", lambda: check(lib.tike_lstsq_gradients(
        p(mid), p(scan), p(psi), p(probe), p(eig), p(w), C, Sm, None,
        p(patches), p(mpu), p(objproj), N, S, pw, HW, HW, st)), N * (T + 2 * P))
    run("scatter_patches", lambda: check(lib.tike_scatter_patches(
        p(objproj), p(scan), p(acc), N, pw, HW, HW, st)), N * P)
    gob = c(1, HW, HW)
    mpu0 = c(pw, pw)
    run("step_stats", lambda: check(lib.tike_lstsq_step_stats(
        p(chi0), p(scan), p(psi), p(gob), p(probe), p(eig), p(w), C, Sm, None,
        p(mpu), p(patches), p(stats), N, S, 1, pw, HW, HW,
        p(eig[0, 0, 0]) if a.eigen else None,
        p(costs) if a.eigen else None, st)), N * 3 * P)
    total = 0.0
    for name, ms, nb in rows:
        print(f"{name:32s} {ms:8.3f} ms   {nb / ms / 1e6:8.1f} GB/s algorithmic")
    print("positions", N, "modes", S, "det", det)


if __name__ == "__main__":
    main()
