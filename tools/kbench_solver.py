#!/usr/bin/env python3
"""Micro-benchmarks of the lstsq kernels: python tools/kbench_solver.py [--n 256 --modes 8]"""
import argparse, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tike_amd._arrays as A  # noqa
from tike_amd._lib import lib, check  # noqa


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=256)
p.add_argument("--modes", type=int, default=8)
p.add_argument("--pw", type=int, default=256)
a = p.parse_args()
N, S, pw = a.n, a.modes, a.pw
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
side = int(np.ceil(np.sqrt(N)))
HW = 8 * side + pw + 8
scan = torch.tensor(1 + rng.random((N, 2)) * (HW - pw - 3), dtype=torch.float32, device=dev)
psi = torch.randn(1, HW, HW, dtype=torch.complex64, device=dev)
probe = torch.randn(1, 1, S, pw, pw, dtype=torch.complex64, device=dev)
chi = torch.randn(N, 1, S, pw, pw, dtype=torch.complex64, device=dev)
data = torch.rand(N, pw, pw, dtype=torch.float32, device=dev)
obj = torch.zeros_like(psi)
proj = torch.zeros(N, pw, pw, dtype=torch.complex64, device=dev)
acc = torch.zeros(2, HW, HW, dtype=torch.float32, device=dev)
amp = torch.rand(pw, pw, dtype=torch.float32, device=dev)
mpu = torch.zeros_like(probe)
costs = torch.zeros(N, device=dev)
stats = torch.zeros(N, 8, device=dev)
st = A.stream_ptr()
gb = N * S * pw * pw * 8 / 1e9
rows = [
    ("lstsq_gradients", lambda: check(lib.tike_lstsq_gradients(chi.data_ptr(), scan.data_ptr(), psi.data_ptr(), probe.data_ptr(), None, None, 0, 0, None, None, mpu.data_ptr(), proj.data_ptr(), N, S, pw, HW, HW, st)), gb),
    ("scatter_patches", lambda: check(lib.tike_scatter_patches(proj.data_ptr(), scan.data_ptr(), acc.data_ptr(), N, pw, HW, HW, st)), N * pw * pw * 8 / 1e9),
    ("probe_grad", lambda: check(lib.tike_probe_grad(chi.data_ptr(), scan.data_ptr(), psi.data_ptr(), None, mpu.data_ptr(), N, S, pw, HW, HW, st)), gb),
    ("farplane_gradient", lambda: check(lib.tike_farplane_gradient(chi.data_ptr(), data.data_ptr(), None, None, costs.data_ptr(), N, S, pw, 0, 1, 1.0, pw * pw, st)), 2 * gb),
    ("step_stats", lambda: check(lib.tike_lstsq_step_stats(chi.data_ptr(), scan.data_ptr(), psi.data_ptr(), obj.data_ptr(), probe.data_ptr(), None, None, 0, 0, None, mpu.data_ptr(), None, stats.data_ptr(), N, S, S, pw, HW, HW, None, None, st)), N * pw * pw * 8 / 1e9),
    ("psi_precond", lambda: check(lib.tike_psi_preconditioner(amp.data_ptr(), scan.data_ptr(), acc.data_ptr(), N, pw, HW, HW, st)), 0),
]
for name, fn, g in rows:
    ms = timeit(fn)
    print(f"{name:20s} {ms:8.3f} ms  {g / ms * 1e3:8.1f} GB/s  ({N} positions, {S} modes)")
