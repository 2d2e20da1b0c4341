#!/usr/bin/env python3
"""Do a write-heavy and a read-heavy kernel of the c3 pipeline run faster
side by side (two streams) than back to back?  Times tike_fwd_pass1 (writes T)
against tike_ifft2_pass2_gradients / tike_fwd_gradient_scale (read T)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import tike_amd._arrays as A  # noqa: E402
from tike_amd._lib import check, lib  # noqa: E402

N, S, det = 500, 8, 256
pw = det
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
side = int(np.ceil(np.sqrt(N)))
ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
              -1).reshape(-1, 2)[:N]
scan = A.to_device((1 + 8.0 * ij + rng.random((N, 2))).astype(np.float32))
HW = int(np.ceil((8 * (side - 1) + pw + 4) / 32.0) * 32)
c = lambda *s: torch.randn(*s, dtype=torch.complex64, device=dev)
psi, probe = c(1, HW, HW), c(1, 1, S, pw, pw)
farA, farB, midB = c(N, 1, S, det, det), c(N, 1, S, det, det), c(N, 1, S, det, det)
data = torch.rand(N, det, det, device=dev)
gscale = torch.empty(N, det, det, device=dev)
patches, objproj, chi0 = c(N, pw, pw), c(N, pw, pw), c(N, pw, pw)
mpu = torch.zeros(1, 1, S, pw, pw, dtype=torch.complex64, device=dev)
p = A.ptr
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def f1(st):
    check(lib.tike_fwd_pass1(p(psi), p(scan), p(probe), 0, None, None, None, 0,
                             0, p(farA), p(patches), N, S, pw, det, HW, HW, st))


def f2(st):
    check(lib.tike_fwd_gradient_scale(p(farB), p(data), 0, None, p(gscale), None,
                                      None, None, N, S, det, 1.0 / det, 0, 1.0,
                                      det * det, st))


def g1(st):
    check(lib.tike_grad_ifft2_pass1(p(farB), p(gscale), None, None, S, p(midB),
                                    N * S, det, 1.0 / det, st))


def p2g(st):
    check(lib.tike_ifft2_pass2_gradients(p(midB), p(patches), p(probe), None,
                                         None, 0, 0, p(objproj), p(chi0),
                                         p(mpu), 1.0, N, S, det, 1.0 / det, st))


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def pair(a, b):
    def both():
        s1.wait_stream(torch.cuda.current_stream())
        s2.wait_stream(torch.cuda.current_stream())
        a(s1.cuda_stream)
        b(s2.cuda_stream)
        torch.cuda.current_stream().wait_stream(s1)
        torch.cuda.current_stream().wait_stream(s2)
    return both


cur = lambda: torch.cuda.current_stream().cuda_stream
names = dict(f1=f1, f2=f2, g1=g1, p2g=p2g)
single = {k: timed(lambda fn=fn: fn(cur())) for k, fn in names.items()}
for k, v in single.items():
    print(f"{k:4s} alone {v:.3f} ms")
for a, b in (("f1", "f2"), ("f1", "p2g"), ("f1", "g1"), ("g1", "p2g"),
             ("f2", "p2g"), ("f1", "f1")):
    t = timed(pair(names[a], names[b]))
    print(f"{a}+{b}: side by side {t:.3f} ms vs back to back "
          f"{single[a] + single[b]:.3f} ms")
