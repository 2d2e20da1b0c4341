"""tike_fwd_grad_ifft2_pass1 vs tike_fwd_gradient_scale + tike_grad_ifft2_pass1:
same intermediate and costs, and the time of each (8000 / S positions, S modes
[argv 1, default 8], 256^2)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tike_amd._arrays as A
from tike_amd._lib import check, lib

S = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N, det = 8000 // S, 256
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
scratch = torch.view_as_complex(torch.rand(N, 1, S, det, det, 2, device=dev, generator=g) - 0.5)
data = torch.rand(N, det, det, device=dev, generator=g) * 30
gs = torch.empty(N, det, det, device=dev)
c1, c2 = torch.empty(N, device=dev), torch.empty(N, device=dev)
w1, w2 = torch.empty_like(scratch), torch.empty_like(scratch)
st = A.stream_ptr()
p = A.ptr

def separate():
    check(lib.tike_fwd_gradient_scale(p(scratch), p(data), 0, None, p(gs), None, p(c1), None,
                                      N, S, det, 1.0 / det, 0, 1.0, det * det, st))
    check(lib.tike_grad_ifft2_pass1(p(scratch), p(gs), None, None, S, p(w1), N * S, det,
                                    1.0 / det, st))

def fused():
    check(lib.tike_fwd_grad_ifft2_pass1(p(scratch), p(data), 0, None, p(c2), p(w2), N, S, det,
                                        1.0 / det, 0, 1.0, det * det, st))

separate(); fused(); torch.cuda.synchronize()
print("costs max rel diff", float(((c1 - c2).abs() / c1.abs()).max()))
print("work max abs diff / max", float((w1 - w2).abs().max()) / float(w1.abs().max()))
for name, f in (("separate", separate), ("fused", fused)):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:10s} {e0.elapsed_time(e1) / 10:.3f} ms")
