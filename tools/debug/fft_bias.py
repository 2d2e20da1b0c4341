import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from tike_amd import _arrays as A
from tike_amd._lib import check, lib
for n in (256, 512, 1024, 384, 768):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((4, n, n)) + 1j * rng.standard_normal((4, n, n))).astype(np.complex64)
    ref = np.fft.fft2(x.astype(np.complex128), norm="ortho")
    xt = A.to_device(x, np.complex64); out = torch.empty_like(xt)
    for name, call in (("tike_fft2", lambda: lib.tike_fft2(A.ptr(xt), A.ptr(out), 4, n, 0, 1.0 / n, A.stream_ptr())),
                       ("general", lambda: lib.tike_fft2_general(A.ptr(xt), A.ptr(out), 4, n, 0, 1.0 / n, 0, 0, A.stream_ptr()))):
        check(call(), name)
        o = out.cpu().numpy().astype(np.complex128)
        I, Ir = np.abs(o)**2, np.abs(ref)**2
        print(n, name, "normwise", np.linalg.norm(o - ref) / np.linalg.norm(ref), "intensity bias", (I.sum() - Ir.sum()) / Ir.sum())
    import scipy.fft
    o = scipy.fft.fft2(x, norm="ortho").astype(np.complex128)
    print(n, "scipy f32", np.linalg.norm(o - ref) / np.linalg.norm(ref), "intensity bias", ((np.abs(o)**2).sum() - (np.abs(ref)**2).sum()) / (np.abs(ref)**2).sum())
