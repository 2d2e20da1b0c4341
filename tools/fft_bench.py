"""Rate of the batched 2-D transform behind Propagation at any tile size
(csrc/fft2.hip register engines for the powers of two, csrc/fft_mixed.hip for
the rest): tiles/s and GB/s of tile bytes (read once + written once = the
algorithmic traffic of an out-of-place 2-D transform).

    gpurun -- python tools/fft_bench.py [n ...]      (HIP events, 20 launches)
    FFT_GROUPS=r,c forces the general engine with r / c lines per group."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tike_amd import _arrays as A  # noqa: E402
from tike_amd._lib import check, lib  # noqa: E402

sizes = [int(v) for v in sys.argv[1:]] or [
    256, 512, 96, 192, 320, 384, 640, 768, 1000, 2048, 127, 509, 1021]
groups = os.environ.get("FFT_GROUPS")
for n in sizes:
    ntile = max(2, min(4096, (1 << 30) // (8 * n * n)))
    x = torch.randn(ntile, n, n, 2, device="cuda").view(torch.float32)
    x = torch.view_as_complex(x.reshape(ntile, n, n, 2))
    out = torch.empty_like(x)

    def run():
        if groups:
            r, c = (int(v) for v in groups.split(","))
            check(lib.tike_fft2_general(A.ptr(x), A.ptr(out), ntile, n, 0,
                                        1.0 / n, r, c, A.stream_ptr()), "fft")
        else:
            check(lib.tike_fft2(A.ptr(x), A.ptr(out), ntile, n, 0, 1.0 / n,
                                A.stream_ptr()), "fft")

    for _ in range(3):
        run()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    reps = 20
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    gb = 2 * 8 * n * n * ntile / 1e9
    ref = np.fft.fft2(x[:1].cpu().numpy().astype(np.complex128), norm="ortho")
    err = np.linalg.norm(out[:1].cpu().numpy() - ref) / np.linalg.norm(ref)
    print(f"n {n:5d}  tiles {ntile:5d}  {ms:8.3f} ms  {ntile / ms / 1e3:9.3f} "
          f"M tiles/s  {gb / ms * 1e3:8.1f} GB/s of tile bytes (in + out)  "
          f"err {err:.1e}", flush=True)
