"""tike_fft2 of round 5 (powers of two on the register engines, everything
else a direct O(n^2) DFT per line, nothing above 1024) beside this tree's
(mixed-radix / Bluestein lines in LDS), same box, same buffers: the two
libraries loaded side by side through ctypes.

    python tools/build_variant.py ...   # or: git archive <r05> | make -> tools/probe/_lib/lib_r05.so
    gpurun -- python tools/fft_before_after.py [n ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

libs = {"round 5": os.path.join(ROOT, "tools/probe/_lib/lib_r05.so"),
        "round 6": os.path.join(ROOT, "tike_amd/csrc/libtike_amd.so")}
sizes = [int(v) for v in sys.argv[1:]] or [96, 127, 192, 320, 384, 640, 768,
                                           1000, 2048]
print(f"{'n':>6s} {'tiles':>6s} " + " ".join(f"{k + ' ms':>14s} {'M tiles/s':>10s}"
                                              for k in libs) + "   speed-up")
for n in sizes:
    ntile = max(2, min(2048, (1 << 29) // (8 * n * n)))
    x = torch.view_as_complex(torch.randn(ntile, n, n, 2, device="cuda"))
    out = torch.empty_like(x)
    row = []
    for name, path in libs.items():
        lib = ctypes.CDLL(path)
        lib.tike_fft2.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                  ctypes.c_void_p]
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: lib.tike_fft2(x.data_ptr(), out.data_ptr(), ntile, n, 0,
                                     1.0 / n, st)
        rc = call()
        if rc:
            row.append(None)
            continue
        torch.cuda.synchronize()
        ref = np.fft.fft2(x[:1].cpu().numpy().astype(np.complex128), norm="ortho")
        err = np.linalg.norm(out[:1].cpu().numpy() - ref) / np.linalg.norm(ref)
        assert err < 2e-6, (name, n, err)
        reps = 3 if name == "round 5" and n not in (256, 512) else 20
        e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / reps)
    cells = " ".join(f"{'refused':>14s} {'-':>10s}" if ms is None else
                     f"{ms:14.3f} {ntile / ms / 1e3:10.4f}" for ms in row)
    gain = "" if None in row else f"{row[0] / row[1]:8.1f} x"
    print(f"{n:6d} {ntile:6d} {cells}   {gain}", flush=True)
