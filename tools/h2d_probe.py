"""Host-to-device rate of pinned copies on this box, one copy stream vs the
chunk split over several (`gpurun -- python tools/h2d_probe.py`)."""
import time

import torch

n = 256 * 2**20 // 4  # 256 MiB of float32: one c3 minibatch of patterns
host = torch.empty(n, dtype=torch.float32).pin_memory()
host.fill_(1.0)
dev = torch.empty(n, dtype=torch.float32, device="cuda")
for streams in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    step = n // streams
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(8):
            for k, s in enumerate(ss):
                with torch.cuda.stream(s):
                    dev[k * step:(k + 1) * step].copy_(
                        host[k * step:(k + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{streams} stream(s): {8 * n * 4 / dt / 1e9:.1f} GB/s", flush=True)
