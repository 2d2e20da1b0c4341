#!/usr/bin/env python3
"""Benchmark of the ptychography hot path on MI355X (see DESIGN.md "Measurement").

python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c5|fwd]

One "step" = one pass of the hot path over one batch of synthetic input
(HBM-resident before the timed region).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default="fwd256x8")
    p.add_argument("--positions", type=int, default=0,
                   help="override the number of scan positions per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    return p.parse_args()


def synthetic(N, S, det, seed=1234, device=None):
    """SURVEY 8(d) generator: raster at 8 px pitch + U[0,1) jitter, pw = det."""
    import torch
    rng = np.random.default_rng(seed)
    pw = det
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:N]
    scan = (1 + 8.0 * ij + rng.random((N, 2))).astype(np.float32)
    rng.shuffle(scan, axis=0)
    HW = int(np.ceil((8 * (side - 1) + pw + 4) / 32.0) * 32)
    psi = ((0.75 + 0.25 * rng.random((1, HW, HW))) * np.exp(
        1j * np.pi * (rng.random((1, HW, HW)) - 0.5))).astype(np.complex64)
    r = (np.arange(pw) + 0.5 - pw / 2) / (pw / 2)
    amp = np.clip(1.25 - np.sqrt(r[:, None]**2 + r[None, :]**2), 0, 1)
    probe = np.stack([
        amp * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
        for m in range(S)
    ])[None, None].astype(np.complex64)
    return dict(scan=scan, psi=psi, probe=probe, HW=HW, pw=pw, det=det)


def cpu_baseline_fwd(p, S, det, seconds=10.0):
    """Oracle forward operator on the host cores (bounded sample)."""
    from oracle import operators as oracle
    cores = os.cpu_count() or 1
    oracle.set_workers(cores)
    n = 16
    t0 = time.perf_counter()
    done = 0
    while True:
        oracle.ptycho_fwd(p["probe"], p["scan"][:n], p["psi"], det)
        done += n
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="patterns/s", cores=cores, kind="port",
                sample=f"oracle Ptycho.fwd on {done} positions x {S} modes "
                f"{det}x{det}, scipy.fft workers={cores}")


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import tike_amd.operators as ops
    import tike_amd._arrays as A

    if a.workload.startswith("fwd"):
        det, S = [int(v) for v in a.workload[3:].split("x")]
        N = a.positions or (2048 if S == 1 else 1024)
        p = synthetic(N, S, det, seed=1234 + rank)
        op = ops.Ptycho(probe_shape=det, detector_shape=det, nz=p["HW"],
                        n=p["HW"])
        scan, psi, probe = (A.to_device(p[k]) for k in ("scan", "psi", "probe"))
        out = torch.empty((N, 1, S, det, det), dtype=torch.complex64,
                          device=psi.device)

        def step():
            op.fwd_device(probe, scan, psi, out=out)

        units = N
        alg_bytes = N * (8 * S * det * det + 8 * det * det + 8) + 8 * S * det * det
        kernel = f"ptycho_fwd_kernel<{det}>"
        launches = 1
    else:
        raise SystemExit(f"unknown workload {a.workload}")

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(
        enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(a.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    kern_ms = ev0.elapsed_time(ev1) / (a.steps * launches)
    if world > 1:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "diffraction patterns/sec/GPU (256x256, 8-mode probe)",
            "value": units * world * a.steps / wall,
            "unit": "patterns/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "c64",
            "data": "synthetic",
            "config": {"workload": a.workload, "positions_per_gpu": units,
                       "modes": S, "detector": det},
            "roofline": {"bound": "hbm", "kernel": kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None},
        }
        if not a.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline_fwd(p, S, det)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
