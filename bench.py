#!/usr/bin/env python3
"""Benchmark of the ptychography hot path on MI355X (DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|fwd256x1|...]

One "step" = one pass of the hot path over the whole synthetic dataset:
  c3 (default, the configuration BASELINE.json's metric is quoted on):
     one lstsq_grad epoch over 10 000 scan positions per GPU, 256x256
     detector, 8 probe modes + eigen-probe correction, 10 minibatches;
  c2: the same with 1 probe mode, no eigen probes;
  fwdDxS: one launch of the fused forward operator (D = detector, S = modes).
Inputs are HBM-resident before the timed region.  Rank 0 prints ONE JSON line.
For N > 1 launch through torch.distributed.run (one rank per GPU, RCCL).
"""
import argparse
import collections
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
METRIC = "diffraction patterns/sec/GPU (256x256, 8-mode probe)"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--workload", default="c3")
    p.add_argument("--positions", type=int, default=0,
                   help="override the number of scan positions per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--breakdown", action="store_true",
                   help="print the per-kernel time breakdown to stderr")
    return p.parse_args()


# ------------------------------------------------------------ synthetic input
def synthetic(N_total, S, det, lo, hi, seed=1234):
    """SURVEY 8(d) generator.  Raster at 8 px pitch + U[0,1) jitter, shuffled
    once; pw = det; psi amplitude 0.75+0.25U, phase pi(U-0.5); probe = radial
    flat-top amplitude x random phase, mode m scaled 1/(m+1).  Every rank
    builds the same global problem and keeps positions [lo, hi) of the
    shuffled order."""
    rng = np.random.default_rng(seed)
    pw = det
    side = int(np.ceil(np.sqrt(N_total)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:N_total]
    scan = (1 + 8.0 * ij + rng.random((N_total, 2))).astype(np.float32)
    rng.shuffle(scan, axis=0)
    HW = int(np.ceil((8 * (side - 1) + pw + 4) / 32.0) * 32)
    amp = 0.75 + 0.25 * rng.random((1, HW, HW), dtype=np.float32)
    ph = np.pi * (rng.random((1, HW, HW), dtype=np.float32) - 0.5)
    psi = (amp * np.exp(1j * ph)).astype(np.complex64)
    r = (np.arange(pw) + 0.5 - pw / 2) / (pw / 2)
    a = np.clip(1.25 - np.sqrt(r[:, None]**2 + r[None, :]**2), 0, 1)
    probe = np.stack([
        a * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
        for m in range(S)
    ])[None, None].astype(np.complex64)
    return dict(scan=scan[lo:hi], psi=psi, probe=probe, HW=HW, pw=pw, det=det)


# ------------------------------------------------------ per-kernel HIP events
class KernelTimers:
    """Brackets every C-ABI launch with HIP events on the launch stream
    (torch's current stream, which is the stream handed to the C ABI)."""

    def __init__(self, lib, names):
        import torch
        self.torch = torch
        self.events = collections.defaultdict(list)
        self.enabled = False
        for name in names:
            fn = getattr(lib, name)

            def wrapper(*args, _fn=fn, _name=name):
                if not self.enabled:
                    return _fn(*args)
                e0 = torch.cuda.Event(enable_timing=True)
                e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = _fn(*args)
                e1.record()
                self.events[_name].append((e0, e1))
                return rc

            setattr(lib, name, wrapper)

    def summary(self):
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[name] = dict(calls=len(ms), total_ms=float(np.sum(ms)),
                             avg_ms=float(np.mean(ms)))
        return out


# -------------------------------------------------------------- CPU baseline
def cpu_baseline_epoch(p, data_np, S, det, seconds=10.0):
    """The oracle's lstsq_grad minibatch on the host cores: a bounded sample
    (as many 32-position minibatches as fit in ~`seconds`)."""
    from oracle import operators as oops
    from oracle import solvers as osol
    cores = os.cpu_count() or 1
    oops.set_workers(cores)
    n = 32
    scan = p["scan"][:n]
    psi0 = np.full_like(p["psi"], 0.5 + 0j)
    pre = osol.psi_preconditioner(psi0, p["probe"], scan)
    t0, done = time.perf_counter(), 0
    while True:
        g = osol.get_nearplane_gradients(
            data_np[:n], psi0, scan, p["probe"], None, None, 0, n,
            num_batch=1, detector_shape=det,
            measured_pixels=np.ones((det, det), dtype=bool))
        osol.precondition_nearplane_gradients(
            g["chi"], scan, g["unique_probe"], p["probe"],
            g["object_upd_sum"], g["m_probe_update"], pre, g["patches"], 0, n)
        done += n
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="patterns/s", cores=cores, kind="port",
                sample=f"oracle lstsq_grad minibatch (gradients + step sizes) "
                f"on {done} positions x {S} modes {det}x{det}, scipy.fft "
                f"workers={cores}")


def cpu_baseline_fwd(p, S, det, seconds=10.0):
    from oracle import operators as oops
    cores = os.cpu_count() or 1
    oops.set_workers(cores)
    n, done, t0 = 16, 0, time.perf_counter()
    while True:
        oops.ptycho_fwd(p["probe"], p["scan"][:n], p["psi"], det)
        done += n
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="patterns/s", cores=cores, kind="port",
                sample=f"oracle Ptycho.fwd on {done} positions x {S} modes "
                f"{det}x{det}, scipy.fft workers={cores}")


# ------------------------------------------------- algorithmic bytes / launch
def algorithmic_bytes(name, n, S, det, pw, C):
    """HBM bytes one launch must move for n positions."""
    T = 8 * S * det * det  # one position's far-plane, bytes
    if name == "tike_ptycho_fwd":
        return n * (T + 8 * pw * pw + 8) + 8 * (S + C) * pw * pw
    if name == "tike_ptycho_fwd_gradient_scale":
        # as below, plus the data read and the gradient-factor write in
        # place of the intensity write
        return n * (T + 8 * pw * pw + 2 * 4 * det * det + 8) + 8 * (S + C) * pw * pw
    if name in ("tike_ptycho_fwd_intensity", "tike_ptycho_fwd_intensity_only"):
        # the intensity-only form hands a far-plane-sized array (the input of
        # its column pass) to tike_grad_ifft2_crop instead of the far plane
        return n * (T + 8 * pw * pw + 4 * det * det + 8) + 8 * (S + C) * pw * pw
    if name in ("tike_ifft2_crop_scaled", "tike_grad_ifft2_crop"):
        return n * (T + 8 * S * pw * pw + 4 * det * det)
    if name == "tike_gradient_scale":
        return n * 3 * 4 * det * det
    if name == "tike_farplane_gradient":
        return n * (2 * T + 4 * det * det + 4)
    if name == "tike_ifft2_crop":
        return n * (T + 8 * S * pw * pw)
    if name == "tike_lstsq_gradients":
        return n * (8 * S * pw * pw + 2 * 8 * pw * pw)
    if name == "tike_scatter_patches":
        return n * (8 * pw * pw + 8 * (pw + 1) * (pw + 1))
    if name == "tike_lstsq_step_stats":
        return n * (3 * 8 * pw * pw + 32)
    return 0


def measured_traffic(workload, kernel, launch_n):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_traffic_<workload>.json: FETCH_SIZE / WRITE_SIZE collected
    in separate --pmc runs of this same command, gfx950-corrected), or None."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles",
                                           f"r*_pmc_traffic_{workload}.json"))):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        k = doc.get("kernels", {}).get(kernel)
        if k and doc.get("positions_per_launch") == launch_n:
            best = k.get("hbm_bytes_per_launch")
    return best


def main():
    a = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    # TIKE_FORCE_COLLECTIVES=1 (under torchrun --nproc-per-node 1) issues the
    # RCCL collectives of the multi-GPU path on a single rank: a dry run of
    # that path on a one-GPU box
    forced = (os.environ.get("TIKE_FORCE_COLLECTIVES") == "1"
              and "MASTER_ADDR" in os.environ)
    if world > 1 or forced:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import tike_amd._arrays as A
    import tike_amd.operators as ops
    import tike_amd.ptycho as tp
    from tike_amd._lib import lib
    from tike_amd.ptycho.solvers.lstsq import chunk_positions

    timers = KernelTimers(lib, [
        "tike_ptycho_fwd", "tike_farplane_gradient", "tike_ifft2_crop",
        "tike_ptycho_fwd_intensity", "tike_gradient_scale",
        "tike_ifft2_crop_scaled", "tike_ptycho_fwd_intensity_only",
        "tike_ptycho_fwd_gradient_scale",
        "tike_grad_ifft2_crop",
        "tike_lstsq_gradients", "tike_scatter_patches",
        "tike_lstsq_step_stats", "tike_psi_preconditioner",
        "tike_probe_preconditioner", "tike_intensity"
    ])
    cpu = None
    C = 0

    if a.workload.startswith("fwd"):
        det, S = [int(v) for v in a.workload[3:].split("x")]
        N = a.positions or max(256, 4096 // S)
        p = synthetic(N * world, S, det, rank * N, (rank + 1) * N)
        op = ops.Ptycho(probe_shape=det, detector_shape=det, nz=p["HW"],
                        n=p["HW"])
        scan, psi, probe = (A.to_device(p[k]) for k in ("scan", "psi", "probe"))
        out = torch.empty((N, 1, S, det, det), dtype=torch.complex64,
                          device=psi.device)

        def step():
            op.fwd_device(probe, scan, psi, out=out)

        units, launch_n, dominant = N, N, "tike_ptycho_fwd"
        if rank == 0 and not a.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_fwd(p, S, det)
        workload = dict(workload=a.workload, positions_per_gpu=N, modes=S,
                        detector=det, solver=None)
    elif a.workload in ("c2", "c3", "c5"):
        # c2: 1 mode; c3 (default, = one GPU's share of c4): 8 modes + eigen
        # probes; c5: 512x512, 4 modes, position correction on
        det = 512 if a.workload == "c5" else 256
        S = {"c2": 1, "c3": 8, "c5": 4}[a.workload]
        N = a.positions or (4000 if a.workload == "c5" else 10000)
        num_batch = 10
        # global problem: N*world positions; this rank's share of every
        # global minibatch, concatenated (see Reconstruction(presharded=True))
        Ng = N * world
        per = Ng // num_batch
        share = per // world
        idx = np.concatenate([
            np.arange(b * per + rank * share, b * per + (rank + 1) * share)
            for b in range(num_batch)
        ])
        full = synthetic(Ng, S, det, 0, Ng)
        p = dict(full, scan=full["scan"][idx])
        N = len(idx)
        np.random.seed(1234 + rank)
        eigen_probe = eigen_weights = None
        if a.workload == "c3":
            import tike_amd.random
            tike_amd.random.randomizer_np = np.random.default_rng(4321)
            eigen_probe, eigen_weights = tp.init_varying_probe(
                p["scan"], p["probe"], num_eigen_probes=2, probes_with_modes=1)
            C = eigen_probe.shape[-4]
        data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
        data_dev = A.to_device(data, np.float32)
        params = tp.PtychoParameters(
            probe=p["probe"].copy(),
            psi=np.full_like(p["psi"], 0.5 + 0j), scan=p["scan"],
            eigen_probe=eigen_probe, eigen_weights=eigen_weights,
            # BASELINE configs[1] names the conjugate-gradient solver
            algorithm_options=(tp.CgradOptions(num_batch=num_batch, cg_iter=4)
                               if a.workload == "c2" else
                               tp.LstsqOptions(num_batch=num_batch,
                                               batch_method="compact")),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions(),
            position_options=tp.PositionOptions(
                p["scan"].copy(), use_adaptive_moment=True,
                update_magnitude_limit=1.0) if a.workload == "c5" else None)
        ctx = tp.Reconstruction(data_dev, params, presharded=True,
                                order=np.arange(N),
                                batches=np.array_split(np.arange(N),
                                                       num_batch))
        ctx.__enter__()

        def step():
            ctx.iterate(1)

        units = N
        launch_n = min(chunk_positions(S, det, True), N // num_batch)
        dominant = None
        if rank == 0 and not a.no_cpu_baseline and world == 1:
            cpu = cpu_baseline_epoch(p, data, S, det)
        workload = dict(workload=a.workload, positions_per_gpu=N, modes=S,
                        detector=det, eigen_probes=C,
                        solver="cgrad (cg_iter=4)" if a.workload == "c2"
                        else "lstsq_grad",
                        num_batch=num_batch, chunk_positions=launch_n,
                        position_correction=a.workload == "c5")
    else:
        raise SystemExit(f"unknown workload {a.workload}")

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timers.enabled = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    timers.enabled = False
    if world > 1:
        t = torch.tensor([wall], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    if rank == 0:
        summ = timers.summary()
        if dominant is None:
            dominant = max(summ, key=lambda k: summ[k]["total_ms"])
        pw = det
        k = summ[dominant]
        nbytes = algorithmic_bytes(dominant, launch_n, S, det, pw, C)
        achieved = nbytes / (k["avg_ms"] * 1e-3) / 1e9
        if a.breakdown:
            tot = sum(v["total_ms"] for v in summ.values())
            for name, v in sorted(summ.items(),
                                  key=lambda kv: -kv[1]["total_ms"]):
                print(f"  {name:28s} calls {v['calls']:5d} avg {v['avg_ms']:8.3f} ms"
                      f" total {v['total_ms']:9.2f} ms ({100 * v['total_ms'] / tot:4.1f}%)",
                      file=sys.stderr)
            print(f"  kernels {tot:.1f} ms of wall {wall * 1e3:.1f} ms",
                  file=sys.stderr)
        line = {
            "metric": METRIC if (det, S) == (256, 8) else
            f"diffraction patterns/sec/GPU ({det}x{det}, {S}-mode probe)",
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
            "config": workload,
            "roofline": {
                "bound": "hbm", "kernel": dominant,
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": measured_traffic(a.workload, dominant, launch_n),
                "algorithmic_bytes": nbytes,
                "avg_launch_ms": k["avg_ms"], "positions_per_launch": launch_n,
                "share_of_kernel_time": k["total_ms"] /
                sum(v["total_ms"] for v in summ.values()),
            },
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
