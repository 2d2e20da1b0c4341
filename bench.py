#!/usr/bin/env python3
"""Benchmark of the ptychography hot path on MI355X (DESIGN.md "Measurement").

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c5|fwdDxS]

One "step" = one pass of the hot path over the whole synthetic dataset:
  c3 (default, the configuration BASELINE.json's metric is quoted on):
     one lstsq_grad epoch over 10 000 scan positions per GPU, 256x256
     detector, 8 probe modes + eigen-probe correction, 10 minibatches;
  c1: BASELINE configs[0] (256 positions, 128x128, 1 mode, cgrad);
  c2: 1 probe mode, cgrad;  c5: 512x512, 4 modes, position correction;
  c3poisson / c3rpie / c3rpie2: c3's shapes with the Poisson noise model /
     solved by rpie / by rpie on a two-slice object (SURVEY 8 rows f2, f3:
     measured, not BASELINE configurations);
  fwdDxS: one launch of the fused forward operator (D = detector, S = modes);
  adjDxS: one call of the fused adjoint operator (stored far plane ->
     object gradient + per-position probe gradients).
Inputs are HBM-resident before the timed region.  Rank 0 prints ONE JSON line.

Multi-GPU: one process per GPU over RCCL.  Either launch through
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...``
(what the driver does) or just run ``python bench.py --gpus N``: with
WORLD_SIZE unset this process -- before it touches any GPU -- starts the N
ranks as a child ``torch.distributed.run`` and exits with its code.  Fewer
visible GPUs than requested is an error, never a silent 1-GPU run.
"""
import argparse
import collections
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: vector FP32 peak
READ_CEILING_GBS = 6450.0  # tools/probe/bw_probe.hip: best pure read stream


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=None,
                   help="timed steps (default 10; forward workloads and c1: 40)")
    p.add_argument("--warmup", type=int, default=None,
                   help="untimed steps in front (default 2; forward workloads and "
                   "c1 10: their steps take 1-3 ms and the clocks of an idle GPU "
                   "need a few of them, profiles/r04_leg_probe.txt)")
    p.add_argument("--workload", default="c3")
    p.add_argument("--positions", type=int, default=0,
                   help="override the number of scan positions per GPU")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-secondary", action="store_true",
                   help="skip the short forward-operator legs")
    p.add_argument("--data-on-host", action="store_true",
                   help="stream the patterns from pinned host memory (the "
                        "PCIe-inclusive rate; never the headline value)")
    p.add_argument("--allreduce-probe", action="store_true",
                   help="after the timed steps, time the object-gradient-sized "
                        "all-reduce alone (100 calls): link time apart from "
                        "compute in a multi-GPU line")
    p.add_argument("--breakdown", action="store_true",
                   help="print the per-kernel time breakdown to stderr")
    a = p.parse_args()
    short = a.workload.startswith(("fwd", "adj")) or a.workload == "c1"  # ms-sized steps
    if a.steps is None:
        a.steps = 40 if short else 10
    if a.warmup is None:
        a.warmup = 10 if short else 2
    return a


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks as a CHILD
    torch.distributed.run.  Runs before anything initialises the GPU in this
    process (device_count() does not), and this process never re-execs."""
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} requested but only {have} GPU(s) "
                 "are visible; refusing to report a smaller run")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


# ------------------------------------------------------------ synthetic input
def synthetic(N_total, S, det, lo, hi, seed=1234, pw=None):
    """SURVEY 8(d) generator (the build's own code).  Raster at 8 px pitch +
    U[0,1) jitter, shuffled once; pw = det (unless given: the off-grid
    workload c3pad has a probe window half the detector, as the reference's
    own test generator does); psi amplitude 0.75+0.25U, phase
    pi(U-0.5); probe mode 0 = flat-top radial amplitude (rin 0.8, rout 1.0)
    x exp(i pi U) with U smoothed by a 5x5 box; mode m > 0 = mode 0 x a
    random linear phase ramp (tilts of up to half a period across the
    window), scaled 1/(m+1).  Every rank builds the same global problem and
    keeps positions [lo, hi) of the shuffled order."""
    import scipy.ndimage
    rng = np.random.default_rng(seed)
    pw = pw or det
    side = int(np.ceil(np.sqrt(N_total)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:N_total]
    scan = (1 + 8.0 * ij + rng.random((N_total, 2))).astype(np.float32)
    rng.shuffle(scan, axis=0)
    HW = int(np.ceil((8 * (side - 1) + pw + 4) / 32.0) * 32)
    amp = 0.75 + 0.25 * rng.random((1, HW, HW), dtype=np.float32)
    ph = np.pi * (rng.random((1, HW, HW), dtype=np.float32) - 0.5)
    psi = (amp * np.exp(1j * ph)).astype(np.complex64)
    # flat top of radius 0.8, linear ramp to zero one pixel beyond radius 1.0
    # (in units of the half width), pixel-centre radii
    r = np.hypot(*np.meshgrid(np.arange(pw) + 0.5 - pw / 2,
                              np.arange(pw) + 0.5 - pw / 2, indexing="ij"))
    reach = (pw - 1) / 2  # sqrt(2)/2 x the pixel-centre half diagonal
    inner, outer = 0.8 * reach, 1.0 * reach + 1.0
    window = np.clip((outer - r) / (outer - inner), 0.0, 1.0)
    phase = scipy.ndimage.uniform_filter(rng.random((pw, pw)), size=5,
                                         mode="wrap")
    mode0 = window * np.exp(1j * np.pi * phase)
    t = (np.arange(pw) + 0.5) / pw - 0.5
    modes = [mode0]
    for m in range(1, S):
        tilt = rng.random(2) - 0.5
        ramp = np.exp(-2j * np.pi * (tilt[0] * t[None, :] + tilt[1] * t[:, None]))
        modes.append(mode0 * ramp / (m + 1))
    probe = np.stack(modes)[None, None].astype(np.complex64)
    return dict(scan=scan[lo:hi], psi=psi, probe=probe, HW=HW, pw=pw, det=det)


# ------------------------------------------------------ per-kernel HIP events
# entry -> index of the output argument whose absence marks a cost-only launch
COST_ONLY_ARG = {"tike_fwd_pass1": 10, "tike_fwd_gradient_scale": 4}
# entry -> index of its probe_per_scan argument: a launch whose probe is one
# wave per position (the slices of a multislice object behind the first) reads
# a far-plane-sized array more: own line (":incident")
INCIDENT_ARG = {"tike_fwd_pass1": 3, "tike_ifft2_pass2_products": 4}
# entry -> (index of its position / tile count, tiles per position or None):
# launches of one entry differ in size in the multislice workload (the object
# preconditioner's chunks, the gradient chunks), so the roofline of such an
# entry is priced on the positions its launches really covered
POSITIONS_ARG = {"tike_fwd_pass1": (11, None), "tike_ifft2_pass2_products": (10, None),
                 "tike_fwd_grad_ifft2_pass1_slices": (6, None),
                 "tike_fresnel_colpass": (4, "S"), "tike_fft2_pass2_inplace": (1, "S"),
                 "tike_fft2_pass2_intensity": (2, None)}
MULTISLICE_DEPTH = 1  # set by the workload (byte model of the sliced gradient pass)


class KernelTimers:
    """Brackets every C-ABI launch with HIP events on the launch stream
    (torch's current stream, which is the stream handed to the C ABI).  The
    events come from a pool allocated before the timed region."""

    def __init__(self, lib, names, pool=8192):
        import torch
        self.events = collections.defaultdict(list)
        self.enabled = False
        self.only = None  # bracket these entries only (None: all of them)
        self.label_cost_only = False  # (cgrad workloads: see COST_ONLY_ARG)
        self.cost_only_suffix = ":cost_only"
        self.modes = 1  # tiles per position of the entries counted in tiles
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(pool)]
        self.next = 0
        for name in names:
            fn = getattr(lib, name)

            def wrapper(*args, _fn=fn, _name=name):
                if not self.enabled or (self.only is not None
                                        and _name not in self.only):
                    return _fn(*args)
                import torch
                if torch.cuda.is_current_stream_capturing():
                    return _fn(*args)  # a graph capture: nothing to time
                e0, e1 = self.pair()
                e0.record()
                rc = _fn(*args)
                e1.record()
                # cost-only launches (cgrad's line search: no patches / no
                # gradient factor stored) move fewer bytes: own line
                out = COST_ONLY_ARG.get(_name) if self.label_cost_only else None
                inc = INCIDENT_ARG.get(_name)
                pos = POSITIONS_ARG.get(_name)
                n = None
                if pos is not None:
                    n = int(args[pos[0]]) / (self.modes if pos[1] else 1)
                if inc is not None and args[inc]:
                    _name = _name + ":incident"
                elif out is not None and args[out] is None:
                    _name = _name + self.cost_only_suffix
                self.events[_name].append((e0, e1, n))
                return rc

            setattr(lib, name, wrapper)

    def pair(self):
        import torch
        if self.next + 2 > len(self.pool):
            self.pool += [torch.cuda.Event(enable_timing=True)
                          for _ in range(1024)]
        e = self.pool[self.next:self.next + 2]
        self.next += 2
        return e

    def wrap_method(self, obj, method, name):
        """Time a Python-level call (the RCCL all-reduce) the same way.  `name`
        may be a function of the call's arguments (the object-gradient sum is
        told apart from the small packed sums by its size)."""
        fn = getattr(obj, method)

        def wrapper(*args, **kw):
            if not self.enabled:
                return fn(*args, **kw)
            e0, e1 = self.pair()
            e0.record()
            out = fn(*args, **kw)
            e1.record()
            self.events[name(*args) if callable(name) else name].append(
                (e0, e1, None))
            return out

        setattr(obj, method, wrapper)

    def reset(self):
        self.events.clear()
        self.next = 0

    def summary(self):
        out = {}
        for name, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b, _ in evs]
            out[name] = dict(calls=len(ms), total_ms=float(np.sum(ms)),
                             avg_ms=float(np.mean(ms)))
            if all(n is not None for _, _, n in evs):
                out[name]["positions"] = float(sum(n for _, _, n in evs))
        return out


# -------------------------------------------------------------- CPU baseline
def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_epoch(p, data_np, S, det, n=256):
    """the oracle's lstsq_grad minibatch (gradients + step sizes)
    on ONE 256-position slice of the workload, host cores, scipy.fft with
    every core -- a bounded sample (~10-30 s)."""
    from oracle import operators as oops
    from oracle import solvers as osol
    cores = os.cpu_count() or 1
    oops.set_workers(cores)
    n = min(n, len(p["scan"]))
    scan = p["scan"][:n]
    psi0 = np.full_like(p["psi"], 0.5 + 0j)
    pre = osol.psi_preconditioner(psi0, p["probe"], scan)
    t0 = time.perf_counter()
    g = osol.get_nearplane_gradients(
        data_np[:n], psi0, scan, p["probe"], None, None, 0, n,
        num_batch=1, detector_shape=det,
        measured_pixels=np.ones((det, det), dtype=bool))
    osol.precondition_nearplane_gradients(
        g["chi"], scan, g["unique_probe"], p["probe"],
        g["object_upd_sum"], g["m_probe_update"], pre, g["patches"], 0, n)
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="patterns/s", cores=cores, kind="port",
                cpu=cpu_model(), seconds=dt,
                sample=f"oracle (NumPy/SciPy restatement) lstsq_grad minibatch "
                f"(gradients + step sizes) on a {n}-position slice x {S} modes "
                f"{det}x{det}, scipy.fft workers={cores}")


def cpu_baseline_c1(p, data_np, det):
    """BASELINE configs[0] IN FULL on the host: one oracle cgrad epoch (cg_iter
    = 4, object then probe) over all 256 positions."""
    from oracle import operators as oops
    from oracle import solvers as osol
    cores = os.cpu_count() or 1
    oops.set_workers(cores)
    n = len(p["scan"])
    state = dict(psi=np.full_like(p["psi"], 0.5 + 0j), probe=p["probe"].copy(),
                 scan=p["scan"].copy(), costs=[])
    t0 = time.perf_counter()
    osol.cgrad(state, data_np, [np.arange(n)], detector_shape=det, cg_iter=4,
               recover_probe=True)
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="patterns/s", cores=cores, kind="port",
                cpu=cpu_model(), seconds=dt,
                sample=f"oracle cgrad epoch (cg_iter=4, object + probe) over all "
                f"{n} positions, 1 mode {det}x{det}, scipy.fft workers={cores}")


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
                cpu=cpu_model(), seconds=dt,
                sample=f"oracle Ptycho.fwd on {done} positions x {S} modes "
                f"{det}x{det}, scipy.fft workers={cores}")


def cpu_baseline_adj(p, S, det, seconds=10.0):
    from oracle import operators as oops
    cores = os.cpu_count() or 1
    oops.set_workers(cores)
    rng = np.random.default_rng(0)
    n, done, t0 = 16, 0, time.perf_counter()
    far = (rng.random((n, 1, S, det, det, 2), dtype=np.float32) - 0.5).view(
        np.complex64)[..., 0]
    probe = np.broadcast_to(p["probe"], (n, 1, S, det, det))
    while True:
        oops.ptycho_adj(far, probe, p["scan"][:n], p["psi"])
        done += n
        if time.perf_counter() - t0 > seconds:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="patterns/s", cores=cores, kind="port",
                cpu=cpu_model(), seconds=dt,
                sample=f"oracle Ptycho.adj on {done} positions x {S} modes "
                f"{det}x{det}, scipy.fft workers={cores}")


# ------------------------------------------------- algorithmic bytes / launch
def fwd_bytes(n, S, det, pw, C=0):
    """SURVEY 8(d) B_fwd: far-plane store + object-patch gather + scan per
    position, shared probe once per call."""
    return n * (8 * S * det * det + 8 * pw * pw + 8) + 8 * (S + C) * pw * pw


def adj_bytes(n, S, det, pw, per_position_probe=False):
    """Algorithmic bytes of Ptycho.adj: far plane in + probe_adj out + object
    patch gather + object read-modify-write (counted once, as SURVEY 8(d)
    counts the gradient scatter) + scan per position; the shared probe once
    per call (a per-position probe: once per position)."""
    per = 8 * S * det * det + 8 * S * pw * pw + 2 * 8 * pw * pw + 8
    if per_position_probe:
        return n * (per + 8 * S * pw * pw)
    return n * per + 8 * S * pw * pw


def algorithmic_bytes(name, n, S, det, pw, C):
    """HBM bytes one launch must move for n positions:
    the operator entries here, every stage entry of a chunk from the byte
    models the GradientPlan carries (tike_amd/ptycho/solvers/_plan.py)."""
    from tike_amd.ptycho.solvers._plan import algorithmic_bytes as stage_bytes
    if name == "tike_ptycho_fwd":
        return fwd_bytes(n, S, det, pw, C)
    if name == "tike_ptycho_adj":
        return adj_bytes(n, S, det, pw)
    return stage_bytes(name, n, S, det, pw, C, depth=MULTISLICE_DEPTH)


def dominant_entry(summary, n, S, det, C, pw=None):
    """The tike_* entry with the most time among those with a byte model (a
    composite such as tike_cgrad_line_search -- many small launches under one
    name -- has none and cannot carry a roofline)."""
    ks = {k: v for k, v in summary.items() if k.startswith("tike_")}
    modelled = [k for k in ks
                if algorithmic_bytes(k, n, S, det, pw or det, C)]
    return max(modelled or ks, key=lambda k: ks[k]["total_ms"])


def iteration_bounds(S, det, pw):
    """compulsory HBM bytes and flops of one lstsq_grad
    iteration per position (intermediates assumed cache-resident)."""
    b_iter = det * det * 4 + 5 * 8 * pw * pw
    f_iter = 2 * S * 5 * det * det * np.log2(det * det) + 60 * S * det * det
    return b_iter, f_iter


def measured_traffic(workload, launch_n):
    """rocprofv3 PMC passes committed under profiles/ (FETCH_SIZE /
    WRITE_SIZE collected in separate --pmc runs of this same command,
    gfx950-corrected by tools/pmc_traffic.py): {kernel: bytes per launch}
    and the sum over one step, from the newest round that has them."""
    import glob
    for f in sorted(glob.glob(os.path.join(
            ROOT, "profiles", f"r*_pmc_traffic_{workload}.json")),
                    reverse=True):
        try:
            doc = json.load(open(f))
        except (OSError, ValueError):
            continue
        if doc.get("positions_per_launch") != launch_n:
            continue
        # the counters belong to the kernels they were collected on: a file
        # whose build id is not the loaded library's is not quoted
        from tike_amd._lib import build_id
        if doc.get("build_id") != build_id():
            return {}, None, os.path.basename(f) + " (STALE: collected on "
            "another build of csrc/; re-run tools/profile_round.sh)"
        per = {k: v.get("hbm_bytes_per_launch")
               for k, v in doc.get("kernels", {}).items()}
        per.update({k: v.get("hbm_bytes_per_launch")
                    for k, v in doc.get("composite", {}).items()})
        return per, doc.get("hbm_bytes_per_step"), os.path.basename(f)
    return {}, None, None


EPOCH_DEFAULTS = {
    # workload: (detector, modes, positions per GPU, minibatches)
    "c1": (128, 1, 256, 1),
    "c2": (256, 1, 10000, 10),
    "c3": (256, 8, 10000, 10),
    # c5 = 80 000 positions over 8 GPUs
    "c5": (512, 4, 10000, 10),
    # SURVEY 8 rows f2 / f3 at the headline shapes (not BASELINE configs; own
    # lines in profiles/): c3 with the Poisson noise model; c3 solved by rpie
    "c3poisson": (256, 8, 10000, 10),
    "c3rpie": (256, 8, 10000, 10),
    # ... and by rpie on a two-slice object (row f3, multislice part)
    "c3rpie2": (256, 8, 10000, 10),
    # ... at the other two tile sizes of the fused multislice chain, and under
    # the Poisson model (round 6; `--workload` legs, c5rpie2 in the default run)
    "c5rpie2": (512, 4, 4000, 10),
    "c128rpie2": (128, 8, 10000, 10),
    # (a rate leg: rpie + Poisson on this problem stops converging after ~10
    # epochs on every route, profiles/r06_experiments.md section 7)
    "c3rpie2poisson": (256, 8, 10000, 10),
    # off-grid shapes (round 6; not BASELINE configurations): what the
    # reference's cuFFT path serves at one speed and the fused pow-2 kernels
    # do not -- a probe window of half the detector, 12 modes, a 384^2 crop
    "c3pad": (256, 8, 10000, 10),
    "c3m12": (256, 12, 10000, 10),
    "c384": (384, 4, 10000, 10),
    # powers of two WITHOUT position-major kernels (not in the default run)
    "c64": (64, 8, 10000, 10),
    "c128": (128, 8, 10000, 10),
    "c1024": (1024, 2, 1000, 10),
}
# The minibatches are contiguous chunks of the ONCE-SHUFFLED scan (SURVEY
# 8(d)): each spans the whole field of view, like the batches the reference's
# default selector `wobbly_center` makes -- and like them they are followed by
# an object update each (`batch_method` names the update rule here; the
# batches themselves are handed over).  TIKE_BENCH_BATCH_RULE=compact: the
# rule for spatially disjoint batches (one summed object update per epoch),
# which rounds 1-4 ran on these overlapping batches -- same kernels, same
# time, but an iteration that stops converging after two epochs
# (tools/soak_costs.py).
BATCH_RULE = os.environ.get("TIKE_BENCH_BATCH_RULE", "wobbly_center")
SOLVER_LABEL = {"c1": "cgrad (cg_iter=4)", "c2": "cgrad (cg_iter=4)",
                "c3poisson": "lstsq_grad (poisson, all_modes)",
                "c3rpie": "rpie", "c3rpie2": "rpie, two-slice object",
                "c5rpie2": "rpie, two-slice object",
                "c128rpie2": "rpie, two-slice object",
                "c3rpie2poisson": "rpie, two-slice object (poisson)"}
MULTISLICE = ("c3rpie2", "c5rpie2", "c128rpie2", "c3rpie2poisson")


def epoch_problem(workload, positions, world, rank, tp, A, data_on_host=False):
    """Synthetic problem + Reconstruction context of one epoch workload.
    c1 = BASELINE configs[0] (256 positions, 128x128, 1 mode, cgrad, one
    minibatch); c2: 1 mode, cgrad; c3 (= one GPU's share of c4): 8 modes +
    eigen probes, lstsq_grad; c5: 512x512, 4 modes, lstsq_grad with position
    correction.  Global problem: N * world positions; this rank holds its
    share of every global minibatch, concatenated
    (Reconstruction(presharded=True))."""
    det, S, N, num_batch = EPOCH_DEFAULTS[workload]
    N = positions or N
    Ng = N * world
    per = Ng // num_batch
    share = per // world
    idx = np.concatenate([
        np.arange(b * per + rank * share, b * per + (rank + 1) * share)
        for b in range(num_batch)
    ])
    full = synthetic(Ng, S, det, 0, Ng, pw=PROBE_WIDTH.get(workload))
    p = dict(full, scan=full["scan"][idx])
    N = len(idx)
    np.random.seed(1234 + rank)
    eigen_probe = eigen_weights = None
    C = 0
    # (the two-slice workload runs without eigen probes, like the reference's
    # multislice test and tools/soak_multislice.py)
    if workload.startswith("c3") and workload not in MULTISLICE:
        import tike_amd.random
        tike_amd.random.randomizer_np = np.random.default_rng(4321)
        eigen_probe, eigen_weights = tp.init_varying_probe(
            p["scan"], p["probe"], num_eigen_probes=2, probes_with_modes=1)
        C = eigen_probe.shape[-4]
    data = tp.simulate(det, p["probe"], p["scan"], p["psi"])
    psi0 = np.full_like(p["psi"], 0.5 + 0j)
    multislice = {}
    if workload in MULTISLICE:
        # two slices, the one behind starts transparent; 0.1 nm, 2 um field
        # of view, 1 um between the slices (tools/soak_multislice.py)
        psi0 = np.concatenate([psi0, np.ones_like(psi0)])
        multislice = dict(probe_wavelength=1e-10,
                          probe_FOV_lengths=(2e-6, 2e-6))
    params = tp.PtychoParameters(
        probe=p["probe"].copy(),
        psi=psi0, scan=p["scan"],
        eigen_probe=eigen_probe, eigen_weights=eigen_weights,
        # BASELINE configs[0] / [1] name the conjugate-gradient solver
        algorithm_options=(tp.CgradOptions(num_batch=num_batch, cg_iter=4)
                           if workload in ("c1", "c2") else
                           tp.RpieOptions(num_batch=num_batch,
                                          batch_method=BATCH_RULE)
                           if workload == "c3rpie" or workload in MULTISLICE
                           else
                           tp.LstsqOptions(num_batch=num_batch,
                                           batch_method=BATCH_RULE)),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool),
            noise_model="poisson")
        if workload in ("c3poisson", "c3rpie2poisson") else
        # (the default mask has the probe's shape, options.py:168: a probe
        # window narrower than the detector must bring its own)
        tp.ExitWaveOptions(measured_pixels=np.ones((det, det), dtype=bool))
        if workload in PROBE_WIDTH else None,
        probe_options=tp.ProbeOptions(force_orthogonality=True, **multislice),
        object_options=tp.ObjectOptions(
            **(dict(multislice_propagation_distance=1e-6)
               if multislice else {})),
        position_options=tp.PositionOptions(
            p["scan"].copy(), use_adaptive_moment=True,
            update_magnitude_limit=1.0) if workload == "c5" else None)
    ctx = tp.Reconstruction(
        data if data_on_host else A.to_device(data, np.float32), params,
        presharded=True, order=np.arange(N),
        batches=np.array_split(np.arange(N), num_batch),
        data_on_host=data_on_host)
    ctx.__enter__()
    return dict(ctx=ctx, p=p, data=data, det=det, S=S, N=N, C=C,
                num_batch=num_batch, eigen_probe=eigen_probe,
                eigen_weights=eigen_weights)


PROBE_WIDTH = {"c3pad": 128}  # probe window narrower than the detector

LEG_EPOCHS = {"c1": 20, "c2": 3, "c5": 3, "c3poisson": 3, "c3rpie": 3, "c3rpie2": 3,
              "c3pad": 3, "c3m12": 3, "c384": 3}
# untimed epochs in front (cgrad: at least two -- its line searches learn their
# slot counts; c1's epochs are 3 ms: ten of them bring the clocks of a GPU
# that idled during the set-up back up, profiles/r04_leg_probe.txt)
LEG_WARM_EPOCHS = {"c1": 10, "c2": 2, "c5": 1}


def epoch_leg(workload, tp, A, torch, positions=0, epochs=None):
    """One short leg of another BASELINE configuration: LEG_WARM_EPOCHS
    untimed epochs, then `epochs` timed ones (wall clock around synchronised
    epochs; default LEG_EPOCHS)."""
    epochs = epochs or LEG_EPOCHS.get(workload, 2)
    built = epoch_problem(workload, positions, 1, 0, tp, A)
    ctx = built["ctx"]
    try:
        ctx.iterate(LEG_WARM_EPOCHS.get(workload, 1))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.iterate(epochs)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        ctx.__exit__(None, None, None)
    det, S, N = built["det"], built["S"], built["N"]
    pw = built["p"]["pw"]
    b_iter, f_iter = iteration_bounds(S, det, pw)
    rate = N * epochs / dt
    leg = dict(workload=workload, positions=N, modes=S, detector=det,
               probe_width=pw,
               solver=SOLVER_LABEL.get(workload, "lstsq_grad"),
               num_batch=built["num_batch"],
               position_correction=workload == "c5", epochs=epochs,
               ms_per_epoch=dt / epochs * 1e3, value=rate, unit="patterns/s")
    if workload not in MULTISLICE:  # (SURVEY 8(d)'s bounds: a single slice's)
        leg.update(iteration_hbm_frac=b_iter * rate / 1e9 / HBM_PEAK_GBS,
                   iteration_fp32_frac=f_iter * rate / 1e12 / FP32_PEAK_TFLOPS)
    return leg


def leg_traffic(workload, entry, N, ms):
    """PMC bytes of one operator call from profiles/ (None while no file of
    this build exists) and the rate they crossed the HBM interface at."""
    per_call, _, source = measured_traffic(workload, N)
    moved = per_call.get(entry)
    out = dict(traffic=moved, traffic_source=source)
    if moved:
        out["traffic_frac"] = moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return out


def forward_leg(ops, A, torch, det, S, N, iters=40, warm=8):
    """One short leg of the forward operator alone (hip-event timed).  The
    GPU has idled while the host generated the inputs: `warm` untimed calls
    bring its clocks back up (with 2 warm-up + 10 timed calls the same
    kernels measured 5 % slower than over 50 calls, tools/leg_probe.py)."""
    p = synthetic(N, S, det, 0, N)
    op = ops.Ptycho(probe_shape=det, detector_shape=det, nz=p["HW"], n=p["HW"])
    scan, psi, probe = (A.to_device(p[k]) for k in ("scan", "psi", "probe"))
    out = torch.empty((N, 1, S, det, det), dtype=torch.complex64,
                      device=psi.device)
    for _ in range(warm):
        op.fwd_device(probe, scan, psi, out=out)
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        op.fwd_device(probe, scan, psi, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = fwd_bytes(N, S, det, det)
    del out
    return dict(workload=f"fwd{det}x{S}", positions=N, ms_per_launch=ms,
                value=N / (ms * 1e-3), unit="patterns/s",
                achieved_GBs=nbytes / (ms * 1e-3) / 1e9,
                frac=nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                **leg_traffic(f"fwd{det}x{S}", "tike_ptycho_fwd", N, ms))


def adjoint_problem(ops, A, torch, det, S, N, lo=0, hi=None, total=None):
    """Inputs of the adjoint operator alone: SURVEY 8(d)'s object, probe and
    scan (listed leaf by leaf of a k-d tree, as the solvers list every
    minibatch: the grouped scatter sums neighbours on chip) and a random far
    plane."""
    from tike_amd.cluster import spatial_order
    total = total or N
    hi = N if hi is None else hi
    p = synthetic(total, S, det, lo, hi)
    p["scan"] = p["scan"][spatial_order(p["scan"])]
    op = ops.Ptycho(probe_shape=det, detector_shape=det, nz=p["HW"], n=p["HW"])
    scan, psi, probe = (A.to_device(p[k]) for k in ("scan", "psi", "probe"))
    n = len(p["scan"])
    g = torch.Generator(device=psi.device).manual_seed(1234)
    far = torch.view_as_complex(
        torch.rand((n, 1, S, det, det, 2), generator=g, device=psi.device) - 0.5)
    out = (torch.empty_like(psi),
           torch.empty((n, 1, S, det, det), dtype=torch.complex64,
                       device=psi.device))
    return p, op, (far, probe, scan, psi), out


def adjoint_leg(ops, A, torch, det, S, N, iters=40, warm=8):
    """One short leg of the adjoint operator alone (hip-event timed, like
    forward_leg): Ptycho.adj of a stored far plane -> (psi_adj, probe_adj)."""
    p, op, args, out = adjoint_problem(ops, A, torch, det, S, N)
    for _ in range(warm):
        op.adj_device(*args, psi_adj=out[0], probe_adj=out[1])
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        op.adj_device(*args, psi_adj=out[0], probe_adj=out[1])
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    nbytes = adj_bytes(N, S, det, det)
    del args, out
    return dict(workload=f"adj{det}x{S}", positions=N, ms_per_launch=ms,
                value=N / (ms * 1e-3), unit="patterns/s",
                achieved_GBs=nbytes / (ms * 1e-3) / 1e9,
                frac=nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                **leg_traffic(f"adj{det}x{S}", "tike_ptycho_adj", N, ms))


def ranks_agreement(ctx, world, torch, dist):
    """Float64 checksums of the replicated state of every rank, gathered:
    `ranks_agree` = the largest relative spread of any checksum over the
    ranks (0.0: bit-identical sums), `rccl_ranks` = what an all-reduce of 1.0
    over the live backend returns (the ranks the collectives really span),
    `collective_backend` its name."""
    prm = ctx.parameters
    psi, probe = prm.psi.to(torch.complex128), prm.probe.to(torch.complex128)
    costs = prm.algorithm_options.costs
    sums = torch.stack([
        psi.abs().square().sum(), psi.real.sum(), psi.imag.sum(),
        probe.abs().square().sum(), probe.real.sum(), probe.imag.sum(),
        torch.tensor(float(costs[-1][0]) if costs else 0.0,
                     dtype=torch.float64, device=psi.device)]).to(torch.float64)
    one = torch.ones(1, dtype=torch.float32, device=psi.device)
    if dist.is_initialized() and dist.get_backend() != "nccl":
        sums, one = sums.cpu(), one.cpu()  # (gloo gathers host tensors)
    if dist.is_initialized():
        gathered = [torch.empty_like(sums) for _ in range(world)]
        dist.all_gather(gathered, sums)
        dist.all_reduce(one)
        backend = dist.get_backend()
    else:
        gathered, backend = [sums], "none"
    table = torch.stack(gathered)  # (world, 7)
    scale = table.abs().amax(dim=0).clamp_min(1e-300)
    spread = ((table.amax(dim=0) - table.amin(dim=0)) / scale).max()
    return dict(ranks_agree=float(spread.item()),
                rccl_ranks=int(round(float(one.item()))),
                collective_backend=backend,
                checksums="sum |psi|^2, Re psi, Im psi, |probe|^2, Re probe, "
                "Im probe, last cost (float64), all-gathered")


def allreduce_probe(nbytes, world, torch, dist, calls=100):
    """The object-gradient-sized sum all-reduce alone, `calls` times back to
    back (HIP events): what a minibatch pays for the link."""
    buf = torch.zeros(nbytes // 4, dtype=torch.float32, device="cuda")
    for _ in range(5):
        dist.all_reduce(buf)
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(2))
    e0.record()
    for _ in range(calls):
        dist.all_reduce(buf)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / calls
    t = torch.tensor([ms], device="cuda", dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t.item())
    return dict(bytes=nbytes, calls=calls, avg_ms=ms,
                algbw_GBs=nbytes / (ms * 1e-3) / 1e9,
                busbw_GBs=(2 * (world - 1) / max(world, 1)) * nbytes /
                (ms * 1e-3) / 1e9, backend=dist.get_backend())


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        launch_ranks(a)  # never returns
    world = int(env_world or "1")
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; launch "
                 "with `python -m torch.distributed.run --nproc-per-node "
                 f"{a.gpus} bench.py --gpus {a.gpus} ...` or plain "
                 f"`python bench.py --gpus {a.gpus}`")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # TIKE_BENCH_SHARE_GPU=1 (tests only): the ranks share the visible GPUs
    # and talk over gloo -- the N > 1 code path of this file (sharding, max
    # over ranks, whole-job value, all-reduce summary) on a one-GPU box; RCCL
    # refuses two ranks on one device, so no such number is ever a result
    share_gpu = os.environ.get("TIKE_BENCH_SHARE_GPU") == "1"
    if share_gpu and torch.cuda.device_count() > 0:
        local %= torch.cuda.device_count()
    if torch.cuda.device_count() <= local:
        sys.exit(f"bench.py: rank {rank} has no GPU {local}")
    torch.cuda.set_device(local)
    # TIKE_FORCE_COLLECTIVES=1 (under torchrun --nproc-per-node 1) issues the
    # RCCL collectives of the multi-GPU path on a single rank: a dry run of
    # that path on a one-GPU box
    forced = (os.environ.get("TIKE_FORCE_COLLECTIVES") == "1"
              and "MASTER_ADDR" in os.environ)
    if share_gpu and world > 1:
        dist.init_process_group("gloo")
    elif world > 1 or forced:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    import tike_amd._arrays as A
    import tike_amd.operators as ops
    import tike_amd.ptycho as tp
    from tike_amd._lib import _PROTOTYPES, lib
    from tike_amd.ptycho.solvers.lstsq import chunk_positions

    timers = KernelTimers(lib, [n for n in _PROTOTYPES if n != "tike_init"])
    timers.label_cost_only = a.workload in ("c1", "c2") + MULTISLICE
    if a.workload in MULTISLICE:
        global MULTISLICE_DEPTH
        MULTISLICE_DEPTH = 2
        timers.cost_only_suffix = ":no_patches"
        timers.modes = EPOCH_DEFAULTS[a.workload][1]
    counts = dict(epoch=0, steps=0, in_minibatch=False)
    cpu = None
    cpu_job = None
    C = 0
    ctx = None

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
            cpu_job = lambda: cpu_baseline_fwd(p, S, det)
        workload = dict(workload=a.workload, positions_per_gpu=N, modes=S,
                        detector=det, solver=None)
    elif a.workload.startswith("adj"):
        det, S = [int(v) for v in a.workload[3:].split("x")]
        N = a.positions or max(256, 4096 // S)
        p, op, adj_args, adj_out = adjoint_problem(
            ops, A, torch, det, S, N, rank * N, (rank + 1) * N, N * world)

        def step():
            op.adj_device(*adj_args, psi_adj=adj_out[0], probe_adj=adj_out[1])

        units, launch_n, dominant = N, N, "tike_ptycho_adj"
        if rank == 0 and not a.no_cpu_baseline and world == 1:
            cpu_job = lambda: cpu_baseline_adj(p, S, det)
        workload = dict(workload=a.workload, positions_per_gpu=N, modes=S,
                        detector=det, solver=None,
                        positions="k-d-tree leaf order (as the solvers list a "
                        "minibatch)")
    elif a.workload in EPOCH_DEFAULTS:
        # c2: 1 mode; c3 (default, = one GPU's share of c4): 8 modes + eigen
        # probes; c5: 512x512, 4 modes, position correction on
        # c1 = BASELINE configs[0], the reference's CPU-runnable case: 256
        # positions, 128x128, 1 mode, cgrad (one minibatch)
        built = epoch_problem(a.workload, a.positions, world, rank, tp, A,
                              data_on_host=a.data_on_host)
        ctx, p, data = built["ctx"], built["p"], built["data"]
        det, S, N, C = built["det"], built["S"], built["N"], built["C"]
        num_batch = built["num_batch"]
        # every all-reduce, timed like a kernel
        n_object = 2 * p["HW"]**2  # float32 words of the object-gradient slice

        def which(*tensors):
            words = sum(t.numel() * (2 if t.is_complex() else 1)
                        for t in tensors)
            return "allreduce:object" if words >= n_object else "allreduce"

        timers.wrap_method(ctx.comm, "Allreduce", which)
        # ... and counted apart where it is issued once per epoch
        # (preconditioners, eigen-weight norms: outside the minibatch loop)
        import tike_amd.ptycho.solvers.lstsq as _L
        _grads = _L._get_nearplane_gradients
        _inner = ctx.comm.Allreduce

        def _count_epoch(*t):
            if timers.enabled and not counts["in_minibatch"]:
                counts["epoch"] += 1
            return _inner(*t)

        def _gradients(*args, **kw):
            counts["in_minibatch"] = True  # (until the epoch's loop ends)
            return _grads(*args, **kw)

        ctx.comm.Allreduce = _count_epoch
        _L._get_nearplane_gradients = _gradients

        def step():
            counts["in_minibatch"] = False
            if timers.enabled:
                counts["steps"] += 1
            ctx.iterate(1)

        units = N
        launch_n = min(chunk_positions(S, det, True), N // num_batch)
        dominant = None
        if (rank == 0 and not a.no_cpu_baseline and world == 1
                and a.workload in ("c1", "c2", "c3", "c5")):
            # (the slice the host leg works on; the leg itself runs AFTER
            # everything timed on the GPU: its 256 FFT threads, run first, cost
            # the epochs behind them 2 % -- launches issued late)
            sample = np.array(data if a.workload == "c1" else data[:256])
            cpu_job = (
                (lambda: cpu_baseline_c1(p, sample, det))
                if a.workload == "c1" else
                (lambda: cpu_baseline_epoch(p, sample, S, det)))
        workload = dict(workload=a.workload, positions_per_gpu=N, modes=S,
                        detector=det, eigen_probes=C,
                        solver=SOLVER_LABEL.get(a.workload, "lstsq_grad"),
                        num_batch=num_batch, chunk_positions=launch_n,
                        position_correction=a.workload == "c5")
        if p["pw"] != det:
            workload["probe_width"] = p["pw"]
        if a.workload not in ("c1", "c2"):
            workload["object_update"] = (
                "after every minibatch (batch_method wobbly_center)"
                if BATCH_RULE != "compact" else
                "once per epoch (batch_method compact)")
        if a.data_on_host:
            workload["data_residency"] = "pinned host, streamed per chunk"
    else:
        raise SystemExit(f"unknown workload {a.workload}")

    # After the warm-up, two more untimed steps bracket every launch with
    # events (the second one counts): it names the dominant kernel and gives the per-kernel shares.
    # The timed steps then bracket only that kernel (and the all-reduce):
    # events around all ~25 launches of a minibatch cost 3 % of a c3 epoch
    # (188 k vs 182 k patterns/s).  --warmup 0: no such step, every launch of
    # the timed steps is bracketed.
    profile = None
    for _ in range(a.warmup):
        step()
    # cgrad replays its CG calls from captured graphs: launches inside a
    # replay cannot be bracketed, so the per-kernel shares (and the dominant
    # kernel's launch time) come from steps launched one by one, the timed
    # steps run as users run them (graphs on, no events)
    import importlib
    _cg = importlib.import_module("tike_amd.ptycho.solvers.cgrad")
    graphs_on = _cg.USE_GRAPHS and a.workload in ("c1", "c2")
    if graphs_on:
        _cg.USE_GRAPHS = False
    if a.warmup > 0:
        # twice: torch creates the HIP event behind a pooled Event object at
        # its first record(), which the first bracketed step pays for
        for _ in range(2):
            timers.reset()
            torch.cuda.synchronize()
            timers.enabled = True
            tp0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            profile = (timers.summary(), time.perf_counter() - tp0)
            timers.enabled = False
        timers.reset()
        counts.update(epoch=0, steps=0)
    if profile is not None:
        if dominant is None:
            dominant = dominant_entry(profile[0], launch_n, S, det, C)
        timers.only = {dominant.split(":")[0]}
    if graphs_on:
        _cg.USE_GRAPHS = True
        if profile is not None:
            timers.only = set()
            step()  # the occurrence that captures
            torch.cuda.synchronize()
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

    plans = list(getattr(getattr(ctx, "operator", None), "_tike_amd_plans",
                         {}).values()) if ctx is not None else []
    # ---- a multi-rank line has to prove itself: the replicated state (object,
    # probe, last cost) must be the SAME on every rank after the timed steps
    # (all-reduced gradients -> identical updates, reference comm.py:96-136),
    # and the collective backend must really span `world` ranks
    agreement = None
    if (world > 1 or forced) and ctx is not None:
        agreement = ranks_agreement(ctx, world, torch, dist)
        if agreement["ranks_agree"] > 1e-6:
            sys.exit(f"bench.py: the ranks hold different states after the "
                     f"timed steps (max relative spread of the checksums "
                     f"{agreement['ranks_agree']:.3e} > 1e-6): no value is "
                     f"reported for a job whose ranks disagree\n{agreement}")
    probe_line = None
    if a.allreduce_probe and (world > 1 or forced):
        probe_line = allreduce_probe(8 * p["HW"]**2, world, torch, dist)

    secondary = None
    if (rank == 0 and world == 1 and a.workload == "c3"
            and not a.no_secondary):
        # the forward operator alone (north_star's 60 % target), driver-timed
        if ctx is not None:
            ctx.__exit__(None, None, None)
            ctx = None
        def guarded(name, leg, *args):
            # a secondary leg must never cost the headline line
            try:
                return leg(*args)
            except Exception as e:  # noqa: BLE001
                torch.cuda.empty_cache()
                return dict(workload=name, error=f"{type(e).__name__}: {e}")

        del data
        torch.cuda.empty_cache()
        secondary = [guarded("fwd256x1", forward_leg, ops, A, torch, 256, 1,
                             4096),
                     guarded("fwd128x1", forward_leg, ops, A, torch, 128, 1,
                             16384),
                     guarded("fwd256x8", forward_leg, ops, A, torch, 256, 8,
                             512),
                     # ... and its adjoint (north_star names both)
                     guarded("adj256x1", adjoint_leg, ops, A, torch, 256, 1,
                             4096),
                     guarded("adj128x1", adjoint_leg, ops, A, torch, 128, 1,
                             16384),
                     guarded("adj256x8", adjoint_leg, ops, A, torch, 256, 8,
                             512)]
        # ... and the other BASELINE configurations, one short leg each
        torch.cuda.empty_cache()
        secondary += [guarded(w, epoch_leg, w, tp, A, torch)
                      for w in ("c1", "c2", "c5")]
        # ... and SURVEY 8 rows f2 / f3 at the headline shapes: the Poisson
        # model, rpie, rpie on a two-slice object
        secondary += [guarded(w, epoch_leg, w, tp, A, torch)
                      for w in ("c3poisson", "c3rpie", "c3rpie2", "c5rpie2")]
        # ... and the off-grid shapes (round 6): a probe window of half the
        # detector, 12 modes, a 384^2 detector -- each with its rate per
        # far-plane byte beside c3's
        for w in ("c3pad", "c3m12", "c384"):
            leg = guarded(w, epoch_leg, w, tp, A, torch)
            if "value" in leg:
                leg["farplane_GBs"] = (8 * leg["modes"] * leg["detector"]**2 *
                                       leg["value"] / 1e9)
            secondary.append(leg)

    if rank == 0:
        summ = timers.summary()  # the timed steps
        # every launch bracketed: the last warm-up step, or (--warmup 0) the
        # timed steps themselves
        full, full_wall = profile if profile is not None else (summ, wall)
        kernels = {k: v for k, v in full.items() if k.startswith("tike_")}
        pw = p.get("pw", det) if isinstance(p, dict) else det
        if dominant is None:
            dominant = dominant_entry(full, launch_n, S, det, C, pw)
        k = summ.get(dominant) or full[dominant]
        if k.get("positions"):
            # launches of different sizes under one entry: the mean launch
            launch_n = k["positions"] / k["calls"]
        nbytes = algorithmic_bytes(dominant, launch_n, S, det, pw, C)
        achieved = nbytes / (k["avg_ms"] * 1e-3) / 1e9
        ktot = sum(v["total_ms"] for v in kernels.values())
        if a.breakdown:
            for name, v in sorted(full.items(),
                                  key=lambda kv: -kv[1]["total_ms"]):
                print(f"  {name:32s} calls {v['calls']:5d} avg {v['avg_ms']:8.3f} ms"
                      f" total {v['total_ms']:9.2f} ms ({100 * v['total_ms'] / ktot:4.1f}%)",
                      file=sys.stderr)
            print(f"  kernels {ktot:.1f} ms of wall {full_wall * 1e3:.1f} ms "
                  f"({100 * ktot / (full_wall * 1e3):.1f} %)", file=sys.stderr)
        per_launch, per_step, pmc_file = measured_traffic(a.workload, launch_n)
        aggregate = units * world * a.steps / wall
        roofline = {
            "bound": "hbm", "kernel": dominant,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": per_launch.get(dominant),
            "algorithmic_bytes": nbytes,
            "avg_launch_ms": k["avg_ms"], "positions_per_launch": launch_n,
            "share_of_kernel_time": full[dominant]["total_ms"] / ktot,
            "kernel_time_share_of_wall": ktot / (full_wall * 1e3),
            "events": "timed steps: this kernel only; shares: an untimed "
                      "step after the warm-up, every launch bracketed"
            if profile is not None else "timed steps: every launch bracketed",
        }
        if roofline["traffic"]:
            # the same launch priced on the bytes the counters saw cross the
            # HBM interface (an algorithmic figure above the stream ceiling --
            # c2's composite entry -- is Infinity-Cache hits, not bandwidth)
            moved = roofline["traffic"] / (k["avg_ms"] * 1e-3) / 1e9
            roofline["traffic_achieved"] = moved
            roofline["traffic_frac"] = moved / HBM_PEAK_GBS
        if graphs_on:
            roofline["events"] = (
                "launch time and shares: an untimed step launched one by one "
                "after the warm-up; the timed steps replay captured graphs "
                "(no events inside a replay)")
        if not nbytes:
            # an entry that runs a data-dependent number of passes (the
            # device-side line search): no byte model, no fraction
            roofline["achieved"] = roofline["frac"] = None
            roofline["note"] = ("no algorithmic-byte model for this entry "
                                "(data-dependent number of trial passes)")
        elif roofline["frac"] > READ_CEILING_GBS / HBM_PEAK_GBS and (
                roofline["traffic"] is None):
            # a rate above what a pure read stream reaches on this chip can only
            # come from cache hits inside the launch: not stated without the
            # counters that show them
            roofline["frac_withheld"] = (
                f"algorithmic bytes / time = {achieved:.0f} GB/s exceeds the "
                f"measured read-stream ceiling ({READ_CEILING_GBS:.0f} GB/s) and "
                "no PMC traffic figure is committed for this kernel")
            roofline["achieved"] = roofline["frac"] = None
        if not a.workload.startswith(("fwd", "adj")):
            # the whole iteration against SURVEY 8(d)'s compulsory bytes and
            # flops (report both fractions)
            b_iter, f_iter = iteration_bounds(S, det, pw)
            per_gpu_rate = units * a.steps / wall
            roofline["iteration_hbm_frac"] = (b_iter * per_gpu_rate / 1e9 /
                                              HBM_PEAK_GBS)
            roofline["iteration_fp32_frac"] = (f_iter * per_gpu_rate / 1e12 /
                                               FP32_PEAK_TFLOPS)
            roofline["traffic_per_step"] = per_step
            roofline["traffic_source"] = pmc_file
        # `value` is the whole-job rate over all ranks (bench contract).  On one
        # GPU that IS BASELINE.json's "patterns/sec/GPU"; on several the label
        # says "whole job" so that nobody reads N x the per-GPU rate under a
        # per-GPU name (the per-GPU rate is `per_gpu`).
        shape = f"({det}x{det}, {S}-mode probe)"
        line = {
            "metric": (f"diffraction patterns/sec/GPU {shape}" if world == 1
                       else f"diffraction patterns/sec, whole job on {world} "
                       f"GPUs {shape}; per-GPU rate = per_gpu"),
            "value": aggregate,
            "per_gpu": aggregate / world,
            "aggregate": aggregate,
            "unit": "patterns/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": wall / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "c64",
            "data": ("synthetic" if not share_gpu else
                     "synthetic; TEST SWITCH: ranks share a GPU over gloo"),
            "config": workload,
            "roofline": roofline,
        }
        if (world > 1 or forced) and ("allreduce" in summ
                                      or "allreduce:object" in summ):
            # every blocking collective of the timed steps (HIP events around
            # Comm.Allreduce): per minibatch the object-gradient slice + two
            # small packed buffers (the probe-gradient slice is started early
            # with Comm.Allreduce_start and hides behind the object scatter:
            # not in these events), per epoch the object preconditioner (+
            # the eigen weight norms)
            small = summ.get("allreduce", dict(calls=0, total_ms=0.0,
                                               avg_ms=0.0))
            big = summ.get("allreduce:object")
            nb = workload.get("num_batch", 1)
            per_epoch = counts["epoch"] / max(counts["steps"], 1)
            calls = small["calls"] + (big["calls"] if big else 0)
            total = small["total_ms"] + (big["total_ms"] if big else 0.0)
            line["allreduce"] = dict(
                calls_per_step=calls / a.steps,
                calls_per_minibatch=(calls / a.steps - per_epoch) / nb,
                started_early_per_minibatch=1,
                avg_ms=total / max(calls, 1), ms_per_step=total / a.steps,
                gradient_bytes=8 * (p["HW"]**2 + S * pw * pw))
            if big:
                nbytes = 8 * p["HW"]**2
                # ring all-reduce: every rank sends and receives
                # 2 (n - 1) / n of the buffer
                line["allreduce"]["object_slice"] = dict(
                    bytes=nbytes, calls_per_step=big["calls"] / a.steps,
                    avg_ms=big["avg_ms"],
                    algbw_GBs=nbytes / (big["avg_ms"] * 1e-3) / 1e9,
                    busbw_GBs=(2 * (world - 1) / max(world, 1)) * nbytes /
                    (big["avg_ms"] * 1e-3) / 1e9)
        if agreement is not None:
            line.update(agreement)
        if probe_line is not None:
            line["allreduce_probe"] = probe_line
        # the route the chunks took and the launches it names (GradientPlan,
        # cached on the operator by the first minibatch)
        if plans:
            line["config"]["route"] = plans[-1].route
            line["config"]["launches"] = list(plans[-1].launches)
        if secondary is not None:
            line["secondary"] = secondary
        if cpu_job is not None:
            cpu = cpu_job()
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line))
    if ctx is not None:
        ctx.__exit__(None, None, None)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
