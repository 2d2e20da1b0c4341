#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE.

Run in the build container only (needs /root/reference, which does not exist on
the GPU box):

    python tests/golden/gen/make_fixtures.py

What it does
------------
1. compiles ``emu.cpp`` (the reference's own ``convolution.cu`` #included by
   path behind a host prelude) into a temp dir,
2. puts ``cupy_shim`` (NumPy-backed stand-in for the absent CuPy) and
   ``/root/reference/src`` on ``sys.path`` and imports the reference,
3. calls the reference's own functions on seeded inputs and stores inputs and
   outputs as ``.npz`` (data only; no reference source is stored),
4. re-encodes the data files the reference's own tests hold
   (``tests/data/ptycho_setup.pickle.lzma``, ``ptycho_gaussian.pickle.lzma``,
   ``tests/ptycho/ortho-{in,out}.mat``) as ``.npz``.
"""
import json
import lzma
import os
import pickle
import subprocess
import sys

sys.dont_write_bytecode = True  # never write into /root/reference
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)
REF = "/root/reference"

tmp = tempfile.mkdtemp(prefix="tike_ref_emu_")
lib = os.path.join(tmp, "libemu.so")
subprocess.check_call(
    ["g++", "-O2", "-shared", "-fPIC", "-o", lib,
     os.path.join(HERE, "emu.cpp")])
os.environ["TIKE_REF_EMU_LIB"] = lib
sys.path.insert(0, os.path.join(HERE, "cupy_shim"))
sys.path.insert(0, os.path.join(REF, "src"))

import cupy as cp  # noqa: E402  (the shim)
import tike.linalg  # noqa: E402
import tike.operators  # noqa: E402
import tike.opt  # noqa: E402
import tike.ptycho  # noqa: E402
import tike.ptycho.solvers.lstsq as ref_lstsq  # noqa: E402
import tike.random  # noqa: E402
from tike.ptycho.solvers._preconditioner import (  # noqa: E402
    _probe_preconditioner, _psi_preconditioner)

PHYS = dict(probe_wavelength=1e-10, probe_FOV_lengths=(1e-5, 1e-5),
            multislice_propagation_distance=1e-8)


PHYS_PROBE = {}
PHYS_OBJECT = {}


def rc(rng, *shape):
    """uniform [-0.5, 0.5) complex64 like tike.random.numpy_complex."""
    return (rng.random((*shape, 2), dtype=np.float32) - 0.5).view(
        np.complex64)[..., 0]


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB")


def A(x):
    return cp.asarray(x)


# ---- 4. reference-held data files ----------------------------------------
with lzma.open(f"{REF}/tests/data/ptycho_setup.pickle.lzma", "rb") as f:
    data, scan, probe, original = pickle.load(f)
save("ref_ptycho_setup.npz", data=data, scan=scan, probe=probe,
     original=original)
with lzma.open(f"{REF}/tests/data/ptycho_gaussian.pickle.lzma", "rb") as f:
    save("ref_ptycho_gaussian.npz", weights=pickle.load(f))
import scipy.io  # noqa: E402

_in = scipy.io.loadmat(f"{REF}/tests/ptycho/ortho-in.mat")
_out = scipy.io.loadmat(f"{REF}/tests/ptycho/ortho-out.mat")
save("ref_ortho.npz", modes=np.rollaxis(_in["modes"], -1, 0),
     pr=np.rollaxis(_out["pr"], -1, 0))

# check: reference simulate() reproduces its own fixture under the shim
sim = tike.ptycho.simulate(detector_shape=data.shape[-1], probe=probe,
                           scan=scan, psi=original, **PHYS)
err = np.abs(np.sqrt(sim) - np.sqrt(data)).max()
print("reference simulate vs its fixture: max|sqrt diff| =", err)
assert err < 1e-6

# ---- operators -------------------------------------------------------------
rng = np.random.default_rng(20241008)

# Patch (tests/operators/test_patch.py shapes, scaled down)
ntheta, nscan, H, W, pw = 3, 7, 48, 40, 12
images = rc(rng, ntheta, H, W)
positions = (rng.random((ntheta, nscan, 2), dtype=np.float32) *
             np.array([H - pw - 2, W - pw - 2], dtype=np.float32))
patches_in = rc(rng, ntheta, nscan, pw, pw)
with tike.operators.Patch() as op:
    fwd1 = op.fwd(images=A(images), positions=A(positions), patch_width=pw)
    padded = cp.zeros((ntheta, nscan * 2, pw + 6, pw + 6), dtype=np.complex64)
    fwd2 = op.fwd(images=A(images), positions=A(positions), patches=padded,
                  patch_width=pw, nrepeat=2)
    adj1 = op.adj(positions=A(positions), patches=A(patches_in),
                  patch_width=pw, height=H, width=W)
    padded_in = rc(rng, ntheta, nscan * 2, pw + 6, pw + 6)
    adj2 = op.adj(positions=A(positions), patches=A(padded_in),
                  patch_width=pw, height=H, width=W, nrepeat=2)
    # K = 1 broadcast (the psi-preconditioner call shape).  Single image only:
    # with nimage > 1 the reference kernel's image_offset assumes N*nrepeat
    # patches per image and reads out of bounds (convolution.cu:96).
    bcast_in = rc(rng, 1, pw, pw)
    adj3 = op.adj(positions=A(positions[0]), patches=A(bcast_in),
                  patch_width=pw, height=H, width=W, nrepeat=1)
save("op_patch.npz", images=images, positions=positions, pw=pw,
     patches_in=patches_in, fwd1=fwd1, fwd2=fwd2, adj1=adj1,
     padded_in=padded_in, adj2=adj2, bcast_in=bcast_in, adj3=adj3)

# Ptycho (tests/operators/test_ptycho.py: pw=15, det=45, nprobe=3, psi 128^2)
for tag, (nscan, pw, det, S, HW, shared) in {
        "odd": (11, 15, 45, 3, 128, False),
        "pow2": (9, 16, 32, 2, 64, True),
        "full": (6, 32, 32, 2, 80, False),
}.items():
    scan_ = (rng.random((nscan, 2), dtype=np.float32) * (HW - pw - 3) +
             1).astype(np.float32)
    probe_ = rc(rng, 1 if shared else nscan, 1, S, pw, pw)
    psi_ = rc(rng, 1, HW, HW)
    far_ = rc(rng, nscan, 1, S, det, det)
    with tike.operators.Ptycho(nscan=nscan, probe_shape=pw, detector_shape=det,
                               nz=HW, n=HW, **PHYS) as op:
        fwd = op.fwd(probe=A(probe_), scan=A(scan_), psi=A(psi_))
        bprobe = np.broadcast_to(probe_, (nscan, 1, S, pw, pw)).copy()
        psi_adj, probe_adj = op.adj(farplane=A(far_), probe=A(bprobe),
                                    scan=A(scan_), psi=A(psi_))
        inten, _ = op._compute_intensity(None, A(psi_), A(scan_), A(probe_))
        data_ = (rng.random((nscan, det, det), dtype=np.float32) *
                 np.asarray(inten).max())
        cost_g = op.cost(A(data_), A(psi_), A(scan_), A(probe_),
                         model="gaussian")
        cost_p = op.cost(A(data_), A(psi_), A(scan_), A(probe_),
                         model="poisson")
        gg = tike.operators.gaussian_grad(A(data_), fwd, inten)
        pg = tike.operators.poisson_grad(A(data_), fwd, inten)
        ge = tike.operators.gaussian_each_pattern(A(data_), inten)
        pe = tike.operators.poisson_each_pattern(A(data_), inten)
    save(f"op_ptycho_{tag}.npz", scan=scan_, probe=probe_, psi=psi_,
         farplane_in=far_, fwd=fwd, psi_adj=psi_adj, probe_adj=probe_adj,
         intensity=inten, data=data_, cost_gaussian=cost_g,
         cost_poisson=cost_p, gaussian_grad=gg, poisson_grad=pg,
         gaussian_each=ge, poisson_each=pe, det=det)


# ---- synthetic ptychography problem ---------------------------------------
def make_problem(rng, N, pw, det, S, pitch=3.0, eigen=0, margin=0,
                 position_error=0.0, eigen_modes=1):
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side),
                              indexing="ij"), -1).reshape(-1, 2)[:N]
    scan = (2 + margin + pitch * ij + rng.random((N, 2))).astype(np.float32)
    HW = int(np.ceil(pitch * (side - 1) + pw + 6 + 2 * margin))
    psi_true = ((0.75 + 0.25 * rng.random((1, HW, HW))) * np.exp(
        1j * np.pi * (rng.random((1, HW, HW)) - 0.5))).astype(np.complex64)
    w = tike.ptycho.probe.gaussian(pw, rin=0.6, rout=1.0)
    probe = np.stack([
        w * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
        for m in range(S)
    ])[None, None].astype(np.complex64)
    scan_true = scan
    if position_error > 0:
        # the data is simulated at perturbed positions; the reconstruction
        # starts from the nominal ones
        scan_true = (scan + position_error *
                     (2 * rng.random((N, 2)) - 1)).astype(np.float32)
    data = tike.ptycho.simulate(detector_shape=det, probe=probe,
                                scan=scan_true, psi=psi_true, **PHYS)
    psi0 = np.full((1, HW, HW), 0.5 + 0j, dtype=np.complex64)
    probe0 = (probe * (1 + 0.1 * rc(rng, *probe.shape))).astype(np.complex64)
    eigen_probe = eigen_weights = None
    if eigen > 0:
        np.random.seed(int(rng.integers(1 << 30)))
        tike.random.randomizer_np = np.random.default_rng(
            int(rng.integers(1 << 30)))
        eigen_probe, eigen_weights = tike.ptycho.probe.init_varying_probe(
            scan, probe0, num_eigen_probes=eigen + 1,
            probes_with_modes=eigen_modes)
    return dict(scan=scan, scan_true=scan_true, psi_true=psi_true,
                probe_true=probe, data=data,
                psi0=psi0, probe0=probe0, eigen_probe=eigen_probe,
                eigen_weights=eigen_weights, det=det)


# ---- lstsq pieces: one minibatch through the reference internals ----------
def lstsq_parts(tag, N, pw, det, S, eigen, slim=False, rng=None):
    """slim: large tiles (128^2 / 256^2) -- keep only what the parity test
    compares (mode 0 of chi, no per-position probe arrays)."""
    rng = rng or globals()["rng"]
    p = make_problem(rng, N, pw, det, S, eigen=eigen, pitch=5.0 if slim else 3.0)
    HW = p["psi0"].shape[-1]
    psi = (p["psi0"] * (1 + 0.2 * rc(rng, 1, HW, HW))).astype(np.complex64)
    measured = np.ones((det, det), dtype=bool)
    params = tike.ptycho.PtychoParameters(
        probe=p["probe0"].copy(), psi=psi.copy(), scan=p["scan"].copy(),
        eigen_probe=None if p["eigen_probe"] is None else
        p["eigen_probe"].copy(),
        eigen_weights=None if p["eigen_weights"] is None else
        p["eigen_weights"].copy(),
        algorithm_options=tike.ptycho.LstsqOptions(num_batch=2),
        probe_options=tike.ptycho.ProbeOptions(),
        object_options=tike.ptycho.ObjectOptions(),
        exitwave_options=tike.ptycho.ExitWaveOptions(measured_pixels=measured),
    ).copy_to_device()
    batches = np.array_split(np.arange(N), 2)
    with tike.operators.Ptycho(probe_shape=pw, detector_shape=det, nz=HW,
                               n=HW, **PHYS) as op:
        psi_pre = _psi_preconditioner(params, [], operator=op)
        probe_pre = _probe_preconditioner(params, [], operator=op)
        out = {}
        bi = 1
        (chi, uprobe, probe_update, object_upd_sum, m_probe_update, costs,
         patches, _, _, _) = ref_lstsq._get_nearplane_gradients(
             p["data"], params.psi, params.scan, params.probe,
             params.eigen_probe, params.eigen_weights, batches, None, None,
             None, [], cp.asarray(measured), psi_pre, batch_index=bi,
             num_batch=2, op=op, recover_psi=True, recover_probe=True,
             recover_positions=False,
             exitwave_options=params.exitwave_options)
        out.update(chi=chi, unique_probe=uprobe, probe_update=probe_update,
                   object_upd_sum=object_upd_sum,
                   m_probe_update=m_probe_update, costs=costs,
                   patches=patches)
        if params.eigen_weights is not None:
            ep, ew = ref_lstsq._update_nearplane(
                chi, probe_update, m_probe_update, params.probe,
                None if params.eigen_probe is None else
                params.eigen_probe.copy(), params.eigen_weights.copy(),
                patches, batches, batch_index=bi, num_batch=2)
            out.update(eigen_probe_out=ep, eigen_weights_out=ew)
        precond, beta_o, beta_p = ref_lstsq._precondition_nearplane_gradients(
            chi, params.scan, uprobe, params.probe, object_upd_sum,
            m_probe_update, psi_pre, patches, batches, batch_index=bi, op=op,
            m=0, recover_psi=True, recover_probe=True,
            probe_options=params.probe_options)
        out.update(object_update_precond=precond, beta_object=beta_o,
                   beta_probe=beta_p)
    if slim:
        out["chi"] = np.asarray(out["chi"])[:, :, :1]
        for k in ("unique_probe", "probe_update", "patches",
                  "object_update_precond"):
            del out[k]
    extra = {}
    if p["eigen_probe"] is not None:
        extra["eigen_probe"] = p["eigen_probe"]
    if p["eigen_weights"] is not None:
        extra["eigen_weights"] = p["eigen_weights"]
    save(f"lstsq_parts_{tag}.npz", data=p["data"], psi=psi,
         probe=p["probe0"], scan=p["scan"], det=det, batch_lo=batches[bi][0],
         batch_hi=batches[bi][-1] + 1, psi_precond=psi_pre,
         probe_precond=probe_pre, **extra, **out)


def multislice_fixture():
    """Multislice / FresnelSpectProp / Ptycho with D = 3 object slices, run by
    the reference (tests/operators/test_multislice.py shapes, scaled down)."""
    r = np.random.default_rng(7)
    D, N, S, pw, HW = 3, 5, 2, 16, 40
    scan_ = (r.random((N, 2), dtype=np.float32) * (HW - pw - 3) + 1).astype(
        np.float32)
    probe_ = rc(r, N, S, pw, pw)
    psi_ = rc(r, D, HW, HW)
    near_ = rc(r, N, S, pw, pw)
    far_ = rc(r, N, 1, S, pw, pw)
    with tike.operators.Multislice(probe_shape=pw, detector_shape=pw, nz=HW,
                                   n=HW, **PHYS) as op:
        prop = op.propagation._create_fresnel_spectrum_propagator(
            (pw, pw), op.propagation.probe_FOV, op.propagation.distance,
            op.propagation.wavelength)
        fresnel_f = op.propagation.fwd(A(near_))
        fresnel_a = op.propagation.adj(A(near_))
        ms_fwd = op.fwd(probe=A(probe_), scan=A(scan_), psi=A(psi_))
        ms_psi_adj, ms_probe_adj = op.adj(nearplane=A(near_), probe=A(probe_),
                                          scan=A(scan_), psi=A(psi_))
        ms_exit, ms_probes = op.fwd_return_intermediate_probes(
            probe=A(probe_[:, None]), scan=A(scan_), psi=A(psi_))
    with tike.operators.Ptycho(probe_shape=pw, detector_shape=pw, nz=HW,
                               n=HW, **PHYS) as op:
        pt_fwd = op.fwd(probe=A(probe_[:, None]), scan=A(scan_), psi=A(psi_))
        pt_psi_adj, pt_probe_adj = op.adj(farplane=A(far_),
                                          probe=A(probe_[:, None]),
                                          scan=A(scan_), psi=A(psi_))
    save("op_multislice.npz", scan=scan_, probe=probe_, psi=psi_,
         nearplane_in=near_, farplane_in=far_, propagator=prop,
         fresnel_fwd=fresnel_f, fresnel_adj=fresnel_a, ms_fwd=ms_fwd,
         ms_psi_adj=ms_psi_adj, ms_probe_adj=ms_probe_adj, ms_exit=ms_exit,
         ms_probes=ms_probes, pt_fwd=pt_fwd, pt_psi_adj=pt_psi_adj,
         pt_probe_adj=pt_probe_adj,
         phys=np.array([PHYS["probe_wavelength"], PHYS["probe_FOV_lengths"][0],
                        PHYS["probe_FOV_lengths"][1],
                        PHYS["multislice_propagation_distance"]]))


ONLY = os.environ.get("TIKE_FIXTURES_ONLY")  # e.g. "big": new fixtures only
if ONLY == "multislice":
    multislice_fixture()
    sys.exit(0)
if ONLY == "big":
    # the tile sizes of the fused FFT kernels (v2 engine, position-major
    # forward, far-plane-free inverse): one minibatch each, own generators so
    # that adding them leaves every other fixture bit-identical
    lstsq_parts("eigen128", N=6, pw=128, det=128, S=8, eigen=1, slim=True,
                rng=np.random.default_rng(128))
    lstsq_parts("eigen256", N=4, pw=256, det=256, S=3, eigen=1, slim=True,
                rng=np.random.default_rng(256))
    sys.exit(0)

lstsq_parts("plain", N=24, pw=16, det=16, S=2, eigen=0)
lstsq_parts("eigen", N=24, pw=16, det=24, S=3, eigen=2)


# ---- full reconstructions (3 epochs, called twice like ReconstructTwice) ---
def recon(tag, N, pw, det, S, eigen, num_batch, batch_method, epochs,
          adaptive=False, orth=False, rng=None, noise_model="gaussian",
          usemodes="all_modes", mask_frac=0.0, scaling=1.0, positions=None,
          position_error=0.0, psi_true_start=False, algo="lstsq", alpha=None,
          no_probe=False, depth=1, probe_extra=None, object_extra=None,
          algo_extra=None, eigen_modes=1):
    rng = globals()["rng"] if rng is None else rng
    p = make_problem(rng, N, pw, det, S, eigen=eigen,
                     margin=8 if positions else 0,
                     position_error=position_error, eigen_modes=eigen_modes)
    if depth > 1:
        # multislice object: every slice starts from the same guess, slightly
        # perturbed so that the slices are distinguishable
        p["psi0"] = (np.repeat(p["psi0"], depth, axis=0) *
                     (1 + 0.05 * rc(rng, depth, *p["psi0"].shape[1:]))).astype(
                         np.complex64)
    if psi_true_start:
        # position correction needs object structure to lock on to
        p["psi0"] = (p["psi_true"] * (1 + 0.05 * rc(rng, *p["psi_true"].shape))
                     ).astype(np.complex64)
    np.random.seed(7)
    tike.random.randomizer_np = np.random.default_rng(11)
    measured = np.ones((det, det), dtype=bool)
    if mask_frac > 0:
        # unmeasured pixels hold NaN, as in the reference's own tests
        # (tests/ptycho/test_ptycho.py:334,553)
        measured = rng.random((det, det)) > mask_frac
        p["data"] = p["data"].copy()
        p["data"][:, ~measured] = np.nan
    params = tike.ptycho.PtychoParameters(
        probe=p["probe0"].copy(), psi=p["psi0"].copy(),
        scan=p["scan"].copy(),
        eigen_probe=None if p["eigen_probe"] is None else
        p["eigen_probe"].copy(),
        eigen_weights=None if p["eigen_weights"] is None else
        p["eigen_weights"].copy(),
        algorithm_options=(tike.ptycho.RpieOptions(
            num_batch=num_batch, batch_method=batch_method, num_iter=epochs,
            **({} if alpha is None else dict(alpha=alpha)))
                           if algo == "rpie" else tike.ptycho.LstsqOptions(
            num_batch=num_batch, batch_method=batch_method, num_iter=epochs,
            **(algo_extra or {}))),
        # (the reference reads the wavelength from probe_options, so "no
        # probe recovery" is a start epoch that is never reached)
        probe_options=tike.ptycho.ProbeOptions(
            force_orthogonality=orth, use_adaptive_moment=adaptive,
            update_start=10**6 if no_probe else 0, **PHYS_PROBE,
            **(probe_extra or {})),
        object_options=tike.ptycho.ObjectOptions(
            use_adaptive_moment=adaptive, **PHYS_OBJECT,
            **(object_extra or {})),
        exitwave_options=tike.ptycho.ExitWaveOptions(
            measured_pixels=measured, noise_model=noise_model,
            step_length_usemodes=usemodes,
            unmeasured_pixels_scaling=scaling),
        position_options=None if positions is None else
        tike.ptycho.PositionOptions(p["scan"].copy(), **positions),
    )
    # record the batches the reference's clustering chooses (host-side,
    # out of scope): one worker => order = concatenated batches
    with tike.ptycho.Reconstruction(p["data"], params, 1, False) as ctx:
        order = np.asarray(ctx.comm.order[0])
        batches = [np.asarray(b) for b in ctx.batches[0]]
        perms = []
        orig_perm = tike.random.randomizer_np.permutation
        ctx.iterate(epochs)
        r1 = ctx.get_result()
    # second call continues from r1 (state round trip), as ReconstructTwice
    np.random.seed(7)
    # == tike.ptycho.reconstruct(p["data"], r1, 1, False), with the second
    # call's clustering recorded (it differs once positions have moved)
    with tike.ptycho.Reconstruction(p["data"], r1, 1, False) as ctx:
        order_2 = np.asarray(ctx.comm.order[0])
        batches_2 = [np.asarray(b) for b in ctx.batches[0]]
        ctx.iterate(r1.algorithm_options.num_iter)
        r2 = ctx.get_result()
    extra = {}
    for k in ("eigen_probe", "eigen_weights"):
        if p[k] is not None:
            extra[k] = p[k]
            extra[k + "_1"] = getattr(r1, k)
    if positions is not None:
        extra.update(
            scan_true=p["scan_true"], scan_1=r1.scan, scan_2=r2.scan,
            transform_1=r1.position_options.transform.asbuffer(),
            transform_2=r2.position_options.transform.asbuffer(),
            momentum_1=r1.position_options._momentum
            if positions.get("use_adaptive_moment") else np.zeros(0),
            position_keys=np.array(sorted(positions)),
            position_vals=np.array([float(positions[k])
                                    for k in sorted(positions)]))
    save(f"{algo}_recon_{tag}.npz", data=p["data"], psi0=p["psi0"],
         probe0=p["probe0"], scan=p["scan"], det=det, order=order,
         batch_sizes=np.array([len(b) for b in batches]), order_2=order_2,
         batch_sizes_2=np.array([len(b) for b in batches_2]),
         num_batch=num_batch, batch_method=batch_method, epochs=epochs,
         adaptive=adaptive, orth=orth, measured=measured,
         noise_model=noise_model, usemodes=usemodes, scaling=scaling,
         alpha=-1.0 if alpha is None else alpha, no_probe=no_probe,
         extras=np.array(json.dumps(dict(probe=probe_extra or {},
                                         object=object_extra or {},
                                         algorithm=algo_extra or {}))),
         phys=np.array([PHYS_PROBE.get("probe_wavelength", np.nan),
                        *PHYS_PROBE.get("probe_FOV_lengths", (np.nan, np.nan)),
                        PHYS_OBJECT.get("multislice_propagation_distance",
                                        1e-9)]),
         psi_1=r1.psi, probe_1=r1.probe,
         costs_1=np.array(r1.algorithm_options.costs[:epochs]),
         costs_2=np.array(r2.algorithm_options.costs), psi_2=r2.psi,
         probe_2=r2.probe, **extra)
    print(tag, "costs:", np.array(r2.algorithm_options.costs).ravel())


if ONLY == "rpie":
    # rpie (solvers/rpie.py:26-612).  With the default
    # alpha = 0.05 the probe step of this snapshot (denominator alpha *
    # max(preconditioner), rpie.py:271-280) diverges, so the
    # fixtures pin (a) alpha = 1 (ePIE: object + probe), (b) the default alpha
    # with the object alone, (c) a two-slice object through the multislice
    # forward model (own generators: the other fixtures stay bit-identical).
    recon("epie", N=40, pw=24, det=32, S=2, eigen=0, num_batch=2,
          batch_method="compact", epochs=3, orth=True, algo="rpie", alpha=1.0,
          rng=np.random.default_rng(99))
    recon("object", N=36, pw=16, det=16, S=1, eigen=0, num_batch=3,
          batch_method="wobbly_center", epochs=3, algo="rpie", no_probe=True,
          rng=np.random.default_rng(98), mask_frac=0.1, scaling=0.9)
    PHYS_PROBE.update(probe_wavelength=1e-10, probe_FOV_lengths=(1e-5, 1e-5))
    PHYS_OBJECT.update(multislice_propagation_distance=2e-4)
    recon("twoslice", N=30, pw=16, det=16, S=2, eigen=0, num_batch=2,
          batch_method="compact", epochs=2, algo="rpie", alpha=1.0, depth=2,
          rng=np.random.default_rng(97))
    sys.exit(0)
if ONLY == "lstsq2":
    # round 4: the reference's "no probe" lstsq_grad test configurations
    # (tests/ptycho/test_ptycho.py:390-407,433-452): object recovery alone,
    # with momentum, random and compact minibatch order
    recon("noprobe", N=40, pw=24, det=32, S=2, eigen=0, num_batch=3,
          batch_method="wobbly_center", epochs=3, adaptive=True, no_probe=True,
          rng=np.random.default_rng(93))
    recon("compact_noprobe", N=36, pw=16, det=16, S=1, eigen=0, num_batch=2,
          batch_method="compact", epochs=4, adaptive=True, no_probe=True,
          rng=np.random.default_rng(92))
    sys.exit(0)
if ONLY == "bootstrap":
    # round 4: the third minibatch selector the reference's options accept
    # (cluster.py:380-462): random 95 % start, wobbly-center for the rest
    recon("bootstrap", N=47, pw=16, det=24, S=2, eigen=0, num_batch=3,
          batch_method="wobbly_center_random_bootstrap", epochs=3,
          rng=np.random.default_rng(90))
    sys.exit(0)
if ONLY == "positions2":
    # round 4: position correction on data with unmeasured detector regions
    # (tests/ptycho/test_position.py:373-412): NaN-masked patterns
    recon("positions_masked", N=36, pw=24, det=24, S=2, eigen=0, num_batch=2,
          batch_method="compact", epochs=3, orth=True,
          rng=np.random.default_rng(91),
          positions=dict(use_adaptive_moment=True,
                         update_magnitude_limit=5), position_error=0.8,
          psi_true_start=True, mask_frac=0.1, scaling=0.9)
    sys.exit(0)
if ONLY == "constraints":
    # round 4: every probe / object constraint of ptycho.py:723-854 switched
    # on in one run (their order of application is part of the result)
    import cupyx.scipy.ndimage as _cnd
    import scipy.ndimage as _snd
    cp.unravel_index = lambda indices, dims, order="C": np.unravel_index(
        indices, dims, order=order)
    _cnd.median_filter = lambda input, size, **kw: _snd.median_filter(
        input, size=tuple(int(v) for v in size), **kw)
    recon("constraints", N=40, pw=24, det=24, S=2, eigen=0, num_batch=2,
          batch_method="compact", epochs=3, orth=True,
          rng=np.random.default_rng(90),
          probe_extra=dict(probe_support=0.3, additional_probe_penalty=0.05,
                           median_filter_abs_probe=True,
                           median_filter_abs_probe_px=(3.0, 3.0),
                           force_centered_intensity=True, force_sparsity=0.1),
          object_extra=dict(positivity_constraint=0.3,
                            smoothness_constraint=0.05, clip_magnitude=True),
          algo_extra=dict(rescale_period=2))
    recon("constraints_photons", N=36, pw=16, det=16, S=2, eigen=0,
          num_batch=2, batch_method="compact", epochs=3, orth=False,
          rng=np.random.default_rng(89),
          probe_extra=dict(probe_photons=5e3, probe_support=0.2),
          object_extra=dict(clip_magnitude=True),
          algo_extra=dict(rescale_method="constant_probe_photons",
                          rescale_period=2))
    sys.exit(0)
if ONLY == "eigen2":
    # round 4: eigen probes on two of three modes (probes_with_modes = 2):
    # the forward model varies both, the update touches mode 0 only
    # (lstsq.py:169,297-364); one and two eigen probes
    recon("eigen_modes2", N=40, pw=24, det=24, S=3, eigen=1, num_batch=2,
          batch_method="compact", epochs=3, orth=True, eigen_modes=2,
          rng=np.random.default_rng(88))
    recon("eigen2_modes2", N=36, pw=16, det=32, S=2, eigen=2, num_batch=2,
          batch_method="wobbly_center", epochs=3, orth=True, eigen_modes=2,
          rng=np.random.default_rng(87))
    sys.exit(0)
if ONLY == "cgrad2":
    # round 4: the cgrad composition with the PROBE step and two minibatches:
    # per minibatch, tike.opt.conjugate_gradient on psi (cost = Ptycho.cost,
    # gradient = Ptycho.adj(gaussian_grad)[0]) then on the probe (gradient =
    # sum over the positions of Ptycho.adj(...)[1]), as
    # lamino/solvers/cgrad.py:58-92 alternates its variables
    rng_c = np.random.default_rng(86)
    p = make_problem(rng_c, 40, 24, 24, 2)
    HW = p["psi0"].shape[-1]
    halves = np.array_split(np.arange(40), 2)
    with tike.operators.Ptycho(probe_shape=24, detector_shape=24, nz=HW, n=HW,
                               **PHYS) as op:
        psi, probe = A(p["psi0"]), A(p["probe0"])
        scan_all, data_all = A(p["scan"]), A(p["data"])
        epoch_costs, psis, probes = [], [], []
        for _ in range(3):
            batch_cost = []
            for b in halves:
                scan, data = scan_all[b], data_all[b]

                def far_gradient(psi_, probe_):
                    inten, far = op._compute_intensity(data, psi_, scan, probe_)
                    return tike.operators.gaussian_grad(data, far, inten)

                def wide(probe_):
                    return cp.asarray(np.broadcast_to(
                        probe_, (len(scan), *probe_.shape[1:])).copy())

                psi, c = tike.opt.conjugate_gradient(
                    cp, x=psi,
                    cost_function=lambda x: float(op.cost(
                        data, x, scan, probe, model="gaussian")),
                    grad=lambda x: [op.adj(farplane=far_gradient(x, probe),
                                           probe=wide(probe), scan=scan,
                                           psi=x)[0]],
                    dir_multi=lambda d: d[0], num_iter=3, step_length=1.0)
                psi = cp.asarray(psi, dtype=np.complex64)
                probe, c = tike.opt.conjugate_gradient(
                    cp, x=probe,
                    cost_function=lambda q: float(op.cost(
                        data, psi, scan, q, model="gaussian")),
                    grad=lambda q: [cp.sum(op.adj(
                        farplane=far_gradient(psi, q), probe=wide(q),
                        scan=scan, psi=psi)[1], axis=0, keepdims=True)],
                    dir_multi=lambda d: d[0], num_iter=3, step_length=1.0)
                probe = cp.asarray(probe, dtype=np.complex64)
                batch_cost.append(float(c))
            epoch_costs.append(float(np.mean(batch_cost)))
            psis.append(np.asarray(psi).copy())
            probes.append(np.asarray(probe).copy())
    save("cgrad_probe.npz", data=p["data"], psi0=p["psi0"],
         probe0=p["probe0"], scan=p["scan"], det=24, cg_iter=3,
         costs=np.array(epoch_costs), psis=np.stack(psis),
         probes=np.stack(probes))
    print("cgrad_probe costs", epoch_costs)
    sys.exit(0)
if ONLY == "rpie2":
    # round 4: the reference's other rpie test configurations
    # (tests/ptycho/test_ptycho.py:490-543,670-700): the Poisson noise model
    # with per-mode step lengths and a variable (eigen) probe, alpha = 1
    recon("poisson", N=40, pw=24, det=32, S=2, eigen=0, num_batch=2,
          batch_method="compact", epochs=3, orth=True, algo="rpie", alpha=1.0,
          rng=np.random.default_rng(96), noise_model="poisson",
          usemodes="dominant_mode")
    recon("poisson_all", N=36, pw=16, det=16, S=3, eigen=0, num_batch=2,
          batch_method="wobbly_center", epochs=3, orth=True, algo="rpie",
          alpha=1.0, rng=np.random.default_rng(95), noise_model="poisson",
          usemodes="all_modes", mask_frac=0.1, scaling=0.9)
    recon("eigen", N=40, pw=24, det=24, S=2, eigen=2, num_batch=2,
          batch_method="compact", epochs=3, orth=True, algo="rpie", alpha=1.0,
          rng=np.random.default_rng(94))
    sys.exit(0)

recon("compact", N=48, pw=24, det=32, S=2, eigen=0, num_batch=2,
      batch_method="compact", epochs=3, adaptive=True, orth=True)
recon("wobbly_eigen", N=40, pw=32, det=32, S=3, eigen=2, num_batch=2,
      batch_method="wobbly_center", epochs=3, orth=True)

# ---- cgrad composition (absent as a solver) ---------------
p = make_problem(rng, 36, 16, 16, 1)
HW = p["psi0"].shape[-1]
with tike.operators.Ptycho(probe_shape=16, detector_shape=16, nz=HW, n=HW,
                           **PHYS) as op:
    probe = A(p["probe_true"])
    scan = A(p["scan"])
    data = A(p["data"])

    def cost_function(psi):
        return float(op.cost(data, psi, scan, probe, model="gaussian"))

    def grad(psi):
        inten, far = op._compute_intensity(data, psi, scan, probe)
        g = tike.operators.gaussian_grad(data, far, inten)
        bp = cp.asarray(
            np.broadcast_to(probe, (len(scan), *probe.shape[1:])).copy())
        return [op.adj(farplane=g, probe=bp, scan=scan, psi=psi)[0]]

    def dir_multi(d):
        return d[0]

    psi = A(p["psi0"])
    costs, psis = [cost_function(psi)], []
    for _ in range(3):
        psi, c = tike.opt.conjugate_gradient(
            cp, x=psi, cost_function=cost_function, grad=grad,
            dir_multi=dir_multi, num_iter=4, step_length=1.0)
        psi = cp.asarray(psi, dtype=np.complex64)
        costs.append(float(c))
        psis.append(np.asarray(psi).copy())
save("cgrad.npz", data=p["data"], psi0=p["psi0"], probe=p["probe_true"],
     scan=p["scan"], det=16, costs=np.array(costs), psis=np.stack(psis))
print("cgrad costs", costs)


# ---- Poisson noise model with per-mode step lengths (SURVEY 8f rank 2;
# lstsq.py:454-489, exitwave.py:122-234), NaN-masked data --------------------
rng_p = np.random.default_rng(4321)
recon("poisson_all", N=40, pw=24, det=32, S=3, eigen=0, num_batch=2,
      batch_method="compact", epochs=3, orth=True, rng=rng_p,
      noise_model="poisson", usemodes="all_modes", mask_frac=0.1, scaling=0.9)
recon("poisson_dominant", N=40, pw=32, det=32, S=2, eigen=0, num_batch=2,
      batch_method="compact", epochs=3, orth=True, rng=rng_p,
      noise_model="poisson", usemodes="dominant_mode")


# ---- position correction inside lstsq_grad (SURVEY 8f rank 1;
# lstsq.py:545-579,764-806; position.py:716-810) ------------------------------
rng_q = np.random.default_rng(8765)
recon("positions_adam", N=36, pw=24, det=24, S=2, eigen=0, num_batch=2,
      batch_method="compact", epochs=3, orth=True, rng=rng_q,
      positions=dict(use_adaptive_moment=True,
                     use_position_regularization=True,
                     update_magnitude_limit=5), position_error=1.0,
      psi_true_start=True)
recon("positions_plain", N=30, pw=16, det=32, S=1, eigen=0, num_batch=3,
      batch_method="wobbly_center", epochs=3, rng=rng_q,
      positions=dict(), position_error=0.7, psi_true_start=True)


# rpie is NOT pinned: at this snapshot the reference's own
# rpie diverges on these problems (costs 0.024 -> 0.69 -> 41 -> ...; NaN with
# eigen probes), so its iterates are no usable golden vectors.
# recon(..., algo="rpie") reproduces that observation.


# ---- reconstruct_multigrid (ptycho.py:975-1047,
# options.py:332-409): coarse-to-fine reconstruction, 2 levels ------------------
def multigrid(tag, interp):
    rng_m = np.random.default_rng(1357)
    p = make_problem(rng_m, N=30, pw=32, det=32, S=2, pitch=4.0, margin=6)
    np.random.seed(7)
    tike.random.randomizer_np = np.random.default_rng(11)
    params = tike.ptycho.PtychoParameters(
        probe=p["probe0"].copy(), psi=p["psi0"].copy(), scan=p["scan"].copy(),
        algorithm_options=tike.ptycho.LstsqOptions(
            num_batch=2, batch_method="compact", num_iter=2),
        probe_options=tike.ptycho.ProbeOptions(force_orthogonality=True),
        object_options=tike.ptycho.ObjectOptions(),
        exitwave_options=tike.ptycho.ExitWaveOptions(
            measured_pixels=np.ones((32, 32), dtype=bool)),
    )
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        coarse = params.resample(0.5, interp)
        r = tike.ptycho.reconstruct_multigrid(p["data"], params, 1, False,
                                              num_levels=2, interp=interp)
    save(f"multigrid_{tag}.npz", data=p["data"], psi0=p["psi0"],
         probe0=p["probe0"], scan=p["scan"], coarse_probe=coarse.probe,
         coarse_psi=coarse.psi, coarse_scan=coarse.scan, psi=r.psi,
         probe=r.probe, costs=np.array(r.algorithm_options.costs))
    print("multigrid", tag, np.array(r.algorithm_options.costs).ravel())


multigrid("fft", None)
