"""GPU parity tests of the operator layer: HIP kernels (through the C ABI and
the tike-compatible Python classes) vs the CPU oracle, the reference-run
fixtures, and the reference's own known-answer / adjoint-identity tests."""
import numpy as np
import pytest

from util import assert_close, relerr, COST_RTOL

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    import tike_amd.operators as o
    return o


@pytest.fixture(scope="module")
def oracle():
    from oracle import operators as o
    return o


def rc(rng, *shape):
    return (rng.random((*shape, 2), dtype=np.float32) - 0.5).view(
        np.complex64)[..., 0]


def inner(x, y):
    return np.sum(x * np.conj(y))


# ---------------------------------------------------------------- Patch
def test_patch_correctness(ops):
    """reference tests/operators/test_patch.py:64-133 (atol 1e-6)."""
    size, win = 256, 8
    rng = np.random.default_rng(0)
    fov = rc(rng, size, size)
    sub = 0.12346789
    positions = np.array([[0, 0], [0, size - win], [size - win, 0],
                          [size - win, size - win],
                          [size // 2 - win // 2, size // 2 - win // 2],
                          [sub, 3]], dtype=np.float32)
    truth = np.stack((
        fov[:win, :win], fov[:win, -win:], fov[-win:, :win], fov[-win:, -win:],
        fov[size // 2 - win // 2:size // 2 + win // 2,
            size // 2 - win // 2:size // 2 + win // 2],
        (1.0 - sub) * fov[0:win, 3:3 + win] + sub * fov[1:1 + win, 3:3 + win],
    ), axis=0)
    with ops.Patch() as op:
        patches = op.fwd(images=fov, positions=positions, patch_width=win)
    np.testing.assert_allclose(patches, truth, atol=1e-6)


def test_patch_correctness_adjoint(ops):
    """reference tests/operators/test_patch.py:136-206 (atol 1e-6)."""
    size, win = 8, 2
    positions = np.array([[0, 0], [0, size - win], [size - win, 0],
                          [size - win, size - win],
                          [size // 2 - win // 2, size // 2 - win // 2],
                          [0.123, 3], [3, 0.123], [5.5, 3.5]],
                         dtype=np.float32)
    fov = np.zeros((size, size), dtype=np.complex64)
    fov[:win, :win] += 1
    fov[:win, -win:] += 1
    fov[-win:, :win] += 1
    fov[-win:, -win:] += 1
    fov[3:5, 3:5] += 1
    fov[0:win, 3:3 + win] += (1 - 0.123)
    fov[1:1 + win, 3:3 + win] += 0.123
    fov[3:3 + win, 0:win] += (1 - 0.123)
    fov[3:3 + win, 1:1 + win] += 0.123
    fov[5:5 + win, 3:3 + win] += 0.25
    fov[6:6 + win, 3:3 + win] += 0.25
    fov[5:5 + win, 4:4 + win] += 0.25
    fov[6:6 + win, 4:4 + win] += 0.25
    with ops.Patch() as op:
        combined = op.adj(
            patches=np.ones((len(positions), win, win), dtype=np.complex64),
            positions=positions, patch_width=win,
            images=np.zeros((size, size), dtype=np.complex64))
    np.testing.assert_allclose(combined, fov, atol=1e-6)


def test_patch_vs_reference_fixture(ops, golden):
    g = golden("op_patch.npz")
    pw = int(g["pw"])
    H, W = g["images"].shape[-2:]
    with ops.Patch() as op:
        assert_close(op.fwd(g["images"], g["positions"], patch_width=pw),
                     g["fwd1"], what="fwd1")
        assert_close(
            op.fwd(g["images"], g["positions"], np.zeros_like(g["fwd2"]),
                   patch_width=pw, nrepeat=2), g["fwd2"], what="fwd2")
        assert_close(
            op.adj(g["positions"], g["patches_in"], patch_width=pw, height=H,
                   width=W), g["adj1"], what="adj1")
        assert_close(
            op.adj(g["positions"], g["padded_in"], patch_width=pw, height=H,
                   width=W, nrepeat=2), g["adj2"], what="adj2")
        assert_close(
            op.adj(g["positions"][0], g["bcast_in"], patch_width=pw, height=H,
                   width=W), g["adj3"], what="adj3 (K=1 broadcast)")


def test_patch_adjoint_identity(ops):
    """reference tests/operators/test_patch.py:19-61 via util.py:42-54."""
    rng = np.random.default_rng(0)
    ntheta, nscan, pw = 3, 7, 16
    scan = (rng.random((ntheta, nscan, 2)) * (256 - pw - 2)).astype(np.float32)
    m = rc(rng, ntheta, 256, 256)
    d = rc(rng, ntheta, nscan, pw, pw)
    with ops.Patch() as op:
        Fm = op.fwd(images=m, positions=scan, patch_width=pw, height=256,
                    width=256)
        Fd = op.adj(patches=d, positions=scan, patch_width=pw, height=256,
                    width=256)
    a, b = inner(Fm, d), inner(m, Fd)
    np.testing.assert_allclose([a.real, a.imag], [b.real, b.imag], rtol=1e-3)


def test_patch_empty(ops):
    with ops.Patch() as op:
        out = op.fwd(images=np.zeros((8, 8), np.complex64),
                     positions=np.zeros((0, 2), np.float32), patch_width=2)
    assert out.shape == (0, 2, 2)


# ---------------------------------------------------------- Convolution
def test_convolution_adjoint_identities(ops, oracle):
    """reference tests/operators/test_convolution.py:20-103."""
    rng = np.random.default_rng(0)
    ntheta, nscan, S, pw, det, HW = 3, 27, 3, 15, 45, 128
    scan = (rng.random((ntheta, nscan, 2)) * (HW - pw - 2) + 1).astype(
        np.float32)
    psi = rc(rng, ntheta, HW, HW)
    probe = rc(rng, ntheta, nscan, S, pw, pw)
    near = rc(rng, ntheta, nscan, S, det, det)
    with ops.Convolution(ntheta=ntheta, nscan=nscan, nz=HW, n=HW,
                         probe_shape=pw, nprobe=S, detector_shape=det) as op:
        d = op.fwd(psi=psi, scan=scan, probe=probe)
        m = op.adj(nearplane=near, scan=scan, probe=probe)
        mp = op.adj_probe(nearplane=near, scan=scan, psi=psi)
    assert d.shape == near.shape and m.shape == psi.shape
    assert mp.shape == probe.shape
    a, b, c = inner(d, near), inner(psi, m), inner(probe, mp)
    np.testing.assert_allclose([a.real, a.imag], [b.real, b.imag], rtol=1e-3)
    np.testing.assert_allclose([a.real, a.imag], [c.real, c.imag], rtol=1e-3)
    for i in range(ntheta):
        assert_close(d[i], oracle.convolution_fwd(psi[i], scan[i], probe[i],
                                                  det), what="conv fwd")
        assert_close(m[i], oracle.convolution_adj(near[i], scan[i], probe[i],
                                                  HW, HW), what="conv adj")
        assert_close(mp[i], oracle.convolution_adj_probe(near[i], scan[i],
                                                         psi[i], pw),
                     what="conv adj_probe")


# ---------------------------------------------------------- Propagation
@pytest.mark.parametrize("n", [32, 64, 128, 256, 512, 1024, 127, 45, 24, 7,
                               # round 6, the shape-general engine: mixed
                               # radix (2, 3, 5, 7, 11, 13) ...
                               96, 160, 192, 320, 384, 640, 768, 1000, 1536,
                               2048, 3072, 4096, 15, 77, 143, 1, 2, 3,
                               # ... and Bluestein (a prime factor > 13)
                               17, 34, 129, 323, 1023, 1234, 2047, 2048 - 31])
@pytest.mark.parametrize("norm", ["ortho", "forward", "backward"])
def test_propagation_vs_numpy(ops, n, norm):
    rng = np.random.default_rng(n)
    batch = 2 if n >= 2000 else 3 if n >= 512 else 13
    if norm != "ortho" and n in (3072, 4096, 1536, 2047, 1234):
        pytest.skip("large sizes: one normalisation")
    x = rc(rng, batch, n, n)
    with ops.Propagation(detector_shape=n, norm=norm) as op:
        f = op.fwd(nearplane=x)
        b = op.adj(farplane=x)
    assert_close(f, np.fft.fft2(x.astype(np.complex128), norm=norm),
                 normwise=2e-6, maxabs=2e-5, what=f"fft2 n={n}")
    assert_close(b, np.fft.ifft2(x.astype(np.complex128), norm=norm),
                 normwise=2e-6, maxabs=2e-5, what=f"ifft2 n={n}")


@pytest.mark.parametrize("n", [16, 32, 64, 128, 256, 512, 1024])
@pytest.mark.parametrize("groups", [(0, 0), (1, 1), (2, 8), (5, 3), (64, 64)])
def test_general_fft_engine_at_the_register_engines_sizes(ops, n, groups):
    """tike_fft2_general (mixed-radix lines in LDS) at the powers of two the
    register engines serve, against float64 NumPy and at every grouping of the
    lines (1 line per workgroup, unequal row / column groups, a request that
    does not fit and is cut down, groups that straddle tiles: 5 tiles of n
    lines in groups of 2..64), out of place and in place."""
    import torch
    from tike_amd import _arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(n)
    x = rc(rng, 5, n, n)
    xt = A.to_device(x, np.complex64)
    out = torch.empty_like(xt)
    for inverse, ref in ((0, np.fft.fft2), (1, np.fft.ifft2)):
        check(lib.tike_fft2_general(A.ptr(xt), A.ptr(out), 5, n, inverse,
                                    1.0 / n, groups[0], groups[1],
                                    A.stream_ptr()), "general fft2")
        assert_close(out.cpu().numpy(),
                     ref(x.astype(np.complex128), norm="ortho"),
                     normwise=2e-6, maxabs=2e-5, what=f"general n={n}")
    inplace = xt.clone()
    check(lib.tike_fft2_general(A.ptr(inplace), A.ptr(inplace), 5, n, 0,
                                1.0 / n, groups[0], groups[1],
                                A.stream_ptr()), "general fft2 in place")
    assert_close(inplace.cpu().numpy(),
                 np.fft.fft2(x.astype(np.complex128), norm="ortho"),
                 normwise=2e-6, maxabs=2e-5, what=f"general in place n={n}")


def test_fft_sizes_supported_and_refused(ops):
    """Every size a detector crop can have has a plan; beyond the engine's
    bounds (a prime factor > 13 above 2048; anything above 4096) the entry
    says so and Propagation raises -- never a wrong answer."""
    from tike_amd._lib import lib
    for n in (1, 7, 45, 96, 127, 384, 1000, 2047, 2048, 3072, 4096, 4095):
        # 4095 = 3^2 5 7 13
        assert lib.tike_fft2_supported(n) == 1, n
    for n in (0, -3, 2049, 4097, 8192, 2053):
        assert lib.tike_fft2_supported(n) == 0, n
    with ops.Propagation(detector_shape=2049) as op:
        with pytest.raises(ValueError, match="unsupported size"):
            op.fwd(nearplane=np.zeros((1, 2049, 2049), np.complex64))


@pytest.mark.parametrize("n", [127, 64, 384])
def test_propagation_adjoint_and_scaled(ops, n):
    """reference tests/operators/test_propagation.py:16-36 (13 waves of 127^2):
    <F m, d> = <m, F* d> and |F* F m| = |m| (rtol 1e-3)."""
    rng = np.random.default_rng(0)
    m, d = rc(rng, 13, n, n), rc(rng, 13, n, n)
    with ops.Propagation(nwaves=13, detector_shape=n) as op:
        Fm = op.fwd(nearplane=m)
        Fd = op.adj(farplane=d)
        FFm = op.adj(farplane=Fm)
    a, b = inner(Fm, d), inner(m, Fd)
    np.testing.assert_allclose([a.real, a.imag], [b.real, b.imag], rtol=1e-3)
    np.testing.assert_allclose(inner(FFm, FFm).real, inner(m, m).real,
                               rtol=1e-3)


def test_propagation_overwrite_and_errors(ops):
    import torch
    rng = np.random.default_rng(1)
    x = rc(rng, 5, 64, 64)
    with ops.Propagation(detector_shape=64) as op:
        t = op.asarray(x)
        out = op.fwd(nearplane=t, overwrite=True)
        assert out.data_ptr() == t.data_ptr()
        assert_close(out.cpu().numpy(), np.fft.fft2(x, norm="ortho"),
                     normwise=2e-6, maxabs=2e-5)
        with pytest.raises(ValueError):
            op.fwd(nearplane=np.zeros((2, 32, 32), np.complex64))


# --------------------------------------------------------------- Ptycho
@pytest.mark.parametrize("tag", ["odd", "pow2", "full"])
def test_ptycho_vs_reference_fixture(ops, golden, tag):
    g = golden(f"op_ptycho_{tag}.npz")
    det = int(g["det"])
    N = len(g["scan"])
    pw = g["probe"].shape[-1]
    HW = g["psi"].shape[-1]
    with ops.Ptycho(nscan=N, probe_shape=pw, detector_shape=det, nz=HW,
                    n=HW) as op:
        fwd = op.fwd(probe=g["probe"], scan=g["scan"], psi=g["psi"])
        assert_close(fwd, g["fwd"], what="fwd")
        bprobe = np.broadcast_to(g["probe"], (N, *g["probe"].shape[1:])).copy()
        psi_adj, probe_adj = op.adj(farplane=g["farplane_in"], probe=bprobe,
                                    scan=g["scan"], psi=g["psi"])
        assert_close(psi_adj, g["psi_adj"], what="psi_adj")
        assert_close(probe_adj, g["probe_adj"], what="probe_adj")
        inten, far = op._compute_intensity(None, g["psi"], g["scan"],
                                           g["probe"])
        assert_close(inten, g["intensity"], what="intensity")
        d = g["data"]
        np.testing.assert_allclose(
            float(op.cost(d, g["psi"], g["scan"], g["probe"],
                          model="gaussian")), g["cost_gaussian"],
            rtol=COST_RTOL)
        np.testing.assert_allclose(
            float(op.cost(d, g["psi"], g["scan"], g["probe"],
                          model="poisson")), g["cost_poisson"], rtol=COST_RTOL)
    assert_close(ops.gaussian_grad(d, g["fwd"], g["intensity"]),
                 g["gaussian_grad"], what="gaussian_grad")
    assert_close(ops.poisson_grad(d, g["fwd"], g["intensity"]),
                 g["poisson_grad"], normwise=1e-4, maxabs=1e-3,
                 what="poisson_grad")
    np.testing.assert_allclose(ops.gaussian_each_pattern(d, g["intensity"]),
                               g["gaussian_each"], rtol=COST_RTOL)
    np.testing.assert_allclose(ops.poisson_each_pattern(d, g["intensity"]),
                               g["poisson_each"], rtol=COST_RTOL)


def test_ptycho_adjoint_identity_reference_shapes(ops):
    """reference tests/operators/test_ptycho.py:18-75."""
    rng = np.random.default_rng(0)
    nscan, pw, S, det = 27, 15, 3, 45
    scan = (rng.random((nscan, 2), dtype=np.float32) * (127 - 16))
    scan = np.maximum(scan, 1).astype(np.float32)
    probe, psi = rc(rng, nscan, 1, S, pw, pw), rc(rng, 1, 128, 128)
    far = rc(rng, nscan, 1, S, det, det)
    with ops.Ptycho(nscan=nscan, probe_shape=pw, detector_shape=det, nz=128,
                    n=128) as op:
        d = op.fwd(scan=scan, probe=probe, psi=psi)
        assert d.shape == far.shape
        m0, m1 = op.adj(farplane=far, scan=scan, probe=probe, psi=psi)
        assert m0.shape == psi.shape and m1.shape == probe.shape
    a, b, c = inner(d, far), inner(psi, m0), inner(probe, m1)
    np.testing.assert_allclose([a.real, a.imag], [b.real, b.imag], rtol=1e-3)
    np.testing.assert_allclose([a.real, a.imag], [c.real, c.imag], rtol=1e-3)


@pytest.mark.parametrize("det,pw,S,shared", [(64, 64, 2, True),
                                             (128, 96, 3, False),
                                             (256, 256, 1, True),
                                             (64, 40, 2, True)])
def test_ptycho_fused_vs_oracle(ops, oracle, det, pw, S, shared):
    rng = np.random.default_rng(det + pw)
    N = 5
    HW = pw + 40
    scan = (rng.random((N, 2)) * 36 + 1.5).astype(np.float32)
    probe = rc(rng, 1 if shared else N, 1, S, pw, pw)
    psi = rc(rng, 1, HW, HW)
    far = rc(rng, N, 1, S, det, det)
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        fwd = op.fwd(probe=probe, scan=scan, psi=psi)
        bprobe = np.broadcast_to(probe, (N, 1, S, pw, pw)).copy()
        psi_adj, probe_adj = op.adj(farplane=far, probe=bprobe, scan=scan,
                                    psi=psi)
    assert_close(fwd, oracle.ptycho_fwd(probe, scan, psi, det), what="fwd")
    o_psi, o_probe = oracle.ptycho_adj(far, bprobe, scan, psi)
    assert_close(psi_adj, o_psi, what="psi_adj")
    assert_close(probe_adj, o_probe, what="probe_adj")


@pytest.mark.parametrize("probes", ["shared", "per_position", "eigen"])
@pytest.mark.parametrize("det,S,N,sub", [(256, 2, 40, 16), (512, 2, 21, 8)])
def test_ptycho_fwd_sub_batches_vs_oracle(ops, oracle, det, S, N, sub, probes):
    """The 256^2 / 512^2 forward operator walks its positions in sub-batches
    (two streaming kernels each): three of them here, the last one ragged, with
    a shared probe, one probe per position, and eigen probes + weights -- the
    scan, probe, weight and far-plane offsets of every sub-batch against the
    oracle, and against the same call as ONE batch."""
    from oracle import solvers as sol
    rng = np.random.default_rng(det + N)
    pw, HW = det, det + 40
    assert N > 2 * sub and N % sub != 0
    scan = (rng.random((N, 2)) * 36 + 1.5).astype(np.float32)
    psi = rc(rng, 1, HW, HW)
    probe = rc(rng, N if probes == "per_position" else 1, 1, S, pw, pw)
    eigen = w = None
    used = probe
    if probes == "eigen":
        eigen = rc(rng, 1, 2, 1, pw, pw)
        w = rng.standard_normal((N, 3, S)).astype(np.float32)
        used = sol.get_varying_probe(probe, eigen, w)
    want = oracle.ptycho_fwd(used, scan, psi, det)
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        dev = [None if x is None else op.asarray(x)
               for x in (probe, scan, psi, eigen, w)]
        got = op.fwd_device(*dev, sub_batch=sub).cpu().numpy()
        one = op.fwd_device(*dev, sub_batch=-1).cpu().numpy()
    assert_close(got, want, what=f"fwd in sub-batches of {sub} ({probes})")
    assert_close(one, want, what=f"fwd as one batch ({probes})")
    assert np.array_equal(got, one)  # the split changes no arithmetic


@pytest.mark.parametrize("det,S,N,sub,shared,layout", [
    (256, 1, 40, 16, True, "raster"),     # MW = 1: a lone mode-wave, no LDS sum
    (256, 2, 21, 8, False, "random"),     # one probe per position
    (256, 3, 19, -1, True, "random"),     # an idle mode-wave (3 modes, 4 waves)
    (256, 5, 11, 4, True, "raster"),      # two modes per wave, one of them idle
    (256, 8, 13, 5, False, "raster"),     # the headline mode count, per-position probes
    (256, 8, 10, 0, True, "random"),
    (128, 1, 50, 16, True, "raster"),
    (128, 4, 23, 9, False, "random"),
    (128, 8, 9, -1, True, "raster"),
    (512, 1, 9, 4, True, "raster"),
    (512, 2, 9, 4, False, "random"),
    (512, 4, 7, 3, True, "raster"),
    (512, 7, 5, -1, True, "random"),
])
def test_ptycho_adj_fused_vs_oracle(ops, oracle, det, S, N, sub, shared, layout):
    """tike_ptycho_adj (inverse pass 1 -> pass 2 in place with both products
    -> grouped scatter) against the oracle's Ptycho.adj and against the general
    kernels: every (mode-waves, modes per wave) instantiation, shared and
    per-position probes, sub-batches with a ragged tail, neighbouring positions
    (summed in LDS by the grouped scatter) and far-apart ones (its fallback),
    integer positions (zero-weight taps) among them."""
    import torch
    rng = np.random.default_rng(11 * det + S)
    pw = det
    if layout == "raster":
        side = int(np.ceil(np.sqrt(N)))
        ij = np.stack(np.meshgrid(np.arange(side), np.arange(side),
                                  indexing="ij"), -1).reshape(-1, 2)[:N]
        scan = (1 + 6.0 * ij + rng.random((N, 2))).astype(np.float32)
        HW = 6 * side + pw + 4
    else:
        HW = pw + 300
        scan = (rng.random((N, 2)) * (HW - pw - 3) + 1).astype(np.float32)
    scan[0] = np.floor(scan[0])          # an integer position
    scan[-1, 1] = np.floor(scan[-1, 1])  # and a half-integer one
    probe = rc(rng, 1 if shared else N, 1, S, pw, pw)
    psi = rc(rng, 1, HW, HW)
    far = rc(rng, N, 1, S, det, det)
    bprobe = np.broadcast_to(probe, (N, 1, S, pw, pw))
    o_psi, o_probe = oracle.ptycho_adj(far, bprobe, scan, psi)
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        assert op.fused_adjoint_shapes(S)
        d = [op.asarray(x) for x in (far, probe, scan, psi)]
        keep = d[0].clone()
        psi_adj, probe_adj = op.adj_device(*d, sub_batch=sub)
        assert torch.equal(d[0], keep)  # the far plane is read only
        # the same through the operator API (numpy in, numpy out)
        api_psi, api_probe = op.adj(farplane=far, probe=probe, scan=scan,
                                    psi=psi)
    assert_close(psi_adj.cpu().numpy(), o_psi, what="psi_adj")
    assert_close(probe_adj.cpu().numpy(), o_probe, what="probe_adj")
    assert_close(api_psi, o_psi, what="psi_adj (API)")
    assert_close(api_probe, o_probe, what="probe_adj (API)")
    assert api_probe.shape == (N, 1, S, pw, pw) and api_psi.shape == psi.shape


def test_ptycho_adj_positions_outside_take_the_general_kernels(ops, oracle):
    """A scan position whose taps leave the image is not for the fused
    adjoint (it drops what falls outside; the reference addresses the taps
    linearly, convolution.cu:113-133): Ptycho.adj must route it to the
    general kernels and still match the oracle."""
    rng = np.random.default_rng(2)
    det = pw = 128
    N, S, HW = 6, 2, 180
    scan = (rng.random((N, 2)) * 40 + 2).astype(np.float32)
    scan[2] = (0.0, HW - pw)  # violates check_allowed_positions on both axes
    probe, psi = rc(rng, N, 1, S, pw, pw), rc(rng, 1, HW, HW)
    far = rc(rng, N, 1, S, det, det)
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        psi_adj, probe_adj = op.adj(farplane=far, probe=probe, scan=scan,
                                    psi=psi)
    o_psi, o_probe = oracle.ptycho_adj(far, probe, scan, psi)
    assert_close(psi_adj, o_psi, what="psi_adj")
    assert_close(probe_adj, o_probe, what="probe_adj")


@pytest.mark.parametrize("n", [256, 128, 512])
@pytest.mark.parametrize("adjoint", [False, True])
def test_fresnel_step_in_three_launches_vs_oracle(oracle, adjoint, n):
    """tike_fft2_pass1 -> tike_fresnel_colpass -> tike_fft2_pass2_inplace ==
    FresnelSpectProp.fwd / .adj (fresnelspectprop.py:52-113) of the oracle,
    for few tiles (k1 split over workgroups) and for many; 128^2 and 512^2
    (round 6): two column stages per 16-row group / two groups per stage."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(4)
    H = oracle.fresnel_spectrum_propagator((n, n), (2e-6, 2e-6), 1e-6, 1e-10)
    st = A.stream_ptr()
    for ntile in (3, 40):
        x = rc(rng, ntile, n, n)
        want = (oracle.fresnel_adj if adjoint else oracle.fresnel_fwd)(x, H)
        xd, Hd = A.to_device(x), A.to_device(H)
        a, b = torch.empty_like(xd), torch.empty_like(xd)
        check(lib.tike_fft2_pass1(A.ptr(xd), A.ptr(a), ntile, n, 0, st))
        check(lib.tike_fresnel_colpass(A.ptr(a), A.ptr(Hd), int(adjoint),
                                       A.ptr(b), ntile, n, 1.0 / n**2, st))
        check(lib.tike_fft2_pass2_inplace(A.ptr(b), ntile, n, 1, 1.0, st))
        assert_close(b.cpu().numpy(), want, what=f"fresnel, {ntile} tiles")


@pytest.mark.parametrize("det", [256, 128, 512])
@pytest.mark.parametrize("S,N,inside", [(8, 5, True), (1, 33, True),
                                        (3, 6, False)])
def test_slice_step_vs_its_two_launches(oracle, S, N, inside, det):
    """tike_slice_step == tike_fft2_pass2_inplace (the probe incident on the
    slice) followed by tike_fwd_pass1 with those per-position probes
    (multislice.py:86-91, :79-85), and both against NumPy; positions whose
    window leaves the image take the general gather."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(S + N)
    HW = det + 44
    scan = (rng.random((N, 2)) * 40 + 1.5).astype(np.float32)
    if not inside:
        scan[1] = (-3.25, 10.5)
        scan[2] = (47.5, 60.75)
    psi = rc(rng, HW, HW)
    x = rc(rng, N, S, det, det)
    scale = 0.81 / det
    st = A.stream_ptr()
    d = {k: A.to_device(v) for k, v in dict(x=x, psi=psi, scan=scan).items()}
    work = torch.empty_like(d["x"])
    check(lib.tike_fft2_pass1(A.ptr(d["x"]), A.ptr(work), N * S, det, 1, st))
    # the two launches
    wave2 = work.clone()
    far2 = torch.empty_like(wave2)
    check(lib.tike_fft2_pass2_inplace(A.ptr(wave2), N * S, det, 1, scale, st))
    check(lib.tike_fwd_pass1(A.ptr(d["psi"]), A.ptr(d["scan"]), A.ptr(wave2), 1,
                             None, None, None, 0, 0, A.ptr(far2), None, N, S,
                             det, det, HW, HW, st))
    # the one
    wave1 = work.clone()
    far1 = torch.empty_like(wave1)
    check(lib.tike_slice_step(A.ptr(wave1), A.ptr(d["psi"]), A.ptr(d["scan"]),
                              A.ptr(far1), N, S, det, HW, HW, scale, st))
    want_wave = (np.fft.ifft2(x.astype(np.complex128), norm="forward")
                 * scale).astype(np.complex64)
    assert_close(wave1.cpu().numpy(), want_wave, what="incident probe")
    assert_close(wave1.cpu().numpy(), wave2.cpu().numpy(), normwise=1e-6,
                 what="incident probe vs pass 2 alone")
    assert_close(far1.cpu().numpy(), far2.cpu().numpy(), normwise=2e-6,
                 what="pass 1 vs tike_fwd_pass1")
    # ... and finished by the column pass, against NumPy
    check(lib.tike_fft2_pass2_inplace(A.ptr(far1), N * S, det, 0, 1.0, st))
    patches = oracle.patch_fwd(psi, scan, patch_width=det)
    want = np.fft.fft2(want_wave.astype(np.complex128) * patches[:, None])
    assert_close(far1.cpu().numpy(), want.astype(np.complex64),
                 what="FFT2(incident x patch)")


@pytest.mark.parametrize("det,S,N", [(256, 8, 5), (256, 1, 33), (128, 3, 7),
                                     (512, 2, 3)])
@pytest.mark.parametrize("inverse", [True, False])
@pytest.mark.parametrize("keep", [True, False])
def test_pass2_with_illumination_vs_oracle(det, S, N, inverse, keep):
    """tike_fft2_pass2_intensity == the second pass of the transform followed
    by sum_s |wave_s|^2 (_preconditioner.py:40-45,86-95); the wave is written
    back iff keep."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(det + S + N)
    x = rc(rng, N, S, det, det)
    scale = 0.37 / det
    # (the unnormalised transform either way, times `scale`)
    f = np.fft.ifft2 if inverse else np.fft.fft2
    want = f(x.astype(np.complex128),
             norm="forward" if inverse else "backward") * scale
    st = A.stream_ptr()
    xd = A.to_device(x)
    work = torch.empty_like(xd)
    check(lib.tike_fft2_pass1(A.ptr(xd), A.ptr(work), N * S, det, int(inverse),
                              st))
    before = work.clone()
    amp = torch.full((N, det, det), -1.0, dtype=torch.float32, device="cuda")
    check(lib.tike_fft2_pass2_intensity(A.ptr(work), A.ptr(amp), N, S, det,
                                        int(inverse), scale, int(keep), st))
    np.testing.assert_allclose(amp.cpu().numpy(),
                               (np.abs(want)**2).sum(axis=1),
                               rtol=2e-4, atol=1e-5 * (np.abs(want)**2).max())
    if keep:
        assert_close(work.cpu().numpy(), want.astype(np.complex64),
                     what="wave in place")
    else:
        assert torch.equal(work, before)  # never written


@pytest.mark.parametrize("det,S,N,shared,keep", [
    (256, 8, 9, False, True), (256, 8, 9, True, False), (256, 3, 7, False, False),
    (256, 1, 11, False, True), (256, 2, 6, True, True), (128, 5, 10, False, True),
    (128, 1, 20, True, False)])
def test_ifft2_pass2_products_vs_oracle(oracle, det, S, N, shared, keep):
    """tike_ifft2_pass2_products (rpie.py:444-472 for one slice): object
    projection, probe numerator (accumulated over the positions, scaled) and
    chi kept in place / mode 0 of chi, against NumPy."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(det + S + N)
    pw, HW = det, det + 60
    scan = (rng.random((N, 2)) * 50 + 1.5).astype(np.float32)
    psi = rc(rng, HW, HW)
    probe = rc(rng, 1 if shared else N, S, pw, pw)
    far = rc(rng, N, S, det, det)
    chi = np.fft.ifft2(far, norm="ortho").astype(np.complex64)
    patches = oracle.patch_fwd(psi, scan, patch_width=pw)  # (N, pw, pw)
    want_obj = np.sum(np.conj(probe) * chi, axis=1)
    want_num = 0.5 * np.sum(np.conj(patches[:, None]) * chi, axis=0)
    st = A.stream_ptr()
    d = {k: A.to_device(v) for k, v in dict(far=far, psi=psi, scan=scan,
                                            probe=probe).items()}
    work = torch.empty_like(d["far"])
    check(lib.tike_fft2_pass1(A.ptr(d["far"]), A.ptr(work), N * S, det, 1, st))
    before = work.clone()
    objproj = torch.empty((N, pw, pw), dtype=torch.complex64, device="cuda")
    num = torch.zeros((S, pw, pw), dtype=torch.complex64, device="cuda")
    chi0 = torch.zeros((N, pw, pw), dtype=torch.complex64, device="cuda")
    check(lib.tike_ifft2_pass2_products(
        A.ptr(work), A.ptr(d["psi"]), A.ptr(d["scan"]), A.ptr(d["probe"]),
        int(not shared), A.ptr(objproj), A.ptr(num), 0.5, A.ptr(chi0),
        int(keep), N, S, det, HW, HW, 1.0 / det, st))
    assert_close(objproj.cpu().numpy(), want_obj, what="objproj")
    assert_close(num.cpu().numpy(), want_num, normwise=2e-5, what="numerator")
    if keep:
        assert_close(work.cpu().numpy(), chi, what="chi in place")
    else:
        assert torch.equal(work, before)
        assert_close(chi0.cpu().numpy(), chi[:, 0], what="chi0")


def test_ptycho_fwd_eigen_probe_on_the_fly(ops):
    """Varying probe synthesised inside the kernel == get_varying_probe."""
    from oracle import solvers as sol
    from oracle import operators as oo
    import torch
    rng = np.random.default_rng(3)
    N, S, C, Sm, pw, det, HW = 6, 3, 2, 1, 32, 64, 80
    scan = (rng.random((N, 2)) * 40 + 2).astype(np.float32)
    probe, psi = rc(rng, 1, 1, S, pw, pw), rc(rng, 1, HW, HW)
    eigen = rc(rng, 1, C, Sm, pw, pw)
    w = rng.standard_normal((N, C + 1, S)).astype(np.float32)
    uprobe = sol.get_varying_probe(probe, eigen, w)
    want = oo.ptycho_fwd(uprobe, scan, psi, det)
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        got = op.fwd_device(op.asarray(probe), op.asarray(scan),
                            op.asarray(psi), op.asarray(eigen), op.asarray(w))
    assert_close(got.cpu().numpy(), want, what="fwd with eigen probes")


@pytest.mark.parametrize("S,det,masked", [(1, 24, False), (3, 24, True),
                                          (8, 32, True), (9, 16, True)])
def test_poisson_step_lengths_vs_oracle(S, det, masked):
    """tike_poisson_steps, all modes (exitwave.py:122-184): S <= 8 runs the
    one-workgroup-per-position kernel (two sweeps), S = 9 the per-tile one;
    masked-out pixels hold NaN counts and must not be touched."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    from oracle import solvers as osol
    rng = np.random.default_rng(7 * S + det)
    N = 5
    far = rc(rng, N, 1, S, det, det)
    inten = np.sum(np.abs(far)**2, axis=2)[:, 0].astype(np.float32)
    data = (inten * rng.uniform(0.5, 1.5, inten.shape)).astype(np.float32)
    mask = np.ones((det, det), bool)
    if masked:
        mask[rng.random((det, det)) < 0.2] = False
        data[:, ~mask] = np.nan
    xi = (1 - data / (inten + 1e-9))[:, None, None]
    want = osol.poisson_steplength_all_modes(
        xi, np.abs(far)**2, inten, data, mask,
        np.full((N, 1, S, 1, 1), 0.7, np.float32), 0.4)
    steps = torch.zeros((N, S), dtype=torch.float32, device="cuda")
    m = A.to_device(mask.astype(np.uint8)) if masked else None
    far_d, inten_d, data_d = (A.to_device(x) for x in (far, inten, data))
    check(lib.tike_poisson_steps(
        A.ptr(far_d), A.ptr(inten_d), A.ptr(data_d), A.ptr(m), A.ptr(steps), N,
        S, det, 0.7, 0.4, 0, A.stream_ptr()), "poisson steps")
    np.testing.assert_allclose(steps.cpu().numpy(), want[:, 0, :, 0, 0],
                               rtol=2e-4)


def test_farplane_gradient_fused(ops, oracle):
    """tike_farplane_gradient vs objective.py + lstsq.py:444-502, with a mask
    whose unmeasured pixels hold NaN data (reference tests put NaN there)."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import lib, check
    rng = np.random.default_rng(5)
    N, S, det = 7, 3, 48
    far = rc(rng, N, 1, S, det, det)
    inten = oracle.intensity_from_farplane(far)
    data = (rng.random((N, det, det), dtype=np.float32) * inten.max())
    mask = rng.random((det, det)) > 0.2
    data_nan = data.copy()
    data_nan[:, ~mask] = np.nan
    for model in ("gaussian", "poisson"):
        want = far.copy()
        grad = getattr(oracle, f"{model}_grad")(data, far, inten)
        want[..., mask] = -grad[..., mask]
        want[..., ~mask] *= np.float32(0.7 - 1.0)
        costs = getattr(oracle, f"{model}_each_pattern")(
            data[:, mask][:, None, :], inten[:, mask][:, None, :])
        f = A.to_device(far, np.complex64)
        I = torch.empty((N, det, det), dtype=torch.float32, device=f.device)
        c = torch.empty(N, dtype=torch.float32, device=f.device)
        d_dev = A.to_device(data_nan, np.float32)  # keep alive over the call
        m_dev = A.to_device(mask.astype(np.uint8))
        check(lib.tike_farplane_gradient(
            A.ptr(f), A.ptr(d_dev), A.ptr(m_dev), A.ptr(I), A.ptr(c), N,
            S, det, 0 if model == "gaussian" else 1, 1, 0.7, int(mask.sum()),
            A.stream_ptr()))
        assert_close(f.cpu().numpy(), want, normwise=1e-4, maxabs=1e-3,
                     what=f"{model} gradient")
        assert_close(I.cpu().numpy(), inten, what="intensity")
        np.testing.assert_allclose(c.cpu().numpy(), costs, rtol=COST_RTOL)


def test_full_size_roundtrip_property(ops):
    """BASELINE size (256^2): F* F = identity on the probe window and
    Parseval, properties the oracle need not be run for."""
    import torch
    rng = np.random.default_rng(9)
    x = rc(rng, 64, 256, 256)
    with ops.Propagation(detector_shape=256) as op:
        t = op.asarray(x)
        f = op.fwd(nearplane=t)
        np.testing.assert_allclose(
            float((f.abs()**2).sum()), float((t.abs()**2).sum()), rtol=1e-5)
        back = op.adj(farplane=f)
    assert relerr(back.cpu().numpy(), x) < 2e-6


@pytest.mark.parametrize("det,pw,S", [(128, 128, 3), (256, 256, 2),
                                      (128, 96, 2), (512, 512, 2),
                                      (512, 384, 1), (256, 192, 3),
                                      (256, 256, 8), (256, 256, 1),
                                      (256, 256, 5), (128, 128, 8),
                                      (128, 128, 1), (512, 512, 4),
                                      (256, 256, 3), (256, 256, 4),
                                      (256, 256, 6), (256, 256, 7)])
def test_position_major_forward_gradient_inverse(ops, oracle, det, pw, S):
    """tike_ptycho_fwd_intensity -> tike_gradient_scale ->
    tike_ifft2_crop_scaled == oracle fwd, intensity, per-pattern cost and
    cropped IFFT2 of the far-plane gradient (lstsq.py:441-507), with eigen
    probes and a mask whose unmeasured pixels hold NaN."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import lib, check
    from oracle import solvers as sol
    rng = np.random.default_rng(det + S)
    N, C, HW = 5, 1, pw + 40
    scan = (rng.random((N, 2)) * 36 + 1.5).astype(np.float32)
    probe, psi = rc(rng, 1, 1, S, pw, pw), rc(rng, 1, HW, HW)
    eigen = rc(rng, 1, C, 1, pw, pw)
    w = (1 + 0.1 * rng.standard_normal((N, C + 1, S))).astype(np.float32)
    uprobe = sol.get_varying_probe(probe, eigen, w)
    want_far = oracle.ptycho_fwd(uprobe, scan, psi, det)
    want_I = oracle.intensity_from_farplane(want_far)
    data = (rng.random((N, det, det), dtype=np.float32) * want_I.max())
    mask = rng.random((det, det)) > 0.1
    data_nan = data.copy()
    data_nan[:, ~mask] = np.nan
    grad = want_far.copy()
    grad[..., mask] = -oracle.gaussian_grad(data, want_far, want_I)[..., mask]
    grad[..., ~mask] *= np.float32(0.5 - 1.0)
    pad = (det - pw) // 2
    want_chi = oracle.propagation_adj(grad)[..., pad:pad + pw, pad:pad + pw]
    want_cost = oracle.gaussian_each_pattern(
        data[:, mask][:, None, :], want_I[:, mask][:, None, :])
    dev = A.current_device()
    t = lambda x, dt=None: A.to_device(x, dt)
    psi_d, scan_d, probe_d, eig_d, w_d = (t(psi), t(scan), t(probe), t(eigen),
                                          t(w))
    far = torch.empty((N, 1, S, det, det), dtype=torch.complex64, device=dev)
    I = torch.empty((N, det, det), dtype=torch.float32, device=dev)
    st = A.stream_ptr()
    uq = torch.empty((N, 1, pw, pw), dtype=torch.complex64, device=dev)
    pat = torch.empty((N, pw, pw), dtype=torch.complex64, device=dev)
    check(lib.tike_varying_probe(A.ptr(probe_d), A.ptr(eig_d), A.ptr(w_d), C, 1,
                                 A.ptr(uq), N, S, pw, st))
    assert_close(uq.cpu().numpy(), uprobe[:, 0, :1], what="varying probe")
    check(lib.tike_ptycho_fwd_intensity(
        A.ptr(psi_d), A.ptr(scan_d), A.ptr(probe_d), 0, A.ptr(uq),
        A.ptr(w_d), C, 1, A.ptr(far), A.ptr(I), A.ptr(pat), N, S, pw, det, HW,
        HW, 1.0 / det, st))
    assert_close(pat.cpu().numpy(), oracle.patch_fwd(psi[0], scan, patch_width=pw),
                 what="patches stored by the forward kernel")
    assert_close(far.cpu().numpy(), want_far, what="farplane")
    assert_close(I.cpu().numpy(), want_I, what="intensity")
    d_d, m_d = t(data_nan, np.float32), t(mask.astype(np.uint8))
    g = torch.empty_like(I)
    costs = torch.empty(N, dtype=torch.float32, device=dev)
    check(lib.tike_gradient_scale(A.ptr(I), A.ptr(d_d), A.ptr(m_d), A.ptr(g),
                                  A.ptr(costs), N, det, 0, 0.5,
                                  int(mask.sum()), st))
    np.testing.assert_allclose(costs.cpu().numpy(), want_cost, rtol=COST_RTOL)
    mid = torch.empty_like(far)
    chi = mid if pw == det else torch.empty((N, 1, S, pw, pw),
                                            dtype=torch.complex64, device=dev)
    check(lib.tike_ifft2_crop_scaled(A.ptr(far), A.ptr(g), S, A.ptr(mid),
                                     A.ptr(chi), N * S, det, pw, 1.0 / det,
                                     st))
    assert_close(chi.cpu().numpy(), want_chi, normwise=1e-4, maxabs=1e-3,
                 what="chi")

    def fused_gradients(work, what):
        """tike_ifft2_pass2_gradients on a pass-1 intermediate == the oracle's
        objproj / m_probe_update / chi mode 0 (lstsq.py:504-539)."""
        proj = torch.full((N, pw, pw), 7.0, dtype=torch.complex64, device=dev)
        chi0 = torch.full_like(proj, 7.0)
        mpu = torch.zeros((S, pw, pw), dtype=torch.complex64, device=dev)
        check(lib.tike_ifft2_pass2_gradients(
            A.ptr(work), A.ptr(pat), A.ptr(probe_d), A.ptr(eig_d), A.ptr(w_d),
            C, 1, A.ptr(proj), A.ptr(chi0), A.ptr(mpu), 1.0, N, S, det,
            1.0 / det, st))
        want_proj = (np.conj(uprobe[:, 0]) * want_chi[:, 0]).sum(axis=1)
        patches = oracle.patch_fwd(psi[0], scan, patch_width=pw)
        want_mpu = (np.conj(patches)[:, None] * want_chi[:, 0]).sum(axis=0)
        assert_close(proj.cpu().numpy(), want_proj, normwise=1e-4, maxabs=1e-3,
                     what=f"objproj ({what})")
        assert_close(chi0.cpu().numpy(), want_chi[:, 0, 0], normwise=1e-4,
                     maxabs=1e-3, what=f"chi0 ({what})")
        assert_close(mpu.cpu().numpy(), want_mpu, normwise=1e-4, maxabs=1e-3,
                     what=f"m_probe_update ({what})")
        # probe gradient only (no object projection): the other instantiation
        mpu2 = torch.zeros_like(mpu)
        check(lib.tike_ifft2_pass2_gradients(
            A.ptr(work), A.ptr(pat), None, None, None, 0, 0, None, None,
            A.ptr(mpu2), 0.5, N, S, det, 1.0 / det, st))
        assert_close(2 * mpu2.cpu().numpy(), want_mpu, normwise=1e-4, maxabs=1e-3,
                     what=f"m_probe_update alone ({what})")

    if pw == det and S <= (4 if det == 512 else 8):
        work = torch.empty_like(far)
        check(lib.tike_ifft2_pass1_scaled(A.ptr(far), A.ptr(g), None, None, S,
                                          A.ptr(work), N * S, det, st))
        fused_gradients(work, "stored far plane")
    if det == 512 and pw == det and S <= 4:
        # far-plane free at 512^2: split forward, then gradient + inverse pass 1
        # straight from the hand-off (radix-32 column pass re-formed in
        # registers), finished by the fused pass 2
        scratch = torch.empty_like(far)
        g5, costs5 = torch.empty_like(g), torch.empty_like(costs)
        check(lib.tike_fwd_pass1(
            A.ptr(psi_d), A.ptr(scan_d), A.ptr(probe_d), 0, None, A.ptr(eig_d),
            A.ptr(w_d), C, 1, A.ptr(scratch), None, N, S, pw, det, HW, HW, st))
        check(lib.tike_fwd_gradient_scale(
            A.ptr(scratch), A.ptr(d_d), 0, A.ptr(m_d), A.ptr(g5), None,
            A.ptr(costs5), None, N, S, det, 1.0 / det, 0, 0.5, int(mask.sum()),
            st))
        np.testing.assert_allclose(costs5.cpu().numpy(), want_cost,
                                   rtol=COST_RTOL)
        work = torch.empty_like(far)
        check(lib.tike_grad_ifft2_pass1(A.ptr(scratch), A.ptr(g), None, None, S,
                                        A.ptr(work), N * S, det, 1.0 / det, st))
        fused_gradients(work, "no far plane, 512")
        # ... and the same intermediate as the stored-far-plane pass 1, also
        # with per-mode steps on measured pixels (poisson form)
        steps = torch.rand((N, S), dtype=torch.float32, device=dev) + 0.5
        for sp, mp in ((None, None), (steps, m_d)):
            w_ref = torch.empty_like(far)
            check(lib.tike_ifft2_pass1_scaled(A.ptr(far), A.ptr(g), A.ptr(sp),
                                              A.ptr(mp), S, A.ptr(w_ref), N * S,
                                              det, st))
            check(lib.tike_grad_ifft2_pass1(A.ptr(scratch), A.ptr(g), A.ptr(sp),
                                            A.ptr(mp), S, A.ptr(work), N * S,
                                            det, 1.0 / det, st))
            assert_close(work.cpu().numpy(), w_ref.cpu().numpy(), normwise=1e-5,
                         maxabs=1e-4, what="inverse intermediate at 512")
    if det == 256:
        # the far-plane-free pipeline: forward for the intensity only, then the
        # gradient and the inverse transform from the column-pass scratch
        scratch = torch.empty_like(far)
        I2 = torch.empty_like(I)
        check(lib.tike_ptycho_fwd_intensity_only(
            A.ptr(psi_d), A.ptr(scan_d), A.ptr(probe_d), 0, A.ptr(uq),
            A.ptr(w_d), C, 1, A.ptr(scratch), A.ptr(I2), N, S, pw, det, HW, HW,
            1.0 / det, st))
        assert_close(I2.cpu().numpy(), want_I, what="intensity (only)")
        # ... and with the gradient factor + costs formed in the same launch
        g2 = torch.empty_like(g)
        costs2 = torch.empty_like(costs)
        scratch2 = torch.empty_like(far)
        pat2 = torch.empty_like(pat)
        check(lib.tike_ptycho_fwd_gradient_scale(
            A.ptr(psi_d), A.ptr(scan_d), A.ptr(probe_d), 0, A.ptr(uq),
            A.ptr(w_d), C, 1, A.ptr(scratch2), None, A.ptr(pat2), A.ptr(d_d),
            A.ptr(m_d), A.ptr(g2), A.ptr(costs2), N, S, pw, det, HW, HW,
            1.0 / det, 0, 0.5, int(mask.sum()), st))
        assert torch.equal(pat2, pat)
        np.testing.assert_allclose(costs2.cpu().numpy(), want_cost,
                                   rtol=COST_RTOL)
        np.testing.assert_allclose(g2.cpu().numpy(), g.cpu().numpy(),
                                   rtol=1e-4, atol=1e-5)
        assert_close(scratch2.cpu().numpy(), scratch.cpu().numpy(),
                     normwise=1e-6, maxabs=1e-5, what="column-pass scratch")
        # ... and as two launches (pass 1, then the streamed column pass)
        scratch3 = torch.empty_like(far)
        pat3, g3, costs3, I3 = (torch.empty_like(pat), torch.empty_like(g),
                                torch.empty_like(costs), torch.empty_like(I))
        check(lib.tike_fwd_pass1(
            A.ptr(psi_d), A.ptr(scan_d), A.ptr(probe_d), 0, A.ptr(uq), None,
            A.ptr(w_d), C, 1, A.ptr(scratch3), A.ptr(pat3), N, S, pw, det, HW,
            HW, st))
        # the varying probe formed on the fly from the eigen probes instead
        scratch4 = torch.empty_like(far)
        check(lib.tike_fwd_pass1(
            A.ptr(psi_d), A.ptr(scan_d), A.ptr(probe_d), 0, None, A.ptr(eig_d),
            A.ptr(w_d), C, 1, A.ptr(scratch4), None, N, S, pw, det, HW, HW,
            st))
        assert_close(scratch4.cpu().numpy(), scratch.cpu().numpy(),
                     normwise=1e-6, maxabs=1e-5, what="scratch (eigen on the fly)")
        check(lib.tike_fwd_gradient_scale(
            A.ptr(scratch3), A.ptr(d_d), 0, A.ptr(m_d), A.ptr(g3), A.ptr(I3),
            A.ptr(costs3), None, N, S, det, 1.0 / det, 0, 0.5, int(mask.sum()),
            st))
        assert_close(pat3.cpu().numpy(), pat.cpu().numpy(), normwise=1e-6,
                     maxabs=1e-6, what="patches (split forward)")
        assert_close(scratch3.cpu().numpy(), scratch.cpu().numpy(),
                     normwise=1e-6, maxabs=1e-5, what="scratch (split forward)")
        assert_close(I3.cpu().numpy(), want_I, what="intensity (split)")
        np.testing.assert_allclose(costs3.cpu().numpy(), want_cost,
                                   rtol=COST_RTOL)
        # (the factor is ill-conditioned where the intensity vanishes)
        np.testing.assert_allclose(g3.cpu().numpy(), g.cpu().numpy(),
                                   rtol=2e-3, atol=1e-5)
        mid2 = torch.empty_like(far)
        chi2 = mid2 if pw == det else torch.empty_like(chi)
        check(lib.tike_grad_ifft2_crop(
            A.ptr(scratch), A.ptr(g), None, None, S, A.ptr(mid2), A.ptr(chi2),
            N * S, det, pw, 1.0 / det, 1.0 / det, st))
        assert_close(chi2.cpu().numpy(), want_chi, normwise=1e-4, maxabs=1e-3,
                     what="chi (no far plane)")
        if pw == det:
            work = torch.empty_like(far)
            check(lib.tike_grad_ifft2_pass1(A.ptr(scratch), A.ptr(g), None,
                                            None, S, A.ptr(work), N * S, det,
                                            1.0 / det, st))
            fused_gradients(work, "no far plane")
            # column pass + factor + inverse pass 1 in ONE launch (two sweeps
            # for S <= 5, column-pass values resident in registers above)
            work1 = torch.full_like(far, 3.0)
            costs1 = torch.empty_like(costs)
            check(lib.tike_fwd_grad_ifft2_pass1(
                A.ptr(scratch), A.ptr(d_d), 0, A.ptr(m_d), A.ptr(costs1),
                A.ptr(work1), N, S, det, 1.0 / det, 0, 0.5, int(mask.sum()),
                st))
            np.testing.assert_allclose(costs1.cpu().numpy(), want_cost,
                                       rtol=COST_RTOL)
            fused_gradients(work1, "one launch")
        # per-mode factor on measured pixels (poisson form)
        steps = torch.rand((N, S), dtype=torch.float32, device=dev) + 0.5
        check(lib.tike_grad_ifft2_crop(
            A.ptr(scratch), A.ptr(g), A.ptr(steps), A.ptr(m_d), S, A.ptr(mid2),
            A.ptr(chi2), N * S, det, pw, 1.0 / det, 1.0 / det, st))
        chi3 = mid if pw == det else torch.empty_like(chi)
        check(lib.tike_ifft2_crop_scaled_modes(
            A.ptr(far), A.ptr(g), A.ptr(steps), A.ptr(m_d), S, A.ptr(mid),
            A.ptr(chi3), N * S, det, pw, 1.0 / det, st))
        assert_close(chi2.cpu().numpy(), chi3.cpu().numpy(), normwise=1e-5,
                     maxabs=1e-4, what="chi with mode steps")


@pytest.mark.parametrize("S,det", [(1, 256), (4, 256), (6, 256), (7, 256),
                                   (8, 256), (1, 512), (3, 512), (4, 512)])
@pytest.mark.parametrize("model,u16,masked", [(0, False, False), (0, True, True),
                                               (1, False, True), (1, True, False)])
def test_one_launch_gradient_pass_matches_the_two_launches(S, det, model, u16,
                                                           masked):
    """tike_fwd_grad_ifft2_pass1 == tike_fwd_gradient_scale +
    tike_grad_ifft2_pass1 on the same hand-off, with enough positions that
    every workgroup walks several work items (the register-resident kernel
    requests the next item's rows while it finishes the current one): both
    noise models, float / 16-bit counts, with and without a mask whose
    unmeasured pixels hold garbage."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import lib, check
    N = 300 if det == 256 else 70  # (512^2: the two-sweep kernel of round 5)
    dev = A.current_device()
    g = torch.Generator(device=dev).manual_seed(S + 10 * model)
    scratch = torch.view_as_complex(
        torch.rand(N, 1, S, det, det, 2, device=dev, generator=g) - 0.5)
    counts = torch.rand(N, det, det, device=dev, generator=g) * 40 * S
    mask = None
    nmeas = det * det
    if masked:
        mask = (torch.rand(det, det, device=dev, generator=g) > 0.1).to(torch.uint8)
        nmeas = int(mask.sum())
    if u16:
        data = counts.to(torch.int32).to(torch.uint16)
        if masked:  # garbage where nothing was measured
            data = torch.where(mask.bool(), data.to(torch.int32),
                               torch.full_like(data, 65535, dtype=torch.int32)
                               ).to(torch.uint16)
    else:
        data = counts
        if masked:
            data = torch.where(mask.bool(), data,
                               torch.full_like(data, float("nan")))
    st = A.stream_ptr()
    gs = torch.empty(N, det, det, device=dev)
    c_ref, c_one = torch.empty(N, device=dev), torch.empty(N, device=dev)
    w_ref, w_one = torch.empty_like(scratch), torch.full_like(scratch, 5.0)
    mp = A.ptr(mask) if masked else None
    check(lib.tike_fwd_gradient_scale(
        A.ptr(scratch), A.ptr(data), int(u16), mp, A.ptr(gs), None,
        A.ptr(c_ref), None, N, S, det, 1.0 / det, model, 0.5, nmeas, st))
    check(lib.tike_grad_ifft2_pass1(A.ptr(scratch), A.ptr(gs), None, None, S,
                                    A.ptr(w_ref), N * S, det, 1.0 / det, st))
    check(lib.tike_fwd_grad_ifft2_pass1(
        A.ptr(scratch), A.ptr(data), int(u16), mp, A.ptr(c_one), A.ptr(w_one),
        N, S, det, 1.0 / det, model, 0.5, nmeas, st))
    torch.cuda.synchronize()
    assert torch.isfinite(torch.view_as_real(w_one)).all()
    np.testing.assert_allclose(c_one.cpu().numpy(), c_ref.cpu().numpy(),
                               rtol=1e-5)
    assert_close(w_one.cpu().numpy(), w_ref.cpu().numpy(), normwise=2e-6,
                 maxabs=2e-5, what="inverse column-pass input")


@pytest.mark.parametrize("layout", ["neighbours", "far_apart", "tall_group",
                                    "border", "ragged"])
def test_grouped_footprint_scatter(oracle, layout):
    """tike_scatter_patches / tike_psi_preconditioner == Patch.adj (oracle) for
    position lists that take the on-chip grouped path (neighbours), the
    per-position fallback (far apart / tall boxes), image-border positions and
    a count that is not a multiple of the group size."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import lib, check
    rng = np.random.default_rng(7)
    pw, H, W = 48, 400, 380
    if layout == "neighbours":
        scan = 60 + rng.random((24, 2)) * 40
    elif layout == "far_apart":
        scan = 1 + rng.random((24, 2)) * np.array([H - pw - 3, W - pw - 3])
    elif layout == "tall_group":  # narrow in x, spread in y beyond the box
        scan = np.stack([1 + rng.random(16) * (H - pw - 3),
                         100 + rng.random(16) * 10], -1)
    elif layout == "border":
        scan = np.array([[1.0, 1.0], [1.5, 2.25], [H - pw - 1.5, W - pw - 1.25],
                         [H - pw - 2.0, W - pw - 2.0], [1.25, W - pw - 1.5]])
    else:
        scan = 30 + rng.random((13, 2)) * 60
    scan = scan.astype(np.float32)
    N = len(scan)
    proj = rc(rng, N, pw, pw)
    want = oracle.patch_adj(patches=proj, images=np.zeros((H, W), np.complex64),
                            positions=scan, nrepeat=1)
    acc = torch.zeros((2, H, W), dtype=torch.float32, device="cuda")
    proj_d, scan_d = A.to_device(proj), A.to_device(scan)  # keep alive
    check(lib.tike_scatter_patches(A.ptr(proj_d), A.ptr(scan_d), A.ptr(acc), N,
                                   pw, H, W, A.stream_ptr()))
    got = (acc[0] + 1j * acc[1]).cpu().numpy()
    assert_close(got, want, normwise=2e-6, maxabs=1e-5, what="object scatter")
    amp = rng.random((pw, pw)).astype(np.float32)
    wantp = oracle.patch_adj(
        patches=np.broadcast_to(amp.astype(np.complex64), (N, pw, pw)).copy(),
        images=np.zeros((H, W), np.complex64), positions=scan, nrepeat=1)
    out = torch.zeros((H, W), dtype=torch.float32, device="cuda")
    amp_d = A.to_device(amp)
    check(lib.tike_psi_preconditioner(A.ptr(amp_d), A.ptr(scan_d), A.ptr(out),
                                      N, pw, H, W, A.stream_ptr()))
    assert_close(out.cpu().numpy(), wantp.real, normwise=2e-6, maxabs=1e-5,
                 what="psi preconditioner")


def test_abi_edge_cases():
    """C-ABI contract on degenerate input: zero positions are a no-op (rc 0,
    no pointer dereferenced), unsupported detector sizes of the specialised
    entries report TIKE_ERR_UNSUPPORTED, bad arguments TIKE_ERR_ARG."""
    import torch
    import tike_amd._arrays as A
    import tike_amd._lib as L
    lib = L.lib
    st = A.stream_ptr()
    z = None
    assert lib.tike_ptycho_fwd_intensity(z, z, z, 0, z, z, 0, 0, z, z, z, 0, 2,
                                         64, 64, 100, 100, 1.0, st) == 0
    assert lib.tike_ptycho_fwd_gradient_scale(
        z, z, z, 0, z, z, 0, 0, z, z, z, z, z, z, z, 0, 2, 256, 256, 400, 400,
        1.0, 0, 1.0, 65536, st) == 0
    assert lib.tike_grad_ifft2_pass1(z, z, z, z, 2, z, 0, 256, 1.0, st) == 0
    assert lib.tike_fwd_pass1(z, z, z, 0, z, z, z, 0, 0, z, z, 0, 2, 256, 256,
                              400, 400, st) == 0
    assert lib.tike_fwd_gradient_scale(z, z, 0, z, z, z, z, z, 0, 2, 256, 1.0, 0,
                                       1.0, 65536, st) == 0
    assert lib.tike_ifft2_pass1_scaled(z, z, z, z, 2, z, 0, 128, st) == 0
    assert lib.tike_ifft2_pass2_gradients(z, z, z, z, z, 0, 0, z, z, z, 1.0, 0,
                                          2, 256, 1.0, st) == 0
    assert lib.tike_grad_ifft2_crop(z, z, z, z, 2, z, z, 0, 256, 256, 1.0, 1.0,
                                    st) == 0
    assert lib.tike_scatter_patches(z, z, z, 0, 64, 100, 100, st) == 0
    assert lib.tike_psi_preconditioner(z, z, z, 0, 64, 100, 100, st) == 0
    assert lib.tike_position_sums(z, z, 1, z, z, z, 0, 0, z, 2, z, z, 0, 1, 64,
                                  st) == 0
    assert lib.tike_poisson_steps(z, z, z, z, z, 0, 2, 64, 0.5, 0.5, 0,
                                  st) == 0
    assert lib.tike_lstsq_step_stats(z, z, z, z, z, z, z, 0, 0, z, z, z, z, 0,
                                     1, 1, 64, 100, 100, z, z, st) == 0
    assert lib.tike_eigen_pixel_update1(z, z, z, z, z, z, 2, z, z, 0, 64, 1,
                                        z, z, 0.0, z, z, z, 100, 100, st) == 0
    assert lib.tike_eigen_position_sums1(z, z, z, z, z, z, 0, 64, 1, z, z, 100,
                                         100, st) == 0
    # specialised entries refuse sizes they do not implement
    x = torch.zeros(4 * 128 * 128, dtype=torch.complex64, device="cuda")
    f = torch.zeros(128 * 128, dtype=torch.float32, device="cuda")
    s = torch.full((1, 2), 5.0, dtype=torch.float32, device="cuda")
    p = lambda t: t.data_ptr()
    assert lib.tike_ptycho_fwd_gradient_scale(
        p(x), p(s), p(x), 0, z, z, 0, 0, p(x), z, z, p(f), z, p(f), z, 1, 1,
        128, 128, 200, 200, 1.0, 0, 1.0, 128 * 128, st) == L.ERR_UNSUPPORTED
    assert lib.tike_grad_ifft2_pass1(p(x), p(f), z, z, 1, p(x) + 8, 1, 128,
                                     1.0, st) == L.ERR_UNSUPPORTED
    assert lib.tike_ifft2_pass2_gradients(p(x), p(x), p(x), z, z, 0, 0, p(x),
                                          z, z, 1.0, 1, 9, 128, 1.0,
                                          st) == L.ERR_UNSUPPORTED  # S > 8
    assert lib.tike_ifft2_pass2_gradients(p(x), p(x), p(x), z, z, 0, 0, p(x),
                                          z, z, 1.0, 1, 1, 64, 1.0,
                                          st) == L.ERR_UNSUPPORTED
    assert lib.tike_grad_ifft2_crop(p(x), p(f), z, z, 1, p(x) + 8, p(x) + 16,
                                    1, 128, 128, 1.0, 1.0,
                                    st) == L.ERR_UNSUPPORTED
    # argument errors: probe wider than the detector, work aliasing the input
    assert lib.tike_ptycho_fwd_intensity(p(x), p(s), p(x), 0, z, z, 0, 0, p(x),
                                         z, z, 1, 1, 256, 128, 300, 300, 1.0,
                                         st) == L.ERR_ARG
    assert lib.tike_ifft2_crop_scaled(p(x), p(f), 1, p(x), p(x), 1, 128, 128,
                                      1.0, st) == L.ERR_ARG
    # the all-at-once line search: a detector size without its kernels, the two
    # far-plane buffers aliased, a missing state
    d = torch.zeros(8, dtype=torch.float64, device="cuda")
    args = lambda det, fa, fb, state: (
        0, p(x), p(x), p(x), p(x), p(s), p(f), 0, fa, 0, fb, p(f), 1, 1, 1, det,
        200, 200, 1.0, 1.0, state, 0, z, st)
    assert lib.tike_cgrad_line_search_linear(
        *args(64, p(x), p(x) + 8, p(d))) == L.ERR_UNSUPPORTED
    assert lib.tike_cgrad_line_search_linear(
        *args(128, p(x), p(x), p(d))) == L.ERR_ARG
    assert lib.tike_cgrad_line_search_linear(
        *args(128, p(x), p(x) + 8, z)) == L.ERR_ARG


def test_comm_abi_single_rank():
    """The RCCL entries of the C ABI on a one-rank communicator: sum
    all-reduce (f32 / f64) and broadcast leave the buffers unchanged, are
    asynchronous on the caller's stream, and argument errors are reported."""
    import ctypes
    import torch
    import tike_amd._arrays as A
    import tike_amd._lib as L
    lib = L.lib
    ident = ctypes.create_string_buffer(L.COMM_ID_BYTES)
    L.check(lib.tike_comm_unique_id(ident), "unique_id")
    comm = ctypes.c_void_p()
    assert lib.tike_comm_create(ident.raw, 1, 1, ctypes.byref(comm)) == L.ERR_ARG
    assert lib.tike_comm_create(None, 1, 0, ctypes.byref(comm)) == L.ERR_ARG
    L.check(lib.tike_comm_create(ident.raw, 1, 0, ctypes.byref(comm)), "create")
    assert comm.value
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(1 << 20, device="cuda", generator=g)
    c = torch.randn(3, 100, 100, dtype=torch.complex64, device="cuda")
    d = torch.randn(5, dtype=torch.float64, device="cuda")
    x0, c0, d0 = x.clone(), c.clone(), d.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        st = A.stream_ptr()
        L.check(lib.tike_comm_allreduce_sum(comm, x.data_ptr(), x.numel(), 0, st))
        L.check(lib.tike_comm_allreduce_sum(comm, c.data_ptr(), 2 * c.numel(), 0,
                                            st))
        L.check(lib.tike_comm_allreduce_sum(comm, d.data_ptr(), d.numel(), 1, st))
        L.check(lib.tike_comm_broadcast(comm, x.data_ptr(), 4 * x.numel(), 0, st))
        assert lib.tike_comm_allreduce_sum(comm, None, 0, 0, st) == 0
        assert lib.tike_comm_allreduce_sum(comm, None, 4, 0, st) == L.ERR_ARG
        assert lib.tike_comm_allreduce_sum(None, x.data_ptr(), 4, 0,
                                           st) == L.ERR_ARG
    side.synchronize()
    assert torch.equal(x, x0) and torch.equal(c, c0) and torch.equal(d, d0)
    L.check(lib.tike_comm_destroy(comm), "destroy")
    assert lib.tike_comm_destroy(None) == 0


def test_multislice_vs_reference_fixture(ops, golden):
    """FresnelSpectProp / Multislice (3 slices) / Ptycho over it on the GPU
    against the reference's own outputs."""
    g = golden("op_multislice.npz")
    wl, fy, fx, dist = (float(v) for v in g["phys"])
    pw, HW = g["probe"].shape[-1], g["psi"].shape[-1]
    phys = dict(probe_wavelength=wl, probe_FOV_lengths=(fy, fx),
                multislice_propagation_distance=dist)
    probe, scan, psi = g["probe"], g["scan"], g["psi"]
    with ops.Multislice(probe_shape=pw, detector_shape=pw, nz=HW, n=HW,
                        **phys) as op:
        assert_close(op.propagation.fwd(g["nearplane_in"]), g["fresnel_fwd"],
                     what="fresnel fwd")
        assert_close(op.propagation.adj(g["nearplane_in"]), g["fresnel_adj"],
                     what="fresnel adj")
        assert_close(op.fwd(probe=probe, scan=scan, psi=psi), g["ms_fwd"],
                     what="multislice fwd")
        exitw, probes = op.fwd_return_intermediate_probes(
            probe=probe[:, None], scan=scan, psi=psi)
        assert_close(exitw, g["ms_exit"], what="exit wave")
        assert_close(probes, g["ms_probes"], what="intermediate probes")
        pa, qa = op.adj(nearplane=g["nearplane_in"], probe=probe, scan=scan,
                        psi=psi)
        assert_close(pa, g["ms_psi_adj"], what="multislice psi_adj")
        assert_close(qa, g["ms_probe_adj"], what="multislice probe_adj")
    with ops.Ptycho(probe_shape=pw, detector_shape=pw, nz=HW, n=HW,
                    **phys) as op:
        assert_close(op.fwd(probe=probe[:, None], scan=scan, psi=psi),
                     g["pt_fwd"], what="ptycho fwd (3 slices)")
        pa, qa = op.adj(farplane=g["farplane_in"], probe=probe[:, None],
                        scan=scan, psi=psi)
        assert_close(pa, g["pt_psi_adj"], what="ptycho psi_adj (3 slices)")
        assert_close(qa, g["pt_probe_adj"], what="ptycho probe_adj (3 slices)")


@pytest.mark.parametrize("depth,pw,distance", [(7, 15, 1e-8), (3, 32, 2e-4),
                                               (2, 128, 1e-4)])
def test_multislice_adjoint_and_oracle(ops, oracle, depth, pw, distance):
    """The reference's TestMultiSlice.test_adjoint (tests/operators/
    test_multislice.py:64-83: depth 7, pw 15, 27 positions, 3 probes per
    position, rtol 1e-3) plus parity with the oracle at propagation distances
    where the Fresnel kernel is far from the identity."""
    rng = np.random.default_rng(depth * 100 + pw)
    nscan, S, HW = 27, 3, pw + 113
    scan = (rng.random((nscan, 2)) * (HW - pw - 2)).astype(np.float32) + 1
    probe = rc(rng, nscan, S, pw, pw)
    psi = rc(rng, depth, HW, HW)
    near = rc(rng, nscan, S, pw, pw)
    phys = dict(probe_wavelength=1e-10, probe_FOV_lengths=(1e-5, 1e-5),
                multislice_propagation_distance=distance)
    with ops.Multislice(nscan=nscan, probe_shape=pw, detector_shape=pw, nz=HW,
                        n=HW, **phys) as op:
        d = op.fwd(probe=probe, scan=scan, psi=psi)
        m0, m1 = op.adj(nearplane=near, probe=probe, scan=scan, psi=psi)
    assert d.shape == near.shape and m0.shape == psi.shape
    assert m1.shape == probe.shape
    H = oracle.fresnel_spectrum_propagator((pw, pw), (1e-5, 1e-5), distance,
                                           1e-10)
    assert np.abs(H - H[0, 0]).max() > (1e-3 if distance > 1e-6 else 0)
    assert_close(d, oracle.multislice_fwd(probe, scan, psi, H),
                 what="multislice fwd")
    o0, o1 = oracle.multislice_adj(near, probe, scan, psi, H)
    assert_close(m0, o0, what="psi_adj")
    assert_close(m1, o1, what="probe_adj")
    # adjoint identities as the reference states them (psi_adj carries the
    # reference's 1/nslices, so <psi, F*d> is compared per slice count)
    a = np.vdot(near, d)
    b = np.vdot(m0, psi)
    c = np.vdot(m1, probe)
    np.testing.assert_allclose([a.real, a.imag], [b.real, b.imag], rtol=1e-3)
    np.testing.assert_allclose([a.real, a.imag], [c.real, c.imag], rtol=1e-3)


def test_extract_patches_and_probe_constraints_on_device(golden):
    """tike.ptycho.learn.extract_patches through the Patch kernel (vs the
    oracle's bilinear patches), and the probe constraints on CUDA tensors
    against the reference-run fixture."""
    import torch
    import tike_amd.ptycho as tp
    from oracle import operators as oop
    rng = np.random.default_rng(9)
    psi = (rng.random((40, 44)) + 1j * rng.random((40, 44))).astype(np.complex64)
    scan = (1 + rng.random((7, 2)) * (20, 24)).astype(np.float32)
    got = tp.learn.extract_patches(psi, scan, 16)
    want = oop.patch_fwd(psi, scan, patch_width=16)
    assert got.shape == (7, 16, 16)
    assert_close(got, want, normwise=1e-6, maxabs=1e-5, what="extract_patches")
    with pytest.raises(ValueError):
        tp.learn.extract_patches(psi, scan + 30, 16)
    g = golden("probe_constraints.npz")
    dev = lambda a: torch.from_numpy(a.copy()).cuda()
    for case in "abcd":
        out = tp.constrain_center_peak(dev(g[f"center_in_{case}"]))
        assert out.is_cuda
        np.testing.assert_array_equal(out.cpu().numpy(),
                                      g[f"center_out_{case}"])
    out = tp.apply_median_filter_abs_probe(dev(g["median_in"]), (2, 4))
    np.testing.assert_allclose(out.cpu().numpy(), g["median_out_2_4"],
                               rtol=2e-6, atol=1e-7)
    out = tp.constrain_probe_sparsity(dev(g["sparse_in"]), 0.6)
    np.testing.assert_array_equal(out.cpu().numpy(), g["sparse_out_60"])


def _patches(oracle, img, scan, pw):
    return oracle.patch_fwd(img, scan, patch_width=pw)


@pytest.mark.parametrize("pw,N,eigen,border", [(256, 7, True, False),
                                                (256, 6, False, True),
                                                (128, 9, True, True),
                                                (64, 5, False, False),
                                                (48, 4, True, False),
                                                (512, 3, False, False)])
def test_step_statistics_on_pairs_vs_numpy(oracle, pw, N, eigen, border):
    """tike_lstsq_step_stats (lstsq.py:641-694, 721-738) straight against
    NumPy: odd and even numbers of positions (the pair kernel's last, single
    position), windows that tile 256 threads and one that does not (48: no row
    walk), positions on the border of the image (the pair falls back)."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(pw + N)
    HW = pw + 40
    scan = (rng.random((N, 2)) * 30 + 2.25).astype(np.float32)
    if border:
        scan[1] = (HW - pw + 2.5, 3.5)  # the window leaves the image
        scan[N - 2] = (-1.75, HW - pw + 1.25)
    psi, gobj = rc(rng, HW, HW), rc(rng, HW, HW)
    probe = rc(rng, 1, 1, 2, pw, pw)
    mpu = rc(rng, 1, 1, 2, pw, pw)
    chi0 = rc(rng, N, pw, pw)
    E = rc(rng, 1, 1, 1, pw, pw) if eigen else None
    w = np.ones((N, 2, 2), np.float32)
    w[:, 1] = 0
    if eigen:
        w = (rng.standard_normal((N, 2, 2)) * 0.3 + 1).astype(np.float32)
    O = _patches(oracle, psi, scan, pw)
    G = _patches(oracle, gobj, scan, pw)
    P0 = probe[0, 0, 0]
    Pn = w[:, 0, 0, None, None] * P0
    if eigen:
        Pn = Pn + w[:, 1, 0, None, None] * E[0, 0, 0]
    dOP, dPO, OP = G * Pn, mpu[0, 0, 0] * O, O * P0
    tot = lambda a: a.reshape(N, -1).sum(axis=1)
    a2 = tot(dOP * np.conj(dPO))
    want = np.stack([tot(np.abs(dOP)**2), tot(np.abs(dPO)**2), a2.real, a2.imag,
                     tot((np.conj(dOP) * chi0).real),
                     tot((np.conj(dPO) * chi0).real),
                     tot((np.conj(OP) * chi0).real), tot(np.abs(OP)**2)], 1)
    d = {k: A.to_device(v) for k, v in dict(
        psi=psi, gobj=gobj, probe=probe, mpu=mpu, chi0=chi0, scan=scan, O=O,
        w=w).items()}
    Ed = None if E is None else A.to_device(E)
    stats = torch.full((N, 8), np.nan, dtype=torch.float32, device="cuda")
    eproj = torch.full((N,), np.nan, dtype=torch.float32, device="cuda")
    check(lib.tike_lstsq_step_stats(
        A.ptr(d["chi0"]), A.ptr(d["scan"]), A.ptr(d["psi"]), A.ptr(d["gobj"]),
        A.ptr(d["probe"]), A.ptr(Ed), A.ptr(d["w"]) if eigen else None,
        1 if eigen else 0, 1 if eigen else 0, None, A.ptr(d["mpu"]),
        A.ptr(d["O"]), A.ptr(stats), N, 2, 1, pw, HW, HW,
        None if Ed is None else A.ptr(Ed[0, 0, 0]),
        A.ptr(eproj) if eigen else None, A.stream_ptr()))
    scale = np.abs(want).max(axis=0)
    np.testing.assert_allclose(stats.cpu().numpy(), want, rtol=2e-4,
                               atol=2e-5 * scale.max())
    if eigen:
        R = np.conj(O) * chi0 - mpu[0, 0, 0]
        np.testing.assert_allclose(
            eproj.cpu().numpy(), tot((np.conj(R) * E[0, 0, 0]).real),
            rtol=2e-4, atol=2e-5 * np.abs(tot(np.abs(R))).max())


@pytest.mark.parametrize("pw,N,border", [(256, 7, False), (256, 4, True),
                                         (128, 5, False), (48, 3, False),
                                         (512, 2, False)])
def test_eigen_position_sums_on_pairs_vs_numpy(oracle, pw, N, border):
    """tike_eigen_position_sums1 (probe.py:437-469 after the eigen probe's
    update) gathering O_n from the object, against NumPy."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(pw + 3 * N)
    HW = pw + 40
    scan = (rng.random((N, 2)) * 30 + 2.25).astype(np.float32)
    if border:
        scan[0] = (HW - pw + 2.5, 3.5)  # the window leaves the image
    psi = rc(rng, HW, HW)
    chi0, mpu, E = rc(rng, N, pw, pw), rc(rng, pw, pw), rc(rng, pw, pw)
    O = _patches(oracle, psi, scan, pw)
    R = np.conj(O) * chi0 - mpu
    phi = O * E
    tot = lambda a: a.reshape(N, -1).sum(axis=1)
    re = tot(R * np.conj(E))
    want = np.stack([tot((np.conj(R) * E).real), tot((chi0 * np.conj(phi)).real),
                     tot(np.abs(phi)**2), re.real, re.imag], 1)
    d = {k: A.to_device(v) for k, v in dict(psi=psi, chi0=chi0, mpu=mpu, E=E,
                                            scan=scan, O=O).items()}
    sums = torch.full((N, 5), np.nan, dtype=torch.float32, device="cuda")
    dsum = torch.zeros(1, dtype=torch.float32, device="cuda")
    check(lib.tike_eigen_position_sums1(
        A.ptr(d["O"]), A.ptr(d["chi0"]), A.ptr(d["mpu"]), A.ptr(d["E"]),
        A.ptr(sums), A.ptr(dsum), N, pw, 1, A.ptr(d["psi"]), A.ptr(d["scan"]),
        HW, HW, A.stream_ptr()))
    np.testing.assert_allclose(sums.cpu().numpy(), want, rtol=2e-4,
                               atol=2e-5 * np.abs(want).max())
    np.testing.assert_allclose(float(dsum), want[:, 2].sum() / (pw * pw),
                               rtol=2e-4)


@pytest.mark.parametrize("pw,N,S", [(512, 3, 2), (256, 6, 1), (128, 7, 3),
                                    (64, 2, 1), (40, 5, 1)])
def test_position_sums_on_pairs_vs_numpy(pw, N, S):
    """tike_position_sums (lstsq.py:545-579; position.py:779-810: gaussian
    derivative of radius 2 along both axes of the patch, central window) against
    NumPy for window widths that tile 256 threads and one that does not."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    from tike_amd.ptycho.position import gaussian_derivative_taps
    rng = np.random.default_rng(pw + N + S)
    O, chi = rc(rng, N, pw, pw), rc(rng, N, S, pw, pw)
    probe = rc(rng, 1, 1, S, pw, pw)
    # g[i] = sum_d t[d] x[i + d], d = -2 .. 2 (the taps themselves are pinned
    # to scipy's gaussian_filter1d by the CPU tests of gaussian_gradient)
    taps, radius = gaussian_derivative_taps(sigma=0.333)
    assert radius == 2
    crop = pw // 4
    win = (slice(None), slice(crop, pw - crop), slice(crop, pw - crop))
    gx = sum(taps[d + 2] * np.roll(O, -d, axis=1) for d in range(-2, 3))[win]
    gy = sum(taps[d + 2] * np.roll(O, -d, axis=2) for d in range(-2, 3))[win]
    P0, c = probe[0, 0, 0][win[1:]], chi[:, 0][win]
    tot = lambda a: a.reshape(N, -1).sum(axis=1)
    num = np.stack([tot((np.conj(gx * P0) * c).real),
                    tot((np.conj(gy * P0) * c).real)], 1)
    den = np.stack([tot(np.abs(gx * P0)**2), tot(np.abs(gy * P0)**2)], 1)
    d = {k: A.to_device(v) for k, v in dict(O=O, chi=chi, probe=probe).items()}
    got_n = torch.full((N, 2), np.nan, dtype=torch.float32, device="cuda")
    got_d = torch.full((N, 2), np.nan, dtype=torch.float32, device="cuda")
    check(lib.tike_position_sums(
        A.ptr(d["O"]), A.ptr(d["chi"]), S, A.ptr(d["probe"]), None, None, 0, 1,
        taps.ctypes.data, 2, A.ptr(got_n), A.ptr(got_d), N, S, pw,
        A.stream_ptr()))
    np.testing.assert_allclose(got_n.cpu().numpy(), num, rtol=2e-4,
                               atol=2e-5 * np.abs(num).max())
    np.testing.assert_allclose(got_d.cpu().numpy(), den, rtol=2e-4)


@pytest.mark.parametrize("pw,N,border", [(256, 70, False), (256, 9, True),
                                         (128, 33, False), (64, 5, True),
                                         (48, 7, False), (512, 4, False)])
def test_probe_preconditioner_vs_numpy(oracle, pw, N, border):
    """tike_probe_preconditioner (_preconditioner.py:106-160): sum_n
    |patch_n(psi)|^2 into the real parts of `out`, for window widths with and
    without the shared-tap-row kernel, several position chunks, windows that
    leave the image."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import check, lib
    rng = np.random.default_rng(pw + N)
    HW = pw + 40
    scan = (rng.random((N, 2)) * 30 + 2.25).astype(np.float32)
    if border:
        scan[1] = (HW - pw + 2.5, 3.5)
        scan[N - 1] = (-1.75, HW - pw + 1.25)
    psi = rc(rng, HW, HW)
    want = (np.abs(_patches(oracle, psi, scan, pw))**2).sum(axis=0)
    out = torch.full((pw, pw), 0.0 + 2.0j, dtype=torch.complex64, device="cuda")
    scan_d, psi_d = A.to_device(scan), A.to_device(psi)  # (kept alive)
    check(lib.tike_probe_preconditioner(A.ptr(scan_d), A.ptr(psi_d), A.ptr(out),
                                        N, pw, HW, HW, A.stream_ptr()))
    got = out.cpu().numpy()
    np.testing.assert_allclose(got.real, want, rtol=1e-4,
                               atol=1e-5 * want.max())
    assert np.all(got.imag == 2.0)  # untouched
