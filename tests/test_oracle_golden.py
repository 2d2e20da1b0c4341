"""Pin the CPU oracle against the reference's golden vectors and against
outputs of the reference itself (tests/golden/gen/make_fixtures.py).

Runs on CPU (-m "not gpu")."""
import numpy as np
import pytest

from oracle import operators as ops
from oracle import solvers as sol
from util import assert_close, relerr, COST_RTOL, SOLVER_NORMWISE


def test_simulate_matches_reference_fixture(golden):
    """tests/ptycho/test_ptycho.py:191-203 (atol 1e-6 on sqrt(I))."""
    g = golden("ref_ptycho_setup.npz")
    data = ops.simulate(g["data"].shape[-1], g["probe"], g["scan"],
                        g["original"])
    assert data.dtype == np.float32 and data.shape == g["data"].shape
    np.testing.assert_allclose(np.sqrt(data), np.sqrt(g["data"]), atol=1e-6)


def test_gaussian_probe_bit_exact(golden):
    """tests/ptycho/test_ptycho.py:80-90."""
    g = golden("ref_ptycho_gaussian.npz")
    np.testing.assert_array_equal(sol.gaussian_probe(15, 0.8, 1.0),
                                  g["weights"])


def test_orthogonalize_eig_reference_mat(golden):
    """tests/ptycho/test_probe.py:138-158 (magnitudes, rtol 1e-4)."""
    g = golden("ref_ortho.npz")
    out, _ = sol.orthogonalize_eig(g["modes"])
    np.testing.assert_allclose(np.abs(out), np.abs(g["pr"]), rtol=1e-4)


def test_patch_known_answer_forward():
    """tests/operators/test_patch.py:64-133."""
    size, win = 256, 8
    rng = np.random.default_rng(0)
    fov = (rng.random((size, size, 2), dtype=np.float32) - 0.5).view(
        np.complex64)[..., 0]
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
    patches = ops.patch_fwd(fov, positions, patch_width=win)
    np.testing.assert_allclose(patches, truth, atol=1e-6)


def test_patch_known_answer_adjoint():
    """tests/operators/test_patch.py:136-206."""
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
    out = ops.patch_adj(positions, np.ones((len(positions), win, win),
                                           dtype=np.complex64),
                        images=np.zeros((size, size), dtype=np.complex64),
                        patch_width=win)
    np.testing.assert_allclose(out, fov, atol=1e-6)


def test_patch_vs_reference(golden):
    g = golden("op_patch.npz")
    pw = int(g["pw"])
    H, W = g["images"].shape[-2:]
    assert_close(ops.patch_fwd(g["images"], g["positions"], patch_width=pw),
                 g["fwd1"], what="fwd1")
    padded = np.zeros_like(g["fwd2"])
    assert_close(
        ops.patch_fwd(g["images"], g["positions"], padded, patch_width=pw,
                      nrepeat=2), g["fwd2"], what="fwd2")
    assert_close(
        ops.patch_adj(g["positions"], g["patches_in"], patch_width=pw,
                      height=H, width=W), g["adj1"], what="adj1")
    assert_close(
        ops.patch_adj(g["positions"], g["padded_in"], patch_width=pw,
                      height=H, width=W, nrepeat=2), g["adj2"], what="adj2")
    assert_close(
        ops.patch_adj(g["positions"][0], g["bcast_in"], patch_width=pw,
                      height=H, width=W), g["adj3"],
        what="adj3 (K=1 broadcast)")


@pytest.mark.parametrize("tag", ["odd", "pow2", "full"])
def test_ptycho_operator_vs_reference(golden, tag):
    g = golden(f"op_ptycho_{tag}.npz")
    det = int(g["det"])
    N = len(g["scan"])
    fwd = ops.ptycho_fwd(g["probe"], g["scan"], g["psi"], det)
    assert_close(fwd, g["fwd"], what="fwd")
    bprobe = np.broadcast_to(g["probe"], (N, *g["probe"].shape[1:]))
    psi_adj, probe_adj = ops.ptycho_adj(g["farplane_in"], bprobe, g["scan"],
                                        g["psi"])
    assert_close(psi_adj, g["psi_adj"], what="psi_adj")
    assert_close(probe_adj, g["probe_adj"], what="probe_adj")
    inten = ops.intensity_from_farplane(fwd)
    assert_close(inten, g["intensity"], what="intensity")
    d = g["data"]
    np.testing.assert_allclose(ops.gaussian(d, inten), g["cost_gaussian"],
                               rtol=COST_RTOL)
    np.testing.assert_allclose(ops.poisson(d, inten), g["cost_poisson"],
                               rtol=COST_RTOL)
    assert_close(ops.gaussian_grad(d, fwd, inten), g["gaussian_grad"],
                 what="gaussian_grad")
    assert_close(ops.poisson_grad(d, fwd, inten), g["poisson_grad"],
                 normwise=1e-4, maxabs=1e-3, what="poisson_grad")
    np.testing.assert_allclose(ops.gaussian_each_pattern(d, inten),
                               g["gaussian_each"], rtol=COST_RTOL)
    np.testing.assert_allclose(ops.poisson_each_pattern(d, inten),
                               g["poisson_each"], rtol=COST_RTOL)


def test_adjoint_identity_reference_shapes():
    """tests/operators/test_ptycho.py:58-75: <Fm,d> = <psi,F*d> = <P,F*d>."""
    rng = np.random.default_rng(0)
    nscan, pw, S, det = 27, 15, 3, 45
    rc = lambda *s: (rng.random((*s, 2), dtype=np.float32) - 0.5).view(
        np.complex64)[..., 0]
    scan = (rng.random((nscan, 2), dtype=np.float32) * (127 - 16 - 2) + 1)
    probe, psi = rc(nscan, 1, S, pw, pw), rc(1, 128, 128)
    far = rc(nscan, 1, S, det, det)
    d = ops.ptycho_fwd(probe, scan, psi, det)
    m0, m1 = ops.ptycho_adj(far, probe, scan, psi)
    a = sol.inner(d, far)
    b = sol.inner(psi, m0)
    c = sol.inner(probe, m1)
    np.testing.assert_allclose([a.real, a.imag], [b.real, b.imag], rtol=1e-3)
    np.testing.assert_allclose([a.real, a.imag], [c.real, c.imag], rtol=1e-3)


@pytest.mark.parametrize("tag", ["plain", "eigen", "eigen128", "eigen256"])
def test_lstsq_parts_vs_reference(golden, tag):
    g = golden(f"lstsq_parts_{tag}.npz")
    det = int(g["det"])
    lo, hi = int(g["batch_lo"]), int(g["batch_hi"])
    ep = g["eigen_probe"] if "eigen_probe" in g else None
    ew = g["eigen_weights"] if "eigen_weights" in g else None
    psi, probe, scan = g["psi"], g["probe"], g["scan"]
    assert_close(sol.psi_preconditioner(psi, probe, scan), g["psi_precond"],
                 what="psi_precond")
    assert_close(sol.probe_preconditioner(psi, probe, scan),
                 g["probe_precond"], what="probe_precond")
    out = sol.get_nearplane_gradients(
        g["data"], psi, scan, probe, ep, ew, lo, hi, num_batch=2,
        detector_shape=det,
        measured_pixels=np.ones((det, det), dtype=bool))
    for k in ("chi", "unique_probe", "probe_update", "object_upd_sum",
              "m_probe_update", "patches"):
        if k not in g:  # the 128^2 / 256^2 fixtures keep mode 0 of chi only
            continue
        want = g[k]
        got = out[k][:, :, :want.shape[2]] if k == "chi" else out[k]
        assert_close(got, want, normwise=2e-5, what=k)
    np.testing.assert_allclose(out["costs"], g["costs"], rtol=COST_RTOL)
    if ew is not None:
        ep2, ew2 = sol.update_nearplane(out, probe,
                                        None if ep is None else ep.copy(),
                                        ew.copy(), lo, hi, num_batch=2)
        assert_close(ep2, g["eigen_probe_out"], normwise=1e-4, maxabs=1e-3,
                     what="eigen_probe")
        assert_close(ew2, g["eigen_weights_out"], normwise=1e-4, maxabs=1e-3,
                     what="eigen_weights")
    precond, bo, bp = sol.precondition_nearplane_gradients(
        out["chi"], scan, out["unique_probe"], probe, out["object_upd_sum"],
        out["m_probe_update"], g["psi_precond"], out["patches"], lo, hi)
    if "object_update_precond" in g:
        assert_close(precond, g["object_update_precond"], normwise=2e-5,
                     what="object_update_precond")
    assert bo.shape == g["beta_object"].shape
    assert bp.shape == g["beta_probe"].shape
    np.testing.assert_allclose(bo, g["beta_object"], rtol=1e-3)
    np.testing.assert_allclose(bp, g["beta_probe"], rtol=1e-3)


def _replay(g, second=False, solver="lstsq_grad"):
    """Replay the reference reconstruction with the oracle epoch driver."""
    det = int(g["det"])
    order = g["order"]
    sizes = g["batch_sizes"]
    ends = np.cumsum(sizes)
    batches = [np.arange(e - s, e) for s, e in zip(sizes, ends)]
    state = dict(psi=g["psi0"].copy(), probe=g["probe0"].copy(),
                 scan=g["scan"][order].copy(), costs=[],
                 eigen_probe=g["eigen_probe"].copy()
                 if "eigen_probe" in g else None,
                 eigen_weights=g["eigen_weights"][order].copy()
                 if "eigen_weights" in g else None)
    data = g["data"][order]
    if "position_keys" in g:
        opts = dict(zip((str(k) for k in g["position_keys"]),
                        (float(v) for v in g["position_vals"])))
        state["position"] = dict(
            initial_scan=g["scan"][order].copy(),
            momentum=np.zeros((len(order), 4), dtype=np.float32), **opts)
    rng = np.random.default_rng(11)
    measured = g["measured"].astype(bool)
    kw = dict(detector_shape=det, batch_method=str(g["batch_method"]),
              force_orthogonality=bool(g["orth"]),
              object_adaptive_moment=bool(g["adaptive"]),
              probe_adaptive_moment=bool(g["adaptive"]), rng=rng,
              measured_pixels=measured, noise_model=str(g["noise_model"]),
              step_length_usemodes=str(g["usemodes"]),
              unmeasured_pixels_scaling=float(g["scaling"]))
    epochs = int(g["epochs"])
    rescale_kw = {}
    if solver != "rpie" and "no_probe" in g and bool(g["no_probe"]):
        # the reference's "no probe" runs: an update start that never comes
        kw.update(probe_update_start=10**6, probe_adaptive_moment=False)
    if solver == "rpie":
        kw.update(solver="rpie", recover_probe=not bool(g["no_probe"]))
        kw.pop("object_adaptive_moment")
        kw.pop("probe_adaptive_moment")
        if float(g["alpha"]) >= 0:
            kw["alpha"] = float(g["alpha"])
        if g["psi0"].shape[0] > 1:
            wl, fy, fx, dist = (float(v) for v in g["phys"])
            pw = g["probe0"].shape[-1]
            kw["propagator"] = ops.fresnel_spectrum_propagator(
                (pw, pw), (fy, fx), dist, wl)
            rescale_kw["propagator"] = kw["propagator"]
    state = sol.rescale_probe(state, data, det, measured_pixels=measured,
                              **rescale_kw)
    state = sol.iterate(state, data, batches, epochs, **kw)
    first = {k: (None if v is None else np.array(v, copy=True))
             for k, v in state.items() if k in ("psi", "probe", "eigen_probe",
                                                "eigen_weights", "scan")}
    if "position" in state:
        first["transform"] = tuple(state["position"]["transform"])
    first["costs"] = list(state["costs"])
    if second:
        order2 = g["order_2"] if "order_2" in g else order
        if not np.array_equal(order2, order):
            # the second call clusters the (moved) positions again: bring the
            # per-position state from the first arrangement into the second
            perm = np.argsort(order)[order2]
            for k in ("scan", "eigen_weights"):
                if state.get(k) is not None:
                    state[k] = state[k][perm]
            if "position" in state:
                for k in ("initial_scan", "momentum"):
                    state["position"][k] = state["position"][k][perm]
            data = g["data"][order2]
            ends2 = np.cumsum(g["batch_sizes_2"])
            batches = [np.arange(e - s, e)
                       for s, e in zip(g["batch_sizes_2"], ends2)]
            order = order2
        state = sol.rescale_probe(state, data, det, measured_pixels=measured,
                                  **rescale_kw)
        state = sol.iterate(state, data, batches, epochs, **kw)
    return first, state, order


@pytest.mark.parametrize("tag", ["compact", "wobbly_eigen", "poisson_all",
                                 "poisson_dominant", "noprobe",
                                 "compact_noprobe", "eigen_modes2",
                                 "eigen2_modes2", "bootstrap"])
def test_lstsq_reconstruction_vs_reference(golden, tag):
    g = golden(f"lstsq_recon_{tag}.npz")
    first, state, order = _replay(g, second=True)
    np.testing.assert_allclose(np.array(first["costs"]), g["costs_1"],
                               rtol=1e-3)
    assert_close(first["psi"], g["psi_1"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="psi after call 1")
    assert_close(first["probe"], g["probe_1"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe after call 1")
    if "eigen_weights" in g:
        inv = np.argsort(g["order"])
        assert_close(first["eigen_weights"][inv], g["eigen_weights_1"],
                     normwise=5e-3, maxabs=5e-2, what="eigen_weights")
    np.testing.assert_allclose(np.array(state["costs"]), g["costs_2"],
                               rtol=5e-3)
    assert_close(state["psi"], g["psi_2"], normwise=5e-3, maxabs=5e-2,
                 what="psi after call 2")


@pytest.mark.parametrize("tag", ["positions_adam", "positions_plain",
                                 "positions_masked"])
def test_position_correction_vs_reference(golden, tag):
    """lstsq_grad with position correction (gaussian-derivative shift
    estimate, trimmed mean, ADAM, affine regularisation with RANSAC draws from
    the shared generator) replayed against the reference's own run."""
    g = golden(f"lstsq_recon_{tag}.npz")
    first, state, order = _replay(g, second=True)
    inv = np.argsort(g["order"])
    np.testing.assert_allclose(np.array(first["costs"]), g["costs_1"],
                               rtol=2e-3)
    # positions in pixels: absolute tolerance
    np.testing.assert_allclose(first["scan"][inv], g["scan_1"], atol=2e-3)
    inv = np.argsort(order)
    np.testing.assert_allclose(first["transform"], g["transform_1"],
                               rtol=1e-3, atol=1e-3)
    assert_close(first["psi"], g["psi_1"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="psi after call 1")
    np.testing.assert_allclose(state["scan"][inv], g["scan_2"], atol=2e-2)
    np.testing.assert_allclose(np.array(state["costs"]), g["costs_2"],
                               rtol=1e-2)


def test_cgrad_vs_reference_composition(golden):
    g = golden("cgrad.npz")
    det = int(g["det"])
    N = len(g["scan"])
    state = dict(psi=g["psi0"].copy(), probe=g["probe"].copy(),
                 scan=g["scan"].copy(), costs=[])
    batches = [np.arange(N)]
    for i in range(3):
        state = sol.cgrad(state, g["data"], batches, detector_shape=det,
                          cg_iter=4)
        np.testing.assert_allclose(state["costs"][-1][0], g["costs"][i + 1],
                                   rtol=2e-3)
        assert relerr(state["psi"], g["psis"][i]) < 2e-3


def test_cgrad_with_probe_vs_reference_composition(golden):
    """The oracle's cgrad with the probe step and two minibatches replays the
    same composition of the reference's own pieces."""
    g = golden("cgrad_probe.npz")
    det, N = int(g["det"]), len(g["scan"])
    state = dict(psi=g["psi0"].copy(), probe=g["probe0"].copy(),
                 scan=g["scan"].copy(), costs=[])
    batches = np.array_split(np.arange(N), 2)
    for i in range(3):
        state = sol.cgrad(state, g["data"], batches, detector_shape=det,
                          cg_iter=int(g["cg_iter"]), recover_probe=True)
        np.testing.assert_allclose(state["costs"][-1][0], g["costs"][i],
                                   rtol=2e-3)
        assert relerr(state["psi"], g["psis"][i]) < 2e-3
        assert relerr(state["probe"], g["probes"][i]) < 2e-3


def test_multislice_vs_reference(golden):
    """FresnelSpectProp, Multislice (3 slices) and Ptycho over it, against
    the reference's own outputs (operators/cupy/multislice.py:69-194,
    fresnelspectprop.py:52-137, ptycho.py:114-176)."""
    g = golden("op_multislice.npz")
    wl, fy, fx, dist = (float(v) for v in g["phys"])
    pw = g["probe"].shape[-1]
    H = ops.fresnel_spectrum_propagator((pw, pw), (fy, fx), dist, wl)
    assert_close(H, g["propagator"], what="propagator")
    assert_close(ops.fresnel_fwd(g["nearplane_in"], H), g["fresnel_fwd"],
                 what="fresnel fwd")
    assert_close(ops.fresnel_adj(g["nearplane_in"], H), g["fresnel_adj"],
                 what="fresnel adj")
    probe, scan, psi = g["probe"], g["scan"], g["psi"]
    assert_close(ops.multislice_fwd(probe, scan, psi, H), g["ms_fwd"],
                 what="multislice fwd")
    exitw, probes = ops.multislice_fwd_intermediate(probe, scan, psi, H)
    assert_close(exitw, g["ms_exit"], what="multislice exit wave")
    assert_close(probes, g["ms_probes"], what="intermediate probes")
    pa, qa = ops.multislice_adj(g["nearplane_in"], probe, scan, psi, H)
    assert_close(pa, g["ms_psi_adj"], what="multislice psi_adj")
    assert_close(qa, g["ms_probe_adj"], what="multislice probe_adj")
    assert_close(ops.ptycho_fwd(probe[:, None], scan, psi, pw, propagator=H),
                 g["pt_fwd"], what="ptycho fwd (3 slices)")
    pa, qa = ops.ptycho_adj(g["farplane_in"], probe[:, None], scan, psi,
                            propagator=H)
    assert_close(pa, g["pt_psi_adj"], what="ptycho psi_adj (3 slices)")
    assert_close(qa, g["pt_probe_adj"], what="ptycho probe_adj (3 slices)")


@pytest.mark.parametrize("tag", ["epie", "object", "twoslice", "poisson",
                                 "poisson_all", "eigen"])
def test_rpie_reconstruction_vs_reference(golden, tag):
    """The oracle's rpie (solvers/rpie.py:26-612 as this snapshot has it)
    replays the reference's own runs: alpha = 1 (ePIE) with object and probe,
    the default alpha with the object alone and NaN-masked data, and a
    two-slice object through the multislice forward model."""
    g = golden(f"rpie_recon_{tag}.npz")
    first, state, order = _replay(g, second=True, solver="rpie")
    np.testing.assert_allclose(np.array(first["costs"]), g["costs_1"],
                               rtol=1e-3)
    assert_close(first["psi"], g["psi_1"], normwise=1e-3, maxabs=1e-2,
                 what="psi after call 1")
    assert_close(first["probe"], g["probe_1"], normwise=1e-3, maxabs=1e-2,
                 what="probe after call 1")
    np.testing.assert_allclose(np.array(state["costs"]), g["costs_2"],
                               rtol=5e-3)
    assert_close(state["psi"], g["psi_2"], normwise=5e-3, maxabs=5e-2,
                 what="psi after call 2")
    assert_close(state["probe"], g["probe_2"], normwise=5e-3, maxabs=5e-2,
                 what="probe after call 2")
