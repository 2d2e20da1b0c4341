"""GPU parity of the update loop: HIP kernels + host driver vs fixtures
recorded from the reference (tests/golden/gen/make_fixtures.py) and vs the
oracle."""
import numpy as np
import pytest

from util import (assert_close, relerr, COST_RTOL, SOLVER_NORMWISE)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tp():
    import tike_amd.ptycho as m
    return m


def test_simulate_matches_reference_fixture(tp, golden):
    """reference tests/ptycho/test_ptycho.py:191-203 (atol 1e-6 on sqrt I)."""
    g = golden("ref_ptycho_setup.npz")
    data = tp.simulate(detector_shape=g["data"].shape[-1], probe=g["probe"],
                       scan=g["scan"], psi=g["original"])
    assert data.dtype == np.float32 and data.shape == g["data"].shape
    np.testing.assert_allclose(np.sqrt(data), np.sqrt(g["data"]), atol=1e-6)


@pytest.mark.parametrize("tag", ["plain", "eigen", "eigen128", "eigen256"])
def test_lstsq_minibatch_kernels_vs_reference(tp, golden, tag):
    """One minibatch through the HIP kernels == the reference's
    _get_nearplane_gradients / _precondition_nearplane_gradients /
    _update_nearplane outputs.  eigen128 (8 modes) / eigen256: the tile sizes
    of the v2 FFT engine, the position-major forward and the far-plane-free
    gradient + inverse kernels, run by the reference itself."""
    import torch
    import tike_amd._arrays as A
    from tike_amd.communicators import Comm
    from tike_amd.operators import Ptycho
    from tike_amd.ptycho.solvers import lstsq as L
    from tike_amd.ptycho.solvers._preconditioner import (
        _psi_preconditioner, _probe_preconditioner)
    g = golden(f"lstsq_parts_{tag}.npz")
    det = int(g["det"])
    lo, hi = int(g["batch_lo"]), int(g["batch_hi"])
    psi, probe, scan = (A.to_device(g[k]) for k in ("psi", "probe", "scan"))
    data = A.to_device(g["data"], np.float32)
    ep = A.to_device(g["eigen_probe"]) if "eigen_probe" in g else None
    ew = A.to_device(g["eigen_weights"]) if "eigen_weights" in g else None
    params = tp.PtychoParameters(
        probe=probe, psi=psi, scan=scan, eigen_probe=ep, eigen_weights=ew,
        algorithm_options=tp.LstsqOptions(num_batch=2),
        probe_options=tp.ProbeOptions(), object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)).copy_to_device())
    comm = Comm()
    HW = psi.shape[-1]
    with Ptycho(probe_shape=probe.shape[-1], detector_shape=det, nz=HW,
                n=HW) as op:
        assert_close(_psi_preconditioner(params, op).cpu().numpy(),
                     g["psi_precond"], what="psi preconditioner")
        assert_close(_probe_preconditioner(params, op).cpu().numpy(),
                     g["probe_precond"], what="probe preconditioner")
        out = L._get_nearplane_gradients(
            data, psi, scan, probe, ep, ew, lo, hi, comm, num_batch=2,
            exitwave_options=params.exitwave_options, op=op, recover_psi=True,
            recover_probe=True)
        assert_close(L.object_upd_sum(out).cpu().numpy(), g["object_upd_sum"],
                     normwise=2e-5, what="object_upd_sum")
        assert_close(out["m_probe_update"].cpu().numpy(), g["m_probe_update"],
                     normwise=2e-5, what="m_probe_update")
        chi0 = out["chi0"]
        if out["chi_modes"] > 1:  # mode 0 read in place from the full chi
            chi0 = chi0[:hi - lo, 0, 0]
        assert_close(chi0.cpu().numpy(), g["chi"][:, 0, 0],
                     normwise=2e-5, what="chi mode 0")
        np.testing.assert_allclose(out["costs"].cpu().numpy(),
                                   np.ravel(g["costs"]), rtol=COST_RTOL)
        if out["patches"] is not None and "patches" in g:
            assert_close(out["patches"].cpu().numpy(), g["patches"][:, 0, 0],
                         what="patches")
        precond = L._precondition_object_update(
            out["object_acc"], A.to_device(g["psi_precond"]))
        if "object_update_precond" in g:
            assert_close(precond.cpu().numpy(), g["object_update_precond"],
                         normwise=2e-5, what="object_update_precond")
        stats = L._step_stats(out, psi, scan, probe, ep, precond, lo, hi,
                              op=op)
        if ew is not None:
            ep2, ew2 = L._update_nearplane(out, stats, probe, ep.clone(),
                                           ew.clone(), lo, hi, comm,
                                           num_batch=2)
            assert_close(ep2.cpu().numpy(), g["eigen_probe_out"],
                         normwise=1e-4, maxabs=1e-3, what="eigen_probe")
            assert_close(ew2.cpu().numpy(), g["eigen_weights_out"],
                         normwise=1e-4, maxabs=1e-3, what="eigen_weights")
        bo, bp, cost = L._solve_steps(stats, out["costs"], out["count"], comm,
                                      pw=probe.shape[-1], recover_psi=True,
                                      recover_probe=True)
        np.testing.assert_allclose(float(cost), g["costs"].mean(),
                                   rtol=COST_RTOL)
        np.testing.assert_allclose(float(bo), g["beta_object"].ravel()[0],
                                   rtol=1e-3)
        np.testing.assert_allclose(float(bp), g["beta_probe"].ravel()[0],
                                   rtol=1e-3)
        if ep is None or ep.shape[-4] == 1:
            # the packed tail lstsq_grad runs for at most one eigen probe,
            # against the same outputs of the reference
            ep3 = None if ep is None else ep.clone()
            ew3 = None if ew is None else ew.clone()
            probe3 = probe.clone()
            steps_row = torch.zeros(5, dtype=torch.float32, device=psi.device)
            norm = None
            if ep is not None:
                norm = torch.square(ew3[lo:hi, 1, 0]).sum().reshape(1)
            bo3, bp3 = L._packed_tail(
                out, psi, scan, probe3, ep3, ew3, precond, lo, hi, comm, op=op,
                num_batch=2, recover_psi=True, recover_probe=True, norm=norm,
                steps_row=steps_row,
                probe_combined_update=torch.zeros_like(probe3))
            np.testing.assert_allclose(float(bo3), g["beta_object"].ravel()[0],
                                       rtol=1e-3)
            np.testing.assert_allclose(float(bp3), g["beta_probe"].ravel()[0],
                                       rtol=1e-3)
            np.testing.assert_allclose(float(steps_row[4]), g["costs"].mean(),
                                       rtol=COST_RTOL)
            if ew is not None:
                assert_close(ew3.cpu().numpy(), g["eigen_weights_out"],
                             normwise=1e-4, maxabs=1e-3,
                             what="eigen_weights (packed tail)")
            if ep is not None:
                assert_close(ep3.cpu().numpy(), g["eigen_probe_out"],
                             normwise=1e-4, maxabs=1e-3,
                             what="eigen_probe (packed tail)")


def _reconstruct_like_reference(tp, g, second, algo="lstsq"):
    import json
    import tike_amd.random
    extras = (json.loads(str(g["extras"])) if "extras" in g else
              dict(probe={}, object={}, algorithm={}))
    for k in ("median_filter_abs_probe_px", "probe_FOV_lengths"):
        if k in extras["probe"]:
            extras["probe"][k] = tuple(extras["probe"][k])
    det = int(g["det"])
    sizes = g["batch_sizes"]
    ends = np.cumsum(sizes)
    batches = [np.arange(e - s, e) for s, e in zip(sizes, ends)]
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    adaptive, orth = bool(g["adaptive"]), bool(g["orth"])
    params = tp.PtychoParameters(
        probe=g["probe0"].copy(), psi=g["psi0"].copy(), scan=g["scan"].copy(),
        eigen_probe=g["eigen_probe"].copy() if "eigen_probe" in g else None,
        eigen_weights=g["eigen_weights"].copy()
        if "eigen_weights" in g else None,
        algorithm_options=(tp.RpieOptions(
            num_batch=int(g["num_batch"]),
            batch_method=str(g["batch_method"]), num_iter=int(g["epochs"]),
            **({} if float(g["alpha"]) < 0 else dict(alpha=float(g["alpha"]))))
                           if algo == "rpie" else tp.LstsqOptions(
            num_batch=int(g["num_batch"]),
            batch_method=str(g["batch_method"]), num_iter=int(g["epochs"]),
            **extras["algorithm"])),
        probe_options=tp.ProbeOptions(
            force_orthogonality=orth, use_adaptive_moment=adaptive,
            **extras["probe"],
            **(dict(update_start=10**6)
               if "no_probe" in g and bool(g["no_probe"]) else {}),
            **(dict(probe_wavelength=float(g["phys"][0]),
                    probe_FOV_lengths=(float(g["phys"][1]),
                                       float(g["phys"][2])))
               if algo == "rpie" and g["psi0"].shape[0] > 1 else {})),
        object_options=tp.ObjectOptions(
            use_adaptive_moment=adaptive, **extras["object"],
            **(dict(multislice_propagation_distance=float(g["phys"][3]))
               if algo == "rpie" and g["psi0"].shape[0] > 1 else {})),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=g["measured"].astype(bool),
            noise_model=str(g["noise_model"]),
            step_length_usemodes=str(g["usemodes"]),
            unmeasured_pixels_scaling=float(g["scaling"])))
    if "position_keys" in g:
        opts = dict(zip((str(k) for k in g["position_keys"]),
                        (float(v) for v in g["position_vals"])))
        for k in ("use_adaptive_moment", "use_position_regularization"):
            if k in opts:
                opts[k] = bool(opts[k])
        params.position_options = tp.PositionOptions(g["scan"].copy(), **opts)
    results = []
    order = g["order"]
    for call in range(2 if second else 1):
        if call == 1 and "order_2" in g:
            # the reference clusters the (moved) positions again
            order = g["order_2"]
            ends = np.cumsum(g["batch_sizes_2"])
            batches = [np.arange(e - s, e)
                       for s, e in zip(g["batch_sizes_2"], ends)]
        with tp.Reconstruction(g["data"], params, order=order,
                               batches=batches) as ctx:
            ctx.iterate(int(g["epochs"]))
            params = ctx.get_result()
        results.append(params)
    return results


@pytest.mark.parametrize("tag", ["compact", "wobbly_eigen", "poisson_all",
                                 "poisson_dominant", "noprobe",
                                 "compact_noprobe", "constraints",
                                 "constraints_photons", "eigen_modes2",
                                 "eigen2_modes2", "bootstrap"])
def test_lstsq_reconstruct_twice_vs_reference(tp, golden, tag):
    """The reference's ReconstructTwice template (tests/ptycho/templates.py:
    115-129), asserted against the reference's own iterates."""
    g = golden(f"lstsq_recon_{tag}.npz")
    r1, r2 = _reconstruct_like_reference(tp, g, second=True)
    epochs = int(g["epochs"])
    # float atomics reorder sums: the eigen quantities and the second call
    # (3 more epochs on top of the first call's round-off) are compared at
    # 5e-3 / 5e-2.  Under TIKE_DETERMINISTIC=1 (ordered sums; the whole file
    # re-run by test_fixtures_under_the_deterministic_switch) they meet the
    # solver tolerance of DESIGN.md section 4, 1e-3 / 1e-2
    from tike_amd import _lib
    loose, loose_abs = ((SOLVER_NORMWISE, 1e-2) if _lib.DETERMINISTIC else
                        (5e-3, 5e-2))
    np.testing.assert_allclose(
        np.array(r1.algorithm_options.costs[:epochs]), g["costs_1"],
        rtol=1e-3)
    assert_close(r1.psi, g["psi_1"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi after call 1")
    assert_close(r1.probe, g["probe_1"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe after call 1")
    if "eigen_weights" in g:
        assert_close(r1.eigen_weights, g["eigen_weights_1"], normwise=loose,
                     maxabs=loose_abs, what="eigen_weights after call 1")
        assert_close(r1.eigen_probe, g["eigen_probe_1"], normwise=loose,
                     maxabs=loose_abs, what="eigen_probe after call 1")
    np.testing.assert_allclose(np.array(r2.algorithm_options.costs),
                               g["costs_2"], rtol=loose)
    assert_close(r2.psi, g["psi_2"], normwise=loose, maxabs=loose_abs,
                 what="psi after call 2")
    assert_close(r2.probe, g["probe_2"], normwise=loose, maxabs=loose_abs,
                 what="probe after call 2")


@pytest.mark.parametrize("tag", ["wobbly_eigen", "positions_adam"])
def test_chunked_minibatches_vs_reference(tp, golden, tag, monkeypatch):
    """Minibatches split into several kernel chunks (7 positions each) give the
    reference's iterates too: exercises the packed mode-0 copy of chi and the
    per-chunk offsets of every per-position array."""
    from tike_amd.ptycho.solvers import lstsq as L
    monkeypatch.setattr(L, "CHUNK_POSITIONS_OVERRIDE", 7)
    g = golden(f"lstsq_recon_{tag}.npz")
    (r1,) = _reconstruct_like_reference(tp, g, second=False)
    epochs = int(g["epochs"])
    np.testing.assert_allclose(
        np.array(r1.algorithm_options.costs[:epochs]), g["costs_1"],
        rtol=2e-3)
    assert_close(r1.psi, g["psi_1"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(r1.probe, g["probe_1"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")
    if "scan_1" in g:
        np.testing.assert_allclose(r1.scan, g["scan_1"], atol=2e-3)


@pytest.mark.parametrize("tag", ["positions_adam", "positions_plain",
                                 "positions_masked"])
def test_position_correction_vs_reference(tp, golden, tag):
    """lstsq_grad with position correction (lstsq.py:545-579,764-806; affine
    regularisation position.py:716-776) against the reference's own run:
    positions, affine transform, ADAM moments, object and costs."""
    g = golden(f"lstsq_recon_{tag}.npz")
    r1, r2 = _reconstruct_like_reference(tp, g, second=True)
    epochs = int(g["epochs"])
    # float atomics reorder sums: the eigen quantities and the second call
    # (3 more epochs on top of the first call's round-off) are compared at
    # 5e-3 / 5e-2.  Under TIKE_DETERMINISTIC=1 (ordered sums; the whole file
    # re-run by test_fixtures_under_the_deterministic_switch) they meet the
    # solver tolerance of DESIGN.md section 4, 1e-3 / 1e-2
    from tike_amd import _lib
    loose, loose_abs = ((SOLVER_NORMWISE, 1e-2) if _lib.DETERMINISTIC else
                        (5e-3, 5e-2))
    np.testing.assert_allclose(
        np.array(r1.algorithm_options.costs[:epochs]), g["costs_1"],
        rtol=2e-3)
    np.testing.assert_allclose(r1.scan, g["scan_1"], atol=2e-3)  # pixels
    np.testing.assert_allclose(r1.position_options.transform.asbuffer(),
                               g["transform_1"], rtol=1e-3, atol=1e-3)
    if g["momentum_1"].size:
        np.testing.assert_allclose(r1.position_options._momentum,
                                   g["momentum_1"], rtol=2e-2, atol=1e-4)
    assert_close(r1.psi, g["psi_1"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi after call 1")
    np.testing.assert_allclose(r2.scan, g["scan_2"], atol=2e-2)
    np.testing.assert_allclose(np.array(r2.algorithm_options.costs),
                               g["costs_2"], rtol=1e-2)


def test_position_sums_kernel_vs_oracle(tp):
    """tike_position_sums vs the oracle's position_update_terms, with a
    varying (eigen) probe and a window where crop < filter radius matters."""
    import torch
    import tike_amd._arrays as A
    from tike_amd._lib import lib, check
    from tike_amd.ptycho.position import gaussian_derivative_taps
    from oracle import position as opos
    from oracle import solvers as osol
    rng = np.random.default_rng(12)
    N, S, pw, C, Sm = 7, 3, 20, 2, 1
    rc = lambda *s: (rng.random(s) - 0.5 + 1j * (rng.random(s) - 0.5)).astype(
        np.complex64)
    patches, chi = rc(N, pw, pw), rc(N, 1, S, pw, pw)
    probe, eigen = rc(1, 1, S, pw, pw), rc(1, C, Sm, pw, pw)
    weights = rng.random((N, C + 1, S)).astype(np.float32)
    unique = osol.get_varying_probe(probe, eigen, weights)
    num, den = opos.position_update_terms(patches[:, None, None], unique, chi)
    taps, r = gaussian_derivative_taps(0.333)
    d = {k: A.to_device(v) for k, v in dict(
        patches=patches, chi=chi, probe=probe, eigen=eigen,
        weights=weights).items()}
    gnum = torch.zeros((N, 2), dtype=torch.float32, device="cuda")
    gden = torch.zeros_like(gnum)
    check(lib.tike_position_sums(
        A.ptr(d["patches"]), A.ptr(d["chi"]), S, A.ptr(d["probe"]),
        A.ptr(d["eigen"]), A.ptr(d["weights"]), C, Sm, taps.ctypes.data, r,
        A.ptr(gnum), A.ptr(gden), N, S, pw, A.stream_ptr()), "position sums")
    np.testing.assert_allclose(gnum.cpu().numpy(), num, rtol=2e-4, atol=1e-5)
    np.testing.assert_allclose(gden.cpu().numpy(), den, rtol=2e-4, atol=1e-6)


def test_cgrad_vs_reference_composition(tp, golden):
    g = golden("cgrad.npz")
    det = int(g["det"])
    N = len(g["scan"])
    params = tp.PtychoParameters(
        probe=g["probe"].copy(), psi=g["psi0"].copy(), scan=g["scan"].copy(),
        algorithm_options=tp.CgradOptions(num_batch=1, cg_iter=4, num_iter=1,
                                          batch_method="contiguous"),
        probe_options=None, object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    for i in range(3):
        params = tp.reconstruct(g["data"], params)
        np.testing.assert_allclose(params.algorithm_options.costs[-1][0],
                                   g["costs"][i + 1], rtol=2e-3)
        assert relerr(params.psi, g["psis"][i]) < 2e-3


def test_cgrad_with_probe_vs_reference_composition(tp, golden):
    """Two minibatches, object then probe per minibatch (3 CG iterations
    each), three epochs, against the same composition of the reference's own
    pieces (tike.opt.conjugate_gradient over Ptycho.cost / Ptycho.adj, the
    probe gradient summed over the positions)."""
    g = golden("cgrad_probe.npz")
    det, N = int(g["det"]), len(g["scan"])
    params = tp.PtychoParameters(
        probe=g["probe0"].copy(), psi=g["psi0"].copy(), scan=g["scan"].copy(),
        algorithm_options=tp.CgradOptions(num_batch=2, cg_iter=int(g["cg_iter"]),
                                          batch_method="contiguous"),
        probe_options=tp.ProbeOptions(init_rescale_from_measurements=False),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    with tp.Reconstruction(g["data"], params, order=np.arange(N),
                           batches=np.array_split(np.arange(N), 2),
                           spatial_sort=False) as ctx:
        for i in range(3):
            ctx.iterate(1)
            r = ctx.get_result()
            np.testing.assert_allclose(r.algorithm_options.costs[-1][0],
                                       g["costs"][i], rtol=5e-3)
            assert relerr(r.psi, g["psis"][i]) < 3e-3
            assert relerr(r.probe, g["probes"][i]) < 3e-3


@pytest.mark.parametrize("det,pw,S,N", [(256, 256, 1, 6), (256, 192, 2, 5),
                                        (128, 128, 1, 8), (512, 512, 2, 3),
                                        (64, 48, 2, 9),
                                        # round 6: sizes off the power-of-two
                                        # grid (prime-factor / mixed-radix /
                                        # Bluestein transforms under cgrad)
                                        (96, 96, 2, 6), (100, 80, 1, 7),
                                        (192, 160, 2, 5), (640, 640, 1, 2),
                                        (300, 300, 2, 4), (400, 320, 1, 4),
                                        (384, 384, 1, 3), (127, 127, 1, 4),
                                        (128, 128, 1, 256)])  # BASELINE configs[0]
def test_cgrad_vs_oracle(tp, det, pw, S, N):
    """cgrad (object then probe, 2 CG iterations each) against the oracle's
    composition: the gradients are lstsq_grad's pipelines (far-plane-free at
    256, split forward + kept far plane at 512, whole-tile kernels at 128,
    generic below), the line-search costs the split forward with nothing but
    the costs stored (256 / 512)."""
    from oracle import solvers as osol
    rng = np.random.default_rng(det + N)
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side),
                              indexing="ij"), -1).reshape(-1, 2)[:N]
    scan = (2 + 7.0 * ij + rng.random((N, 2))).astype(np.float32)
    HW = 7 * (side - 1) + pw + 8
    psi_true = ((0.75 + 0.25 * rng.random((1, HW, HW))) * np.exp(
        1j * np.pi * (rng.random((1, HW, HW)) - 0.5))).astype(np.complex64)
    w = tp.gaussian(pw, rin=0.6)
    probe = np.stack([w * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
                      for m in range(S)])[None, None].astype(np.complex64)
    data = tp.simulate(det, probe, scan, psi_true)
    psi0 = np.full_like(psi_true, 0.5)
    params = tp.PtychoParameters(
        probe=probe.copy(), psi=psi0.copy(), scan=scan.copy(),
        algorithm_options=tp.CgradOptions(num_batch=1, cg_iter=2, num_iter=1,
                                          batch_method="contiguous"),
        probe_options=tp.ProbeOptions(init_rescale_from_measurements=False),
        object_options=tp.ObjectOptions())
    batches = [np.arange(N)]
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=batches) as ctx:
        ctx.iterate(2)
        got = ctx.get_result()
    state = dict(psi=psi0.copy(), probe=probe.copy(), scan=scan.copy(),
                 costs=[])
    for _ in range(2):
        state = osol.cgrad(state, data, batches, detector_shape=det, cg_iter=2,
                           recover_probe=True)
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(state["costs"]), rtol=2e-3)
    assert_close(got.psi, state["psi"], normwise=2e-3, maxabs=2e-2, what="psi")
    assert_close(got.probe, state["probe"], normwise=2e-3, maxabs=2e-2,
                 what="probe")


@pytest.mark.parametrize("det,S,N,slots,most", [
    (256, 1, 12, (8, 4), 30),
    (256, 2, 7, (1, 1), 30),  # out of slots -> repeated with more slots
    (256, 2, 7, (1, 1), 1),  # ... no more to be had -> host-side search
    (512, 2, 4, (8, 4), 30),
    (128, 1, 10, (8, 4), 30),  # configs[0] size
    (128, 2, 6, (8, 4), 30)])
def test_cgrad_device_line_search_equals_host_line_search(tp, monkeypatch, det,
                                                          S, N, slots, most):
    """The line search decided on the device (tike_cgrad_line_search: trials
    enqueued ahead, skipped once one is accepted) follows opt.line_search
    (opt.py:216-278) trial for trial: same accepted step lengths, therefore
    the same iterates as the host-side search that reads every cost back.
    slots = (1, 1): most searches run out of slots; the call is then repeated
    with every slot the entry allows (and the reconstruction remembers what
    its searches needed), or -- `most` = 1: there are no more -- handed to the
    host-side search.  Every route must give the same result."""
    import importlib
    C = importlib.import_module("tike_amd.ptycho.solvers.cgrad")
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=det + 3 * S, eigen=False)
    results = []
    for on_device in (True, False):
        monkeypatch.setattr(C, "DEVICE_LINE_SEARCH", on_device)
        monkeypatch.setattr(C, "LINE_SEARCH_SLOTS", slots)
        monkeypatch.setattr(C, "MAX_SLOTS", most)
        monkeypatch.setattr(C, "USE_GRAPHS", False)  # every call through the spy
        monkeypatch.setattr(C, "LINEAR_LINE_SEARCH", False)  # (its own test)
        calls = []
        real = C._cg_device

        def spy(*a, **k):
            r = real(*a, **k)
            calls.append(r is not None)
            return r

        monkeypatch.setattr(C, "_cg_device", spy)
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=tp.CgradOptions(num_batch=2, cg_iter=3,
                                              num_iter=2,
                                              batch_method="compact"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=np.array_split(np.arange(N), 2)) as ctx:
            ctx.iterate(2)
            results.append(ctx.get_result())
        monkeypatch.setattr(C, "_cg_device", real)
        if on_device:
            # 2 epochs x 2 minibatches x (object, probe) = 8 CG calls
            if slots == (8, 4):
                assert calls == [True] * 8  # every step found in its slots
            elif most == 1:
                assert len(calls) == 8 and not all(calls)  # host fallback ran
            else:
                # a call that ran out was repeated with more slots and then
                # succeeded; what it needed is remembered, so that the last
                # calls succeed at once
                assert calls.count(True) == 8 and not all(calls)
                assert calls[-2:] == [True, True]
        else:
            assert not calls
    a, b = results
    np.testing.assert_allclose(np.array(a.algorithm_options.costs),
                               np.array(b.algorithm_options.costs), rtol=1e-5)
    assert_close(a.psi, b.psi, normwise=1e-5, maxabs=1e-4, what="psi")
    assert_close(a.probe, b.probe, normwise=1e-5, maxabs=1e-4, what="probe")


@pytest.mark.parametrize("det,S,N,chunk,step", [
    (256, 1, 12, None, 1.0), (256, 2, 7, None, 1.0), (256, 8, 6, None, 1.0),
    (256, 2, 9, 2, 1.0),  # several kernel chunks per minibatch: F(x) formed anew
    (512, 2, 4, None, 1.0), (128, 1, 10, None, 1.0), (128, 3, 6, None, 1.0),
    (128, 2, 10, 3, 1.0), (512, 1, 5, 2, 1.0),
    # first step far too long: accepted in the SECOND pass of 8 step lengths
    (256, 2, 7, None, 1024.0), (256, 2, 9, 2, 1024.0), (128, 1, 10, None, 1024.0),
    # ... beyond all 16: the trial-by-trial search (30 slots) takes the call over
    (256, 1, 8, None, 1e6)])
def test_cgrad_all_steps_at_once_equals_trial_by_trial(tp, monkeypatch, det, S,
                                                        N, chunk, step):
    """tike_cgrad_line_search_linear (the far plane is linear in the variable
    a search moves along: one forward pass of the direction, the costs of 16
    step lengths from one pass over two hand-offs) takes the decisions of the
    trial-by-trial searches -- the device one (tike_cgrad_line_search) and
    opt.line_search on the host (opt.py:216-278): same costs, same iterates,
    up to float32 rounding."""
    import importlib
    C = importlib.import_module("tike_amd.ptycho.solvers.cgrad")
    L = importlib.import_module("tike_amd.ptycho.solvers.lstsq")
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=det + 5 * S, eigen=False)
    if chunk:
        monkeypatch.setattr(L, "CHUNK_POSITIONS_OVERRIDE", chunk)
    monkeypatch.setattr(C, "USE_GRAPHS", False)
    results, used = [], []
    real = C._cg_device

    def spy(*a, **k):
        r = real(*a, **k)
        used.append((bool(k.get("linear")), r is not None))
        return r

    monkeypatch.setattr(C, "_cg_device", spy)
    for linear, on_device in ((True, True), (False, True), (False, False)):
        monkeypatch.setattr(C, "LINEAR_LINE_SEARCH", linear)
        monkeypatch.setattr(C, "DEVICE_LINE_SEARCH", on_device)
        del used[:]
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=tp.CgradOptions(num_batch=2, cg_iter=3,
                                              num_iter=2, step_length=step,
                                              batch_method="compact"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=np.array_split(np.arange(N), 2)) as ctx:
            ctx.iterate(2)
            results.append(ctx.get_result())
        if linear and step < 1e5:
            # every one of the 8 CG calls found its steps among the 16
            assert used == [(True, True)] * 8, used
        elif linear:
            # a call whose search ran out is redone by the trial-by-trial search
            assert (True, False) in used and (False, True) in used, used
        elif on_device:
            assert used and not any(lin for lin, _ in used)
    a = results[0]
    # (a first step of 1e6: accepted steps are 1e-5 of it and smaller, the
    # float32 rounding of x + step d and of the costs is felt)
    tol = 2e-5 if step < 1e5 else 3e-4
    for b in results[1:]:
        np.testing.assert_allclose(np.array(a.algorithm_options.costs),
                                   np.array(b.algorithm_options.costs),
                                   rtol=tol)
        assert_close(a.psi, b.psi, normwise=tol, maxabs=10 * tol, what="psi")
        assert_close(a.probe, b.probe, normwise=tol, maxabs=10 * tol,
                     what="probe")


@pytest.mark.parametrize("det,S,N,chunk", [(256, 2, 9, None), (256, 2, 9, 4),
                                           (128, 1, 10, None)])
def test_cgrad_staged_search_equals_the_one_call_search(tp, monkeypatch, det,
                                                        S, N, chunk):
    """Several ranks run the all-at-once search in four stages (cost pass ->
    [all-reduce of the row sums] -> decision, twice): on one rank, where the
    all-reduce is the identity, the stages must take the decisions of the
    one-call search (stage 0) -- same step lengths, costs and iterates."""
    import importlib
    import socket
    import torch.distributed as dist
    C = importlib.import_module("tike_amd.ptycho.solvers.cgrad")
    L = importlib.import_module("tike_amd.ptycho.solvers.lstsq")
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=det + 7 * S, eigen=False)
    if chunk:
        monkeypatch.setattr(L, "CHUNK_POSITIONS_OVERRIDE", chunk)
    monkeypatch.setattr(C, "USE_GRAPHS", False)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    results = []
    for staged in (False, True):
        if staged:
            # a one-rank process group whose collectives are really issued
            # (TIKE_FORCE_COLLECTIVES): every all-reduce is the identity
            monkeypatch.setenv("TIKE_FORCE_COLLECTIVES", "1")
            dist.init_process_group("gloo", rank=0, world_size=1,
                                    init_method=f"tcp://127.0.0.1:{port}")
        try:
            results += _staged_runs(tp, data, scan, psi_true, probe0, N)
        finally:
            if staged:
                dist.destroy_process_group()
    for a, b in zip(results[:2], results[2:]):
        np.testing.assert_allclose(np.array(a.algorithm_options.costs),
                                   np.array(b.algorithm_options.costs),
                                   rtol=1e-6)
        assert_close(a.psi, b.psi, normwise=1e-6, maxabs=1e-5, what="psi")
        assert_close(a.probe, b.probe, normwise=1e-6, maxabs=1e-5,
                     what="probe")


def _staged_runs(tp, data, scan, psi_true, probe0, N):
    results = []
    if True:
        for step in (1.0, 1024.0):  # first- and second-pass acceptance
            params = tp.PtychoParameters(
                probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
                scan=scan.copy(),
                algorithm_options=tp.CgradOptions(num_batch=1, cg_iter=3,
                                                  num_iter=2,
                                                  step_length=step),
                probe_options=tp.ProbeOptions(force_orthogonality=True),
                object_options=tp.ObjectOptions())
            with tp.Reconstruction(data, params, order=np.arange(N),
                                   batches=[np.arange(N)]) as ctx:
                ctx.iterate(2)
                results.append(ctx.get_result())
    return results


@pytest.mark.parametrize("det,S,N", [(128, 1, 10), (256, 2, 7)])
def test_cgrad_graph_replay_equals_eager_launches(tp, monkeypatch, det, S, N):
    """From its second occurrence on a CG call is replayed from a captured HIP
    graph (`_CgGraph`): four epochs with replay == four epochs launched one by
    one, and the graphs were really used."""
    import importlib
    C = importlib.import_module("tike_amd.ptycho.solvers.cgrad")
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=det + 5 * S, eigen=False)
    results, replays = [], []
    for graphs in (True, False):
        monkeypatch.setattr(C, "USE_GRAPHS", graphs)
        count = [0]
        real = C._CgGraph.__call__

        def counted(self, x, other):
            count[0] += 1
            return real(self, x, other)

        monkeypatch.setattr(C._CgGraph, "__call__", counted)
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=tp.CgradOptions(num_batch=2, cg_iter=3,
                                              batch_method="compact"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=np.array_split(np.arange(N), 2)) as ctx:
            ctx.iterate(4)
            results.append(ctx.get_result())
        monkeypatch.setattr(C._CgGraph, "__call__", real)
        replays.append(count[0])
    # 4 epochs x 2 minibatches x (object, probe) = 16 calls; a call is eager
    # the first time its (minibatch, variable, slot counts) occurs
    assert replays[0] >= 6 and replays[1] == 0, replays
    a, b = results
    np.testing.assert_allclose(np.array(a.algorithm_options.costs),
                               np.array(b.algorithm_options.costs), rtol=1e-5)
    assert_close(a.psi, b.psi, normwise=1e-5, maxabs=1e-4, what="psi")
    assert_close(a.probe, b.probe, normwise=1e-5, maxabs=1e-4, what="probe")


@pytest.mark.parametrize("planar", [True, False])
def test_cgrad_direction_entry_equals_direction_dy(tp, planar):
    """tike_cgrad_direction (negated update -> gradient, the Dai-Yuan sums and
    the new direction in two kernels, + the mean cost on the first iteration)
    against tike_amd.opt.direction_dy, which is pinned to the reference's
    (tests/test_host_golden_cpu.py)."""
    import torch
    import tike_amd._arrays as A
    from tike_amd import opt
    from tike_amd._lib import check, lib
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device="cpu").manual_seed(5)
    n = 70 * 93
    c64 = lambda: torch.randn(n, 2, generator=gen).to(dev)
    ups = [c64() for _ in range(3)]  # accumulated updates of three iterations
    costs = torch.rand(1000, generator=gen).to(dev)
    count = 1000.0
    gradient = torch.empty(n, dtype=torch.complex64, device=dev)
    direction = torch.empty(n, dtype=torch.complex64, device=dev)
    state = torch.zeros(5, dtype=torch.float64, device=dev)
    sums = torch.empty(4, dtype=torch.float64, device=dev)
    g0 = d0 = None
    for i, up in enumerate(ups):
        if planar:
            buf = up.t().contiguous()  # (2, n): real plane, imaginary plane
            args = (A.ptr(buf), None)
        else:
            buf = torch.view_as_complex(up.contiguous())
            args = (None, A.ptr(buf))
        check(lib.tike_cgrad_direction(*args, A.ptr(gradient), A.ptr(direction),
                                       n, int(i == 0), A.ptr(costs), 1000,
                                       count, A.ptr(state), A.ptr(sums),
                                       A.stream_ptr()), "direction")
        g1 = -torch.view_as_complex(up.contiguous())
        want = (opt.direction_dy(torch, [g1]) if i == 0 else
                opt.direction_dy(torch, [g1], [g0], [d0]))[0]
        assert_close(gradient.cpu().numpy(), g1.cpu().numpy(), normwise=1e-7,
                     what="gradient")
        assert_close(direction.cpu().numpy(), want.cpu().numpy(),
                     normwise=2e-6, maxabs=2e-5, what=f"direction {i}")
        g0, d0 = g1, direction.clone()
    assert float(state[0]) == pytest.approx(
        float(costs.sum(dtype=torch.float64)) / count, rel=1e-12)
    # exactly one form of the update must be given
    assert lib.tike_cgrad_direction(None, None, A.ptr(gradient),
                                    A.ptr(direction), n, 1, None, 0, 0.0, None,
                                    A.ptr(sums), A.stream_ptr()) == 1000001


def test_lstsq_converges_and_resumes(tp):
    """Cost decreases monotonically on a clean synthetic problem and state
    round-trips through PtychoParameters (larger, pow-2 sizes)."""
    rng = np.random.default_rng(0)
    N, S, pw = 100, 2, 64
    side = 10
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side),
                              indexing="ij"), -1).reshape(-1, 2)
    scan = (2 + 6.0 * ij + rng.random((N, 2))).astype(np.float32)
    HW = 6 * (side - 1) + pw + 8
    psi_true = ((0.75 + 0.25 * rng.random((1, HW, HW))) * np.exp(
        1j * np.pi * (rng.random((1, HW, HW)) - 0.5))).astype(np.complex64)
    w = tp.gaussian(pw, rin=0.6)
    probe = np.stack([w * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
                      for m in range(S)])[None, None].astype(np.complex64)
    data = tp.simulate(pw, probe, scan, psi_true)
    params = tp.PtychoParameters(
        probe=probe * (1 + 0.05 * rng.standard_normal(probe.shape)).astype(
            np.float32), psi=np.full_like(psi_true, 0.5), scan=scan,
        algorithm_options=tp.LstsqOptions(num_batch=4, num_iter=4,
                                          batch_method="compact"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions())
    np.random.seed(0)
    params = tp.reconstruct(data, params)
    np.random.seed(0)
    params = tp.reconstruct(data, params)
    costs = [c[0] for c in params.algorithm_options.costs]
    assert len(costs) == 8 and len(params.algorithm_options.times) == 8
    assert all(np.isfinite(costs))
    assert costs[-1] < 0.5 * costs[0], costs


def _oracle_state(g_psi, g_probe, scan, order):
    return dict(psi=g_psi.copy(), probe=g_probe.copy(),
                scan=scan[order].copy(), costs=[], eigen_probe=None,
                eigen_weights=None)


@pytest.mark.parametrize("det,pw,S,N,num_batch,masked,model", [
    (128, 128, 2, 37, 3, True, "gaussian"),   # position-major kernels, ragged batches, mask
    (128, 96, 3, 30, 2, False, "gaussian"),   # zero-padded probe window (pw < det)
    (64, 64, 1, 9, 4, True, "gaussian"),      # v1 engine path, tiny batches (2-3 positions)
    (48, 32, 2, 12, 1, False, "gaussian"),    # non-power-of-two detector: generic DFT path
    (128, 128, 3, 24, 2, True, "poisson:all_modes"),      # fused inverse with per-mode steps
    (256, 256, 2, 10, 1, False, "poisson:dominant_mode"),
    (48, 32, 2, 12, 2, True, "poisson:all_modes"),        # generic path: tike_scale_modes
    (512, 512, 2, 6, 2, True, "gaussian"),                # config-5 size: 512^2 position-major
    (256, 256, 3, 8, 2, True, "poisson:all_modes"),       # 256^2: per-mode steps from the hand-off
    (512, 512, 4, 5, 1, False, "poisson:dominant_mode"),
    (256, 256, 8, 6, 1, False, "poisson:all_modes"),      # ... at the headline mode count
    (512, 512, 2, 5, 1, True, "poisson:all_modes"),       # ... and at 512^2
    (256, 256, 3, 8, 2, True, "poisson:all_modes:kept"),  # the stored-far-plane pipeline
    (256, 256, 3, 9, 2, False, "poisson:all_modes"),      # every pixel measured: steps applied by pass 2
    (256, 256, 5, 7, 1, False, "poisson:all_modes:sweeps"),  # ... and its three-read form
    (256, 256, 6, 5, 1, False, "poisson:all_modes"),      # resident sweeps, three modes per half
    (256, 256, 7, 4, 2, False, "poisson:all_modes"),      # ... four, the last one empty
    (256, 256, 8, 5, 1, True, "poisson:all_modes:unit"),  # a mask with the default unmeasured scaling 1:
    (256, 256, 3, 7, 2, True, "poisson:all_modes:unit"),  # still linear in the steps (resident / two sweeps)
])
def test_lstsq_epochs_vs_oracle(tp, det, pw, S, N, num_batch, masked, model):
    """Two lstsq_grad epochs against the CPU oracle on seeded problems that
    exercise ragged minibatches, unmeasured pixels holding NaN (reference
    tests/ptycho/test_ptycho.py:334,553), pw < det and every FFT path."""
    from oracle import solvers as osol
    import importlib
    import pytest as _pytest
    L = importlib.import_module("tike_amd.ptycho.solvers.lstsq")
    mp = _pytest.MonkeyPatch()
    request_cleanup = mp.undo
    # tags behind the model: the pipeline to force / the unmeasured scaling
    unmeasured_scaling = 0.9
    model, *tags = model.split(":")[0:1] + model.split(":")[1:]
    usemodes = tags.pop(0) if tags and tags[0] in ("all_modes",
                                                   "dominant_mode") else None
    for tag in tags:
        if tag == "kept":
            mp.setattr(L, "POISSON_FROM_HANDOFF", False)
        elif tag == "sweeps":
            mp.setattr(L, "POISSON_STEPS_IN_PASS2", False)
        elif tag == "unit":
            unmeasured_scaling = 1.0
        else:
            raise AssertionError(tag)
    model = model if usemodes is None else f"{model}:{usemodes}"
    rng = np.random.default_rng(det * 7 + N)
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side),
                              indexing="ij"), -1).reshape(-1, 2)[:N]
    scan = (2 + 5.0 * ij + rng.random((N, 2))).astype(np.float32)
    HW = 5 * (side - 1) + pw + 8
    psi_true = ((0.75 + 0.25 * rng.random((1, HW, HW))) * np.exp(
        1j * np.pi * (rng.random((1, HW, HW)) - 0.5))).astype(np.complex64)
    w = tp.gaussian(pw, rin=0.6)
    probe = np.stack([w * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
                      for m in range(S)])[None, None].astype(np.complex64)
    data = tp.simulate(det, probe, scan, psi_true)
    mask = np.ones((det, det), dtype=bool)
    if masked:
        mask = rng.random((det, det)) > 0.15
        data[:, ~mask] = np.nan
    psi0 = np.full_like(psi_true, 0.5)
    probe0 = (probe * (1 + 0.05 * rng.standard_normal(probe.shape))).astype(
        np.complex64)
    batches = np.array_split(np.arange(N), num_batch)
    order = np.arange(N)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=num_batch, num_iter=2,
                                          batch_method="compact"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=mask,
            unmeasured_pixels_scaling=unmeasured_scaling,
            noise_model=model.split(":")[0],
            step_length_usemodes=(model.split(":") + ["all_modes"])[1]))
    import warnings
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # NaN in data warns, as the reference
            with tp.Reconstruction(data, params, order=order,
                                   batches=batches) as ctx:
                ctx.iterate(2)
                got = ctx.get_result()
    finally:
        request_cleanup()
    state = _oracle_state(psi0, probe0, scan, order)
    odata = np.where(mask, np.nan_to_num(data), 0).astype(np.float32)
    state = osol.rescale_probe(state, odata, det, measured_pixels=mask)
    state = osol.iterate(state, odata, batches, 2, detector_shape=det,
                         batch_method="compact", force_orthogonality=True,
                         measured_pixels=mask,
                         unmeasured_pixels_scaling=unmeasured_scaling,
                         noise_model=model.split(":")[0],
                         step_length_usemodes=(model.split(":") +
                                               ["all_modes"])[1])
    # (the poisson cost mean(I - d log I) cancels heavily: a looser bound)
    np.testing.assert_allclose(
        np.array(got.algorithm_options.costs), np.array(state["costs"]),
        rtol=2e-3 if model.startswith("poisson") else 1e-3)
    assert_close(got.psi, state["psi"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")


def test_full_size_operator_properties(tp):
    """BASELINE size (256x256, 8 modes): size-independent properties of the
    fused kernels -- adjoint identity <F m, d> = <m, F* d> for psi and probe,
    linearity in psi, and Parseval between chi and the far-plane gradient."""
    import torch
    import tike_amd.operators as ops
    rng = np.random.default_rng(1)
    N, S, pw, det = 96, 8, 256, 256
    HW = 420
    rc = lambda *s: torch.from_numpy((rng.random((*s, 2), dtype=np.float32) -
                                      0.5).view(np.complex64)[..., 0]).cuda()
    scan = torch.from_numpy((rng.random((N, 2)) * (HW - pw - 3) + 1).astype(
        np.float32)).cuda()
    psi, psi2 = rc(1, HW, HW), rc(1, HW, HW)
    probe = rc(1, 1, S, pw, pw)
    d = rc(N, 1, S, det, det)
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        Fm = op.fwd(probe=probe, scan=scan, psi=psi)
        Fm2 = op.fwd(probe=probe, scan=scan, psi=psi2)
        Fsum = op.fwd(probe=probe, scan=scan, psi=psi + 2 * psi2)
        assert float((Fsum - Fm - 2 * Fm2).abs().max()) < 1e-3 * float(
            Fsum.abs().max())
        bprobe = probe.expand(N, 1, S, pw, pw).contiguous()
        m0, m1 = op.adj(farplane=d, probe=bprobe, scan=scan, psi=psi)
    a = (Fm * d.conj()).sum()
    b = (psi * m0.conj()).sum()
    c = (bprobe * m1.conj()).sum()
    for x in (b, c):
        np.testing.assert_allclose([float(a.real), float(a.imag)],
                                   [float(x.real), float(x.imag)], rtol=1e-3)
    # Parseval: ||F m||^2 = ||pad(patch * probe)||^2 (ortho norm)
    near = (Fm.abs()**2).sum()
    with ops.Propagation(detector_shape=det) as prop:
        back = prop.adj(farplane=Fm)
    np.testing.assert_allclose(float(near), float((back.abs()**2).sum()),
                               rtol=1e-4)


def test_reconstruct_multigrid_vs_reference(tp, golden):
    """Two-level coarse-to-fine reconstruction (ptycho.py:975-1047) against
    the reference's own run: costs of all four epochs, final object, probe."""
    import tike_amd.random
    import warnings
    g = golden("multigrid_fft.npz")
    np.random.seed(7)
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    params = tp.PtychoParameters(
        probe=g["probe0"].copy(), psi=g["psi0"].copy(), scan=g["scan"].copy(),
        algorithm_options=tp.LstsqOptions(num_batch=2, batch_method="compact",
                                          num_iter=2),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((32, 32), dtype=bool)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # "< 64 pixels wide" at this test size
        r = tp.reconstruct_multigrid(g["data"], params, num_levels=2)
    np.testing.assert_allclose(np.array(r.algorithm_options.costs), g["costs"],
                               rtol=2e-3)
    assert_close(r.psi, g["psi"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(r.probe, g["probe"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="probe")


def _headline_problem(tp, det, S, N, seed, eigen, pitch=7.0, margin=8,
                      pw=None):
    """Small problem with the shapes of BASELINE configs[2] / [4] (pw: a
    probe window narrower than the detector)."""
    import tike_amd.random
    rng = np.random.default_rng(seed)
    pw = pw or det
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side),
                              indexing="ij"), -1).reshape(-1, 2)[:N]
    scan = (2 + (margin - 8) // 2 + pitch * ij +
            rng.random((N, 2))).astype(np.float32)
    HW = int(pitch * (side - 1)) + pw + margin
    psi_true = ((0.75 + 0.25 * rng.random((1, HW, HW))) * np.exp(
        1j * np.pi * (rng.random((1, HW, HW)) - 0.5))).astype(np.complex64)
    w = tp.gaussian(pw, rin=0.6)
    probe = np.stack([w * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
                      for m in range(S)])[None, None].astype(np.complex64)
    ep = ew = None
    if eigen:
        # (eigen = (probes incl. the shared one, modes owning eigen probes),
        # True: (2, 1) -- one eigen probe on the first mode)
        K, Mm = eigen if isinstance(eigen, tuple) else (2, 1)
        np.random.seed(seed)
        tike_amd.random.randomizer_np = np.random.default_rng(seed + 1)
        ep, ew = tp.init_varying_probe(scan, probe, num_eigen_probes=K,
                                       probes_with_modes=Mm)
        # weights large enough for the eigen probe to matter in the forward
        ew[:, 1, 0] = 0.05 * rng.standard_normal(N).astype(np.float32)
        if (K, Mm) != (2, 1):
            ew[:, 1:, :Mm] = 0.05 * rng.standard_normal(
                ew[:, 1:, :Mm].shape).astype(np.float32)
    data = tp.simulate(det, probe, scan, psi_true, eigen_probe=ep,
                       eigen_weights=ew)
    probe0 = (probe * (1 + 0.05 * rng.standard_normal(probe.shape))).astype(
        np.complex64)
    return scan, psi_true, probe0, ep, ew, data


@pytest.mark.parametrize("spatial_sort,batch_method", [
    (False, "compact"), (True, "compact"),
    # the update rule bench.py runs since round 4 (an object update after
    # every minibatch, minibatches in a random order, lstsq.py:134-205)
    (True, "wobbly_center")])
@pytest.mark.parametrize("tag,det,S,N,num_batch", [
    ("c3", 256, 8, 20, 2),  # BASELINE configs[2]: 8 modes + eigen probe, far-plane-free
    ("c3-4modes", 256, 4, 12, 2),
    ("c5", 512, 4, 8, 2),   # BASELINE configs[4]: 512^2, 4 modes, position correction
    # round 6, off-grid detector sizes through the whole solver: 192 = 3 x 64
    # (prime-factor launches), 300 = 20 x 15 (LDS line engine), 100 (below
    # GENERAL_MIN_DETECTOR: the unfused kernels on the new transforms), all
    # with eigen probes
    ("off-grid pfa", 192, 3, 12, 2),
    ("off-grid lds", 300, 2, 8, 2),
    ("off-grid unfused", 100, 2, 10, 2),
])
def test_lstsq_headline_shapes_vs_oracle(tp, tag, det, S, N, num_batch,
                                         spatial_sort, batch_method,
                                         monkeypatch):
    """The code path bench.py times (c3: 256^2, S = 8, one eigen probe,
    several minibatches; c5: 512^2, S = 4, position correction with ADAM and
    affine regularisation): two epochs against the CPU oracle, under both
    object-update rules."""
    import tike_amd.random
    from oracle import solvers as osol
    eigen = tag.startswith(("c3", "off-grid"))
    positions = tag == "c5"
    if tag.startswith("off-grid"):
        from tike_amd.ptycho.solvers import lstsq as L
        from tike_amd.ptycho.solvers._plan import GradientPlan
        assert L.pfa_gradients(S, det, det) == (tag == "off-grid pfa")
        assert L.general_gradients(S, det, det)
        # (the LDS line engine is no longer anybody's default route: this case
        # asks for it from 256 pixels a side, round 6's first rule)
        monkeypatch.setattr(L, "GENERAL_MIN_DETECTOR", 256)
        seen = []
        real = GradientPlan.gradients
        monkeypatch.setattr(GradientPlan, "gradients",
                            lambda self, c, k: (seen.append(self.route),
                                                real(self, c, k))[1])
    # (position correction under the per-minibatch rule moves the corner
    # positions further: more room around the scan, or check_allowed_positions
    # -- the reference's rule -- stops the run)
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=det + S, eigen=eigen,
        margin=24 if positions and batch_method != "compact" else 8)
    psi0 = np.full_like(psi_true, 0.5)
    batches = np.array_split(np.arange(N), num_batch)
    order = np.arange(N)
    popts = dict(use_adaptive_moment=True, update_magnitude_limit=1.0,
                 use_position_regularization=True)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        eigen_probe=None if ep is None else ep.copy(),
        eigen_weights=None if ew is None else ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=num_batch, num_iter=2,
                                          batch_method=batch_method),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        position_options=tp.PositionOptions(scan.copy(), **popts)
        if positions else None,
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    # spatial_sort=True is the default and what the bench runs: the positions
    # of a minibatch are re-listed along a space-filling curve (membership,
    # and therefore every sum over the minibatch, unchanged)
    with tp.Reconstruction(data, params, order=order, batches=batches,
                           spatial_sort=spatial_sort) as ctx:
        ctx.iterate(2)
        got = ctx.get_result()
    if tag.startswith("off-grid"):
        assert set(seen) == {dict(pfa="pfa", lds="general",
                                  unfused="unfused")[tag.split()[-1]]}
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=None if ep is None else ep.copy(),
                 eigen_weights=None if ew is None else ew.copy())
    if positions:
        state["position"] = dict(
            initial_scan=scan.copy(),
            momentum=np.zeros((N, 4), dtype=np.float32), **popts)
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, 2, detector_shape=det,
                         batch_method=batch_method, force_orthogonality=True,
                         rng=np.random.default_rng(11))
    np.testing.assert_allclose(
        np.array(got.algorithm_options.costs), np.array(state["costs"]),
        rtol=1e-3)
    assert_close(got.psi, state["psi"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")
    if eigen:
        # (5e-3 with float atomics; the solver tolerance under the
        # deterministic switch, test_fixtures_under_the_deterministic_switch)
        from tike_amd import _lib
        tol = dict(normwise=SOLVER_NORMWISE, maxabs=1e-2) if (
            _lib.DETERMINISTIC) else dict(normwise=5e-3, maxabs=5e-2)
        assert_close(got.eigen_probe, state["eigen_probe"],
                     what="eigen_probe", **tol)
        assert_close(got.eigen_weights, state["eigen_weights"],
                     what="eigen_weights", **tol)
    if positions:
        assert np.abs(got.scan - scan).max() > 0.02  # positions did move
        np.testing.assert_allclose(got.scan, state["scan"], atol=5e-3)


def test_bench_c3_one_epoch_vs_oracle(tp):
    """ONE epoch of bench.py's own c3 problem -- its generator (SURVEY 8(d):
    ramp modes, shuffled raster), its eigen-probe set-up, its update rule, its
    presharded Reconstruction -- at 200 positions against the oracle's epoch.
    (From the second epoch on, product and oracle can part by several per cent
    of the cost on these inputs WITH eigen probes -- and so does the oracle
    from its own state perturbed by 4e-6: `orthogonalize_eig` takes LAPACK's
    eigenvectors, whose phase convention makes the last component real, and
    for the dominant mode of the ramp modes that component is 8e-7 of the
    vector (tests/test_oracle_sensitivity_cpu.py,
    profiles/r06_oracle_sensitivity.txt; tools/eigen_diag.py: with the
    oracle's own simulated patterns the two agree to 1e-3 for three epochs,
    with the product's they flip in the second).  The first epoch is what can
    be, and is, pinned -- eigen probe included.)"""
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench
    import tike_amd._arrays as A
    import tike_amd.random
    from oracle import solvers as osol
    built = bench.epoch_problem("c3", 200, 1, 0, tp, A)
    ctx, p, det = built["ctx"], built["p"], built["det"]
    assert built["S"] == 8 and det == 256 and built["num_batch"] == 10
    assert built["eigen_probe"] is not None and built["C"] == 1
    N = built["N"]
    batches = np.array_split(np.arange(N), built["num_batch"])
    psi0 = np.full_like(p["psi"], 0.5 + 0j)
    state = dict(psi=psi0.copy(), probe=p["probe"].copy(),
                 scan=p["scan"].copy(), costs=[],
                 eigen_probe=built["eigen_probe"].copy(),
                 eigen_weights=built["eigen_weights"].copy())
    try:
        tike_amd.random.randomizer_np = np.random.default_rng(11)
        ctx.iterate(1)
        got = ctx.get_result()
    finally:
        ctx.__exit__(None, None, None)
    state = osol.rescale_probe(state, built["data"], det)
    state = osol.iterate(state, built["data"], batches, 1, detector_shape=det,
                         batch_method=bench.BATCH_RULE,
                         force_orthogonality=True,
                         rng=np.random.default_rng(11))
    np.testing.assert_allclose(
        np.array(got.algorithm_options.costs), np.array(state["costs"]),
        rtol=1e-3)
    assert_close(got.psi, state["psi"], normwise=1e-3, maxabs=1e-2, what="psi")
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")
    from tike_amd import _lib
    tol = dict(normwise=SOLVER_NORMWISE, maxabs=1e-2) if _lib.DETERMINISTIC else (
        dict(normwise=5e-3, maxabs=5e-2))
    assert_close(got.eigen_weights, state["eigen_weights"],
                 what="eigen_weights", **tol)
    assert_close(got.eigen_probe, state["eigen_probe"], what="eigen_probe",
                 **tol)


def test_fixtures_under_the_deterministic_switch():
    """The reference-run fixtures and the oracle comparisons at the headline
    shapes once more in a process with TIKE_DETERMINISTIC=1, where the eigen
    quantities and the second ReconstructTwice call are held to the solver
    tolerance 1e-3 (with float atomics: 5e-3) -- the tolerances are
    parametrised on the switch, not loosened."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, TIKE_DETERMINISTIC="1")
    env.pop("TIKE_CHUNK_POSITIONS", None)
    out = subprocess.run(
        [sys.executable, "-m", "pytest", os.path.join(here,
                                                      "test_solvers_gpu.py"),
         "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "-k",
         "reconstruct_twice_vs_reference or headline_shapes_vs_oracle or "
         "bench_c3_one_epoch"],
        capture_output=True, text=True, env=env, timeout=1500)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert " passed" in out.stdout and "failed" not in out.stdout


@pytest.mark.parametrize("det,S,N,eigen", [(256, 8, 9, True), (256, 4, 7, True),
                                           (512, 4, 5, False),
                                           (128, 8, 11, True),
                                           (256, 5, 6, False),
                                           # one / two modes WITH eigen probes:
                                           # their LDS slices need spare
                                           # mode-waves at 512^2 (raised
                                           # "unsupported size" until round 6,
                                           # found by tools/fuzz_routes.py)
                                           (512, 1, 4, True), (512, 2, 4, True),
                                           (256, 1, 6, True), (128, 1, 7, True)])
def test_lstsq_minibatch_kernels_vs_oracle(tp, det, S, N, eigen):
    """One minibatch through the HIP kernels at the headline shapes (every
    specialisation on the number of modes the bench runs: S = 8 at 256^2,
    S = 4 at 512^2) against the oracle's _get_nearplane_gradients /
    _precondition_nearplane_gradients / _update_nearplane."""
    _minibatch_vs_oracle(tp, det, S, N, eigen)


@pytest.mark.parametrize("det,pw,S,N,eigen", [
    (256, 128, 8, 9, True),    # bench c3pad: probe window = half the detector
    (256, 256, 12, 7, True),   # bench c3m12: more modes than the fused kernels take
    (384, 384, 4, 5, False),   # bench c384: a detector size with a factor 3
    (384, 192, 3, 5, True),
    (96, 64, 2, 11, True), (160, 100, 3, 6, False),  # radix 5; odd padding
    (45, 31, 2, 13, True),     # 3 * 3 * 5, odd window, odd padding
    (64, 64, 16, 6, True),     # a power of two the fused kernels do not serve
    (1024, 1024, 1, 2, False), (640, 512, 2, 3, False),
    (768, 768, 2, 2, False), (192, 192, 9, 4, True), (320, 300, 2, 5, True),
    (448, 448, 2, 3, True), (224, 200, 3, 4, False),  # 7 x 2^k
    (896, 896, 1, 2, False),   # 7 x 128: sub-tiles gathered and transformed in LDS
    (1536, 1536, 1, 2, False),  # 3 x 512 sub-tiles
    (2048, 1500, 2, 2, False),  # the largest transform size
])
@pytest.mark.parametrize("engine", ["pfa", "lds"])
def test_general_shape_launches_vs_oracle(tp, det, pw, S, N, eigen, engine,
                                          monkeypatch):
    """The three shape-general launches (csrc/general.hip: tike_gen_fwd_rows ->
    tike_gen_cols_gradient -> tike_gen_inv_rows_gradients) on shapes the fused
    power-of-two kernels refuse -- probe window < detector, 12 and 16 modes,
    detector sizes with factors 3 and 5, odd windows -- against the oracle's
    minibatch (gradients, chi0, patches, costs, position sums, step lengths,
    eigen update, packed tail).  The reference runs all of them through cuFFT
    at one speed (ptycho/solvers/lstsq.py:422-579)."""
    from tike_amd.ptycho.solvers import lstsq as L
    assert not L.fused_gradients(S, pw, det)
    assert L.general_gradients(S, pw, det)
    monkeypatch.setattr(L, "MODE_GROUPS", False)  # (9 ... 16 modes: not in groups here)
    # engine: the prime-factor launches (csrc/pfa.hip: detector sizes 3 x 2^k
    # and 5 x 2^k on the power-of-two register engine) or the LDS line engine
    # (csrc/general.hip) -- the same shapes through both where both serve
    if engine == "pfa" and not L.pfa_gradients(S, pw, det):
        pytest.skip("no prime-factor decomposition of this detector size")
    saved = L.GENERAL_FUSED, L.PFA_ROUTE
    L.GENERAL_FUSED = "always"  # also where position-major kernels exist
    L.PFA_ROUTE = engine == "pfa"
    try:
        _minibatch_vs_oracle(tp, det, S, N, eigen, pw=pw)
        from tike_amd.operators import Ptycho  # the plan really took the route
    finally:
        L.GENERAL_FUSED, L.PFA_ROUTE = saved


@pytest.mark.parametrize("det,S,N,eigen", [(256, 12, 7, True),   # bench c3m12
                                           (256, 9, 6, False), (256, 16, 5, True),
                                           (128, 10, 9, True), (128, 13, 6, False),
                                           (256, 19, 4, True),   # three groups
                                           (512, 5, 4, True), (512, 8, 3, False),
                                           (512, 11, 2, False)])
def test_mode_groups_vs_oracle(tp, det, S, N, eigen):
    """9 ... 16 modes at 128^2 / 256^2 on the far-plane-free kernels with the
    inverse's second pass in two groups of modes
    (`tike_ifft2_pass2_gradients_modes`: the second group adds its share of the
    object projection to the first's): the oracle's minibatch
    (ptycho/solvers/lstsq.py:422-579), and the plan really took that route."""
    from tike_amd.ptycho.solvers import lstsq as L
    from tike_amd.ptycho.solvers._plan import GradientPlan
    assert not L.fused_gradients(S, det, det)
    groups = L.mode_groups(S, det, det, 1)
    cap = 4 if det == 512 else 8
    assert len(groups) == -(-S // cap) and sum(c for _, c in groups) == S
    assert all(2 <= c <= cap for _, c in groups)
    calls = []
    real = GradientPlan.gradients

    def spy(self, c, k):
        calls.append((self.route, self.groups))
        return real(self, c, k)

    GradientPlan.gradients = spy
    try:
        _minibatch_vs_oracle(tp, det, S, N, eigen)
    finally:
        GradientPlan.gradients = real
    assert calls and all(g == groups for _, g in calls)
    assert all(r == ("pos_major" if det == 128 else "no_farplane")
               for r, _ in calls)


@pytest.mark.parametrize("det,S,N,eigen", [
    (256, 4, 8, (3, 2)),    # 2 eigen probes x 2 modes: 4 slices fit pass 2's LDS
    (256, 6, 6, (3, 3)),    # 6 do not: chi is stored (found by tools/fuzz_routes.py:
    (512, 3, 4, (3, 3)),    # "unsupported size" until round 6)
    (128, 8, 8, (3, 3)),
    (256, 12, 6, (3, 2)),   # ... with the modes in two groups
    (192, 3, 8, (3, 2)), (300, 2, 6, (3, 2))])  # prime-factor / LDS line engine
def test_several_eigen_probes_vs_oracle(tp, det, S, N, eigen):
    """More than one eigen probe, on more than one mode (probe.py:272-476,
    lstsq.py:297-364): two epochs against the oracle on every route."""
    import tike_amd.random
    from oracle import solvers as osol
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=det + S, eigen=eigen)
    assert ep.shape[-4] == eigen[0] - 1 and ep.shape[-3] == eigen[1]
    psi0 = np.full_like(psi_true, 0.5)
    batches = np.array_split(np.arange(N), 2)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        eigen_probe=ep.copy(), eigen_weights=ew.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2,
                                          batch_method="compact"),
        probe_options=tp.ProbeOptions(force_orthogonality=False),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    tike_amd.random.randomizer_np = np.random.default_rng(11)
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=batches) as ctx:
        ctx.iterate(2)
        got = ctx.get_result()
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=ep.copy(), eigen_weights=ew.copy())
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, 2, detector_shape=det,
                         batch_method="compact", force_orthogonality=False,
                         rng=np.random.default_rng(11))
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(state["costs"]), rtol=1e-3)
    assert_close(got.psi, state["psi"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")
    assert_close(got.eigen_probe, state["eigen_probe"], normwise=5e-3,
                 maxabs=5e-2, what="eigen probes")
    np.testing.assert_allclose(got.eigen_weights, state["eigen_weights"],
                               rtol=5e-3, atol=1e-4)


@pytest.mark.parametrize("det,S,N", [(256, 10, 8), (128, 12, 9), (256, 6, 8)])
def test_probe_only_reconstruction_vs_oracle(tp, det, S, N):
    """`object_options=None` (the object is not recovered, lstsq.py:383-420
    with recover_psi False): no object projection is formed -- with more
    than 8 modes the groups of `tike_ifft2_pass2_gradients_modes` run without
    it -- two epochs against the oracle."""
    from oracle import solvers as osol
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=det + S, eigen=False)
    batches = np.array_split(np.arange(N), 2)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi_true.copy(), scan=scan.copy(),
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2,
                                          batch_method="compact"),
        # (no orthogonalisation: with the object at its true value the modes'
        # powers come close and the eigenbasis of `orthogonalize_eig` is then
        # a matter of rounding -- product and oracle pick different ones)
        probe_options=tp.ProbeOptions(force_orthogonality=False),
        object_options=None,
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    with tp.Reconstruction(data, params, order=np.arange(N),
                           batches=batches) as ctx:
        ctx.iterate(2)
        got = ctx.get_result()
    state = dict(psi=psi_true.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=None, eigen_weights=None)
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, 2, detector_shape=det,
                         batch_method="compact", force_orthogonality=False,
                         recover_psi=False)
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(state["costs"]), rtol=1e-3)
    np.testing.assert_array_equal(got.psi, psi_true)
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")


def test_general_shape_launches_equal_the_unfused_kernels(tp):
    """... and equal the unfused round-1 kernels they replace (stored far
    plane, generic transforms, per-tap gradients) on a masked problem with
    NaN counts at the unmeasured pixels."""
    import tike_amd._arrays as A
    from tike_amd.communicators import Comm
    from tike_amd.operators import Ptycho
    from tike_amd.ptycho.solvers import lstsq as L
    det, pw, S, N = 96, 64, 3, 9
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=11, eigen=True, pw=pw)
    rng = np.random.default_rng(3)
    mask = rng.random((det, det)) > 0.1
    data = data.copy()
    data[:, ~mask] = np.nan
    params = tp.PtychoParameters(
        probe=probe0, psi=psi_true, scan=scan, eigen_probe=ep,
        eigen_weights=ew, algorithm_options=tp.LstsqOptions(num_batch=2),
        exitwave_options=tp.ExitWaveOptions(measured_pixels=mask,
                                            unmeasured_pixels_scaling=0.9))
    d = {k: A.to_device(v) for k, v in dict(psi=psi_true, probe=probe0,
                                            scan=scan, ep=ep, ew=ew).items()}
    data_d = A.to_device(data, np.float32)
    HW = psi_true.shape[-1]
    outs = []
    for general in ("always", False):
        saved = L.GENERAL_FUSED
        L.GENERAL_FUSED = general
        try:
            with Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
                out = L._get_nearplane_gradients(
                    data_d, d["psi"], d["scan"], d["probe"], d["ep"], d["ew"],
                    0, N, Comm(), num_batch=2,
                    exitwave_options=params.exitwave_options, op=op,
                    recover_psi=True, recover_probe=True)
                chi0 = out["chi0"]
                if out["chi_modes"] > 1:
                    chi0 = chi0[:N, 0, 0]
                outs.append(dict(
                    obj=L.object_upd_sum(out).cpu().numpy(),
                    mpu=out["m_probe_update"].cpu().numpy(),
                    chi0=chi0.cpu().numpy().copy(),
                    costs=out["costs"].cpu().numpy().copy()))
        finally:
            L.GENERAL_FUSED = saved
    for k in ("obj", "mpu", "chi0"):
        assert np.isfinite(outs[0][k]).all()
        assert_close(outs[0][k], outs[1][k], normwise=2e-5, what=k)
    np.testing.assert_allclose(outs[0]["costs"], outs[1]["costs"],
                               rtol=COST_RTOL)


def _minibatch_vs_oracle(tp, det, S, N, eigen, pw=None):
    pw = pw or det
    # the cost of these nearly fitted patterns (psi0 = truth x (1 + 10 %
    # noise)) is a mean of squared DIFFERENCES of nearly equal amplitudes,
    # sqrt(I) - sqrt(d) ~ 2 % of sqrt(I): a float32 transform error of 2e-7
    # sqrt(log2 n) in I (product and oracle alike, tools/debug/fft_bias.py) is
    # 1e-5 ... 1e-4 of such a cost and grows with the line length -- 1.5e-4 at
    # 640, 5.5e-4 at 1024 (gradients, chi and step lengths agree to 2e-5 there)
    # ... 1.8e-3 at 1536, 3e-3 at 1792, 4.5e-3 at 2048
    cost_rtol = COST_RTOL if det <= 512 else 1e-3 if det <= 1024 else 6e-3
    import tike_amd._arrays as A
    from tike_amd.communicators import Comm
    from tike_amd.operators import Ptycho
    from tike_amd.ptycho.solvers import lstsq as L
    from oracle import solvers as osol
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=3 * det + S, eigen=eigen, pw=pw)
    rng = np.random.default_rng(5)
    psi0 = (psi_true * (1 + 0.1 * rng.standard_normal(psi_true.shape))
            ).astype(np.complex64)
    mask = np.ones((det, det), dtype=bool)
    o = osol.get_nearplane_gradients(
        data, psi0, scan, probe0, ep, ew, 0, N, num_batch=2,
        detector_shape=det, measured_pixels=mask, recover_positions=True)
    d = {k: (None if v is None else A.to_device(v)) for k, v in dict(
        psi=psi0, probe=probe0, scan=scan, ep=ep, ew=ew).items()}
    data_d = A.to_device(data, np.float32)
    params = tp.PtychoParameters(
        probe=probe0, psi=psi0, scan=scan, eigen_probe=ep, eigen_weights=ew,
        algorithm_options=tp.LstsqOptions(num_batch=2),
        exitwave_options=tp.ExitWaveOptions(measured_pixels=mask))
    comm = Comm()
    HW = psi0.shape[-1]
    with Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        pos_terms = (A.to_device(np.zeros_like(scan)),
                     A.to_device(np.zeros_like(scan)))
        out = L._get_nearplane_gradients(
            data_d, d["psi"], d["scan"], d["probe"], d["ep"], d["ew"], 0, N,
            comm, num_batch=2, exitwave_options=params.exitwave_options,
            op=op, recover_psi=True, recover_probe=True,
            position_terms=pos_terms)
        assert_close(L.object_upd_sum(out).cpu().numpy(), o["object_upd_sum"],
                     normwise=2e-5, what="object_upd_sum")
        assert_close(out["m_probe_update"].cpu().numpy(), o["m_probe_update"],
                     normwise=2e-5, what="m_probe_update")
        chi0 = out["chi0"]
        if out["chi_modes"] > 1:
            chi0 = chi0[:N, 0, 0]
        assert_close(chi0.cpu().numpy(), o["chi"][:, 0, 0], normwise=2e-5,
                     what="chi mode 0")
        np.testing.assert_allclose(out["costs"].cpu().numpy(),
                                   np.ravel(o["costs"]), rtol=cost_rtol)
        assert_close(out["patches"].cpu().numpy(), o["patches"][:, 0, 0],
                     what="patches")
        np.testing.assert_allclose(pos_terms[0].cpu().numpy(),
                                   o["position_numerator"], rtol=2e-3,
                                   atol=1e-4 * np.abs(
                                       o["position_numerator"]).max())
        np.testing.assert_allclose(pos_terms[1].cpu().numpy(),
                                   o["position_denominator"], rtol=2e-3)
        pre = osol.psi_preconditioner(psi0, probe0, scan)
        precond_o, bo_o, bp_o = osol.precondition_nearplane_gradients(
            o["chi"], scan, o["unique_probe"], probe0, o["object_upd_sum"],
            o["m_probe_update"], pre, o["patches"], 0, N)
        precond = L._precondition_object_update(out["object_acc"],
                                                A.to_device(pre))
        assert_close(precond.cpu().numpy(), precond_o, normwise=2e-5,
                     what="object_update_precond")
        stats = L._step_stats(out, d["psi"], d["scan"], d["probe"], d["ep"],
                              precond, 0, N, op=op)
        bo, bp, cost = L._solve_steps(stats, out["costs"], out["count"], comm,
                                      pw=pw, recover_psi=True,
                                      recover_probe=True)
        np.testing.assert_allclose(float(cost), o["costs"].mean(),
                                   rtol=cost_rtol)
        np.testing.assert_allclose(float(bo), np.ravel(bo_o)[0], rtol=1e-3)
        np.testing.assert_allclose(float(bp), np.ravel(bp_o)[0], rtol=1e-3)
        if eigen:
            og = dict(o)
            ep_o, ew_o = osol.update_nearplane(og, probe0, ep.copy(),
                                               ew.copy(), 0, N, num_batch=2)
            ep2, ew2 = L._update_nearplane(out, stats, d["probe"],
                                           d["ep"].clone(), d["ew"].clone(),
                                           0, N, comm, num_batch=2)
            assert_close(ep2.cpu().numpy(), ep_o, normwise=1e-4, maxabs=1e-3,
                         what="eigen_probe")
            assert_close(ew2.cpu().numpy(), ew_o, normwise=1e-4, maxabs=1e-3,
                         what="eigen_weights")
        # the packed tail (what lstsq_grad runs for at most one eigen probe):
        # same statistics, eigen update, step lengths and probe update in five
        # launches, against the same oracle values
        import torch
        ep3 = None if ep is None else d["ep"].clone()
        ew3 = None if ew is None else d["ew"].clone()
        probe3 = d["probe"].clone()
        combined = torch.zeros_like(probe3)
        steps_row = torch.zeros(5, dtype=torch.float32, device=probe3.device)
        norm = None
        if eigen:
            norm = torch.square(ew3[:, 1, 0]).sum().reshape(1).contiguous()
        bo3, bp3 = L._packed_tail(
            out, d["psi"], d["scan"], probe3, ep3, ew3, precond, 0, N, comm,
            op=op, num_batch=2, recover_psi=True, recover_probe=True,
            norm=norm, steps_row=steps_row, probe_combined_update=combined)
        np.testing.assert_allclose(float(bo3), np.ravel(bo_o)[0], rtol=1e-3)
        np.testing.assert_allclose(float(bp3), np.ravel(bp_o)[0], rtol=1e-3)
        np.testing.assert_allclose(float(steps_row[4]), o["costs"].mean(),
                                   rtol=cost_rtol)
        want_probe = probe0 + np.ravel(bp_o)[0] * o["m_probe_update"]
        assert_close(probe3.cpu().numpy(), want_probe, normwise=2e-5,
                     what="probe after the packed tail")
        assert_close(combined.cpu().numpy(), (want_probe - probe0) / 2,
                     normwise=1e-3, what="combined probe update")
        if eigen:
            assert_close(ep3.cpu().numpy(), ep_o, normwise=1e-4, maxabs=1e-3,
                         what="eigen_probe (packed tail)")
            assert_close(ew3.cpu().numpy(), ew_o, normwise=1e-4, maxabs=1e-3,
                         what="eigen_weights (packed tail)")


@pytest.mark.parametrize("u16_and_mask", [False, True])
def test_resident_gradient_kernel_vs_oracle_beyond_one_wave(tp, u16_and_mask):
    """The kernel of record (`fwd_grad_ifft2_pass1_resident_kernel`, what
    `tike_fwd_grad_ifft2_pass1` runs at 256^2 with 8 modes) against the ORACLE
    at a size where its rotated prefetch is live: 64 positions = 1024 work
    items for 256 resident workgroups, so every workgroup requests the rows of
    its next item while it finishes the current one.  (The smaller oracle
    cases launch fewer work items than workgroups; the bench-sized cases
    compare with the two-launch path, not with the oracle.)  Variant: uint16
    counts + a mask with unmeasured pixels that hold garbage."""
    import tike_amd._arrays as A
    from tike_amd.communicators import Comm
    from tike_amd.operators import Ptycho
    from tike_amd.ptycho.solvers import lstsq as L
    from oracle import solvers as osol
    det, S, N = 256, 8, 64
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=77, eigen=True)
    rng = np.random.default_rng(9)
    psi0 = (psi_true * (1 + 0.1 * rng.standard_normal(psi_true.shape))
            ).astype(np.complex64)
    mask = np.ones((det, det), dtype=bool)
    data_in = data
    if u16_and_mask:
        mask = rng.random((det, det)) > 0.1
        data = np.round(data * (20000.0 / data.max())).astype(np.uint16)
        data_in = data.copy()
        data_in[:, ~mask] = 65535  # never read: the mask selects
    o = osol.get_nearplane_gradients(
        data.astype(np.float32), psi0, scan, probe0, ep, ew, 0, N,
        num_batch=1, detector_shape=det, measured_pixels=mask)
    calls = []
    entry = L.lib.tike_fwd_grad_ifft2_pass1

    def counted(*args):
        calls.append(args[6])  # positions in this launch
        return entry(*args)

    L.lib.tike_fwd_grad_ifft2_pass1 = counted
    try:
        HW = psi0.shape[-1]
        with Ptycho(probe_shape=det, detector_shape=det, nz=HW, n=HW) as op:
            out = L._get_nearplane_gradients(
                A.data_to_device(data_in), A.to_device(psi0),
                A.to_device(scan), A.to_device(probe0), A.to_device(ep),
                A.to_device(ew), 0, N, Comm(), num_batch=1,
                exitwave_options=tp.ExitWaveOptions(measured_pixels=mask),
                op=op, recover_psi=True, recover_probe=True)
    finally:
        L.lib.tike_fwd_grad_ifft2_pass1 = entry
    assert calls == [N], calls  # one resident launch over all 64 positions
    np.testing.assert_allclose(out["costs"].cpu().numpy(),
                               np.ravel(o["costs"]), rtol=COST_RTOL)
    assert_close(L.object_upd_sum(out).cpu().numpy(), o["object_upd_sum"],
                 normwise=2e-5, what="object_upd_sum")
    assert_close(out["m_probe_update"].cpu().numpy(), o["m_probe_update"],
                 normwise=2e-5, what="m_probe_update")
    chi0 = out["chi0"]
    if out["chi_modes"] > 1:
        chi0 = chi0[:N, 0, 0]
    assert_close(chi0.cpu().numpy(), o["chi"][:, 0, 0], normwise=2e-5,
                 what="chi mode 0")
    assert_close(out["patches"].cpu().numpy(), o["patches"][:, 0, 0],
                 what="patches")


def test_invalid_patterns_warn_like_the_reference(tp):
    """ptycho.py:392-397: patterns that are negative or not finite draw a
    UserWarning when the context is entered (checked on the device for
    resident float data, on the host for streamed data); clean data and
    16-bit counts draw none."""
    import warnings
    det, S, N = 32, 1, 6
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=3, eigen=False)

    def enter(d, **kw):
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=tp.LstsqOptions(num_batch=1),
            probe_options=tp.ProbeOptions(), object_options=tp.ObjectOptions(),
            exitwave_options=tp.ExitWaveOptions(
                measured_pixels=np.ones((det, det), dtype=bool)))
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            with tp.Reconstruction(d, params, **kw):
                pass
        return [w for w in seen if "invalid data" in str(w.message)]

    bad_nan, bad_neg = data.copy(), data.copy()
    bad_nan[2, 5, 7] = np.nan
    bad_neg[1, 0, 0] = -1.0
    assert not enter(data) and not enter(data, data_on_host=True)
    assert not enter(np.round(data * 100).astype(np.uint16))
    for bad in (bad_nan, bad_neg):
        assert len(enter(bad)) == 1
        assert len(enter(bad, data_on_host=True)) == 1


def test_bench_whole_job_line_of_two_ranks():
    """The N > 1 code path of bench.py itself -- sharding of every global
    minibatch, barrier + max over ranks, the whole-job value, the all-reduce
    summary -- with two ranks sharing the test box's GPU over gloo
    (TIKE_BENCH_SHARE_GPU=1; RCCL needs a device per rank, the driver's
    8-GPU run is the measurement).  One JSON line, from rank 0."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TIKE_BENCH_SHARE_GPU="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29541", os.path.join(root, "bench.py"), "--gpus",
         "2", "--steps", "2", "--warmup", "1", "--positions", "160",
         "--allreduce-probe"],
        env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    # the line proves itself: every rank holds the same object, probe and
    # cost after the timed steps (float64 checksums, all-gathered), and an
    # all-reduce of 1.0 over the live backend counts both ranks
    assert line["ranks_agree"] <= 1e-6, line["ranks_agree"]
    assert line["rccl_ranks"] == 2 and line["collective_backend"] == "gloo"
    # the object-gradient-sized all-reduce timed on its own
    assert line["allreduce_probe"]["bytes"] == line["allreduce"][
        "object_slice"]["bytes"]
    assert line["allreduce_probe"]["calls"] == 100
    assert line["allreduce_probe"]["avg_ms"] > 0
    assert line["config"]["positions_per_gpu"] == 160
    # whole-job value = all ranks' positions over the slowest rank's time
    np.testing.assert_allclose(
        line["value"], 2 * 160 / (line["ms_per_step"] * 1e-3), rtol=1e-6)
    np.testing.assert_allclose(line["per_gpu"] * 2, line["value"], rtol=1e-9)
    assert "whole job on 2" in line["metric"]
    assert "cpu_baseline" not in line or line["cpu_baseline"] is None
    # the collectives of the shared minibatches were issued and timed
    assert line["allreduce"]["calls_per_minibatch"] >= 3
    assert line["allreduce"]["object_slice"]["bytes"] > 0


def test_example_script_runs_and_converges():
    """examples/reconstruct_synthetic.py: the drop-in use shown to a tike user
    (simulate + reconstruct through the public API) runs and lowers the cost."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(
        [sys.executable, os.path.join(root, "examples",
                                      "reconstruct_synthetic.py"),
         "--positions", "100", "--width", "64", "--epochs", "4"],
        env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "cost per epoch" in r.stdout


def test_bench_launcher_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` with fewer than N GPUs must fail loudly
    (never a silent 1-GPU number), before touching the GPU."""
    import os
    import subprocess
    import sys
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"),
                        "--gpus", str(n)], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "refusing" in r.stderr
    assert not r.stdout.strip()


def test_bench_rccl_path_through_the_launcher():
    """One rank started by torch.distributed.run, RCCL collectives forced: the
    multi-GPU code path of bench.py (process group, presharded minibatches,
    timed gradient all-reduce) on the one-GPU test box."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TIKE_FORCE_COLLECTIVES"] = "1"
    r = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node=1", "--master-addr", "127.0.0.1", "--master-port",
         "29533", os.path.join(root, "bench.py"), "--gpus", "1", "--steps",
         "1", "--warmup", "1", "--positions", "200", "--no-cpu-baseline",
         "--no-secondary"], env=env, capture_output=True, text=True,
        timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    # per minibatch: the object-gradient slice + two small packed buffers
    # block the stream (the probe-gradient slice is started early)
    assert line["allreduce"]["calls_per_step"] >= 10
    assert line["allreduce"]["calls_per_minibatch"] <= 3
    big = line["allreduce"]["object_slice"]
    assert 10 <= big["calls_per_step"] <= 11 and big["bytes"] > 0
    assert big["avg_ms"] > 0 and big["busbw_GBs"] == 0  # one rank: no traffic
    # the self-checks of a multi-rank line, through RCCL itself (one rank)
    assert line["collective_backend"] == "nccl" and line["rccl_ranks"] == 1
    assert line["ranks_agree"] == 0.0
    assert 0 < line["roofline"]["iteration_hbm_frac"] < 1


@pytest.mark.parametrize("tag", ["epie", "object", "twoslice", "poisson",
                                 "poisson_all", "eigen"])
def test_rpie_reconstruct_twice_vs_reference(tp, golden, tag):
    """rpie (solvers/rpie.py:26-612) against the reference's own runs:
    alpha = 1 (ePIE, object + probe, compact batches), the default alpha with
    the object alone on NaN-masked data (wobbly_center batches), a two-slice
    object through Multislice / FresnelSpectProp, and (round 4) the
    reference's remaining rpie test configurations
    (tests/ptycho/test_ptycho.py:490-543,670-700): the Poisson model with
    dominant-mode and per-mode step lengths (NaN-masked data) and a variable
    (eigen) probe."""
    import warnings
    g = golden(f"rpie_recon_{tag}.npz")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # NaN in masked data warns
        r1, r2 = _reconstruct_like_reference(tp, g, second=True, algo="rpie")
    epochs = int(g["epochs"])
    np.testing.assert_allclose(
        np.array(r1.algorithm_options.costs[:epochs]), g["costs_1"],
        rtol=1e-3)
    assert r1.psi.shape == g["psi_1"].shape
    assert_close(r1.psi, g["psi_1"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi after call 1")
    assert_close(r1.probe, g["probe_1"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe after call 1")
    np.testing.assert_allclose(np.array(r2.algorithm_options.costs),
                               g["costs_2"], rtol=5e-3)
    assert_close(r2.psi, g["psi_2"], normwise=5e-3, maxabs=5e-2,
                 what="psi after call 2")
    assert_close(r2.probe, g["probe_2"], normwise=5e-3, maxabs=5e-2,
                 what="probe after call 2")
    if "eigen_weights_1" in g:
        # rpie.py:209-214 divides the weights by their rms over the positions:
        # the columns of modes without eigen probes are 0 / 0 = NaN in the
        # reference as well
        want = g["eigen_weights_1"]
        finite = np.isfinite(want)
        assert np.array_equal(np.isfinite(r1.eigen_weights), finite)
        np.testing.assert_allclose(r1.eigen_weights[finite], want[finite],
                                   rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("det,S,N,eigen", [(256, 8, 12, True), (128, 2, 10, False),
                                           # off-grid: prime-factor launches
                                           (160, 3, 9, True)])
def test_rpie_epochs_vs_oracle(tp, det, S, N, eigen):
    """rpie on the fused-kernel sizes (256^2 with 8 modes and eigen-probe
    weights: the far-plane-free pipeline) against the CPU oracle."""
    import tike_amd.random
    from oracle import solvers as osol
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=7 * det + S, eigen=eigen)
    psi0 = np.full_like(psi_true, 0.5)
    batches = np.array_split(np.arange(N), 2)
    params = tp.PtychoParameters(
        probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
        eigen_probe=None if ep is None else ep.copy(),
        eigen_weights=None if ew is None else ew.copy(),
        algorithm_options=tp.RpieOptions(num_batch=2, num_iter=2,
                                         batch_method="compact", alpha=1.0),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        exitwave_options=tp.ExitWaveOptions(
            measured_pixels=np.ones((det, det), dtype=bool)))
    with tp.Reconstruction(data, params, order=np.arange(N), batches=batches,
                           spatial_sort=(det == 256)) as ctx:
        ctx.iterate(2)
        got = ctx.get_result()
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=None if ep is None else ep.copy(),
                 eigen_weights=None if ew is None else ew.copy())
    state = osol.rescale_probe(state, data, det)
    state = osol.iterate(state, data, batches, 2, detector_shape=det,
                         solver="rpie", alpha=1.0, batch_method="compact",
                         force_orthogonality=True)
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(state["costs"]), rtol=1e-3)
    assert_close(got.psi, state["psi"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")
    if eigen:
        # rpie.py:209-214 divides by the rms over positions: the columns of
        # modes without eigen probes are 0 / 0 = NaN in the reference too
        want = state["eigen_weights"]
        finite = np.isfinite(want)
        assert np.array_equal(np.isfinite(got.eigen_weights), finite)
        assert finite[:, 0, :].all()
        np.testing.assert_allclose(got.eigen_weights[finite], want[finite],
                                   rtol=5e-3, atol=1e-4)


@pytest.mark.parametrize("step_back", ["frequency", "as written",
                                       "frequency, separate slice passes",
                                       "frequency, first slice by products"])
@pytest.mark.parametrize("depth,S,N,eigen,u16", [(2, 8, 10, False, False),
                                                  (3, 2, 9, False, True),
                                                  (2, 1, 12, False, False),
                                                  (2, 5, 7, True, False),
                                                  (4, 6, 5, False, False)])
def test_rpie_multislice_fused_vs_oracle(tp, depth, S, N, eigen, u16, step_back):
    """A multislice object at 256^2 on the fused kernels (`tike_fwd_pass1`
    with the incident probes -> `tike_fresnel_colpass` ->
    `tike_fft2_pass2_inplace`; `tike_ifft2_pass2_products` on the way back):
    two rpie epochs against the CPU oracle (rpie.py:367-495,
    multislice.py:69-92, fresnelspectprop.py:52-113) and against the
    slice-by-slice composition of the general operators.  step_back: the
    steps back through the slices as extra outputs of the last slice's
    gradient pass (`tike_fwd_grad_ifft2_pass1_slices`, the default) or as the
    reference writes them (`tike_fft2_pass1` -> `tike_fresnel_colpass`)."""
    import importlib
    import tike_amd.random
    from oracle import operators as oops
    from oracle import solvers as osol
    R = importlib.import_module("tike_amd.ptycho.solvers.rpie")
    det = 256
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=13 * depth + S, eigen=eigen)
    phys = dict(wavelength=1e-10, fov=(2e-6, 2e-6), distance=1e-6)
    psi0 = np.repeat(np.full_like(psi_true, 0.5), depth, axis=0)
    psi0[1:] = 1.0
    if u16:
        data = np.round(data * (20000.0 / data.max())).astype(np.uint16)
    batches = np.array_split(np.arange(N), 2)

    def run(fused):
        R.FUSED_MULTISLICE = fused
        R.STEP_BACK_IN_FREQUENCY = step_back.startswith("frequency")
        R.SLICE_STEP_FUSED = "separate" not in step_back
        R.FIRST_SLICE_STORED_PATCHES = "products" not in step_back
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
            eigen_probe=None if ep is None else ep.copy(),
            eigen_weights=None if ew is None else ew.copy(),
            algorithm_options=tp.RpieOptions(num_batch=2, num_iter=2,
                                             batch_method="compact",
                                             alpha=1.0),
            probe_options=tp.ProbeOptions(
                force_orthogonality=True, probe_wavelength=phys["wavelength"],
                probe_FOV_lengths=phys["fov"]),
            object_options=tp.ObjectOptions(
                multislice_propagation_distance=phys["distance"]),
            exitwave_options=tp.ExitWaveOptions(
                measured_pixels=np.ones((det, det), dtype=bool)))
        try:
            with tp.Reconstruction(data, params, order=np.arange(N),
                                   batches=batches) as ctx:
                ctx.iterate(2)
                return ctx.get_result()
        finally:
            R.FUSED_MULTISLICE = True
            R.STEP_BACK_IN_FREQUENCY = True
            R.SLICE_STEP_FUSED = True
            R.FIRST_SLICE_STORED_PATCHES = True

    got, slow = run(True), run(False)
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(slow.algorithm_options.costs),
                               rtol=1e-4)
    assert_close(got.psi, slow.psi, normwise=2e-4, maxabs=2e-3,
                 what="psi vs slice by slice")
    assert_close(got.probe, slow.probe, normwise=2e-4, maxabs=2e-3,
                 what="probe vs slice by slice")
    propagator = oops.fresnel_spectrum_propagator(
        (det, det), phys["fov"], phys["distance"], phys["wavelength"])
    fdata = data.astype(np.float32)
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=None if ep is None else ep.copy(),
                 eigen_weights=None if ew is None else ew.copy())
    state = osol.rescale_probe(state, fdata, det, propagator=propagator)
    state = osol.iterate(state, fdata, batches, 2, detector_shape=det,
                         solver="rpie", alpha=1.0, batch_method="compact",
                         force_orthogonality=True, propagator=propagator)
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(state["costs"]), rtol=1e-3)
    assert_close(got.psi, state["psi"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")


@pytest.mark.parametrize("det,depth,S,N,model,slice_step,u16", [
    (128, 2, 3, 10, "gaussian", True, False),
    (128, 3, 1, 9, "gaussian", False, True),
    (128, 2, 8, 7, "gaussian", True, False),
    (512, 2, 2, 6, "gaussian", True, False),
    (512, 3, 1, 5, "gaussian", False, False),
    (512, 2, 5, 4, "gaussian", True, True),
    (256, 2, 4, 8, "poisson", True, False),
    (256, 3, 1, 7, "poisson", False, True),
    (128, 2, 2, 8, "poisson", True, False),
    (512, 2, 1, 5, "poisson", True, False),
])
def test_rpie_multislice_fused_sizes_and_models(tp, det, depth, S, N, model,
                                                slice_step, u16):
    """Round 6: the fused multislice chain at 128^2 and 512^2
    (`tike_fresnel_colpass`, `tike_slice_step` and `tike_fwd_pass1` at those
    sizes; `tike_ifft2_pass2_products` with the numerators at 512^2) and under
    the Poisson model at all three (the last slice with its far field stored,
    `rpie._stored_farplane_gradient`): two rpie epochs against the CPU oracle
    (rpie.py:367-495, multislice.py:69-92, fresnelspectprop.py:52-113,
    exitwave.py:122-234) and against the slice-by-slice composition of the
    general operators."""
    import importlib
    from oracle import operators as oops
    from oracle import solvers as osol
    from tike_amd import _lib
    R = importlib.import_module("tike_amd.ptycho.solvers.rpie")
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=7 * depth + S + det, eigen=False)
    phys = dict(wavelength=1e-10, fov=(2e-6, 2e-6), distance=1e-6)
    psi0 = np.repeat(np.full_like(psi_true, 0.5), depth, axis=0)
    psi0[1:] = 1.0
    if u16 or model == "poisson":
        data = np.round(data * (20000.0 / data.max()))
        data = data.astype(np.uint16 if u16 else np.float32)
    batches = np.array_split(np.arange(N), 2)
    seen = []
    real = R._gradients_multislice_fused

    def spy(*a, **k):
        seen.append(1)
        return real(*a, **k)

    def run(fused):
        R.FUSED_MULTISLICE = fused
        R.SLICE_STEP_FUSED = slice_step
        R._gradients_multislice_fused = spy
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=psi0.copy(), scan=scan.copy(),
            algorithm_options=tp.RpieOptions(num_batch=2, num_iter=2,
                                             batch_method="compact",
                                             alpha=1.0),
            probe_options=tp.ProbeOptions(
                force_orthogonality=True, probe_wavelength=phys["wavelength"],
                probe_FOV_lengths=phys["fov"]),
            object_options=tp.ObjectOptions(
                multislice_propagation_distance=phys["distance"]),
            exitwave_options=tp.ExitWaveOptions(
                measured_pixels=np.ones((det, det), dtype=bool),
                noise_model=model))
        try:
            with tp.Reconstruction(data, params, order=np.arange(N),
                                   batches=batches) as ctx:
                ctx.iterate(2)
                return ctx.get_result()
        finally:
            R.FUSED_MULTISLICE = True
            R.SLICE_STEP_FUSED = True
            R._gradients_multislice_fused = real

    got = run(True)
    assert len(seen) == 4, "the fused chain ran for every minibatch"
    slow = run(False)
    assert len(seen) == 4
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(slow.algorithm_options.costs),
                               rtol=2e-4)
    assert_close(got.psi, slow.psi, normwise=3e-4, maxabs=3e-3,
                 what="psi vs slice by slice")
    assert_close(got.probe, slow.probe, normwise=3e-4, maxabs=3e-3,
                 what="probe vs slice by slice")
    propagator = oops.fresnel_spectrum_propagator(
        (det, det), phys["fov"], phys["distance"], phys["wavelength"])
    fdata = data.astype(np.float32)
    state = dict(psi=psi0.copy(), probe=probe0.copy(), scan=scan.copy(),
                 costs=[], eigen_probe=None, eigen_weights=None)
    state = osol.rescale_probe(state, fdata, det, propagator=propagator)
    state = osol.iterate(state, fdata, batches, 2, detector_shape=det,
                         solver="rpie", alpha=1.0, batch_method="compact",
                         force_orthogonality=True, propagator=propagator,
                         noise_model=model)
    np.testing.assert_allclose(np.array(got.algorithm_options.costs),
                               np.array(state["costs"]), rtol=1e-3)
    assert_close(got.psi, state["psi"], normwise=SOLVER_NORMWISE, maxabs=1e-2,
                 what="psi")
    assert_close(got.probe, state["probe"], normwise=SOLVER_NORMWISE,
                 maxabs=1e-2, what="probe")


@pytest.mark.parametrize("det,S,model", [(256, 2, "gaussian"),
                                         (64, 1, "gaussian"),
                                         (256, 3, "poisson")])
def test_uint16_data_stays_16_bit_and_matches_float(tp, det, S, model):
    """Detector counts arriving as uint16 are kept as uint16 in HBM (reference
    ptycho.py:383-390) and give the iterates of the same counts as float32
    (256^2: the 16-bit loader of the streamed column pass -- gaussian: the
    one-launch gradient pass; poisson with per-mode steps: the two column
    passes of tike_poisson_steps_handoff --; 64^2: the converted-chunk
    path)."""
    import torch
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, 8, seed=det + 11, eigen=False)
    counts = np.round(data * (20000.0 / data.max())).astype(np.uint16)
    results = []
    for d in (counts, counts.astype(np.float32)):
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2,
                                              batch_method="compact"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions(),
            exitwave_options=tp.ExitWaveOptions(
                measured_pixels=np.ones((det, det), dtype=bool),
                noise_model=model))
        with tp.Reconstruction(d, params, order=np.arange(8),
                               batches=np.array_split(np.arange(8), 2)) as ctx:
            assert ctx.data.dtype == (torch.uint16 if d.dtype == np.uint16
                                      else torch.float32)
            ctx.iterate(2)
            results.append(ctx.get_result())
    a, b = results
    np.testing.assert_allclose(np.array(a.algorithm_options.costs),
                               np.array(b.algorithm_options.costs), rtol=1e-5)
    assert_close(a.psi, b.psi, normwise=1e-5, maxabs=1e-4, what="psi")
    assert_close(a.probe, b.probe, normwise=1e-5, maxabs=1e-4, what="probe")


@pytest.mark.parametrize("det,S,u16,solver", [(256, 2, True, "lstsq_grad"),
                                              (64, 1, False, "lstsq_grad"),
                                              (128, 1, False, "rpie"),
                                              (256, 1, True, "cgrad")])
def test_data_streamed_from_pinned_host_matches_resident(tp, monkeypatch, det,
                                                         S, u16, solver):
    """`data_on_host=True` (datasets larger than HBM; the reference's
    stream_and_modify2, communicators/stream.py:285-404): the patterns stay in
    pinned host memory and reach the kernels chunk by chunk, prefetched on a
    copy stream.  Same kernels, same numbers: the iterates must agree with the
    HBM-resident run to summation order, and most chunks must have been on
    their way before they were asked for."""
    import tike_amd.ptycho.solvers.lstsq as L
    monkeypatch.setattr(L, "CHUNK_POSITIONS_OVERRIDE", 3)
    N = 16
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=det + 5, eigen=False)
    if u16:
        data = np.round(data * (20000.0 / data.max())).astype(np.uint16)
    options = dict(lstsq_grad=tp.LstsqOptions, rpie=tp.RpieOptions,
                   cgrad=tp.CgradOptions)[solver]
    results = []
    for on_host in (False, True):
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=options(num_batch=2, num_iter=2,
                                      batch_method="compact"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions(),
            exitwave_options=tp.ExitWaveOptions(
                measured_pixels=np.ones((det, det), dtype=bool)))
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=np.array_split(np.arange(N), 2),
                               data_on_host=on_host) as ctx:
            ctx.iterate(2)
            results.append(ctx.get_result())
            if on_host:
                assert type(ctx.data).__name__ == "PinnedData"
                assert ctx.data.copies > 0
                assert ctx.data.hits >= ctx.data.copies // 3
    a, b = results
    np.testing.assert_allclose(np.array(a.algorithm_options.costs),
                               np.array(b.algorithm_options.costs), rtol=1e-5)
    assert_close(a.psi, b.psi, normwise=1e-5, maxabs=1e-4, what="psi")
    assert_close(a.probe, b.probe, normwise=1e-5, maxabs=1e-4, what="probe")


@pytest.mark.parametrize("solver", ["lstsq_grad", "rpie"])
def test_streamed_patterns_follow_the_random_minibatch_order(tp, monkeypatch,
                                                             solver):
    """The solvers visit the minibatches of an epoch in a random permutation
    (lstsq.py:128, rpie.py:93) and tell the pinned-host prefetcher that order
    (`PinnedData.hint`): with 6 minibatches of 2 kernel chunks each, nearly
    every chunk must already be on its way when it is asked for -- the rows
    behind the current chunk would be the right guess for half of them -- and
    the iterates are those of the resident run."""
    import tike_amd.ptycho.solvers.lstsq as L
    monkeypatch.setattr(L, "CHUNK_POSITIONS_OVERRIDE", 5)  # chunks of 5 + 3
    det, S, N, nb = 128, 2, 48, 6
    scan, psi_true, probe0, _, _, data = _headline_problem(
        tp, det, S, N, seed=det + 9, eigen=False)
    options = dict(lstsq_grad=tp.LstsqOptions, rpie=tp.RpieOptions)[solver]
    results = []
    for on_host in (False, True):
        tike_amd_random = __import__("tike_amd.random").random
        tike_amd_random.randomizer_np = np.random.default_rng(3)
        params = tp.PtychoParameters(
            probe=probe0.copy(), psi=np.full_like(psi_true, 0.5),
            scan=scan.copy(),
            algorithm_options=options(num_batch=nb, num_iter=3,
                                      batch_method="wobbly_center"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions(),
            exitwave_options=tp.ExitWaveOptions(
                measured_pixels=np.ones((det, det), dtype=bool)))
        with tp.Reconstruction(data, params, order=np.arange(N),
                               batches=np.array_split(np.arange(N), nb),
                               data_on_host=on_host) as ctx:
            ctx.iterate(3)
            results.append(ctx.get_result())
            if on_host:
                d = ctx.data
                # 3 epochs x 6 minibatches x 2 chunks, each copied ONCE
                # (plus whatever the set-up read)
                assert d.copies <= 36 + 14, (d.copies, d.hits)
                assert d.hits >= 30, (d.copies, d.hits)
    a, b = results
    np.testing.assert_allclose(np.array(a.algorithm_options.costs),
                               np.array(b.algorithm_options.costs), rtol=1e-5)
    assert_close(a.psi, b.psi, normwise=1e-5, maxabs=1e-4, what="psi")


@pytest.mark.parametrize("det,S,N", [(256, 8, 1000), (512, 4, 400)])
def test_full_size_fused_gradient_adjoint(tp, det, S, N):
    """The bench's launch sizes (c3: 1000 positions x 8 modes x 256^2; c5: 400
    x 4 x 512^2), checked through a size-independent property: the fused
    gradient kernels (inverse pass 1 -> pass 2 + both gradients -> grouped
    scatter) are the adjoint of the forward operator,
        <A(psi) , G> = <psi , A^H G>   and   <A(probe) , G> = <probe , A^H G>,
    for a random far-plane array G (gradient factor 1)."""
    import torch
    import tike_amd._arrays as A
    import tike_amd.operators as ops
    from tike_amd._lib import check, lib
    from tike_amd import cluster
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(det)
    pw = det
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:N]
    scan_np = (1 + 8.0 * ij + rng.random((N, 2))).astype(np.float32)
    scan_np = scan_np[cluster.spatial_order(scan_np)]
    HW = int(np.ceil((8 * (side - 1) + pw + 4) / 32.0) * 32)
    g = torch.Generator(device=dev).manual_seed(det)
    rc = lambda *s: torch.view_as_complex(
        torch.rand(*s, 2, device=dev, generator=g) - 0.5)
    psi, probe = rc(1, HW, HW), rc(1, 1, S, pw, pw)
    scan = A.to_device(scan_np)
    G = rc(N, 1, S, det, det)
    st = A.stream_ptr()
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        far = torch.empty_like(G)
        op.fwd_device(probe, scan, psi, out=far)
    lhs = (far * G.conj()).sum()
    del far
    ones = torch.ones(N, det, det, device=dev)
    work = torch.empty_like(G)
    check(lib.tike_ifft2_pass1_scaled(A.ptr(G), A.ptr(ones), None, None, S,
                                      A.ptr(work), N * S, det, st))
    patches = torch.empty(N, pw, pw, dtype=torch.complex64, device=dev)
    scratch = torch.empty_like(G)
    check(lib.tike_fwd_pass1(A.ptr(psi), A.ptr(scan), A.ptr(probe), 0, None,
                             None, None, 0, 0, A.ptr(scratch), A.ptr(patches),
                             N, S, pw, det, HW, HW, st))
    del scratch
    objproj = torch.empty_like(patches)
    chi0 = torch.empty_like(patches)
    mpu = torch.zeros(1, 1, S, pw, pw, dtype=torch.complex64, device=dev)
    check(lib.tike_ifft2_pass2_gradients(
        A.ptr(work), A.ptr(patches), A.ptr(probe), None, None, 0, 0,
        A.ptr(objproj), A.ptr(chi0), A.ptr(mpu), 1.0, N, S, det, 1.0 / det, st))
    acc = torch.zeros(2, HW, HW, device=dev)
    check(lib.tike_scatter_patches(A.ptr(objproj), A.ptr(scan), A.ptr(acc), N,
                                   pw, HW, HW, st))
    adj_psi = torch.complex(acc[0], acc[1])
    rhs_psi = (psi[0] * adj_psi.conj()).sum()
    rhs_probe = (probe * mpu.conj()).sum()
    scale = float(lhs.abs())
    for name, rhs in (("psi", rhs_psi), ("probe", rhs_probe)):
        err = float((lhs - rhs).abs()) / scale
        assert err < 2e-4, (name, err, complex(lhs), complex(rhs))


@pytest.mark.parametrize("det,S,N", [(256, 8, 1000), (512, 4, 400)])
def test_full_size_far_plane_free_chain_matches_stored_far_plane(tp, det, S, N):
    """At the bench's launch sizes the far-plane-free chain (forward pass 1 ->
    streamed column pass + gradient factor -> gradient + inverse pass 1) hands
    pass 2 the same intermediate, costs and factor as the stored-far-plane
    kernels (forward operator -> gradient scale -> scaled inverse pass 1), and
    the costs are the gaussian objective of the stored far plane."""
    import torch
    import tike_amd._arrays as A
    import tike_amd.operators as ops
    from tike_amd._lib import check, lib
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(det + 1)
    pw = det
    side = int(np.ceil(np.sqrt(N)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"),
                  -1).reshape(-1, 2)[:N]
    scan = A.to_device((1 + 8.0 * ij + rng.random((N, 2))).astype(np.float32))
    HW = int(np.ceil((8 * (side - 1) + pw + 4) / 32.0) * 32)
    g = torch.Generator(device=dev).manual_seed(det)
    rc = lambda *s: torch.view_as_complex(
        torch.rand(*s, 2, device=dev, generator=g) - 0.5)
    psi, probe = rc(1, HW, HW) + 1, rc(1, 1, S, pw, pw)
    data = torch.rand(N, det, det, device=dev, generator=g) * 50
    st = A.stream_ptr()
    with ops.Ptycho(probe_shape=pw, detector_shape=det, nz=HW, n=HW) as op:
        far = torch.empty(N, 1, S, det, det, dtype=torch.complex64, device=dev)
        op.fwd_device(probe, scan, psi, out=far)
    inten = (far.abs()**2).sum(dim=(1, 2))
    want_cost = ((inten.sqrt() - data.sqrt())**2).mean(dim=(-2, -1))
    scratch = torch.empty_like(far)
    gs = torch.empty(N, det, det, device=dev)
    costs = torch.empty(N, device=dev)
    check(lib.tike_fwd_pass1(A.ptr(psi), A.ptr(scan), A.ptr(probe), 0, None,
                             None, None, 0, 0, A.ptr(scratch), None, N, S, pw,
                             det, HW, HW, st))
    check(lib.tike_fwd_gradient_scale(A.ptr(scratch), A.ptr(data), 0, None,
                                      A.ptr(gs), None, A.ptr(costs), None, N, S,
                                      det, 1.0 / det, 0, 1.0, det * det, st))
    torch.testing.assert_close(costs, want_cost, rtol=2e-4, atol=1e-6)
    work = torch.empty_like(far)
    check(lib.tike_grad_ifft2_pass1(A.ptr(scratch), A.ptr(gs), None, None, S,
                                    A.ptr(work), N * S, det, 1.0 / det, st))
    del scratch
    ref = torch.empty_like(far)
    check(lib.tike_ifft2_pass1_scaled(A.ptr(far), A.ptr(gs), None, None, S,
                                      A.ptr(ref), N * S, det, st))
    if det == 256:
        # The kernel of record -- tike_fwd_grad_ifft2_pass1, the one-launch
        # column pass + gradient factor + inverse pass 1 the bench times -- at
        # its bench launch size (N = 1000, S = 8: the register-resident kernel
        # with the rotated prefetch), with float32 counts and no mask, and with
        # uint16 counts and unmeasured pixels holding values that must never be
        # read as data.  Its intermediate and costs must be those of the two
        # launches above on the same hand-off, mask and counts.
        scratch2 = torch.empty_like(far)
        check(lib.tike_fwd_pass1(A.ptr(psi), A.ptr(scan), A.ptr(probe), 0, None,
                                 None, None, 0, 0, A.ptr(scratch2), None, N, S,
                                 pw, det, HW, HW, st))
        mask = (torch.rand(det, det, device=dev, generator=g) > 0.1)
        mask_u8 = mask.to(torch.uint8).contiguous()
        nmeas = int(mask.sum().item())
        counts = A.data_to_device(
            np.round(data.cpu().numpy() * 400).astype(np.uint16))
        assert counts.dtype == torch.uint16
        counts_f = torch.from_numpy(
            np.round(data.cpu().numpy() * 400).astype(np.float32)).to(dev)
        for tag, d_k, d_f, u16, m_u8, nm in (
                ("f32", data, data, 0, None, det * det),
                ("u16+mask", counts, counts_f, 1, mask_u8, nmeas)):
            one = torch.empty_like(far)
            costs1 = torch.empty(N, device=dev)
            check(lib.tike_fwd_grad_ifft2_pass1(
                A.ptr(scratch2), A.ptr(d_k), u16, A.ptr(m_u8), A.ptr(costs1),
                A.ptr(one), N, S, det, 1.0 / det, 0, 0.25, nm, st))
            if m_u8 is None:
                want_c, two = costs, work
            else:
                gs2 = torch.empty_like(gs)
                want_c = torch.empty(N, device=dev)
                check(lib.tike_fwd_gradient_scale(
                    A.ptr(scratch2), A.ptr(d_k), u16, A.ptr(m_u8), A.ptr(gs2),
                    None, A.ptr(want_c), None, N, S, det, 1.0 / det, 0, 0.25, nm,
                    st))
                two = torch.empty_like(far)
                check(lib.tike_grad_ifft2_pass1(
                    A.ptr(scratch2), A.ptr(gs2), None, None, S, A.ptr(two),
                    N * S, det, 1.0 / det, st))
                # ... and the masked costs / factor are the objective's
                sq = (inten.sqrt() - d_f.sqrt())**2
                torch.testing.assert_close(
                    want_c, (sq * mask).sum(dim=(-2, -1)) / nmeas, rtol=2e-4,
                    atol=1e-6)
                g_want = torch.where(
                    mask, -(1 - d_f.sqrt() / (inten.sqrt() + 1e-9)),
                    torch.full_like(inten, 0.25 - 1.0))
                torch.testing.assert_close(gs2, g_want, rtol=2e-4, atol=2e-5)
                del gs2, g_want, sq
            torch.testing.assert_close(costs1, want_c, rtol=1e-5, atol=1e-7)
            err1 = float((one - two).abs().max()) / float(two.abs().max())
            assert err1 < 2e-6, (tag, err1)
            del one
            if m_u8 is not None:
                del two
        del scratch2
    if det == 256:
        # different (equivalent) factorisations of the inverse at 256^2: the
        # intermediates differ, what pass 2 makes of them must not
        outs = []
        patches = torch.zeros(N, pw, pw, dtype=torch.complex64, device=dev)
        for w in (work, ref):
            chi0 = torch.empty_like(patches)
            check(lib.tike_ifft2_pass2_gradients(
                A.ptr(w), A.ptr(patches), None, None, None, 0, 0, None,
                A.ptr(chi0), None, 1.0, N, S, det, 1.0 / det, st))
            outs.append(chi0)
        work, ref = outs
    err = float((work - ref).abs().max()) / float(ref.abs().max())
    assert err < 2e-5, err


@pytest.mark.parametrize("det,S,N,eigen,u16", [(256, 3, 7, True, False),
                                               (256, 8, 5, False, True),
                                               (512, 2, 4, False, False)])
def test_chunk_gradients_entry_matches_the_staged_pipeline(tp, det, S, N, eigen,
                                                           u16):
    """tike_lstsq_chunk_gradients (the chunk body of _get_nearplane_gradients,
    lstsq.py:422-579, in one C call) == the five stage entries as the solver
    issues them (themselves checked against the oracle and the reference)."""
    import torch
    import tike_amd._arrays as A
    import tike_amd.ptycho.solvers.lstsq as L
    from tike_amd._lib import check, lib
    from tike_amd.communicators import Comm
    from tike_amd.operators import Ptycho
    from tike_amd.operators.propagation import fft_scales
    scan, psi_true, probe0, ep, ew, data = _headline_problem(
        tp, det, S, N, seed=det + S, eigen=eigen)
    if u16:
        data = np.round(data * (20000.0 / data.max())).astype(np.uint16)
    dev = torch.device("cuda", 0)
    psi = A.to_device(np.full_like(psi_true, 0.5) + 0.1 * psi_true)
    probe, scan_d = A.to_device(probe0), A.to_device(scan)
    data_d = A.data_to_device(data)
    ep_d = None if ep is None else A.to_device(ep)
    ew_d = None if ew is None else A.to_device(ew.astype(np.float32))
    H, W = psi.shape[-2:]
    opts = tp.ExitWaveOptions(measured_pixels=np.ones((det, det), dtype=bool))
    with Ptycho(probe_shape=det, detector_shape=det, nz=H, n=W) as op:
        g = L._get_nearplane_gradients(
            data_d, psi, scan_d, probe, ep_d, ew_d, 0, N, Comm(), num_batch=2,
            exitwave_options=opts, op=op, recover_psi=True, recover_probe=True)
        want = {k: g[k].clone() for k in ("object_acc", "m_probe_update",
                                          "costs", "chi0")}
    c64 = lambda *s: torch.empty(*s, dtype=torch.complex64, device=dev)
    scratch, work = c64(N, 1, S, det, det), c64(N, 1, S, det, det)
    gscale = torch.empty(N, det, det, device=dev)
    patches, objproj, chi0 = c64(N, det, det), c64(N, det, det), c64(N, det, det)
    costs = torch.empty(N, device=dev)
    mpu = torch.zeros(1, 1, S, det, det, dtype=torch.complex64, device=dev)
    acc = torch.zeros(2, H, W, device=dev)
    C = Sm = 0
    if ep_d is not None:
        C, Sm = ep_d.shape[-4], ep_d.shape[-3]
    fwd_scale, inv_scale = fft_scales(det, "ortho")
    check(lib.tike_lstsq_chunk_gradients(
        A.ptr(psi), A.ptr(scan_d), A.ptr(probe), A.ptr(ep_d), A.ptr(ew_d), C, Sm,
        A.ptr(data_d), int(u16), None, 0, 1.0, det * det, A.ptr(scratch),
        A.ptr(work), A.ptr(gscale), A.ptr(patches), A.ptr(costs),
        A.ptr(objproj), A.ptr(chi0), A.ptr(mpu), 0.5, A.ptr(acc), N, S, det, H,
        W, fwd_scale, inv_scale, A.stream_ptr()), "chunk gradients")
    torch.testing.assert_close(costs, want["costs"], rtol=1e-5, atol=1e-7)
    assert_close(acc.cpu().numpy(), want["object_acc"].cpu().numpy(),
                 normwise=1e-5, maxabs=1e-4, what="object gradient")
    assert_close(mpu.cpu().numpy(), want["m_probe_update"].cpu().numpy(),
                 normwise=1e-5, maxabs=1e-4, what="probe gradient")
    assert_close(chi0.cpu().numpy(), want["chi0"].cpu().numpy(), normwise=1e-6,
                 maxabs=1e-5, what="chi0")
    # unsupported shapes are refused, not mis-computed
    assert lib.tike_lstsq_chunk_gradients(
        A.ptr(psi), A.ptr(scan_d), A.ptr(probe), None, None, 0, 0,
        A.ptr(data_d), int(u16), None, 0, 1.0, det * det, A.ptr(scratch),
        A.ptr(work), A.ptr(gscale), A.ptr(patches), A.ptr(costs),
        A.ptr(objproj), A.ptr(chi0), A.ptr(mpu), 0.5, A.ptr(acc), N, 9, det, H,
        W, fwd_scale, inv_scale, A.stream_ptr()) == 1000002


def test_data_upload_moves_only_the_rows_it_is_asked_for(monkeypatch):
    """data_to_device (round-4 advisor finding): a rank's share of the rows,
    a full permutation, the identity and a row named twice all arrive in their
    final order, block by block -- with blocks of three rows here -- and the
    upload of a share never touches the rows of other ranks."""
    import torch
    import tike_amd._arrays as A
    rng = np.random.default_rng(0)
    a = rng.random((23, 5, 7)).astype(np.float64)  # widened/narrowed on the way
    monkeypatch.setattr(A, "_upload_block_rows", lambda arr: 3)
    seen = []
    real = torch.from_numpy

    def spy(x):
        if x.ndim == 3:
            seen.append(x.shape[0])
        return real(x)

    monkeypatch.setattr(A.torch, "from_numpy", spy)
    share = np.array([20, 1, 2, 3, 11, 22, 0])
    for order in (share, rng.permutation(23), None, np.array([4, 4, 9])):
        seen.clear()
        t = A.data_to_device(a, order=order)
        want = a if order is None else a[order]
        assert t.dtype == torch.float32 and t.is_contiguous()
        np.testing.assert_array_equal(t.cpu().numpy(), want.astype(np.float32))
        assert sum(seen) == len(want)  # no other row crossed to the device
    bad = a.astype(np.float32)
    assert not A.has_invalid_counts(A.data_to_device(bad))
    bad[17, 2, 3] = -1.0
    monkeypatch.setattr(A.torch, "from_numpy", real)
    assert A.has_invalid_counts(A.data_to_device(bad), block_bytes=3 * 140)
    bad[17, 2, 3] = np.nan
    assert A.has_invalid_counts(A.data_to_device(bad), block_bytes=3 * 140)


def test_deterministic_mode_gives_bit_identical_runs():
    """TIKE_DETERMINISTIC=1 (include/tike_amd.h `tike_set_deterministic`):
    fixed-order sums instead of float atomics.  Two fresh processes produce
    bit-identical objects, probes, eigen probes, weights and costs -- on the
    general-shape kernels (the reference's ReconstructTwice run, fixture
    lstsq_recon_compact, both calls) and on the fused 256^2 kernels (8 modes +
    eigen probe; two rpie epochs on a two-slice object) -- and the SECOND call
    of the fixture, whose tolerance is 5e-3 with atomics, agrees with the
    reference's run to 1e-3."""
    import json
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))

    def child(det):
        env = dict(os.environ, TIKE_DETERMINISTIC="1" if det else "0")
        env.pop("TIKE_CHUNK_POSITIONS", None)
        out = subprocess.run(
            [sys.executable, os.path.join(here, "_deterministic_child.py")],
            capture_output=True, text=True, env=env, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")]
        return json.loads(line[-1][len("RESULT "):])

    a, b, free = child(True), child(True), child(False)
    assert a["compact"] == b["compact"]
    assert a["headline"] == b["headline"]
    assert a["headline_cost"] == b["headline_cost"]
    assert a["multislice"] == b["multislice"]
    np.testing.assert_allclose(a["multislice_cost"], free["multislice_cost"],
                               rtol=1e-4)
    # the same algorithm: the atomics run agrees to float32 round-off
    np.testing.assert_allclose(a["headline_cost"], free["headline_cost"],
                               rtol=1e-4)
    np.testing.assert_allclose(a["headline_psi_norm"],
                               free["headline_psi_norm"], rtol=1e-5)
    # round 6: cgrad (direction sums, all-steps-at-once line search; 256^2 and
    # 128^2) and the Poisson model with per-mode step lengths
    for key in ("cgrad256", "cgrad128", "poisson", "groups", "pfa", "offgrid",
                "multislice128p"):
        assert a[key] == b[key], key
        np.testing.assert_allclose(a[key + "_cost"], free[key + "_cost"],
                                   rtol=2e-3)
    assert max(a["compact_err2"]) < 1e-3, a["compact_err2"]
    assert a["compact_cost2"] < 1e-3, a["compact_cost2"]
