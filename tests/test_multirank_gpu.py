"""Two ranks (one process each, gloo, sharing cuda:0 on the single-GPU test
box) reconstruct the same problem as one rank: with all-reduced gradients the
P-rank iterates equal the 1-rank iterates up to summation order
(DESIGN.md section 5)."""
import os
import socket

import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


SMALL = dict(N=64, S=2, pw=32, side=8, pitch=5.0)
# the headline (c3) shape: 256^2 far-plane-free kernels, 8 modes + one eigen
# probe, two minibatches of 64 positions = 32 per rank and minibatch
C3 = dict(N=128, S=8, pw=256, side=12, pitch=8.0)


def _problem(eigen, shape=SMALL):
    import tike_amd.ptycho as tp
    import tike_amd.random
    rng = np.random.default_rng(3)
    N, S, pw, side, pitch = (shape[k] for k in ("N", "S", "pw", "side",
                                                "pitch"))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side),
                              indexing="ij"), -1).reshape(-1, 2)[:N]
    scan = (2 + pitch * ij + rng.random((N, 2))).astype(np.float32)
    HW = int(pitch * (side - 1)) + pw + 8
    psi_true = ((0.75 + 0.25 * rng.random((1, HW, HW))) * np.exp(
        1j * np.pi * (rng.random((1, HW, HW)) - 0.5))).astype(np.complex64)
    w = tp.gaussian(pw, rin=0.6)
    probe = np.stack([w * np.exp(1j * np.pi * rng.random((pw, pw))) / (m + 1)
                      for m in range(S)])[None, None].astype(np.complex64)
    data = tp.simulate(pw, probe, scan, psi_true)
    ep = ew = None
    if eigen:
        np.random.seed(5)
        tike_amd.random.randomizer_np = np.random.default_rng(6)
        ep, ew = tp.init_varying_probe(scan, probe, 2, 1)
    return data, scan, probe, np.full_like(psi_true, 0.5), ep, ew


def _reconstruct(eigen, method, positions=False, rank=0, shape=SMALL,
                 num_iter=3):
    import tike_amd.ptycho as tp
    import tike_amd.random
    data, scan, probe, psi0, ep, ew = _problem(eigen, shape)
    # only rank 0 starts from the single-rank run's generator states: the
    # library must hand them to the other ranks (Comm.sync_random)
    np.random.seed(1 + 17 * rank)
    tike_amd.random.randomizer_np = np.random.default_rng(2 + 17 * rank)
    params = tp.PtychoParameters(
        probe=probe.copy(), psi=psi0.copy(), scan=scan.copy(), eigen_probe=ep,
        eigen_weights=ew,
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=num_iter,
                                          batch_method=method),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        position_options=tp.PositionOptions(
            scan.copy(), use_adaptive_moment=True,
            use_position_regularization=True, update_magnitude_limit=2)
        if positions else None)
    return tp.reconstruct(data, params)


def _worker(rank, world, port, eigen, method, ret, positions=False,
            shape=SMALL, num_iter=3):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # every float32 sum the solver issues, in order: ("start" | "sum", numel)
    from tike_amd.communicators import Comm
    log = []
    plain, early = Comm.Allreduce, Comm.Allreduce_start

    def logged_sum(self, *tensors):
        log.append(("sum", sum(t.numel() * (2 if t.is_complex() else 1)
                               for t in tensors)))
        return plain(self, *tensors)

    def logged_start(self, flat):
        log.append(("start", flat.numel()))
        return early(self, flat)

    Comm.Allreduce, Comm.Allreduce_start = logged_sum, logged_start
    try:
        r = _reconstruct(eigen, method, positions, rank=rank, shape=shape,
                         num_iter=num_iter)
        ret[rank] = (r.psi, r.probe, r.eigen_weights, r.scan,
                     np.array(r.algorithm_options.costs), log,
                     r.psi.shape[-2] * r.psi.shape[-1], r.probe.size)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("eigen,method,positions,shape", [
    (False, "compact", False, SMALL), (True, "wobbly_center", False, SMALL),
    (False, "compact", True, SMALL), (False, "wobbly_center", True, SMALL),
    # the kernels the headline number comes from (256^2 far-plane-free chain,
    # grouped scatter on a sharded, spatially sorted minibatch, eigen path at
    # S = 8), two epochs of two minibatches
    (True, "compact", False, C3)])
def test_two_ranks_match_one_rank(eigen, method, positions, shape):
    import torch.multiprocessing as mp
    num_iter = 2 if shape is C3 else 3
    single = _reconstruct(eigen, method, positions, shape=shape,
                          num_iter=num_iter)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker,
                         args=(r, 2, port, eigen, method, ret, positions,
                               shape, num_iter))
             for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    for rank in range(2):
        psi, probe, ew, scan, costs, log, npix, nprobe = ret[rank]
        # the same collectives in the same order on both ranks, and per
        # minibatch: the probe-gradient slice started early (it travels
        # during the object scatter), then the object slice, then the two
        # small packed sums of the tail
        assert log == ret[0][5]
        starts = [i for i, (kind, n) in enumerate(log)
                  if kind == "start" and n == 2 * nprobe]
        assert len(starts) == 2 * num_iter  # two minibatches per epoch
        for i in starts:
            assert log[i + 1] == ("sum", 2 * npix), log[i:i + 4]
        if shape is C3:  # packed tail: exactly two more sums per minibatch
            for a, b in zip(starts, starts[1:] + [len(log)]):
                tail = [x for x in log[a + 2:b] if x[0] == "sum"]
                small = [x for x in tail if x[1] <= 2 * 256 * 256 + 4]
                assert len(small) >= 2, log[a:b]
        np.testing.assert_allclose(
            costs, np.array(single.algorithm_options.costs), rtol=1e-3)
        assert_close(psi, single.psi, normwise=1e-3, maxabs=1e-2,
                     what=f"psi rank {rank}")
        assert_close(probe, single.probe, normwise=1e-3, maxabs=1e-2,
                     what=f"probe rank {rank}")
        if positions:
            # corrected positions (pixels); moved by up to ~1 px per epoch
            assert np.abs(single.scan - _problem(eigen, shape)[1]).max() > 0.05
            np.testing.assert_allclose(scan, single.scan, atol=5e-3)
        else:
            np.testing.assert_allclose(scan, single.scan, atol=1e-5)
        if eigen:
            assert_close(ew, single.eigen_weights, normwise=5e-3, maxabs=5e-2,
                         what=f"eigen weights rank {rank}")


def _rccl_worker(port, ret, backend=""):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["TIKE_FORCE_COLLECTIVES"] = "1"
    os.environ["TIKE_COMM_BACKEND"] = backend
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        r = _reconstruct(True, "wobbly_center", positions=True)
        ret[0] = (r.psi, r.probe, r.eigen_weights, r.scan,
                  np.array(r.algorithm_options.costs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", ["", "cabi"])
def test_rccl_collectives_single_rank(backend):
    """The production backend: every collective of the solver (packed f32
    all-reduce, f64 scalar all-reduces, max all-reduce, row all-gather,
    position gather) issued through RCCL ("nccl") on a one-rank group must
    leave the result unchanged.  backend "cabi": the float32 all-reduces go
    through tike_comm_allreduce_sum (the library's own RCCL communicator)."""
    import torch.multiprocessing as mp
    single = _reconstruct(True, "wobbly_center", positions=True)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), ret, backend))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    psi, probe, ew, scan, costs = ret[0]
    # two runs differ by the order of the float atomics only; the normalised
    # ADAM position steps amplify that a little
    np.testing.assert_allclose(costs, np.array(single.algorithm_options.costs),
                               rtol=1e-4)
    assert_close(psi, single.psi, normwise=1e-4, maxabs=1e-3, what="psi")
    assert_close(probe, single.probe, normwise=1e-4, maxabs=1e-3, what="probe")
    np.testing.assert_allclose(scan, single.scan, atol=1e-3)


def test_reconstruct_num_gpu_hands_uint16_counts_through_shared_memory(
        monkeypatch):
    """16-bit detector counts reach the spawned ranks as they are (shared
    memory, viewed back as uint16) and stay 16-bit in HBM there: the two-rank
    result equals the one-rank result on the same counts."""
    import tike_amd.ptycho as tp
    data, scan, probe, psi0, _, _ = _problem(False)
    counts = np.round(data * (20000.0 / data.max())).astype(np.uint16)

    def run(num_gpu):
        np.random.seed(3)
        params = tp.PtychoParameters(
            probe=probe.copy(), psi=psi0.copy(), scan=scan.copy(),
            algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=2,
                                              batch_method="compact"),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())
        return tp.reconstruct(counts, params, num_gpu=num_gpu)

    one = run(None)
    monkeypatch.setenv("TIKE_AMD_OVERSUBSCRIBE", "1")
    two = run(2)
    np.testing.assert_allclose(np.array(two.algorithm_options.costs),
                               np.array(one.algorithm_options.costs),
                               rtol=1e-3)
    assert_close(two.psi, one.psi, normwise=1e-3, maxabs=1e-2, what="psi")
    assert_close(two.probe, one.probe, normwise=1e-3, maxabs=1e-2,
                 what="probe")


def test_reconstruct_num_gpu_starts_the_ranks_itself(monkeypatch):
    """The reference's one-call contract (ptycho.py:182-187,371-381):
    ``reconstruct(data, parameters, num_gpu=2)`` from a plain process uses two
    ranks -- here two spawned children that share the test box's GPU over
    gloo (TIKE_AMD_OVERSUBSCRIBE) -- and returns the one-rank iterates."""
    import tike_amd.ptycho as tp
    import tike_amd.random
    single = _reconstruct(True, "wobbly_center", positions=True)
    data, scan, probe, psi0, ep, ew = _problem(True)
    np.random.seed(1)
    tike_amd.random.randomizer_np = np.random.default_rng(2)
    params = tp.PtychoParameters(
        probe=probe.copy(), psi=psi0.copy(), scan=scan.copy(), eigen_probe=ep,
        eigen_weights=ew,
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=3,
                                          batch_method="wobbly_center"),
        probe_options=tp.ProbeOptions(force_orthogonality=True),
        object_options=tp.ObjectOptions(),
        position_options=tp.PositionOptions(
            scan.copy(), use_adaptive_moment=True,
            use_position_regularization=True, update_magnitude_limit=2))
    monkeypatch.setenv("TIKE_AMD_OVERSUBSCRIBE", "1")
    r = tp.reconstruct(data, params, num_gpu=2)
    np.testing.assert_allclose(np.array(r.algorithm_options.costs),
                               np.array(single.algorithm_options.costs),
                               rtol=1e-3)
    assert_close(r.psi, single.psi, normwise=1e-3, maxabs=1e-2, what="psi")
    assert_close(r.probe, single.probe, normwise=1e-3, maxabs=1e-2,
                 what="probe")
    np.testing.assert_allclose(r.scan, single.scan, atol=5e-3)
    assert r.eigen_weights.shape == single.eigen_weights.shape
    # the caller's generators advanced as in an in-process call
    after = tike_amd.random.randomizer_np.bit_generator.state
    assert after != np.random.default_rng(2).bit_generator.state


@pytest.mark.parametrize("solver", ["cgrad", "rpie"])
def test_other_solvers_on_two_ranks_match_one_rank(monkeypatch, solver):
    """cgrad (cost and gradient summed over the ranks, every rank takes the
    same step: the pattern of lamino/solvers/cgrad.py:58-92; its line
    searches run on the device, the cost sums of each pass all-reduced
    between the pass and its decision) and rpie on two spawned ranks give
    the one-rank iterates."""
    import tike_amd.ptycho as tp
    import tike_amd.random
    data, scan, probe, psi0, _, _ = _problem(False)
    options = dict(cgrad=tp.CgradOptions(num_batch=2, num_iter=2, cg_iter=2),
                   rpie=tp.RpieOptions(num_batch=2, num_iter=3,
                                       batch_method="wobbly_center"))[solver]
    results = []
    for num_gpu in (None, 2):
        np.random.seed(1)
        tike_amd.random.randomizer_np = np.random.default_rng(2)
        params = tp.PtychoParameters(
            probe=probe.copy(), psi=psi0.copy(), scan=scan.copy(),
            algorithm_options=__import__("copy").deepcopy(options),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())
        if num_gpu:
            monkeypatch.setenv("TIKE_AMD_OVERSUBSCRIBE", "1")
        results.append(tp.reconstruct(data, params, num_gpu=num_gpu))
    a, b = results
    np.testing.assert_allclose(np.array(b.algorithm_options.costs),
                               np.array(a.algorithm_options.costs), rtol=1e-3)
    assert_close(b.psi, a.psi, normwise=1e-3, maxabs=1e-2, what="psi")
    assert_close(b.probe, a.probe, normwise=1e-3, maxabs=1e-2, what="probe")


def test_reconstruction_context_refuses_num_gpu_it_cannot_honour(monkeypatch):
    import tike_amd.ptycho as tp
    data, scan, probe, psi0, _, _ = _problem(False)
    params = tp.PtychoParameters(
        probe=probe, psi=psi0, scan=scan,
        algorithm_options=tp.LstsqOptions(num_batch=2, num_iter=1),
        probe_options=tp.ProbeOptions(), object_options=tp.ObjectOptions())
    monkeypatch.setenv("TIKE_AMD_OVERSUBSCRIBE", "1")
    with pytest.raises(ValueError, match="num_gpu"):
        tp.Reconstruction(data, params, num_gpu=2)
    monkeypatch.delenv("TIKE_AMD_OVERSUBSCRIBE")
    # fewer GPUs than requested: the reference's rule -- warn, use what is there
    with pytest.warns(UserWarning, match="GPU"):
        r = tp.reconstruct(data, params, num_gpu=64)
    assert len(r.algorithm_options.costs) == 1


@pytest.mark.timeout(600)
@pytest.mark.parametrize("eigen,solver", [(False, "lstsq_grad"),
                                          (True, "lstsq_grad"),
                                          (True, "rpie"), (False, "cgrad")])
def test_a_rank_with_an_empty_share_issues_the_same_collectives(monkeypatch,
                                                                eigen, solver):
    """Round-4 advisor finding: a minibatch of ONE position leaves rank 1 of
    two with an empty share.  That rank never enters the chunk loop, where
    the early probe-gradient all-reduce used to be started: it issued one sum
    where rank 0 issued two (a hang under RCCL, an error under gloo).  The
    order of the collectives now depends on rank-invariant facts only, and the
    cached per-minibatch counts are keyed on the minibatch (two empty shares
    have the same local bounds): the two-rank run completes and reproduces the
    one-rank iterates."""
    import tike_amd.ptycho as tp
    import tike_amd.random
    data, scan, probe, psi0, ep, ew = _problem(eigen)
    N = 11
    data, scan = data[:N], scan[:N]
    if eigen:
        ew = ew[:N]
    order = np.arange(N)
    sizes = [1, 1, 5, 4]  # two one-position minibatches: two empty shares
    ends = np.cumsum(sizes)
    batches = [np.arange(e - s, e) for s, e in zip(sizes, ends)]

    def run(num_gpu):
        np.random.seed(1)
        tike_amd.random.randomizer_np = np.random.default_rng(2)
        params = tp.PtychoParameters(
            probe=probe.copy(), psi=psi0.copy(), scan=scan.copy(),
            eigen_probe=None if ep is None else ep.copy(),
            eigen_weights=None if ew is None else ew.copy(),
            algorithm_options=dict(
                lstsq_grad=lambda: tp.LstsqOptions(
                    num_batch=len(sizes), num_iter=2,
                    batch_method="wobbly_center"),
                rpie=lambda: tp.RpieOptions(
                    num_batch=len(sizes), num_iter=2, alpha=1.0,
                    batch_method="wobbly_center"),
                cgrad=lambda: tp.CgradOptions(num_batch=len(sizes), num_iter=2,
                                              cg_iter=2))[solver](),
            probe_options=tp.ProbeOptions(force_orthogonality=True),
            object_options=tp.ObjectOptions())
        return tp.reconstruct(data, params, num_gpu=num_gpu, order=order,
                              batches=batches)

    single = run(None)
    monkeypatch.setenv("TIKE_AMD_OVERSUBSCRIBE", "1")
    two = run(2)
    np.testing.assert_allclose(np.array(two.algorithm_options.costs),
                               np.array(single.algorithm_options.costs),
                               rtol=1e-3)
    assert_close(two.psi, single.psi, normwise=1e-3, maxabs=1e-2, what="psi")
    assert_close(two.probe, single.probe, normwise=1e-3, maxabs=1e-2,
                 what="probe")
