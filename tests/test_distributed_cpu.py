"""World-size-2 gloo tests (CPU) of the multi-GPU plumbing: the packed
all-reduce of `Comm`, the scalar all-reduce, and the position sharding of
`Reconstruction._shard` (every global minibatch split contiguously over ranks,
every position owned exactly once)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fn, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _run(fn, world=2):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), fn, ret), nprocs=world,
             join=True)
    return [ret[r] for r in range(world)]


def _allreduce(rank, world):
    from tike_amd.communicators import Comm
    comm = Comm()
    a = torch.full((3, 4), rank + 1.0) * (1 + 2j)
    a = a.to(torch.complex64)
    b = torch.arange(5, dtype=torch.float32) * (rank + 1)
    comm.Allreduce(a, b)
    s = comm.Allreduce_scalars([rank + 1, torch.tensor(2.0 * rank)], "cpu")
    single = comm.Allreduce(torch.ones(2, dtype=torch.complex64) * (rank + 1))
    mx = comm.Allreduce_max(torch.tensor(3.0 - rank))
    rows = comm.Allgather_rows(
        torch.full((2 + rank, 2), float(rank), dtype=torch.float32))
    # a collective started early and waited for later, then a plain one
    early = torch.arange(6, dtype=torch.float32) * (rank + 1)
    handle = comm.Allreduce_start(early[2:])
    late = comm.Allreduce(torch.ones(3) * (rank + 1))
    handle.wait()
    assert torch.equal(early, torch.tensor([0., rank + 1, 6, 9, 12, 15]))
    assert torch.equal(late, torch.full((3,), 3.0))
    return (a.numpy(), b.numpy(), s.numpy(), single.numpy(), comm.size,
            comm.rank, float(mx), rows.numpy())


def test_comm_allreduce_gloo():
    out = _run(_allreduce)
    for r, (a, b, s, single, size, rank, mx, rows) in enumerate(out):
        assert size == 2 and rank == r
        assert mx == 3.0
        # ragged gather: 2 rows of rank 0 then 3 rows of rank 1
        np.testing.assert_array_equal(rows[:, 0], [0, 0, 1, 1, 1])
        np.testing.assert_allclose(a, np.full((3, 4), 3 * (1 + 2j)))
        np.testing.assert_allclose(b, np.arange(5) * 3.0)
        np.testing.assert_allclose(s, [3.0, 2.0])
        np.testing.assert_allclose(single, [3, 3])


def _shard(rank, world):
    import tike_amd.ptycho as tp
    from tike_amd.communicators import Comm
    rng = np.random.default_rng(0)
    N, pw = 50, 8
    scan = (rng.random((N, 2)) * 20 + 2).astype(np.float32)
    params = tp.PtychoParameters(
        probe=np.ones((1, 1, 1, pw, pw), np.complex64),
        psi=np.ones((1, 40, 40), np.complex64), scan=scan,
        algorithm_options=tp.LstsqOptions(num_batch=3,
                                          batch_method="wobbly_center"))
    rec = tp.Reconstruction.__new__(tp.Reconstruction)
    rec._parameters_in = params
    rec._presharded = False
    rec._order_in = rec._batches_in = None
    rec._spatial_sort = True
    rec.comm = Comm()
    order, local, batches = rec._shard(N)
    return order, local, [b.tolist() for b in batches]


def test_position_sharding_covers_every_position_once():
    out = _run(_shard)
    (order0, local0, b0), (order1, local1, b1) = out
    np.testing.assert_array_equal(order0, order1)  # same global clustering
    both = np.concatenate([local0, local1])
    assert sorted(both.tolist()) == list(range(50))
    # local batches are contiguous ranges of the local arrays and every
    # global batch is split (nearly) evenly
    for batches, local in ((b0, local0), (b1, local1)):
        flat = [i for b in batches for i in b]
        assert flat == list(range(len(local)))
    for x, y in zip(b0, b1):
        assert abs(len(x) - len(y)) <= 1


def _shard_unsynchronised_generators(rank, world):
    """Ranks start from DIFFERENT generator states (what separately launched
    processes have); Reconstruction.__enter__ calls Comm.sync_random() before
    sharding, reproduced here."""
    import tike_amd.ptycho as tp
    import tike_amd.random as trandom
    from tike_amd.communicators import Comm
    np.random.seed(100 + rank)
    trandom.randomizer_np = np.random.default_rng(200 + rank)
    rng = np.random.default_rng(0)
    N, pw = 60, 8
    scan = (rng.random((N, 2)) * 20 + 2).astype(np.float32)
    params = tp.PtychoParameters(
        probe=np.ones((1, 1, 1, pw, pw), np.complex64),
        psi=np.ones((1, 40, 40), np.complex64), scan=scan,
        algorithm_options=tp.LstsqOptions(num_batch=4,
                                          batch_method="compact"))
    rec = tp.Reconstruction.__new__(tp.Reconstruction)
    rec._parameters_in = params
    rec._presharded = False
    rec._order_in = rec._batches_in = None
    rec._spatial_sort = True
    rec.comm = Comm()
    rec.comm.sync_random()
    order, local, batches = rec._shard(N)
    draws = (trandom.randomizer_np.permutation(7).tolist(),
             np.random.rand(3).tolist())
    return order, local, [b.tolist() for b in batches], draws


def test_ranks_with_different_seeds_share_one_clustering():
    (o0, l0, b0, d0), (o1, l1, b1, d1) = _run(_shard_unsynchronised_generators)
    np.testing.assert_array_equal(o0, o1)
    assert sorted(np.concatenate([l0, l1]).tolist()) == list(range(60))
    assert d0 == d1  # later draws (minibatch permutation, RANSAC) agree too


def test_injected_batches_are_validated():
    from tike_amd.ptycho.ptycho import _check_batches
    order = np.random.default_rng(0).permutation(10)
    _check_batches(order, np.array_split(np.arange(10), 3), 10)
    with pytest.raises(ValueError):  # reference-style index batches
        _check_batches(order, [order[:5], order[5:]], 10)
    with pytest.raises(ValueError):  # not a tiling
        _check_batches(order, [np.arange(0, 4), np.arange(5, 10)], 10)
    with pytest.raises(ValueError):
        _check_batches(np.arange(9), [np.arange(10)], 10)


def test_single_rank_comm_is_identity():
    from tike_amd.communicators import Comm
    comm = Comm()
    assert comm.size == 1 and comm.rank == 0
    t = torch.ones(3)
    assert comm.Allreduce(t) is t
    np.testing.assert_allclose(
        comm.Allreduce_scalars([1.5, torch.tensor(2.0)], "cpu").numpy(),
        [1.5, 2.0])


def _resolve_in_group(rank, world):
    from tike_amd.ptycho import _spawn
    out = [_spawn.resolve(None), _spawn.resolve(world), _spawn.resolve((0, 1))]
    try:
        _spawn.resolve(1)
    except ValueError as e:
        out.append(str(e))
    return out


def test_num_gpu_is_honoured_or_refused(monkeypatch):
    """`num_gpu` (reference ptycho.py:182-187): left out = this process / this
    job; N from a plain process = N spawned ranks; anything that contradicts
    the running process group raises."""
    from tike_amd.ptycho import _spawn
    assert _spawn.requested_devices(None) is None
    assert _spawn.requested_devices(3) == (0, 1, 2)
    assert _spawn.requested_devices((4, 2)) == (4, 2)
    with pytest.raises(ValueError):
        _spawn.requested_devices(0)
    with pytest.raises(ValueError):
        _spawn.requested_devices((0, -1))
    # plain process
    assert _spawn.resolve(None) == ("here", None)
    assert _spawn.resolve(1) == ("here", None)
    assert _spawn.resolve((3,)) == ("here", 3)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert _spawn.resolve(4) == ("spawn", (0, 1, 2, 3))
    assert _spawn.resolve((6, 7)) == ("spawn", (6, 7))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    with pytest.warns(UserWarning, match="only 2 GPU"):
        assert _spawn.resolve(4) == ("spawn", (0, 1))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    with pytest.warns(UserWarning):
        assert _spawn.resolve(4) == ("here", 0)
    monkeypatch.setenv("TIKE_AMD_OVERSUBSCRIBE", "1")
    assert _spawn.resolve(2) == ("spawn", (0, 0))
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 0)
    with pytest.raises(RuntimeError, match="GPU"):
        _spawn.resolve(2)
    # inside a two-rank job
    for out in _run(_resolve_in_group):
        assert out[:3] == [("here", None)] * 3
        assert "world" in out[3] or "job of 2" in out[3]
