"""N > 1 path on CPU: two gloo ranks shard sequences, gather padded trajectories, agree on the max time."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import parallel


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_rows(seq_id, n):
    rng = np.random.default_rng(sequence_seed := parallel.sequence_seed(seq_id))
    rows = np.zeros((n, 8))
    rows[:, 0] = 1000.0 + 0.1 * np.arange(n)
    rows[:, 1:4] = np.cumsum(rng.normal(0, 0.1, (n, 3)), axis=0)
    q = rng.normal(0, 1, (n, 4))
    rows[:, 4:8] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return rows


def _worker(rank, world, port, n_seq, t_max, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = parallel.shard_sequences(n_seq, rank, world)
    rows = torch.zeros((len(mine), t_max, 8), dtype=torch.float64)
    counts = []
    for j, s in enumerate(mine):
        n = 5 + 3 * s  # ragged lengths
        rows[j, :n] = torch.from_numpy(_fake_rows(s, n))
        counts.append(n)
    got = parallel.gather_trajectories(rows, counts, dist)
    mx = parallel.max_over_ranks(0.5 + rank, dist)
    ok = mx == 0.5 + (world - 1)
    for r in range(world):
        for j, s in enumerate(parallel.shard_sequences(n_seq, r, world)):
            ok &= np.array_equal(got[(r, j)], _fake_rows(s, 5 + 3 * s))
    q.put((rank, bool(ok), len(got)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather():
    world, n_seq, t_max = 2, 4, 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_seq, t_max, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == n_seq for r in res)


def test_sharding_and_rows():
    assert parallel.shard_sequences(8, 3, 8) == [3]
    assert parallel.shard_sequences(8, 1, 2) == [1, 3, 5, 7]
    assert sorted(sum((parallel.shard_sequences(10, r, 4) for r in range(4)), [])) == list(range(10))
    rows = _fake_rows(0, 6)
    t, T = parallel.rows_to_poses(rows)
    assert np.array_equal(t, rows[:, 0]) and np.allclose(T[:, :3, 3], rows[:, 1:4])
    assert np.allclose(np.einsum("nij,nkj->nik", T[:, :3, :3], T[:, :3, :3]), np.eye(3))


def _id_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    made = []

    def make():  # stands in for ptl_comm_unique_id (which needs a GPU): only rank 0 may be asked
        made.append(rank)
        return bytes((7 * i + 3) % 256 for i in range(parallel.Comm.ID_BYTES))

    got = parallel.broadcast_id(dist, dist.group.WORLD, make)
    # the communicator itself needs a HIP device: on this box the constructor must fail loudly, not fall back to anything
    err = None
    try:
        parallel.Comm(got, world, rank, 0)
    except Exception as e:  # noqa: BLE001
        err = str(e)
    q.put((rank, got.hex(), made, err))
    dist.barrier()
    dist.destroy_process_group()


def test_communicator_id_travels_over_the_control_plane():
    """the library does no bootstrap: rank 0 makes the 128-byte id, the control plane (gloo here) carries it, every rank
    constructs ptl_comm with the same bytes; without a HIP device the constructor raises (no CPU stand-in for the gather)"""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = bytes((7 * i + 3) % 256 for i in range(128)).hex()
    assert [r[1] for r in res] == [want, want]
    assert res[0][2] == [0] and res[1][2] == []  # made on rank 0 only
    if not torch.cuda.is_available():
        assert all(r[3] and "HIP device" in r[3] for r in res), res
    import pytest
    with pytest.raises(ValueError):
        parallel.Comm(b"short", 1, 0)


def _id_fail_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    idg = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=60))  # the id's own group, as bench.py does

    def make():
        raise OSError("librccl.so could not be opened")

    err = None
    try:
        parallel.broadcast_id(dist, idg, make)
    except RuntimeError as e:
        err = str(e)
    # the control plane is still in step on every rank: the collective that follows (bench.py: has anybody failed?) completes
    worst = parallel.max_over_ranks(1.0 if err else 0.0, dist, device="cpu", group=dist.group.WORLD)
    q.put((rank, err, worst))
    dist.barrier()
    dist.destroy_process_group()


def test_a_failing_id_on_rank_0_raises_on_every_rank_and_leaves_the_control_plane_in_step():
    """rank 0's make_id() raising (librccl cannot be opened - the case bench.py's gloo fallback exists for) used to leave the
    other ranks blocked in the broadcast while rank 0 went on to an all_reduce of the same group.  The error now travels IN the broadcast:
    every rank raises together, and the next collective of the control plane lines up."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_fail_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] and "librccl.so could not be opened" in r[1] for r in res), res
    assert [r[2] for r in res] == [1.0, 1.0]
