"""The one collective of the path through the C-ABI (include/ptudes_mi.h ptl_comm_*, ptl_gather_trajectories):
a communicator of ONE rank on the test box's GPU - ncclGetUniqueId, ncclCommInitRank, ncclAllGather all execute, the rows come back as
the filter kernel wrote them.  (Two ranks need two devices: RCCL refuses duplicate devices; the id's way over a control plane is a
CPU test, tests/test_distributed_gloo.py.)"""
import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import core, parallel, synth

pytestmark = pytest.mark.gpu


def test_one_rank_gather_of_a_batch_and_of_a_single_sequence():
    S, n = 3, 5
    seqs = [synth.make_sequence(seed=1900 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True)
    one = core.SeqRunner(n, seqs[1].H * seqs[1].W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=32, gn_lanes_per_point=8, gn_threads=512)
    for s, sq in enumerate(seqs):
        ends = [sq.imu_range_for_scan(k)[1] for k in range(n)]
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
            if s == 1:
                one.upload_scan(k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu], ends)
        if s == 1:
            one.upload_imu(sq.imu[:n_imu], ends)
    b.run(n - 1)  # (one scan short: the count travels with the rows)
    one.run()
    uid = parallel.Comm.unique_id()
    assert len(uid) == parallel.Comm.ID_BYTES and any(uid)
    comm = parallel.Comm(uid, 1, 0, 0)
    got = comm.gather_batch(b)
    assert sorted(got) == [(0, j) for j in range(S)]
    for j in range(S):
        o = b.results(j)
        rows = got[(0, j)]
        assert rows.shape == (n - 1, 8)
        assert np.array_equal(rows[:, 0], o["res_t"]) and np.array_equal(rows[:, 1:4], o["res_poses"][:, :3, 3])
        t, T = parallel.rows_to_poses(rows)
        assert np.abs(T[:, :3, :3] - o["res_poses"][:, :3, :3]).max() < 1e-12  # (quaternion rows <-> the filter's rotation)
    # the generic entry: any (S, T, 8) device rows - here the single-sequence runner's
    ptr, n_rows = one.traj_device()
    got1 = comm.gather_rows(ptr, 1, n, [n_rows])
    o1 = one.results()
    assert got1[(0, 0)].shape == (n, 8) and np.array_equal(got1[(0, 0)][:, 0], o1["res_t"])
    assert np.array_equal(got1[(0, 0)][: n - 1], got[(0, 1)])  # the batch member and the single run of the same sequence (32 workgroups each)
    with pytest.raises(ValueError):
        comm.gather_rows(ptr, 1, n, [n + 1])  # a count beyond T
    # ... a failure on this rank behind communicator creation ABORTS the communicator (the peers' collective fails instead of waiting for
    # ever); the handle then refuses further calls instead of entering a collective alone
    with pytest.raises(RuntimeError, match="aborted"):
        comm.gather_rows(ptr, 1, n, [n_rows])
    comm.close()
    # a second communicator in the same process (a new id)
    c2 = parallel.Comm(parallel.Comm.unique_id(), 1, 0, 0)
    assert np.array_equal(c2.gather_batch(b)[(0, 2)], got[(0, 2)])
    c2.close()
