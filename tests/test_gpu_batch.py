"""Batched runner: S sequences on one GPU, one XCD each (kx_gn_loop: the workgroups with blockIdx & 7 == s run sequence s's
Gauss-Newton loop with one-hop exchange inside the XCD) == S independent device-resident runs with gn_workgroups / 8
workgroups each, bit for bit."""
import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import core, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_imu,S,lanes", [(True, 3, 8), (False, 3, 8), (True, 8, 8), (True, 3, 32), (True, 12, 8), (True, 20, 8)])
def test_batch_equals_independent_runs(use_imu, S, lanes):
    n = 10 if S <= 8 else 6
    seqs = [synth.make_sequence(seed=1010 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=use_imu, with_ekf=True, gn_lanes_per_point=lanes, gn_threads=512 if lanes == 8 else 1024)
    singles = []
    for s, sq in enumerate(seqs):
        # one sequence of the batch runs on 256 / 8 = 32 workgroups (16 / 8 when two / four sequences share an XCD): the
        # independent run it must equal uses as many
        r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=use_imu, with_ekf=True, gn_workgroups=32 if S <= 8 else 16 if S <= 16 else 8,
                           gn_lanes_per_point=lanes, gn_threads=512 if lanes == 8 else 1024)
        ends = [sq.imu_range_for_scan(k)[1] for k in range(n)]
        for k in range(n):
            x = sq.scan(k)
            b.upload_scan(s, k, x)
            r.upload_scan(k, x)
        b.upload_imu(s, sq.imu[:n_imu], ends)
        r.upload_imu(sq.imu[:n_imu], ends)
        r.run()
        singles.append(r.results())
    b.run(4)       # cold start + 4 scans ...
    b.enqueue(n - 4)  # ... then the rest without a reset
    b.wait()
    for s in range(S):
        out = b.results(s)
        assert np.array_equal(out["kiss_poses"], singles[s]["kiss_poses"]), s
        assert np.array_equal(out["res_poses"], singles[s]["res_poses"]), s
        assert np.array_equal(out["res_t"], singles[s]["res_t"])
        assert out["stats"] == singles[s]["stats"]
    # sequences differ from each other (the batch is not computing one thing S times)
    assert not np.array_equal(b.results(0)["kiss_poses"], b.results(1)["kiss_poses"])


def test_batch_rejects_missing_imu():
    b = core.BatchRunner(2, 3, 1024, 4, with_ekf=True, max_points_per_scan=1024, scan_cols=64)
    with pytest.raises(ValueError):
        b.upload_imu(0, np.zeros((4, 7)), [2, 2, 4])  # no IMU between scans 0 and 1


def test_batch_rejects_bad_workgroup_count():
    with pytest.raises(ValueError):
        core.BatchRunner(2, 3, 1024, 4, with_ekf=True, max_points_per_scan=1024, scan_cols=64, gn_workgroups=100)


def test_points_per_voxel_above_32_is_rejected_and_32_matches_the_oracle():
    """the search maps one lane of a 32-lane group to one stored point of a voxel: 33 is refused; at 32 a
    voxel's 32nd point is found"""
    from oracle import cpu as orc
    with pytest.raises(ValueError):
        core.Icp(70.0, 1.0, max_points_per_voxel=33)
    rng = np.random.default_rng(5)
    pts = rng.uniform(0.0, 0.69, (400, 3)) + np.array([7.0, 7.0, 0.7])  # one voxel, far more than 32 candidates
    src = pts[31] + np.array([1e-3, 0.0, 0.0])                          # nearest stored point is the 32nd
    icp = core.Icp(70.0, 1.0, max_points_per_voxel=32)
    icp.map_add(pts)
    m = orc.Map(0.7, 70.0, 32)
    m.add_points(pts)
    assert icp.map_size() == (m.num_voxels, m.num_points) == (1, 32)
    sums, nc, cand = icp.linear_system(src[None], 2.0, 0.5)
    ref = m.linear_system(src[None], 2.0, 0.5)
    assert nc == 1 and cand == 32
    assert np.allclose(sums, ref[0], rtol=1e-12, atol=1e-15)
