"""Batched runner: S sequences in lockstep on one GPU == S independent device-resident runs, bit for bit."""
import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import core, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_imu", [True, False])
def test_batch_equals_independent_runs(use_imu):
    S, n = 3, 10
    seqs = [synth.make_sequence(seed=1010 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=use_imu, with_ekf=True)
    singles = []
    for s, sq in enumerate(seqs):
        r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=use_imu, with_ekf=True)
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
