"""Pins the CPU oracle's ES-EKF restatement against outputs of the reference itself
(tests/golden/ekf_*.npz, produced by tests/golden/gen_golden.py from
/root/reference/src/ptudes/ins/es_ekf.py)."""
import os

import numpy as np
import pytest

from oracle import cpu as orc

TOL = 1e-9  # EKF teacher-forced |dstate| <= 1e-9, |dP| <= 1e-9 relative


def _quat_close(a, b, tol):
    a = np.asarray(a)
    b = np.asarray(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max()) <= tol


def _nav_close(a, b, tol=TOL):
    assert np.abs(a[:3] - b[:3]).max() <= tol
    assert _quat_close(a[3:7], b[3:7], tol)
    assert np.abs(a[7:] - b[7:]).max() <= tol


@pytest.mark.parametrize("variant", ["default", "init", "cov"])
def test_ekf_step_by_step(golden_dir, variant):
    g = np.load(os.path.join(golden_dir, f"ekf_steps_{variant}.npz"))
    default = variant != "init"
    ekf = orc.EKF(None if default else g["init_grav"], None if default else g["init_bacc"],
                  None if default else g["init_bgyr"])
    assert np.abs(ekf.cov - g["cov0"]).max() <= 1e-12
    upd = {int(i): k for k, i in enumerate(g["upd_idx"])}
    ki = 0
    for i in range(len(g["imu_ts"])):
        ekf.process_imu(g["imu_lacc"][i], g["imu_avel"][i], g["imu_ts"][i])
        _nav_close(ekf.nav, g["nav_after_imu"][i])
        if i % 25 == 0:
            ref = g["cov_imu_every25"][ki]
            assert np.abs(ekf.cov - ref).max() <= TOL * max(1.0, np.abs(ref).max())
            ki += 1
        if i in upd:
            k = upd[i]
            ref = g["cov_pre"][k]
            assert np.abs(ekf.cov - ref).max() <= TOL * max(1.0, np.abs(ref).max())
            ekf.process_pose(g["upd_pose"][k], g["upd_cov"][k] if bool(g["has_cov"]) else None)
            ref = g["cov_post"][k]
            assert np.abs(ekf.cov - ref).max() <= TOL * max(1.0, np.abs(ref).max())
            _nav_close(ekf.nav, g["nav_after_upd"][k])
    assert ekf.ts == float(g["final_ts"])


def test_ekf_bench_sim(golden_dir):
    """`ptudes ekf-bench sim -t 2.0` (reference cli/ekf_bench.py:107-179) replayed on the oracle."""
    g = np.load(os.path.join(golden_dir, "ekf_sim.npz"))
    ekf_gt, ekf = orc.EKF(), orc.EKF()
    ts = g["ts"]
    last_corr, n_upd = ts[0], 0
    gt_poses, poses = [], []
    for i in range(len(ts)):
        ekf_gt.process_imu(g["ideal_lacc"][i], g["ideal_avel"][i], ts[i])
        ekf.process_imu(g["noisy_lacc"][i], g["noisy_avel"][i], ts[i])
        if ts[i] - last_corr > 0.1:
            ekf.process_pose(ekf_gt.pose_mat())
            last_corr = ts[i]
            assert abs(ts[i] - g["upd_ts"][n_upd]) < 1e-12
            n_upd += 1
            gt_poses.append(ekf_gt.pose_mat())
            poses.append(ekf.pose_mat())
    assert n_upd == int(g["n_updates"]) == 19
    _nav_close(ekf_gt.nav, g["nav_gt"])
    _nav_close(ekf.nav, g["nav"])
    assert np.abs(ekf.cov - g["cov"]).max() <= TOL * np.abs(g["cov"]).max()
    ate_r, ate_t = orc.calc_ate(poses, gt_poses)
    assert abs(ate_r - float(g["ate_rot"])) <= 1e-9
    assert abs(ate_t - float(g["ate_trans"])) <= 1e-9


def test_calc_ate(golden_dir):
    g = np.load(os.path.join(golden_dir, "calc_ate.npz"))
    for k in range(3):
        r, t = orc.calc_ate(g[f"a{k}"], g[f"b{k}"])
        assert abs(r - g[f"ate{k}"][0]) <= 1e-9 * max(1.0, g[f"ate{k}"][0])
        assert abs(t - g[f"ate{k}"][1]) <= 1e-9 * max(1.0, g[f"ate{k}"][1])
