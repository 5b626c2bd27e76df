"""CPU restatement of the posed-scan arithmetic of the map fly-by.  TEST INFRASTRUCTURE ONLY
(only tests/ imports it).

* `poses_at`: ouster.sdk.pose_util.TrajectoryEvaluator.poses_at as the reference uses it (reference utils.py:368,
  cli/ekf_bench.py:489, :537): third-party and absent from /root/reference and from this image, so this is the
  published behaviour from memory - parity unpinned - the SE(3) geodesic between the bracketing knots,
  P_i Exp(alpha Log(P_i^-1 P_i+1)), the end segments extended by `time_bounds`, ValueError further out.  Pinned only by
  algebraic known answers (tests/test_dewarp.py: knots reproduce themselves, a constant twist is followed exactly, pure
  translations interpolate linearly).
* `dewarp`: ouster client.dewarp(xyz, column_poses=...): every pixel with its column's pose (no masking).
"""
import numpy as np
from scipy.linalg import expm, logm


def _log(T):
    return np.real(logm(T))


def poses_at(knots, ts, time_bounds=0.0):
    """knots: list of (t, 4x4); ts: (n,) -> (n, 4, 4)"""
    kt = np.array([k[0] for k in knots], dtype=np.float64)
    kp = [np.asarray(k[1], dtype=np.float64) for k in knots]
    tb = (time_bounds, time_bounds) if np.isscalar(time_bounds) else time_bounds
    out = np.empty((len(ts), 4, 4))
    for j, t in enumerate(np.asarray(ts, dtype=np.float64)):
        if t < kt[0] - tb[0] or t > kt[-1] + tb[1]:
            raise ValueError(f"timestamp {t} outside the trajectory ({kt[0]} .. {kt[-1]}, bounds {tb})")
        i = int(np.clip(np.searchsorted(kt, t, side="right") - 1, 0, len(kt) - 2))
        alpha = (t - kt[i]) / (kt[i + 1] - kt[i])
        out[j] = kp[i] @ np.real(expm(alpha * _log(np.linalg.inv(kp[i]) @ kp[i + 1])))
    return out


def dewarp(xyz_hw3, col_poses):
    """xyz_hw3: (H, W, 3) sensor-frame points; col_poses: (W, 4, 4) -> (H, W, 3)"""
    R, t = col_poses[:, :3, :3], col_poses[:, :3, 3]
    return np.einsum("wij,hwj->hwi", R, xyz_hw3) + t[None, :, :]
