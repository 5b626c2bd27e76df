"""IMU / NavState types, block helpers and the trajectory metric.

Host-side mirror of reference src/ptudes/ins/data.py (same names, argument meaning and results), written
for this package: per-instance array defaults (the reference shares one class-level ndarray per field,
SURVEY.md App. C1/C2), no ouster-sdk import.  `StreamStatsTracker` (reference :207-369)
reduces each scan's range image on the device (`ptl_range_stats`) and pools the moments on the host.
"""
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np
from scipy.spatial.transform import Rotation

GRAV = 9.782940329221166  # reference ins/data.py:10


def _zeros3():
    return np.zeros(3)


@dataclass
class IMU:
    """One inertial sample: specific force (m/s^2), angular rate (rad/s), time (s).  reference :12-31"""
    lacc: np.ndarray = field(default_factory=_zeros3)
    avel: np.ndarray = field(default_factory=_zeros3)
    ts: float = 0
    dt: float = 0

    @staticmethod
    def from_packet(imu_packet, dt: float = 0.01, _intr_rot: Optional[np.ndarray] = None) -> "IMU":
        """Ouster ImuPacket (accel in g, angular_vel in deg/s, sys_ts in ns) -> SI units (reference :18-31)"""
        lacc = GRAV * np.asarray(imu_packet.accel, dtype=np.float64)
        avel = np.pi * np.asarray(imu_packet.angular_vel, dtype=np.float64) / 180.0
        if _intr_rot is not None:
            lacc = _intr_rot @ lacc
            avel = _intr_rot @ avel
        return IMU(lacc=lacc, avel=avel, ts=imu_packet.sys_ts / 10**9, dt=dt)


@dataclass
class NavState:
    """Navigation state; the attitude is STORED as an xyzw quaternion (reference :34-104)"""
    pos: np.ndarray = field(default_factory=_zeros3)
    att_q: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0, 1.0]))
    vel: np.ndarray = field(default_factory=_zeros3)
    bias_gyr: np.ndarray = field(default_factory=_zeros3)
    bias_acc: np.ndarray = field(default_factory=_zeros3)
    grav: np.ndarray = field(default_factory=lambda: GRAV * np.array([0.0, 0.0, -1.0]))
    update: bool = False
    # log-only attachments used by the reference's plotting code
    cov: Optional[np.ndarray] = None
    scan: Optional[object] = None
    kiss_pose: Optional[np.ndarray] = None

    @staticmethod
    def from_vector(v) -> "NavState":
        """19 doubles as the C-ABI returns them: pos, quat_xyzw, vel, bias_gyr, bias_acc, grav"""
        v = np.asarray(v, dtype=np.float64)
        return NavState(pos=v[0:3].copy(), att_q=v[3:7].copy(), vel=v[7:10].copy(), bias_gyr=v[10:13].copy(),
                        bias_acc=v[13:16].copy(), grav=v[16:19].copy())

    def pose_mat(self) -> np.ndarray:
        T = np.eye(4)
        T[:3, :3] = self.att_h
        T[:3, 3] = self.pos
        return T

    @property
    def att_h(self) -> np.ndarray:
        return Rotation.from_quat(self.att_q).as_matrix()

    @att_h.setter
    def att_h(self, val: np.ndarray):
        self.att_q = Rotation.from_matrix(val).as_quat()

    @property
    def att_v(self) -> np.ndarray:
        return Rotation.from_quat(self.att_q).as_rotvec()

    @att_v.setter
    def att_v(self, val: np.ndarray):
        self.att_q = Rotation.from_rotvec(val).as_quat()

    def __repr__(self) -> str:
        tag = " (S)" if self.scan else ""
        return (f"NavState{tag}:\n  pos: {self.pos}\n  vel: {self.vel}\n  att_v: {self.att_v}\n"
                f"  bg: {self.bias_gyr}\n  ba: {self.bias_acc}\n  grav: {self.grav}\n")


def set_blk(m: np.ndarray, row_id: int, col_id: int, b: np.ndarray) -> np.ndarray:
    r, c = b.shape
    m[row_id:row_id + r, col_id:col_id + c] = b
    return m


def blk(m: np.ndarray, row_id: int, col_id: int, nrows: int, ncols: Optional[int] = None) -> np.ndarray:
    return m[row_id:row_id + nrows, col_id:col_id + (nrows if ncols is None else ncols)]


def calc_ate(navs_poses, gt_poses) -> Tuple[float, float]:
    """Reference-style "ATE" (reference :124-153): trajectories aligned on their first pose; returns
    (mean of squared rotation angles * 180/pi, mean of squared translation distances [m^2]).
    NOTE: these are mean squares, not RMSE; see `rmse_ate` for the usual metric."""
    a = np.asarray(navs_poses, dtype=np.float64)
    g = np.asarray(gt_poses, dtype=np.float64)
    assert len(a) == len(g)
    assert len(a)
    g = (a[0] @ np.linalg.inv(g[0])) @ g
    dt = np.linalg.norm(g[:, :3, 3] - a[:, :3, 3], axis=1)
    rel = np.einsum("nji,njk->nik", a[:, :3, :3], g[:, :3, :3])
    dr = np.linalg.norm(Rotation.from_matrix(rel).as_rotvec(), axis=1)
    ate_t = float(np.sum(np.square(dt)) / len(dt))
    ate_r = float(np.sum(np.square(dr)) / len(dr)) * 180 / np.pi
    return ate_r, ate_t


def rmse_ate(navs_poses, gt_poses) -> Tuple[float, float]:
    """(rotation RMSE [rad], translation RMSE [m]) with the same first-pose alignment"""
    r, t = calc_ate(navs_poses, gt_poses)
    return float(np.sqrt(r * np.pi / 180)), float(np.sqrt(t))


def calc_ate_from_navs(navs, gt_poses) -> Tuple[float, float]:
    return calc_ate([n.pose_mat() for n in navs], gt_poses)


def _collect_navs_from_gt(ekf_gt, ekf) -> Tuple[List, List, List]:
    """Pair every pose-update knot of `ekf` with the state `ekf_gt` logged at the same time (reference :170-193)"""
    gt_by_t = {}
    for n, t in zip(ekf_gt._navs, ekf_gt._navs_t):
        gt_by_t[t] = n  # the reference walks backwards and stops at the first match = the LAST entry with that t
    ts, navs_gt, navs = [], [], []
    for idx in ekf._nav_update_idxs:
        t = ekf._navs_t[idx]
        ts.append(t)
        navs.append(ekf._navs[idx])
        navs_gt.append(gt_by_t[t])
    return ts, navs_gt, navs


def ekf_traj_ate(ekf_gt, ekf):
    """ATE between two filters at the update knots (reference :196-204)"""
    _, navs_gt, navs = _collect_navs_from_gt(ekf_gt, ekf)
    return calc_ate([n.pose_mat() for n in navs], [n.pose_mat() for n in navs_gt])


class StreamStatsTracker:
    """Tracks the mean / std of the scans' ranges and of the IMU samples (reference ins/data.py:207-369, same
    methods, properties and report).  `trackScan` takes an Ouster LidarScan (when ouster-sdk is importable), or a
    (H, W) uint32 range image with the last valid column timestamp in ns; the per-scan count / mean / variance /
    min / max come from the HIP reduction `ptl_range_stats`, the pooling formula is the reference's (:310-316)."""

    def __init__(self, use_beams_num: Optional[int] = None, metadata=None, device_id: int = 0):
        self._metadata, self._device_id, self._use_beams_num = metadata, device_id, use_beams_num
        # ranges: pooled mean / variance over all valid returns seen so far, and their extremes
        self._mean = self._sigma_sq = 0
        self._min_range = self._max_range = 0
        self._points_num = self._scans_num = 0
        # IMU: running mean and sum of squared deviations per axis (rows: accelerometer, gyroscope)
        self._imu_mean = np.zeros((2, 3))
        self._imu_m2 = np.zeros((2, 3))
        self._imu_num = 0
        self._t_span = None  # [first, last] timestamp of anything tracked

    # mm -> m; the 15-bit low-data-rate profile stores ranges in units of 8 mm (:242-252)
    def _range_to_m(self) -> float:
        profile = str(getattr(getattr(self._metadata, "format", None), "udp_profile_lidar", ""))
        return 0.008 if profile.endswith("RNG15_RFL8_NIR8") else 0.001

    def _stretch_span(self, ts: float):  # :254-260
        self._t_span = [ts, ts] if self._t_span is None else [min(self._t_span[0], ts), max(self._t_span[1], ts)]

    def trackImu(self, imu: IMU):
        """Update IMU mean / sigma (:266-282): Welford's update, accelerometer and gyroscope side by side"""
        x = np.stack([imu.lacc, imu.avel])
        before = self._imu_mean.copy()
        self._imu_mean += (x - self._imu_mean) / (self._imu_num + 1)
        self._imu_m2 += (x - before) * (x - self._imu_mean)
        self._stretch_span(imu.ts)
        self._imu_num += 1

    def trackScan(self, ls, last_valid_column_ts_ns: Optional[int] = None):
        """Update the range mean / sigma with one scan (:284-321)"""
        import ctypes as C
        from .. import _lib as L
        if hasattr(ls, "field") and hasattr(ls, "h"):  # ouster LidarScan, only when ouster-sdk is present
            from ouster import client  # noqa: WPS433 (guarded import)
            rng = np.ascontiguousarray(ls.field(client.ChanField.RANGE), dtype=np.uint32)
            if last_valid_column_ts_ns is None:
                last_valid_column_ts_ns = client.last_valid_column_ts(ls)
        else:
            rng = np.ascontiguousarray(ls, dtype=np.uint32)
        if rng.ndim != 2:
            raise ValueError("range image must be (H, W)")
        if last_valid_column_ts_ns is None:
            raise ValueError("a raw range image needs last_valid_column_ts_ns")
        out = np.zeros(5)
        L.check(L.lib().ptl_range_stats(self._device_id, rng.ctypes.data_as(C.c_void_p), rng.shape[0], rng.shape[1],
                                        int(self._use_beams_num or 0), self._range_to_m(), L.dptr(out)))
        n, m, v, lo, hi = int(out[0]), out[1], out[2], out[3], out[4]
        if n == 0:
            raise ValueError("scan has no valid returns")  # the reference divides by zero here
        seen = self._points_num
        if seen:  # :262-264
            lo, hi = min(self._min_range, lo), max(self._max_range, hi)
        self._min_range, self._max_range = lo, hi
        # pooled variance of two samples (:310-316): within-sample parts plus the term for the distance of the means
        within = (seen - 1) * self._sigma_sq if seen else 0
        between = seen * n * np.square(self._mean - m) / ((seen + n) * (seen + n - 1))
        self._sigma_sq = (within + n * v) / (seen + n - 1) + between
        self._mean = (self._mean * seen + m * n) / (seen + n)
        self._stretch_span(last_valid_column_ts_ns * 1e-9)
        self._scans_num += 1
        self._points_num = seen + n

    range_mean = property(lambda self: self._mean)
    range_std = property(lambda self: np.sqrt(self._sigma_sq))
    acc_mean = property(lambda self: self._imu_mean[0])
    gyr_mean = property(lambda self: self._imu_mean[1])
    acc_std = property(lambda self: np.sqrt(self._imu_m2[0] / self._imu_num))
    gyr_std = property(lambda self: np.sqrt(self._imu_m2[1] / self._imu_num))
    dt = property(lambda self: 0 if self._t_span is None else self._t_span[1] - self._t_span[0])

    def _formatted_str(self) -> str:  # the reference's report, line for line (:351-366)
        lo3 = max(self._min_range, self.range_mean - 3 * self.range_std)
        hi3 = min(self._max_range, self.range_mean + 3 * self.range_std)
        rows = [f"StreamStatsTracker[dt: {self.dt:.04f} s, imus: {self._imu_num}, scans: {self._scans_num}]:",
                f"  range_mean: {self.range_mean:.03f} m,",
                f"  range_std: {self.range_std:.03f} m (s3 span: [{lo3:.03f} - {hi3:.03f} m])",
                f"  range min max: {self._min_range:.03f} - {self._max_range:.03f} m",
                f"  acc_mean: {self.acc_mean} m/s^2",
                f"  acc_std: {self.acc_std}",
                f"  gyr_mean: {self.gyr_mean} rad/s",
                f"  gyr_std: {self.gyr_std}"]
        return "\n".join(rows)

    def __repr__(self):
        return self._formatted_str()
