"""IMU / NavState types, block helpers and the trajectory metric.

Host-side mirror of reference src/ptudes/ins/data.py (same names, argument meaning and results), written
for this package: per-instance array defaults (the reference shares one class-level ndarray per field,
SURVEY.md App. C1/C2), no ouster-sdk import.  `StreamStatsTracker` (range / IMU running statistics that
are only printed) is not part of the pose path and is not provided.
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
