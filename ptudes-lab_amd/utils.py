"""Pose-file surface of the path: KITTI / Newer-College-GT writers and reader, timestamp matching.

Host-side mirror of the pose-file part of reference src/ptudes/utils.py (:22-36, :191-341): same
function names, argument meaning and byte-for-byte identical files, so `ptudes flyby --kitti-poses /
--nc-gt-poses` and `ptudes ekf-bench cmp` keep consuming what this package writes.  The viz / packet-IO
helpers of that file (PointViz, pcap/bag openers) are outside the pose path and are not provided.
"""
from typing import List, Optional, Tuple

import numpy as np
from scipy.spatial.transform import Rotation

# Newer College 2021 extrinsics (reference utils.py:20-26): IMU -> sensor, sensor -> base, both pure translations
NC_OS_IMU_TO_OS_SENSOR = np.eye(4)
NC_OS_IMU_TO_OS_SENSOR[:3, 3] = [-0.014, 0.012, 0.015]
NC_OS_SENSOR_TO_BASE = np.eye(4)
NC_OS_SENSOR_TO_BASE[:3, 3] = [0.001, 0.000, 0.091]
NC_OS_IMU_TO_BASE = NC_OS_SENSOR_TO_BASE @ NC_OS_IMU_TO_OS_SENSOR


def read_metadata_json(meta_path: str):
    """SensorInfo from a metadata .json (reference utils.py:157-168, including the lidar_mode back-fill for the
    Newer College 2020 beam_intrinsics files).  Needs ouster-sdk."""
    import json
    from ouster import client  # guarded: raises ImportError where ouster-sdk is absent
    with open(meta_path) as f:
        js = json.loads(f.read())
    if "beam_altitude_angles" in js and "beam_azimuth_angles" in js and "lidar_mode" not in js:
        print(f"WARNING: lidar_mode is not present in legacy metadata '{meta_path}' so using lidar_mode: 1024x10")
        js["lidar_mode"] = "1024x10"
    return client.SensorInfo(json.dumps(js))


def read_packet_source(file_path: str, meta=None):
    """Open PCAP or BAG based Ouster raw packet source (reference utils.py:171-187): a .pcap through ouster-sdk, a .bag
    or a directory of .bag files (sorted by name) through `bag.OusterRawBagSource`.  Both need ouster-sdk for the
    packet types; like the reference, anything else yields None."""
    import glob
    from pathlib import Path
    file = Path(file_path)
    if file.is_file():
        if file.suffix == ".pcap":
            from ouster import pcap  # guarded
            return pcap.Pcap(file_path, meta)
        elif file.suffix == ".bag":
            from .bag import OusterRawBagSource
            return OusterRawBagSource(file, meta)
    elif file.is_dir():
        from .bag import OusterRawBagSource
        bags_paths = sorted([Path(p) for p in glob.glob(str(Path(file) / "*.bag"))])
        return OusterRawBagSource(bags_paths, meta)


def vee(vec: np.ndarray) -> np.ndarray:
    """3-vector -> skew-symmetric matrix, hat(v) w = v x w (reference utils.py:28-36 names it `vee`)"""
    x, y, z = vec[0], vec[1], vec[2]
    return np.array([[0.0, -z, y], [z, 0.0, -x], [-y, x, 0.0]])


def save_poses_kitti_format(filename: str, poses: List[np.ndarray], header: str = ""):
    """One row per pose: the top three rows of the 4x4, 12 numbers (reference utils.py:191-196)"""
    rows = np.array([np.asarray(p)[:3, :].reshape(12) for p in poses])
    np.savetxt(fname=filename, X=rows, header=header)


def save_poses_nc_gt_format(filename: str, t: List[float], poses: List[np.ndarray], header: str = ""):
    """Newer College ground-truth CSV: sec, nsec, x, y, z, qx, qy, qz, qw in the BASE frame; incoming poses
    are in the IMU (nav) frame (reference utils.py:199-228)."""
    t_arr = np.asarray(t, dtype=np.float64)
    base = np.einsum("nij,jk->nik", np.asarray(poses, dtype=np.float64), np.linalg.inv(NC_OS_IMU_TO_BASE))
    out = np.zeros((len(t_arr), 9))
    out[:, 0] = np.floor(t_arr)
    out[:, 1] = np.floor((t_arr - out[:, 0]) * 1e+9)
    out[:, 2:5] = base[:, :3, 3]
    out[:, 5:9] = Rotation.from_matrix(base[:, :3, :3]).as_quat()
    if header:
        header += "\n\n" + "sec,nsec,x,y,z,qx,qy,qz,qw"
    np.savetxt(fname=filename, X=out, delimiter=", ", header=header)


def read_newer_college_gt(data_path: str, to_os_imu: bool = True) -> List[Tuple[float, np.ndarray]]:
    """[(ts, pose4x4)] from a Newer College GT CSV, moved to the Ouster IMU nav frame (reference utils.py:231-252)"""
    d = np.loadtxt(data_path, delimiter=",")
    d = d.reshape(-1, 9) if d.ndim == 1 else d
    ts = d[:, 0] + d[:, 1] * 1e-9
    T = np.tile(np.eye(4), (len(d), 1, 1))
    T[:, :3, 3] = d[:, 2:5]
    T[:, :3, :3] = Rotation.from_quat(d[:, 5:9]).as_matrix()
    if to_os_imu:
        T = np.einsum("nij,jk->nik", T, NC_OS_IMU_TO_BASE)
    return list(zip(ts, T))


class _Cursor:
    """forward-only reader that raises StopIteration when exhausted (the reference walks Python iterators)"""

    def __init__(self, items):
        self.items, self.i = items, 0

    def take(self):
        if self.i >= len(self.items):
            raise StopIteration
        self.i += 1
        return self.items[self.i - 1]


def filter_nc_gt_by_close_ts(nc_gt, gt_t):
    """Greedy nearest-timestamp pairing of two non-decreasing streams (reference utils.py:255-302), including
    its corner behaviour: a candidate that loses the "closer than my successor" test is dropped, and the
    walk stops as soon as either stream runs out."""
    if not len(nc_gt):
        return nc_gt
    if not len(gt_t):
        return []
    nc_t = np.array([g[0] for g in nc_gt])
    tol = min(np.min(nc_t[1:] - nc_t[:-1]), np.min(np.array(gt_t[1:]) - np.array(gt_t[:-1])))
    a, b = _Cursor(list(nc_gt)), _Cursor(list(gt_t))
    out_gt, out_t = [], []
    try:
        n, g = a.take(), b.take()
        while True:
            while abs(n[0] - g) > tol:
                while n[0] < g - tol:
                    n = a.take()
                while g < n[0] - tol:
                    g = b.take()
            if n[0] < g:
                n2 = a.take()
                if abs(n[0] - g) < abs(n2[0] - g):
                    out_gt.append(n)
                    out_t.append(g)
                    n = n2
                    g = b.take()
            else:
                g2 = b.take()
                if abs(n[0] - g) < abs(n[0] - g2):
                    out_gt.append(n)
                    out_t.append(g)
                    n = a.take()
                    g = g2
    except StopIteration:
        pass
    return out_gt, out_t


def filter_nc_gt_by_cmp(nc_gt, nc_gt_cmp):
    """Closest subset of nc_gt_cmp in nc_gt, as two equally long [(t, pose)] lists (reference utils.py:305-325)"""
    cmp_t = [g[0] for g in nc_gt_cmp]
    matched_gt, matched_t = filter_nc_gt_by_close_ts(nc_gt, cmp_t)
    poses, idx = [], 0
    for t in matched_t:
        while cmp_t[idx] != t:
            idx += 1
        poses.append(nc_gt_cmp[idx][1])
        idx += 1
    assert len(poses) == len(matched_t)
    return matched_gt, list(zip(matched_t, poses))


def active_beam_rows(h: int, beams_num: int) -> np.ndarray:
    """rows kept by `reduce_active_beams` (reference utils.py:336): beams_num uniformly spread rows of h"""
    return np.linspace(0, h, num=beams_num, endpoint=False, dtype=int)


def reduce_active_beams(scan, beams_num: int):
    """Zero the RANGE (or xyz) of every row outside `active_beam_rows` (reference utils.py:328-341).
    `scan` is a (H, W) range image, a (H, W, 3) xyz image, or an ouster LidarScan."""
    if hasattr(scan, "field") and hasattr(scan, "h"):  # ouster LidarScan, only when ouster-sdk is present
        from ouster import client  # noqa: WPS433 (guarded import)
        arr, h = scan.field(client.ChanField.RANGE), scan.h
    else:
        arr, h = scan, scan.shape[0]
    drop = np.ones(h, dtype=bool)
    drop[active_beam_rows(h, beams_num)] = False
    arr[drop] = 0
    return scan


class TrajectoryEvaluator:
    """Poses along a time-stamped trajectory: the part of ouster.sdk.pose_util.TrajectoryEvaluator the reference uses
    (utils.py:368 `pose_scans_from_nc_gt`, cli/ekf_bench.py:489 / :537 `--use-gt-guess`; third-party, behaviour from its
    published description): SE(3) geodesic between the bracketing knots, the end segments extended by `time_bounds`
    seconds (a number, or (before, after)), ValueError further out.  The interpolation runs on the GPU
    (`ptl_traj_poses_at`): a sweep asks for 1024 or 2048 poses at once."""

    def __init__(self, poses, time_bounds=0.0, device_id: int = 0):
        poses = list(poses)
        if len(poses) < 2:
            raise ValueError("a trajectory needs at least two (ts, pose) knots")
        self._ts = np.array([p[0] for p in poses], dtype=np.float64)
        self._poses = np.array([np.asarray(p[1], dtype=np.float64) for p in poses])
        self._bounds = (float(time_bounds), float(time_bounds)) if np.isscalar(time_bounds) else tuple(map(float, time_bounds))
        self._device_id = device_id

    def poses_at(self, ts) -> np.ndarray:
        from . import core
        ts = np.atleast_1d(np.asarray(ts, dtype=np.float64))
        out, outside = core.traj_poses_at(self._ts, self._poses, ts, self._bounds[0], self._bounds[1], self._device_id)
        if outside:
            raise ValueError(f"{outside} of {len(ts)} timestamps lie outside the trajectory "
                             f"({self._ts[0]} .. {self._ts[-1]}, bounds {self._bounds})")
        return out

    def pose_at(self, ts: float) -> np.ndarray:
        return self.poses_at([ts])[0]

    def __call__(self, scan, col_ts=None):
        """write the pose of every column of `scan` (ts of its columns in seconds) into scan.pose, like upstream"""
        col_ts = np.asarray(scan.timestamp, dtype=np.float64) * 1e-9 if col_ts is None else np.asarray(col_ts, dtype=np.float64)
        scan.pose[:] = self.poses_at(col_ts)


def pose_scans_from_nc_gt(source, nc_gt_poses_file: Optional[str] = None, nc_gt_poses=None, device_id: int = 0):
    """Give every scan of `source` the ground-truth pose of each of its columns (reference utils.py:344-392): scans whose
    columns are not all within 1.5 s of the trajectory are left out, and counted in the closing note.  A scan is
    anything with `timestamp` (W ns) and `pose` (W, 4, 4) - an ouster LidarScan, or `fly.PosedScan`."""
    gts = read_newer_college_gt(nc_gt_poses_file) if nc_gt_poses_file else nc_gt_poses
    traj_eval = TrajectoryEvaluator(gts, time_bounds=1.5, device_id=device_id)
    skipped_scans = 0
    for scan in source:
        if not (hasattr(scan, "timestamp") and hasattr(scan, "pose")):
            continue
        try:
            traj_eval(scan, col_ts=np.asarray(scan.timestamp, dtype=np.float64) * 1e-9)
        except ValueError:
            skipped_scans += 1
            continue
        yield scan
    print(f"NOTE: Therere where {skipped_scans} skipped scans that wasn't "
          "because they were outside of the NC GT poses available")
