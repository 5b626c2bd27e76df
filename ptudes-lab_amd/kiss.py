"""KissICPWrapper: drop-in for reference src/ptudes/kiss.py:18-166, computed on the GPU.

Same constructor, `register_frame(scan, initial_guess)`, properties (`pose`, `poses`, `poses_ts`, `velocity`,
`local_map_points`, `_config`) and the members the reference's driver reads directly (`_kiss.poses`,
`_kiss.get_prediction_model()`, `_err_dt`, `_err_drot`, `_sigmas`, `_poses_ts`).  The whole per-scan pipeline
(kiss.py:83-131: deskew, range filter, two voxel downsamplings, adaptive threshold, ICP, map update) is one
call into `ptl_icp_register_frame` (include/ptudes_mi.h).

Scans: an Ouster `LidarScan` when ouster-sdk is importable (the reference's only input type), or - since that
SDK may be absent - any object with `.xyz` ((H, W, 3) or (N, 3)) and optionally `.range` / `.ts`, or a plain
(H*W, 3) / (N, 3) array.  `register_points` takes what `_kiss_register_frame` takes.
"""
from types import SimpleNamespace
from typing import List, Optional

import numpy as np

from . import core


class _KissState:
    """the two members of kiss_icp.KissICP the reference's driver touches (ekf_bench.py:545-547)"""

    def __init__(self, icp: core.Icp, cfg):
        self._icp, self.config = icp, cfg
        self.poses: List[np.ndarray] = []

    def get_prediction_model(self) -> np.ndarray:
        if len(self.poses) < 2:
            return np.eye(4)
        return self._icp.prediction()


class KissICPWrapper:
    def __init__(self, metadata, *, _min_range: float = 5, _max_range: float = 100, _use_extrinsics: bool = False,
                 device_id: int = 0, lazy_map_stats: bool = False, **icp_over):
        self._metadata = metadata
        self._use_extrinsics = _use_extrinsics
        w = metadata.format.columns_per_frame
        h = metadata.format.pixels_per_column
        self._w, self._h = w, h
        self._xyz_lut = None
        try:  # only with ouster-sdk: range image -> xyz
            import ouster.client as client  # noqa: WPS433
            self._xyz_lut = client.XYZLut(metadata, use_extrinsics=_use_extrinsics)
        except Exception:
            pass
        # device LUT when the metadata carries the beam geometry (SensorInfo field names); then a scan that only
        # has a `.range` image is converted and masked on the GPU
        self._dev_lut = None
        if self._xyz_lut is None and hasattr(metadata, "beam_altitude_angles"):
            l2s = np.array(getattr(metadata, "lidar_to_sensor_transform", np.eye(4)), dtype=np.float64)
            ext = np.array(metadata.extrinsic, dtype=np.float64) if (_use_extrinsics and hasattr(metadata, "extrinsic")) else None
            self._dev_lut = core.Lut(h, w, metadata.beam_altitude_angles, metadata.beam_azimuth_angles,
                                     getattr(metadata, "lidar_origin_to_beam_origin_mm", 0.0), l2s, ext,
                                     device_id=device_id)
        # per-pixel normalised time, reference kiss.py:34-35
        self._timestamps = np.tile(np.linspace(0, 1.0, w, endpoint=False), (h, 1))
        self._max_range, self._min_range = _max_range, _min_range
        # lazy_map_stats: register_frame returns with the pose (what the reference's returns), the map update of the scan still under
        # way on the device; the per-scan `stats` rows then carry no map size (core.Icp)
        # (the per-call handle keeps a 2^22-slot map table - 64 MB, cleared at every construction - instead of the batch default's 2^24: one
        # sequence over the whole chip holds its probe results in registers and gains little from a sparser table)
        # ... unless the caller sizes the block pool up (small voxels, a dense map): the table raises its capacity flag at 3/4 of its slots,
        # tombstones included - the default follows the pool, the next power of two >= 8 slots per block
        blocks = int(icp_over.get("map_block_capacity", 0))
        table = 1 << 22
        while table < 8 * blocks:
            table <<= 1
        icp_over.setdefault("map_table_capacity", table)
        self._icp = core.Icp(_max_range, _min_range, device_id=device_id, scan_cols=w,
                             max_points_per_scan=max(h * w, 1024), lazy_map_stats=lazy_map_stats, **icp_over)
        c = self._icp.cfg
        self._kiss_config = SimpleNamespace(
            data=SimpleNamespace(max_range=c.max_range, min_range=c.min_range, deskew=bool(c.deskew), preprocess=True),
            mapping=SimpleNamespace(voxel_size=c.voxel_size, max_points_per_voxel=c.max_points_per_voxel),
            adaptive_threshold=SimpleNamespace(initial_threshold=c.initial_threshold, min_motion_th=c.min_motion_th,
                                               fixed_threshold=None))
        self._kiss = _KissState(self._icp, self._kiss_config)
        self._poses_ts: List[float] = []
        self._err_dt: List[float] = []
        self._err_drot: List[float] = []
        self._sigmas: List[float] = []

    # ------------------------------------------------------------------ registration
    def register_frame(self, scan, initial_guess: Optional[np.ndarray] = None) -> np.ndarray:
        """reference kiss.py:54-74"""
        if self._dev_lut is not None and hasattr(scan, "range") and not hasattr(scan, "xyz"):
            ts = float(getattr(scan, "ts", 0.0))
            pose = self._icp.register_range(self._dev_lut, scan.range, initial_guess, ts)
            self._log_pose(pose, ts)
            return self.pose
        xyz, t, ts = self._scan_arrays(scan)
        self.register_points(xyz, t, ts, initial_guess=initial_guess)
        return self.pose

    def register_points(self, frame, timestamps, ts: float = 0.0, initial_guess: Optional[np.ndarray] = None):
        """reference kiss.py:83-131 on (N, 3) points + per-point normalised times (None => column-implicit)"""
        pose = self._icp.register_frame(frame, timestamps, initial_guess, ts)
        self._log_pose(pose, ts)
        return pose

    def _log_pose(self, pose, ts):
        st = self._icp.stats[-1]  # reference kiss.py:116-124, :130, :72
        self._err_dt.append(st["err_dt"])
        self._err_drot.append(st["err_drot"])
        self._sigmas.append(st["sigma"])
        self._kiss.poses.append(pose)
        self._poses_ts.append(ts)

    def deskew(self, frame, timestamps) -> np.ndarray:
        """motion-compensate `frame` with the last two poses (reference kiss.py:76-78)"""
        return self._icp.deskew(frame, timestamps)

    def _scan_arrays(self, scan):
        if self._xyz_lut is not None and hasattr(scan, "field"):
            import ouster.client as client  # noqa: WPS433
            sel = scan.field(client.ChanField.RANGE) != 0
            return self._xyz_lut(scan)[sel], self._timestamps[sel], client.last_valid_column_ts(scan) * 1e-09
        xyz = np.asarray(scan.xyz if hasattr(scan, "xyz") else scan)
        ts = float(getattr(scan, "ts", 0.0))
        if xyz.ndim == 3:
            xyz = xyz.reshape(-1, 3)
        if len(xyz) == self._h * self._w:
            rng = getattr(scan, "range", None)
            if rng is not None:  # reference kiss.py:59-61: drop RANGE == 0 pixels and their times
                sel = np.asarray(rng).reshape(-1) != 0
                return xyz[sel], self._timestamps.reshape(-1)[sel], ts
            return xyz, None, ts  # full image: invalid returns are (0,0,0) and fall to the range filter
        t = getattr(scan, "t01", None)
        if t is None:
            raise ValueError("a point list that is not a full H*W image needs per-point times (.t01)")
        return xyz, np.asarray(t), ts

    # ------------------------------------------------------------------ reference properties
    @property
    def velocity(self) -> np.ndarray:
        if len(self.poses) < 2:
            return np.zeros(3)
        dt = self.poses_ts[-1] - self.poses_ts[-2]
        return self._kiss.get_prediction_model()[:3, 3] / dt

    @property
    def pose(self) -> np.ndarray:
        return np.eye(4) if not self.poses else self.poses[-1]

    @property
    def poses(self) -> List[np.ndarray]:
        return self._kiss.poses

    @property
    def poses_ts(self) -> List[float]:
        return self._poses_ts

    @property
    def local_map_points(self) -> np.ndarray:
        return self._icp.map_points()

    @property
    def _config(self):
        return self._kiss_config

    @property
    def stats(self):
        """per-scan counters (N_v, N_d, N_s, iterations, sum C_i, M_v): the byte model of SURVEY.md 8(d)"""
        return self._icp.stats
