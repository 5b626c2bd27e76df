"""The compute half of the reference's map fly-by (src/ptudes/fly.py:75-111, src/ptudes/cli/flyby.py): scans that
carry one pose per column are de-warped into the world frame and accumulated into a point map.  The reference hands
both to ouster-sdk's `ScansAccumulator` and draws the result (the viewer is out of scope here); this module keeps the
arithmetic: column poses from a time-stamped trajectory (`utils.TrajectoryEvaluator`, `utils.pose_scans_from_nc_gt`),
`client.dewarp` on the GPU (`ptl_lut_dewarp`), and the map as the voxel-hash map of the registration path
(`ptl_icp_map_add`: a voxel keeps its first points, deterministic) instead of ScansAccumulator's random subsample."""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import core


@dataclass
class PosedScan:
    """What the fly-by needs of an ouster LidarScan: the range image, column timestamps (ns) and column poses"""
    range_mm: np.ndarray                 # (H, W) uint32, 0 = no return
    timestamp: np.ndarray                # (W,) ns
    pose: Optional[np.ndarray] = field(default=None)  # (W, 4, 4), world <- sensor at the column's firing time

    def __post_init__(self):
        self.range_mm = np.ascontiguousarray(self.range_mm, dtype=np.uint32)
        if self.pose is None:
            self.pose = np.tile(np.eye(4), (self.range_mm.shape[1], 1, 1))


class MapAccumulator:
    """Accumulates posed scans into a world-frame voxel map on the GPU (ScansAccumulator's map, deterministic form)."""

    def __init__(self, lut: core.Lut, voxel_size: float = 0.5, max_points_per_voxel: int = 20, max_range: float = 1.0e9,
                 device_id: int = 0, map_block_capacity: int = 1 << 21, map_table_capacity: int = 1 << 23):
        self.lut = lut
        # only the map of the handle is used: no registration happens here, the poses are given
        self._icp = core.Icp(max_range, 0.0, voxel_size=voxel_size, max_points_per_voxel=max_points_per_voxel,
                             scan_cols=lut.W, max_points_per_scan=lut.H * lut.W, device_id=device_id,
                             map_block_capacity=map_block_capacity, map_table_capacity=map_table_capacity)
        self.scans = 0
        self.returns = 0

    def update(self, scan) -> np.ndarray:
        """de-warp one posed scan and add its returns to the map; gives back their world coordinates"""
        xyz, n_valid = self.lut.dewarp(scan.range_mm, scan.pose)
        keep = np.asarray(scan.range_mm).reshape(-1) != 0
        pts = xyz[keep]
        self._icp.map_add(pts)
        self.scans += 1
        self.returns += n_valid
        return pts

    def map_size(self):
        return self._icp.map_size()

    def map_points(self) -> np.ndarray:
        return self._icp.map_points()
