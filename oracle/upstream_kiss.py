"""The hook that pins the ICP half of the oracle the day the upstream package exists.  TEST INFRASTRUCTURE ONLY (tests/ and
bench.py's cpu_baseline leg; the product never imports oracle/).

All ICP arithmetic of the reference lives in the third-party package `kiss-icp` (reference setup.py:23, effective pin 0.2.10,
SURVEY.md App. A), which is neither in /root/reference nor installable in this environment (no network, no Eigen / Sophus / TBB):
`oracle/oracle_icp.c` restates its published algorithm and is "parity unpinned" (oracle/oracle.h).  This module drives the REAL
package - if `import kiss_icp` succeeds - through exactly the calls the reference makes (reference src/ptudes/kiss.py:7-10 imports,
:40-43 configuration, :83-131 the per-scan sequence), on plain arrays, so that tests/test_oracle_vs_upstream_kiss.py and
bench.py's optional "upstream" baseline leg can hold the oracle against it.  Nothing of the reference travels: the calls below
are upstream's public 0.2.x API.
"""
import numpy as np

SUPPORTED = ("0.2.9", "0.2.10")


def available():
    """(module, version) of an importable kiss_icp whose API the reference can drive (0.2.9 / 0.2.10), else (None, reason)"""
    try:
        import kiss_icp  # noqa: WPS433
    except Exception as e:  # noqa: BLE001
        return None, f"import kiss_icp failed: {e!r}"
    ver = getattr(kiss_icp, "__version__", None)
    if ver is None:
        try:
            from importlib.metadata import version
            ver = version("kiss-icp")
        except Exception:  # noqa: BLE001
            ver = "unknown"
    if not any(ver == v or ver.startswith(v + ".") for v in SUPPORTED):
        return None, f"kiss_icp {ver} is importable but is not {' / '.join(SUPPORTED)} (reference kiss.py:7-10 needs that API)"
    return kiss_icp, ver


class Upstream:
    """One kiss_icp.KissICP driven like reference KissICPWrapper._kiss_register_frame (kiss.py:83-131)"""

    def __init__(self, max_range=100.0, min_range=5.0):
        from kiss_icp.config import load_config  # kiss.py:7
        from kiss_icp.kiss_icp import KissICP  # kiss.py:9
        cfg = load_config(None, deskew=True, max_range=max_range)  # kiss.py:40-42
        cfg.data.min_range = min_range  # kiss.py:43
        self.k = KissICP(config=cfg)
        self.cfg = cfg
        self.sigmas, self.err_dt, self.err_drot = [], [], []
        self.last_frame_down = self.last_source = None

    @property
    def poses(self):
        return self.k.poses

    def register_frame(self, xyz, t01, guess=None):
        from kiss_icp.registration import register_frame as upstream_register  # kiss.py:8
        from scipy.spatial.transform import Rotation
        k = self.k
        frame = k.compensator.deskew_scan(np.asarray(xyz, np.float64), k.poses, np.asarray(t01, np.float64))  # :90
        frame = k.preprocess(frame)  # :93
        source, frame_down = k.voxelize(frame)  # :96
        sigma = k.get_adaptive_threshold()  # :99
        if guess is None:  # :102-105
            last = k.poses[-1] if k.poses else np.eye(4)
            guess = last @ k.get_prediction_model()
        pose = upstream_register(points=source, voxel_map=k.local_map, initial_guess=guess,
                                 max_correspondance_distance=3 * sigma, kernel=sigma / 3)  # :108-114
        gain = np.linalg.inv(guess) @ pose  # :116
        self.err_dt.append(float(np.linalg.norm(gain[:3, 3])))
        self.err_drot.append(float(np.linalg.norm(Rotation.from_matrix(gain[:3, :3]).as_rotvec())))
        self.sigmas.append(float(sigma))
        k.adaptive_threshold.update_model_deviation(gain)  # :128
        k.local_map.update(frame_down, pose)  # :129
        k.poses.append(pose)  # :130
        self.last_frame_down, self.last_source = np.asarray(frame_down), np.asarray(source)
        return np.asarray(pose)

    def map_points(self):
        return np.asarray(self.k.local_map.point_cloud())  # kiss.py:160-161


def voxel_down_sample(xyz, voxel):
    from kiss_icp.voxelization import voxel_down_sample as f
    return np.asarray(f(np.asarray(xyz, np.float64), float(voxel)))


def new_map(voxel_size, max_distance, max_points_per_voxel=20):
    from kiss_icp.mapping import VoxelHashMap
    return VoxelHashMap(voxel_size, max_distance, max_points_per_voxel)


def rows_sorted(a):
    """a cloud as a canonical (lexicographically sorted) array: upstream's container order is unspecified (robin_map iteration)"""
    a = np.asarray(a, np.float64).reshape(-1, 3)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
