"""Posed scans of the map fly-by (reference utils.py:344-392, fly.py:75-111): trajectory
interpolation and dewarp.  The CPU tests pin the numpy oracle by algebraic known answers (the upstream routine is
third-party and absent); the GPU tests hold the HIP kernels to the oracle."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

from oracle import dewarp as od


def _pose(rv, t):
    T = np.eye(4)
    T[:3, :3] = Rot.from_rotvec(rv).as_matrix()
    T[:3, 3] = t
    return T


def _knots(n=12, seed=3):
    rng = np.random.default_rng(seed)
    ts = 100.0 + np.cumsum(rng.uniform(0.05, 0.15, n))
    T = np.eye(4)
    out = []
    for t in ts:
        T = T @ _pose(rng.normal(0, 0.05, 3), rng.normal(0, 0.3, 3))
        out.append((float(t), T.copy()))
    return out


def test_oracle_interpolation_known_answers():
    kn = _knots()
    ts = np.array([k[0] for k in kn])
    P = od.poses_at(kn, ts)
    assert np.abs(P - np.array([k[1] for k in kn])).max() < 1e-12  # knots reproduce themselves
    # a constant twist is followed exactly, also beyond the ends (within the bounds)
    xi = np.zeros((4, 4))
    xi[:3, :3] = [[0, -0.3, 0.1], [0.3, 0, -0.2], [-0.1, 0.2, 0]]
    xi[:3, 3] = [1.0, -0.5, 0.25]
    from scipy.linalg import expm
    k2 = [(float(t), expm(t * xi)) for t in (0.0, 0.4, 1.0, 1.3)]
    q = np.array([-0.2, 0.1, 0.7, 1.1, 1.5])
    assert np.abs(od.poses_at(k2, q, time_bounds=0.25) - np.array([expm(t * xi) for t in q])).max() < 1e-12
    with pytest.raises(ValueError):
        od.poses_at(k2, [1.6], time_bounds=0.25)
    # pure translations interpolate linearly
    k3 = [(0.0, _pose([0, 0, 0], [0, 0, 0])), (2.0, _pose([0, 0, 0], [4.0, -2.0, 1.0]))]
    assert np.allclose(od.poses_at(k3, [0.5])[0][:3, 3], [1.0, -0.5, 0.25], atol=1e-14)


def test_oracle_dewarp_is_per_column():
    rng = np.random.default_rng(1)
    H, W = 4, 6
    xyz = rng.normal(0, 5, (H, W, 3))
    poses = np.array([_pose(rng.normal(0, 0.2, 3), rng.normal(0, 1, 3)) for _ in range(W)])
    out = od.dewarp(xyz, poses)
    for u in range(H):
        for v in range(W):
            assert np.allclose(out[u, v], poses[v, :3, :3] @ xyz[u, v] + poses[v, :3, 3], atol=1e-14)


@pytest.mark.gpu
def test_device_trajectory_interpolation_vs_oracle():
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import utils as pu
    kn = _knots(40, seed=5)
    rng = np.random.default_rng(6)
    ts = np.sort(rng.uniform(kn[0][0] - 0.9, kn[-1][0] + 0.9, 2048))
    ev = pu.TrajectoryEvaluator(kn, time_bounds=1.0)
    got = ev.poses_at(ts)
    ref = od.poses_at(kn, ts, time_bounds=1.0)
    assert np.abs(got - ref).max() < 1e-10
    assert np.abs(ev.pose_at(kn[7][0]) - kn[7][1]).max() < 1e-12
    with pytest.raises(ValueError):
        ev.poses_at([kn[0][0] - 1.5])
    with pytest.raises(ValueError):
        pu.TrajectoryEvaluator(kn[:1])


@pytest.mark.gpu
def test_device_dewarp_and_map_accumulation_vs_oracle():
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import core, fly, synth, utils as pu
    from oracle import lut as olut
    H, W, n = 32, 256, 6
    seq = synth.make_sequence(seed=31, n_scans=n, H=H, W=W)
    alt = np.linspace(45.0, -45.0, H)
    az = np.zeros(H)
    lut = core.Lut(H, W, alt, az, 0.0)
    # ground truth as a trajectory with one knot per 20 ms, sweeps as range images on the ouster column convention
    kt = np.arange(0, n * seq.scan_dt + 0.3, 0.02)
    knots = [(seq.t_base + float(t), seq.pose_at(np.array([t]))[0]) for t in kt]
    scans = []
    for k in range(n):
        x = seq.scan(k).astype(np.float64).reshape(H, W, 3)
        rng_mm = np.round(np.linalg.norm(x, axis=2) * 1000.0).astype(np.uint32)
        rng_mm = rng_mm[:, (W - np.arange(W)) % W]  # ouster column v looks along 2 pi (1 - v / W)
        col_t = seq.t_base + (k + ((W - np.arange(W)) % W) / W) * seq.scan_dt
        scans.append(fly.PosedScan(rng_mm, (col_t * 1e9).astype(np.int64)))
    acc = fly.MapAccumulator(lut, voxel_size=0.5, map_block_capacity=1 << 16, map_table_capacity=1 << 18)
    posed = list(pu.pose_scans_from_nc_gt(scans, nc_gt_poses=knots))
    assert len(posed) == n
    total = 0
    for sc in posed:
        ref_poses = od.poses_at(knots, np.asarray(sc.timestamp) * 1e-9, time_bounds=1.5)
        assert np.abs(sc.pose - ref_poses).max() < 1e-9
        xyz_s = olut.apply(*olut.xyz_lut(H, W, alt, az, 0.0, np.eye(4)), sc.range_mm).reshape(H, W, 3)
        ref = od.dewarp(xyz_s, ref_poses)[sc.range_mm != 0]
        pts = acc.update(sc)
        assert pts.shape == ref.shape and np.abs(pts - ref).max() < 1e-9
        total += len(pts)
    assert acc.returns == total and acc.scans == n
    vox, npts = acc.map_size()
    assert 0 < vox <= npts <= total
    # de-warped with the true column poses the static scene lands on itself: walls are thin in the map
    m = acc.map_points()
    assert np.isfinite(m).all() and len(m) == npts
    # a scan beyond the trajectory is skipped, as in the reference
    late = fly.PosedScan(scans[0].range_mm, scans[0].timestamp + int(60e9))
    assert list(pu.pose_scans_from_nc_gt([late], nc_gt_poses=knots)) == []
