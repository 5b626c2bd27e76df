"""numpy restatement of ouster-sdk `client.XYZLut` as the reference uses it (kiss.py:28-29, 59-60).
TEST INFRASTRUCTURE ONLY.  ouster-sdk (setup.py:20, >= 0.10.0) is a third-party dependency absent from
/root/reference and from this image; the formula is its published one - parity unpinned."""
import numpy as np


def xyz_lut(H, W, alt_deg, az_deg, n_mm, lidar_to_sensor_mm, extrinsic_m=None):
    """returns (direction, offset), each (H*W, 3), in metres per mm / metres: xyz = range_mm * direction + offset"""
    u = np.repeat(np.arange(H), W)
    v = np.tile(np.arange(W), H)
    enc = 2.0 * np.pi * (1.0 - v / W)
    az = -np.asarray(az_deg, dtype=np.float64)[u] * np.pi / 180.0
    phi = np.asarray(alt_deg, dtype=np.float64)[u] * np.pi / 180.0
    d = np.stack([np.cos(enc + az) * np.cos(phi), np.sin(enc + az) * np.cos(phi), np.sin(phi)], axis=1)
    o = np.stack([n_mm * (np.cos(enc) - d[:, 0]), n_mm * (np.sin(enc) - d[:, 1]), -n_mm * d[:, 2]], axis=1)
    T = np.array(lidar_to_sensor_mm, dtype=np.float64)
    if extrinsic_m is not None:
        E = np.array(extrinsic_m, dtype=np.float64)
        E[:3, 3] *= 1e3
        T = E @ T
    return (d @ T[:3, :3].T) * 1e-3, (o @ T[:3, :3].T + T[:3, 3]) * 1e-3


def apply(direction, offset, range_mm):
    r = np.asarray(range_mm, dtype=np.float64).reshape(-1, 1)
    return np.where(r != 0, r * direction + offset, 0.0)
