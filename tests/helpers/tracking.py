"""Helpers of the tracking-diagnosis tests: single registrations under controlled motion (no sweep distortion)."""
import ctypes as C

import numpy as np
from scipy.spatial.transform import Rotation as Rot

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import synth
from oracle import cpu as orc


def pose(dx=0.0, dy=0.0, dz=0.0, roll=0.0, pitch=0.0, yaw=0.0):
    T = np.eye(4)
    T[:3, :3] = Rot.from_euler("xyz", [roll, pitch, yaw], degrees=True).as_matrix()
    T[:3, 3] = [dx, dy, dz]
    return T


def static_scan(seq, T, seed=7):
    """one sweep of `seq`'s scene from the FIXED pose T (every column fired from the same place: nothing to deskew)"""
    W = seq.W
    cp = np.ascontiguousarray(np.tile(np.concatenate([T[:3, :3].reshape(9), T[:3, 3]]), (W, 1)))
    out = np.empty((seq.H * W, 3), dtype=np.float32)
    s = synth._Sensor(seq.H, W, 45.0, -45.0, seq.min_range, seq.max_range, seq.noise_std, seq.dropout, seq.rough_amp, seq.rough_len, getattr(seq, 'ray_jitter_deg', 0.0))
    p = synth._p
    synth._l().ptl_synth_render(p(np.ascontiguousarray(seq.room)), p(np.ascontiguousarray(seq.boxes)), len(seq.boxes),
                                p(np.ascontiguousarray(seq.cyls)), len(seq.cyls), C.byref(s), p(cp), C.c_uint64(seed),
                                out.ctypes.data_as(C.POINTER(C.c_float)))
    x = out.astype(np.float64)
    return x[np.linalg.norm(x, axis=1) > 0]


def voxelize(x, max_range=70.0, min_range=1.0):
    """KissICP.preprocess + voxelize at the CLI defaults (voxel 0.7 m): frame_downsample, source"""
    x = orc.preprocess(x, max_range, min_range)
    fd = orc.voxel_downsample(x, 0.35)
    return fd, orc.voxel_downsample(fd, 1.05)


def perfectly_deskewed(seq, k):
    """sweep k in its mid-sweep frame, every column moved there with its GROUND-TRUTH pose"""
    x = seq.scan(k).astype(np.float64)
    W = seq.W
    Tc = seq.pose_at((k + np.arange(W) / W) * seq.scan_dt)
    Tm = seq.pose_at(np.array([(k + 0.5) * seq.scan_dt]))[0]
    rel = np.einsum("ij,njk->nik", np.linalg.inv(Tm), Tc)
    cols = np.tile(np.arange(W), seq.H)
    keep = np.linalg.norm(x, axis=1) > 0
    y = np.einsum("nij,nj->ni", rel[cols, :3, :3], x) + rel[cols, :3, 3]
    return y[keep]


def err_of(T_est, T_true):
    d = np.linalg.inv(T_true) @ T_est  # (a lost track may hand back something that is no rotation any more: no scipy here)
    return float(np.linalg.norm(d[:3, 3])), float(np.degrees(np.arccos(np.clip((np.trace(d[:3, :3]) - 1.0) / 2.0, -1.0, 1.0))))
