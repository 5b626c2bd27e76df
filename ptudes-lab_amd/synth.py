"""Synthetic spinning-lidar sequences: 128x1024 / 64x2048 sweeps with random-walk SE(3) ground
truth, per-column motion distortion and a 100 Hz IMU.

Test / bench data tooling: the reference ships no sample data and the datasets its
README uses are not available offline.  Rendering is done by csrc/synth.c (host C + OpenMP).

Frames: world = the sensor(=IMU) frame at t = 0 (level, at rest), which is what the reference's
EKF assumes of its first IMU frame (es_ekf.py:75).
"""
import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np
from scipy.spatial.transform import Rotation

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "csrc", "synth.c")
_LIB = os.path.join(_HERE, "csrc", "libptl_synth.so")

GRAV = 9.782940329221166  # reference ins/data.py:10


class _Sensor(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("el_top_deg", C.c_double), ("el_bot_deg", C.c_double),
                ("min_range", C.c_double), ("max_range", C.c_double), ("noise_std", C.c_double),
                ("dropout", C.c_double), ("rough_amp", C.c_double), ("rough_len", C.c_double),
                ("ray_jitter_deg", C.c_double)]


_lib = None


def build(force=False):
    if force or not os.path.exists(_LIB) or os.path.getmtime(_SRC) > os.path.getmtime(_LIB):
        subprocess.run(["gcc", "-O2", "-fPIC", "-fopenmp", "-shared", "-o", _LIB, _SRC, "-lm"], check=True)
    return _LIB


def _l():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        dp = C.POINTER(C.c_double)
        _lib.ptl_synth_render.restype = None
        _lib.ptl_synth_render.argtypes = [dp, dp, C.c_int32, dp, C.c_int32, C.POINTER(_Sensor), dp,
                                          C.c_uint64, C.POINTER(C.c_float)]
        _lib.ptl_synth_trajectory.restype = None
        _lib.ptl_synth_trajectory.argtypes = [dp, dp, C.c_int32, dp, dp, C.c_int32, C.c_double, C.c_double,
                                              C.c_int64, dp, C.c_double, dp, dp, dp, dp]
        _lib.ptl_synth_set_threads.restype = None
        _lib.ptl_synth_set_threads.argtypes = [C.c_int32]
        _lib.ptl_synth_set_threads(min(usable_cores(), 16))
    return _lib


def set_threads(n: int):
    """threads the renderer uses per sweep (default: the usable cores, at most 16); ranks that share a host divide them"""
    _l().ptl_synth_set_threads(max(1, int(n)))


def usable_cores():
    """cores this process may actually use: the affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(q) // int(per)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


@dataclass
class Sequence:
    """One synthetic sequence.  Scans are rendered lazily (`scan(k)`) or all at once (`scans()`)."""
    H: int
    W: int
    n_scans: int
    seed: int
    scan_dt: float
    t_base: float
    room: np.ndarray
    boxes: np.ndarray
    cyls: np.ndarray
    min_range: float
    max_range: float
    noise_std: float
    dropout: float
    rough_amp: float = 0.0
    rough_len: float = 1.5
    ray_jitter_deg: float = 0.0
    # fine trajectory (1 kHz)
    traj_dt: float = 1e-3
    traj_R: np.ndarray = field(default=None, repr=False)   # (n,3,3) world<-body
    traj_p: np.ndarray = field(default=None, repr=False)   # (n,3)
    traj_vw: np.ndarray = field(default=None, repr=False)
    traj_wb: np.ndarray = field(default=None, repr=False)
    # 100 Hz IMU: (n_imu, 7) = ts, lacc(3), avel(3)
    imu: np.ndarray = field(default=None, repr=False)
    imu_bias_acc: np.ndarray = None
    imu_bias_gyr: np.ndarray = None

    # ---------------------------------------------------------------- poses
    def pose_at(self, t):
        """world<-body 4x4 at relative time(s) t (seconds from sequence start); t scalar or (n,)"""
        t = np.atleast_1d(np.asarray(t, dtype=np.float64))
        u = np.clip(t / self.traj_dt, 0, len(self.traj_p) - 1 - 1e-9)
        i = np.floor(u).astype(np.int64)
        f = (u - i)[:, None]
        p = (1 - f) * self.traj_p[i] + f * self.traj_p[i + 1]
        r0 = Rotation.from_matrix(self.traj_R[i])
        r1 = Rotation.from_matrix(self.traj_R[i + 1])
        rel = (r0.inv() * r1).as_rotvec()
        R = (r0 * Rotation.from_rotvec(rel * f)).as_matrix()
        T = np.tile(np.eye(4), (len(t), 1, 1))
        T[:, :3, :3] = R
        T[:, :3, 3] = p
        return T

    def gt_poses(self, where=0.5):
        """GT pose per scan at fraction `where` of the sweep (0.5 = mid-sweep, the instant KISS-ICP's
        deskew refers every point to)"""
        k = np.arange(self.n_scans)
        return self.pose_at((k + where) * self.scan_dt)

    # ---------------------------------------------------------------- scans
    def scan(self, k):
        """(H*W, 3) float32 xyz, row-major beam-outer; (0,0,0) = no return"""
        t = (k + np.arange(self.W) / self.W) * self.scan_dt
        T = self.pose_at(t)
        cp = np.ascontiguousarray(np.concatenate([T[:, :3, :3].reshape(self.W, 9), T[:, :3, 3]], axis=1))
        out = np.empty((self.H * self.W, 3), dtype=np.float32)
        s = _Sensor(self.H, self.W, 45.0, -45.0, self.min_range, self.max_range, self.noise_std, self.dropout,
                    self.rough_amp, self.rough_len, self.ray_jitter_deg)
        boxes = np.ascontiguousarray(self.boxes, dtype=np.float64)
        cyls = np.ascontiguousarray(self.cyls, dtype=np.float64)
        room = np.ascontiguousarray(self.room, dtype=np.float64)
        _l().ptl_synth_render(_p(room), _p(boxes), len(boxes), _p(cyls), len(cyls), C.byref(s), _p(cp),
                              C.c_uint64(self.seed * 1000003 + k), out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def scans(self, n=None):
        n = self.n_scans if n is None else n
        out = np.empty((n, self.H * self.W, 3), dtype=np.float32)
        for k in range(n):
            out[k] = self.scan(k)
        return out

    def column_times(self):
        """per-point normalised time in [0,1): col / W, tiled like reference kiss.py:34-35"""
        return np.tile(np.linspace(0, 1.0, self.W, endpoint=False), (self.H, 1)).reshape(-1)

    # ---------------------------------------------------------------- event stream (reference feed order)
    def imu_range_for_scan(self, k):
        """indices [a, b) of the IMU samples that precede scan k's completion and follow scan k-1's"""
        per = int(round(self.scan_dt * 100))
        return k * per, (k + 1) * per

    def events(self, n_scans=None, with_points=True):
        """Yield ("imu", lacc, avel, ts) / ("scan", xyz, t01) in packet order: the IMU samples of a
        sweep arrive before the sweep completes (reference data.py:31-77, ekf_bench.py:493-563)."""
        n = self.n_scans if n_scans is None else n_scans
        t01 = self.column_times()
        for k in range(n):
            a, b = self.imu_range_for_scan(k)
            for i in range(a, b):
                yield ("imu", self.imu[i, 1:4], self.imu[i, 4:7], self.imu[i, 0])
            if with_points:
                yield ("scan", self.scan(k), t01)
            else:
                yield ("scan", k, t01)


def make_sequence(seed=1000, n_scans=100, H=128, W=1024, *, min_range=1.0, max_range=70.0, noise_std=0.01,
                  dropout=0.02, scan_hz=10.0, t_base=1626432000.0, room_size=(80.0, 60.0, 15.0),
                  n_boxes=40, n_cyls=20, imu_noise=(0.05, 0.005), rough_amp=0.15, rough_len=1.5, ray_jitter_deg=0.0):
    """Scene + random-walk trajectory + IMU for sequence `seed` (seeds 1000..1007)."""
    rng = np.random.default_rng(seed)
    lx, ly, lz = room_size
    room_scene = np.array([-lx / 2, lx / 2, -ly / 2, ly / 2, 0.0, lz])
    # obstacles standing on the floor, sizes U[0.5, 6] m
    boxes = np.zeros((n_boxes, 6))
    for b in range(n_boxes):
        h = rng.uniform(0.5, 6.0, 3) / 2
        boxes[b] = [rng.uniform(-lx / 2 + 3, lx / 2 - 3), rng.uniform(-ly / 2 + 3, ly / 2 - 3), h[2], *h]
    cyls = np.zeros((n_cyls, 4))
    for c in range(n_cyls):
        cyls[c] = [rng.uniform(-lx / 2 + 3, lx / 2 - 3), rng.uniform(-ly / 2 + 3, ly / 2 - 3),
                   rng.uniform(0.5, 6.0) / 2, rng.uniform(0.5, 6.0)]
    # start near the centre, clear of obstacles
    obst = np.concatenate([np.c_[boxes[:, 0], boxes[:, 1], np.hypot(boxes[:, 3], boxes[:, 4])],
                           np.c_[cyls[:, 0], cyls[:, 1], cyls[:, 2]]])
    height = 1.5
    p0 = np.array([0.0, 0.0, height])
    for _ in range(200):
        cand = np.array([rng.uniform(-10, 10), rng.uniform(-8, 8), height])
        if np.all(np.hypot(obst[:, 0] - cand[0], obst[:, 1] - cand[1]) > obst[:, 2] + 2.5):
            p0 = cand
            break
    # AR(1) velocity / rate commands at the sweep rate (start from rest: the EKF assumes v0 = 0)
    scan_dt = 1.0 / scan_hz
    n_knots = n_scans + 3
    v = np.zeros((n_knots, 3))
    w = np.zeros((n_knots, 3))
    for k in range(1, n_knots):
        v[k] = 0.98 * v[k - 1] + rng.normal(0, 0.15, 3)
        nv = np.linalg.norm(v[k])
        if nv > 3.0:
            v[k] *= 3.0 / nv
        w[k] = 0.95 * w[k - 1] + rng.normal(0, 0.05, 3)
        nw = np.linalg.norm(w[k])
        if nw > 0.8:
            w[k] *= 0.8 / nw
    traj_dt = 1e-3
    n_steps = int(round((n_scans + 1) * scan_dt / traj_dt)) + 2
    R = np.empty((n_steps, 9))
    p = np.empty((n_steps, 3))
    vw = np.empty((n_steps, 3))
    wb = np.empty((n_steps, 3))
    obst_c = np.ascontiguousarray(obst)
    _l().ptl_synth_trajectory(_p(room_scene), _p(obst_c), len(obst_c), _p(v), _p(w), n_knots, scan_dt, traj_dt,
                              n_steps, _p(p0), height, _p(R), _p(p), _p(vw), _p(wb))
    # shift to world = body frame at t=0 (R0 = I already)
    room = room_scene.copy()
    room[0:2] -= p0[0]
    room[2:4] -= p0[1]
    room[4:6] -= p0[2]
    boxes_w = boxes.copy()
    boxes_w[:, :3] -= p0
    cyls_w = cyls.copy()
    cyls_w[:, :2] -= p0[:2]
    p_w = p - p0
    # IMU at 100 Hz: f = R^T (a - g) + b_a + n, w + b_g + n   (g along -Z of the first frame)
    a_w = np.gradient(vw, traj_dt, axis=0)
    Rm = R.reshape(-1, 3, 3)
    g = np.array([0.0, 0.0, -GRAV])
    b_a = np.array([0.05, -0.02, 0.03])
    b_g = np.array([0.002, -0.001, 0.0015])
    stride = int(round(0.01 / traj_dt))
    idx = np.arange(0, int(round(n_scans * scan_dt / 0.01))) * stride
    f_b = np.einsum("nji,nj->ni", Rm[idx], a_w[idx] - g)
    imu = np.empty((len(idx), 7))
    imu[:, 0] = t_base + idx * traj_dt
    imu[:, 1:4] = f_b + b_a + rng.normal(0, imu_noise[0], (len(idx), 3))
    imu[:, 4:7] = wb[idx] + b_g + rng.normal(0, imu_noise[1], (len(idx), 3))
    return Sequence(H=H, W=W, n_scans=n_scans, seed=seed, scan_dt=scan_dt, t_base=t_base, room=room,
                    boxes=boxes_w, cyls=cyls_w, min_range=min_range, max_range=max_range,
                    noise_std=noise_std, dropout=dropout, rough_amp=rough_amp, rough_len=rough_len, ray_jitter_deg=ray_jitter_deg,
                    traj_dt=traj_dt, traj_R=Rm, traj_p=p_w, traj_vw=vw,
                    traj_wb=wb, imu=imu, imu_bias_acc=b_a, imu_bias_gyr=b_g)


def make_path_sequence(seed=2000, n_scans=100, H=128, W=1024, *, step_m=0.5, static_sweeps=0, ramp_sweeps=0, yaw_rate=0.0,
                       min_range=1.0, max_range=70.0, noise_std=0.0, dropout=0.0, scan_hz=10.0, t_base=1626432000.0,
                       room_size=None, n_boxes=None, n_cyls=None, rough_amp=0.15, rough_len=1.5, imu_noise=(0.0, 0.0),
                       clear_halfwidth=4.0, wobble_deg=0.0, wobble_hz=1.3, heave_m=0.0, ray_jitter_deg=0.0):
    """A controlled sequence for tracking experiments (not SURVEY.md 8(d)'s random walk): the sensor drives along +x at
    `step_m` metres per sweep (0.05 = walking pace, 1.0 = 10 m/s vehicle speed), optionally standing still for the first
    `static_sweeps` sweeps (no un-deskewed motion baked into the first map), accelerating over `ramp_sweeps` sweeps
    (smoothstep; 0 = the speed is there from t = 0), optionally yawing at `yaw_rate` rad/s and rocking like a sprung
    vehicle: roll / pitch of `wobble_deg` amplitude at `wobble_hz` (and 0.77 of it), heave of `heave_m` - scaled by the
    current speed fraction, so a standing vehicle stands still.  The scene is a hall long
    enough for the drive (default: path length + 80 m by 60 m by 15 m) with the obstacle density of make_sequence and a
    clear lane of +-`clear_halfwidth` m around the path.  Range noise and dropout default to none."""
    rng = np.random.default_rng(seed)
    scan_dt = 1.0 / scan_hz
    path_len = step_m * max(n_scans + 1 - static_sweeps, 0)
    if room_size is None:
        room_size = (max(80.0, path_len + 80.0), 60.0, 15.0)
    lx, ly, lz = room_size
    area_ratio = (lx * ly) / (80.0 * 60.0)
    n_boxes = int(round(40 * area_ratio)) if n_boxes is None else n_boxes
    n_cyls = int(round(20 * area_ratio)) if n_cyls is None else n_cyls
    height = 1.5
    p0 = np.array([-lx / 2 + 40.0, 0.0, height])
    boxes = np.zeros((n_boxes, 6))
    for b in range(n_boxes):
        h = rng.uniform(0.5, 6.0, 3) / 2
        while True:
            cy = rng.uniform(-ly / 2 + 3, ly / 2 - 3)
            if abs(cy) - h[1] > clear_halfwidth:
                break
        boxes[b] = [rng.uniform(-lx / 2 + 3, lx / 2 - 3), cy, h[2], *h]
    cyls = np.zeros((n_cyls, 4))
    for c in range(n_cyls):
        r = rng.uniform(0.5, 6.0) / 2
        while True:
            cy = rng.uniform(-ly / 2 + 3, ly / 2 - 3)
            if abs(cy) - r > clear_halfwidth:
                break
        cyls[c] = [rng.uniform(-lx / 2 + 3, lx / 2 - 3), cy, r, rng.uniform(0.5, 6.0)]
    traj_dt = 1e-3
    n_steps = int(round((n_scans + 1) * scan_dt / traj_dt)) + 2
    t = np.arange(n_steps) * traj_dt
    v_full = step_m / scan_dt
    t0 = static_sweeps * scan_dt
    if ramp_sweeps > 0:
        u = np.clip((t - t0) / (ramp_sweeps * scan_dt), 0.0, 1.0)
        speed = v_full * u * u * (3 - 2 * u)
    else:
        speed = np.where(t >= t0, v_full, 0.0)
    x = np.concatenate([[0.0], np.cumsum(0.5 * (speed[1:] + speed[:-1]) * traj_dt)])
    p_w = np.zeros((n_steps, 3))
    p_w[:, 0] = x
    vw = np.zeros((n_steps, 3))
    vw[:, 0] = speed
    yaw = yaw_rate * np.maximum(t - t0, 0.0)
    frac = speed / v_full if v_full > 0 else np.zeros_like(speed)
    roll = np.radians(wobble_deg) * frac * np.sin(2 * np.pi * wobble_hz * t + 0.7)
    pitch = np.radians(wobble_deg) * frac * np.sin(2 * np.pi * 0.77 * wobble_hz * t + 2.1)
    rot = Rotation.from_euler("zyx", np.stack([yaw, pitch, roll], axis=1))
    Rm = rot.as_matrix()
    p_w[:, 2] = heave_m * frac * np.sin(2 * np.pi * 1.19 * wobble_hz * t)
    vw = np.gradient(p_w, traj_dt, axis=0)
    # body angular rate from consecutive attitudes
    rel = (rot[:-1].inv() * rot[1:]).as_rotvec() / traj_dt
    wb = np.concatenate([rel, rel[-1:]], axis=0)
    room = np.array([-lx / 2, lx / 2, -ly / 2, ly / 2, 0.0, lz])
    room[0:2] -= p0[0]
    room[2:4] -= p0[1]
    room[4:6] -= p0[2]
    boxes[:, :3] -= p0
    cyls[:, :2] -= p0[:2]
    a_w = np.gradient(vw, traj_dt, axis=0)
    g = np.array([0.0, 0.0, -GRAV])
    b_a = np.array([0.05, -0.02, 0.03])
    b_g = np.array([0.002, -0.001, 0.0015])
    stride = int(round(0.01 / traj_dt))
    idx = np.arange(0, int(round(n_scans * scan_dt / 0.01))) * stride
    f_b = np.einsum("nji,nj->ni", Rm[idx], a_w[idx] - g)
    imu = np.empty((len(idx), 7))
    imu[:, 0] = t_base + idx * traj_dt
    imu[:, 1:4] = f_b + b_a + rng.normal(0, 1.0, (len(idx), 3)) * imu_noise[0]
    imu[:, 4:7] = wb[idx] + b_g + rng.normal(0, 1.0, (len(idx), 3)) * imu_noise[1]
    return Sequence(H=H, W=W, n_scans=n_scans, seed=seed, scan_dt=scan_dt, t_base=t_base, room=room, boxes=boxes, cyls=cyls,
                    min_range=min_range, max_range=max_range, noise_std=noise_std, dropout=dropout, rough_amp=rough_amp,
                    rough_len=rough_len, ray_jitter_deg=ray_jitter_deg, traj_dt=traj_dt, traj_R=Rm, traj_p=p_w, traj_vw=vw,
                    traj_wb=wb, imu=imu,
                    imu_bias_acc=b_a, imu_bias_gyr=b_g)
