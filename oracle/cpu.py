"""ctypes face of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package never does (see oracle/oracle.h for what is restated and how it is pinned).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

c_d_p = C.POINTER(C.c_double)
c_i64_p = C.POINTER(C.c_int64)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("oracle_icp.c", "oracle_ekf.c", "oracle.h", "oracle_math.h")]
    if (force or not os.path.exists(_LIB_PATH)
            or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src if os.path.exists(s))):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True, capture_output=True)
    return _LIB_PATH


class IcpCfg(C.Structure):
    _fields_ = [("max_range", C.c_double), ("min_range", C.c_double), ("voxel_size", C.c_double),
                ("max_points_per_voxel", C.c_int32), ("initial_threshold", C.c_double),
                ("min_motion_th", C.c_double), ("deskew", C.c_int32), ("max_iterations", C.c_int32),
                ("convergence", C.c_double)]


class IcpStats(C.Structure):
    _fields_ = [("sigma", C.c_double), ("err_dt", C.c_double), ("err_drot", C.c_double),
                ("iterations", C.c_int32), ("n_corr_last", C.c_int32), ("n_in", C.c_int64),
                ("n_valid", C.c_int64), ("n_down", C.c_int64), ("n_src", C.c_int64),
                ("sum_cand", C.c_int64), ("map_voxels", C.c_int64), ("map_points", C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        sig = {
            "orc_se3_exp": (None, [c_d_p, c_d_p]), "orc_se3_log": (None, [c_d_p, c_d_p]),
            "orc_se3_inv": (None, [c_d_p, c_d_p]), "orc_se3_mul": (None, [c_d_p, c_d_p, c_d_p]),
            "orc_rot_angle": (C.c_double, [c_d_p]),
            "orc_icp_default_cfg": (None, [C.POINTER(IcpCfg), C.c_double, C.c_double]),
            "orc_deskew": (None, [c_d_p, c_d_p, C.c_int64, c_d_p, c_d_p, c_d_p]),
            "orc_preprocess": (C.c_int64, [c_d_p, C.c_int64, C.c_double, C.c_double, c_d_p]),
            "orc_voxel_downsample": (C.c_int64, [c_d_p, C.c_int64, C.c_double, c_d_p, c_i64_p]),
            "orc_map_create": (vp, [C.c_double, C.c_double, C.c_int32]),
            "orc_map_destroy": (None, [vp]),
            "orc_map_add_points": (None, [vp, c_d_p, C.c_int64]),
            "orc_map_prune": (None, [vp, c_d_p]),
            "orc_map_update": (None, [vp, c_d_p, C.c_int64, c_d_p]),
            "orc_map_num_voxels": (C.c_int64, [vp]), "orc_map_num_points": (C.c_int64, [vp]),
            "orc_map_export": (C.c_int64, [vp, c_d_p]),
            "orc_map_linear_system": (None, [vp, c_d_p, C.c_int64, C.c_double, C.c_double, c_d_p,
                                             c_i64_p, c_i64_p, c_d_p]),
            "orc_register": (None, [vp, c_d_p, C.c_int64, c_d_p, C.c_double, C.c_double, C.c_int32,
                                    C.c_double, c_d_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), c_i64_p]),
            "orc_solve6": (None, [c_d_p, c_d_p]),
            "orc_icp_create": (vp, [C.POINTER(IcpCfg)]), "orc_icp_destroy": (None, [vp]),
            "orc_icp_register_frame": (C.c_int, [vp, c_d_p, c_d_p, C.c_int64, c_d_p, c_d_p,
                                                 C.POINTER(IcpStats)]),
            "orc_icp_num_poses": (C.c_int64, [vp]),
            "orc_icp_get_pose": (None, [vp, C.c_int64, c_d_p]),
            "orc_icp_prediction": (None, [vp, c_d_p]),
            "orc_icp_sigma": (C.c_double, [vp]), "orc_icp_has_moved": (C.c_int, [vp]),
            "orc_icp_map": (vp, [vp]),
            "orc_icp_last_frame_down": (C.c_int64, [vp, c_d_p]),
            "orc_icp_last_source": (C.c_int64, [vp, c_d_p]),
            "orc_ekf_create": (vp, [c_d_p, c_d_p, c_d_p]), "orc_ekf_destroy": (None, [vp]),
            "orc_ekf_process_imu": (None, [vp, c_d_p, c_d_p, C.c_double]),
            "orc_ekf_process_pose": (None, [vp, c_d_p, c_d_p]),
            "orc_ekf_get_nav": (None, [vp, c_d_p]), "orc_ekf_get_cov": (None, [vp, c_d_p]),
            "orc_ekf_pose_mat": (None, [vp, c_d_p]), "orc_ekf_ts": (C.c_double, [vp]),
            "orc_calc_ate": (C.c_int, [c_d_p, c_d_p, C.c_int64, c_d_p]),
            "orc_set_threads": (None, [C.c_int]), "orc_get_threads": (C.c_int, []),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def set_threads(n: int):
    """multi-core timing mode of the ICP loops (see oracle.h); 1 = the sequential pass every parity test uses"""
    lib().orc_set_threads(int(n))


def get_threads() -> int:
    return lib().orc_get_threads()


def _d(a):
    """contiguous float64 view + pointer (keeps the array alive through the returned tuple)"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(c_d_p)


def _opt(a):
    if a is None:
        return None, None
    return _d(a)


# ------------------------------------------------------------------ SE(3)
def se3_exp(xi):
    x, xp = _d(xi)
    T = np.empty((4, 4))
    lib().orc_se3_exp(xp, T.ctypes.data_as(c_d_p))
    return T


def se3_log(T):
    t, tp = _d(T)
    xi = np.empty(6)
    lib().orc_se3_log(tp, xi.ctypes.data_as(c_d_p))
    return xi


def rot_angle(T):
    t, tp = _d(T)
    return lib().orc_rot_angle(tp)


# ------------------------------------------------------------------ pipeline pieces
def deskew(xyz, t01, start, finish):
    x, xp = _d(xyz)
    t, tp = _d(t01)
    s, sp = _d(start)
    f, fp = _d(finish)
    out = np.empty_like(x)
    lib().orc_deskew(xp, tp, len(x), sp, fp, out.ctypes.data_as(c_d_p))
    return out


def preprocess(xyz, max_range, min_range):
    x, xp = _d(xyz)
    out = np.empty_like(x)
    m = lib().orc_preprocess(xp, len(x), max_range, min_range, out.ctypes.data_as(c_d_p))
    return out[:m].copy()


def voxel_downsample(xyz, voxel, return_index=False):
    x, xp = _d(xyz)
    out = np.empty_like(x)
    idx = np.empty(len(x), dtype=np.int64)
    m = lib().orc_voxel_downsample(xp, len(x), voxel, out.ctypes.data_as(c_d_p),
                                   idx.ctypes.data_as(c_i64_p))
    return (out[:m].copy(), idx[:m].copy()) if return_index else out[:m].copy()


def solve6(sums):
    s, sp = _d(sums)
    dx = np.empty(6)
    lib().orc_solve6(sp, dx.ctypes.data_as(c_d_p))
    return dx


class Map:
    """VoxelHashMap restatement"""

    def __init__(self, voxel_size, max_distance, max_points_per_voxel=20, _borrowed=None):
        self._own = _borrowed is None
        self._h = _borrowed if _borrowed is not None else lib().orc_map_create(
            voxel_size, max_distance, max_points_per_voxel)

    def __del__(self):
        if getattr(self, "_own", False) and self._h:
            lib().orc_map_destroy(self._h)
            self._h = None

    def add_points(self, xyz):
        x, xp = _d(xyz)
        lib().orc_map_add_points(self._h, xp, len(x))

    def prune(self, origin):
        o, op = _d(origin)
        lib().orc_map_prune(self._h, op)

    def update(self, xyz, pose):
        x, xp = _d(xyz)
        p, pp = _d(pose)
        lib().orc_map_update(self._h, xp, len(x), pp)

    @property
    def num_voxels(self):
        return lib().orc_map_num_voxels(self._h)

    @property
    def num_points(self):
        return lib().orc_map_num_points(self._h)

    def points(self):
        out = np.empty((self.num_points, 3))
        lib().orc_map_export(self._h, out.ctypes.data_as(c_d_p))
        return out

    def linear_system(self, src, max_dist, kernel, want_targets=False):
        s, sp = _d(src)
        sums = np.empty(27)
        nc, cand = C.c_int64(0), C.c_int64(0)
        tgt = np.empty_like(s) if want_targets else None
        lib().orc_map_linear_system(self._h, sp, len(s), max_dist, kernel, sums.ctypes.data_as(c_d_p),
                                    C.byref(nc), C.byref(cand),
                                    tgt.ctypes.data_as(c_d_p) if want_targets else None)
        return (sums, nc.value, cand.value, tgt) if want_targets else (sums, nc.value, cand.value)

    def register(self, frame, guess, max_dist, kernel, max_iter=500, conv=1e-4):
        f, fp = _d(frame)
        g, gp = _d(guess)
        out = np.empty((4, 4))
        it, nc, cand = C.c_int32(0), C.c_int32(0), C.c_int64(0)
        lib().orc_register(self._h, fp, len(f), gp, max_dist, kernel, max_iter, conv,
                           out.ctypes.data_as(c_d_p), C.byref(it), C.byref(nc), C.byref(cand))
        return out, it.value, nc.value, cand.value


class ICP:
    """Per-scan pipeline of reference kiss.py:83-131 on (N,3) points + per-point t in [0,1)."""

    def __init__(self, max_range=100.0, min_range=5.0, **over):
        self.cfg = IcpCfg()
        lib().orc_icp_default_cfg(C.byref(self.cfg), max_range, min_range)
        for k, v in over.items():
            setattr(self.cfg, k, v)
        self._h = lib().orc_icp_create(C.byref(self.cfg))
        self.stats = []

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_icp_destroy(self._h)
            self._h = None

    def register_frame(self, xyz, t01, guess=None):
        x, xp = _d(xyz)
        t, tp = _opt(t01)
        g, gp = _opt(guess)
        out = np.empty((4, 4))
        st = IcpStats()
        rc = lib().orc_icp_register_frame(self._h, xp, tp, len(x), gp, out.ctypes.data_as(c_d_p), C.byref(st))
        if rc != 0:
            raise MemoryError("oracle register_frame failed")
        self.stats.append(st.as_dict())
        return out

    @property
    def num_poses(self):
        return lib().orc_icp_num_poses(self._h)

    def pose(self, idx=-1):
        out = np.empty((4, 4))
        lib().orc_icp_get_pose(self._h, idx, out.ctypes.data_as(c_d_p))
        return out

    def poses(self):
        return [self.pose(i) for i in range(self.num_poses)]

    def prediction(self):
        out = np.empty((4, 4))
        lib().orc_icp_prediction(self._h, out.ctypes.data_as(c_d_p))
        return out

    def has_moved(self):
        return bool(lib().orc_icp_has_moved(self._h))

    @property
    def map(self):
        return Map(0, 0, _borrowed=lib().orc_icp_map(self._h))

    def last_frame_down(self):
        n = lib().orc_icp_last_frame_down(self._h, None)
        out = np.empty((n, 3))
        lib().orc_icp_last_frame_down(self._h, out.ctypes.data_as(c_d_p))
        return out

    def last_source(self):
        n = lib().orc_icp_last_source(self._h, None)
        out = np.empty((n, 3))
        lib().orc_icp_last_source(self._h, out.ctypes.data_as(c_d_p))
        return out


class EKF:
    """ES-EKF restatement (reference ins/es_ekf.py)"""

    def __init__(self, init_grav=None, init_bacc=None, init_bgyr=None):
        a, ap = _opt(init_grav)
        b, bp = _opt(init_bacc)
        c, cp = _opt(init_bgyr)
        self._h = lib().orc_ekf_create(ap, bp, cp)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_ekf_destroy(self._h)
            self._h = None

    def process_imu(self, lacc, avel, ts):
        a, ap = _d(lacc)
        w, wp = _d(avel)
        lib().orc_ekf_process_imu(self._h, ap, wp, float(ts))

    def process_pose(self, pose, meas_cov=None):
        p, pp = _d(pose)
        c, cp = _opt(meas_cov)
        lib().orc_ekf_process_pose(self._h, pp, cp)

    @property
    def nav(self):
        out = np.empty(19)
        lib().orc_ekf_get_nav(self._h, out.ctypes.data_as(c_d_p))
        return out

    @property
    def cov(self):
        out = np.empty((18, 18))
        lib().orc_ekf_get_cov(self._h, out.ctypes.data_as(c_d_p))
        return out

    def pose_mat(self):
        out = np.empty((4, 4))
        lib().orc_ekf_pose_mat(self._h, out.ctypes.data_as(c_d_p))
        return out

    @property
    def ts(self):
        return lib().orc_ekf_ts(self._h)


def calc_ate(nav_poses, gt_poses):
    a, ap = _d(np.asarray(nav_poses))
    b, bp = _d(np.asarray(gt_poses))
    assert a.shape == b.shape and len(a)
    out = np.empty(2)
    rc = lib().orc_calc_ate(ap, bp, len(a), out.ctypes.data_as(c_d_p))
    assert rc == 0
    return out[0], out[1]


def run_sequence(events, *, max_range=70.0, min_range=1.0, use_imu_prediction=False, with_ekf=True,
                 max_scans=None, **icp_over):
    """The reference's driver loop (cli/ekf_bench.py:493-563) over an event list.

    events: iterable of ("imu", lacc, avel, ts) and ("scan", xyz(N,3), t01(N,)).
    Returns dict(res_t, res_poses, kiss_poses, stats).
    """
    icp = ICP(max_range=max_range, min_range=min_range, **icp_over)
    ekf = EKF() if with_ekf else None
    res_t, res_poses, kiss_poses = [], [], []
    imus_per_scan = 1  # ekf_bench.py:491
    n_scans = 0
    for ev in events:
        if ev[0] == "imu":
            if ekf is not None:
                ekf.process_imu(ev[1], ev[2], ev[3])
            imus_per_scan += 1
        else:
            if with_ekf and not imus_per_scan:  # ekf_bench.py:512-518
                continue
            imus_per_scan = 0
            guess = ekf.pose_mat() if (use_imu_prediction and ekf is not None) else None
            pose = icp.register_frame(ev[1], ev[2], guess)
            kiss_poses.append(pose)
            if ekf is not None:
                ekf.process_pose(pose)
                res_poses.append(ekf.pose_mat())
                res_t.append(ekf.ts)
            n_scans += 1
            if max_scans is not None and n_scans >= max_scans:
                break
    return dict(res_t=np.array(res_t), res_poses=np.array(res_poses), kiss_poses=np.array(kiss_poses),
                stats=icp.stats, icp=icp)
