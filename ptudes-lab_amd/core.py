"""Thin object layer over the C-ABI: `Icp` (one KissICP-equivalent on one GPU), `Ekf`, `SeqRunner`
(the reference's driver loop, cli/ekf_bench.py:493-563, on a sequence resident in HBM)."""
import ctypes as C

import numpy as np

from . import _lib as L


def icp_cfg(max_range=100.0, min_range=5.0, **over):
    """Reference defaults for (max_range, min_range) (kiss.py:21-43) with overrides by field name."""
    cfg = L.IcpCfg()
    L.check(L.lib().ptl_icp_default_cfg(C.byref(cfg), float(max_range), float(min_range)))
    for k, v in over.items():
        if not hasattr(cfg, k):
            raise ValueError(f"unknown icp cfg field {k}")
        setattr(cfg, k, v)
    return cfg


def ekf_cfg(init_grav=None, init_bacc=None, init_bgyr=None, device_id=0):
    cfg = L.EkfCfg()
    L.check(L.lib().ptl_ekf_default_cfg(C.byref(cfg)))
    for name, val in (("init_grav", init_grav), ("init_bacc", init_bacc), ("init_bgyr", init_bgyr)):
        if val is not None:
            v = np.asarray(val, dtype=np.float64).reshape(3)
            setattr(cfg, name, (C.c_double * 3)(*v))
    cfg.device_id = device_id
    return cfg


class Icp:
    def __init__(self, max_range=100.0, min_range=5.0, lazy_map_stats=False, **over):
        """lazy_map_stats: registrations return when the pose is there, before the scan's map update is complete (the per-scan rows then
        carry map_voxels = map_points = -1; `map_size()` gives them on demand) - include/ptudes_mi.h ptl_icp_set_lazy_map_stats"""
        self.cfg = icp_cfg(max_range, min_range, **over)
        self._h = C.c_void_p()
        L.check(L.lib().ptl_icp_create(C.byref(self.cfg), C.byref(self._h)))
        if lazy_map_stats:
            L.check(L.lib().ptl_icp_set_lazy_map_stats(self._h, 1))
        self.stats = []

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ptl_icp_destroy(self._h)
            self._h = None

    __del__ = close

    def register_frame(self, xyz, t01=None, guess=None, scan_ts=0.0):
        xyz = np.asarray(xyz)
        if xyz.dtype == np.float32:
            x = np.ascontiguousarray(xyz)
            dt = L.PTL_F32
        else:
            x = L.as_f64(xyz)
            dt = L.PTL_F64
        if x.ndim != 2 or x.shape[1] != 3:
            raise ValueError("xyz must be (N, 3)")
        t = None if t01 is None else L.as_f64(t01)
        if t is not None and len(t) != len(x):
            raise ValueError("t01 must have one entry per point")
        g = None if guess is None else L.as_f64(guess).reshape(16)
        out = np.empty((4, 4))
        st = L.IcpStats()
        L.check(L.lib().ptl_icp_register_frame(self._h, x.ctypes.data_as(C.c_void_p), dt, len(x),
                                               None if t is None else L.dptr(t), float(scan_ts),
                                               None if g is None else L.dptr(g), L.dptr(out), C.byref(st)))
        self.stats.append(st.as_dict())
        return out

    def register_range(self, lut, range_mm, guess=None, scan_ts=0.0):
        """register a raw range image (H*W u32 mm); xyz, the RANGE != 0 mask and column times happen on device"""
        r = np.ascontiguousarray(range_mm, dtype=np.uint32).reshape(-1)
        if r.size != lut.H * lut.W:
            raise ValueError("range image size mismatch")
        g = None if guess is None else L.as_f64(guess).reshape(16)
        out = np.empty((4, 4))
        st = L.IcpStats()
        L.check(L.lib().ptl_icp_register_range(self._h, lut._h, r.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               float(scan_ts), None if g is None else L.dptr(g), L.dptr(out),
                                               C.byref(st)))
        self.stats.append(st.as_dict())
        return out

    def set_active_beams(self, H, beams_num):
        L.check(L.lib().ptl_icp_set_active_beams(self._h, int(H), int(beams_num)))

    @property
    def num_poses(self):
        n = C.c_int64()
        L.check(L.lib().ptl_icp_num_poses(self._h, C.byref(n)))
        return n.value

    def poses(self):
        n = self.num_poses
        out = np.empty((max(n, 1), 4, 4))
        w = C.c_int64()
        L.check(L.lib().ptl_icp_get_poses(self._h, L.dptr(out), n, C.byref(w)))
        return out[:w.value]

    def prediction(self):
        out = np.empty((4, 4))
        L.check(L.lib().ptl_icp_get_prediction(self._h, L.dptr(out)))
        return out

    def map_size(self):
        v, p = C.c_int64(), C.c_int64()
        L.check(L.lib().ptl_icp_map_size(self._h, C.byref(v), C.byref(p)))
        return v.value, p.value

    def map_points(self):
        _, p = self.map_size()
        out = np.empty((max(p, 1), 3))
        w = C.c_int64()
        L.check(L.lib().ptl_icp_map_points(self._h, L.dptr(out), p, C.byref(w)))
        return out[:w.value]

    def _cloud(self, fn):
        cap = int(self.cfg.max_points_per_scan)
        out = np.empty((cap, 3))
        w = C.c_int64()
        L.check(fn(self._h, L.dptr(out), cap, C.byref(w)))
        return out[:w.value].copy()

    def last_frame_down(self):
        return self._cloud(L.lib().ptl_icp_last_frame_down)

    def last_source(self):
        return self._cloud(L.lib().ptl_icp_last_source)

    def deskew(self, xyz, t01):
        x, t = L.as_f64(xyz), L.as_f64(t01)
        out = np.empty_like(x)
        L.check(L.lib().ptl_icp_deskew(self._h, L.dptr(x), L.dptr(t), len(x), L.dptr(out)))
        return out

    # stage-level entry points (teacher-forced parity)
    def map_add(self, xyz_world, origin=None):
        x = L.as_f64(xyz_world)
        o = None if origin is None else L.as_f64(origin)
        L.check(L.lib().ptl_icp_map_add(self._h, L.dptr(x), len(x), None if o is None else L.dptr(o),
                                        0 if o is None else 1))

    def linear_system(self, src_world, max_dist, kernel):
        s = L.as_f64(src_world)
        sums = np.empty(27)
        nc, cand = C.c_int64(), C.c_int64()
        L.check(L.lib().ptl_icp_linear_system(self._h, L.dptr(s), len(s), max_dist, kernel, L.dptr(sums),
                                              C.byref(nc), C.byref(cand)))
        return sums, nc.value, cand.value

    def align(self, frame, guess, max_dist, kernel):
        f = L.as_f64(frame)
        g = L.as_f64(guess).reshape(16)
        out = np.empty((4, 4))
        it = C.c_int32()
        L.check(L.lib().ptl_icp_align(self._h, L.dptr(f), len(f), L.dptr(g), max_dist, kernel, L.dptr(out),
                                      C.byref(it)))
        return out, it.value


class Lut:
    """range image -> xyz on device (ouster client.XYZLut equivalent, reference kiss.py:28-29)"""

    def __init__(self, H, W, beam_altitude_deg, beam_azimuth_deg, lidar_origin_to_beam_origin_mm=0.0,
                 lidar_to_sensor_mm=None, extrinsic_m=None, device_id=0):
        self.H, self.W = int(H), int(W)
        alt, az = L.as_f64(beam_altitude_deg), L.as_f64(beam_azimuth_deg)
        if len(alt) != H or len(az) != H:
            raise ValueError("need one altitude / azimuth angle per row")
        T = L.as_f64(np.eye(4) if lidar_to_sensor_mm is None else lidar_to_sensor_mm).reshape(16)
        E = None if extrinsic_m is None else L.as_f64(extrinsic_m).reshape(16)
        self._h = C.c_void_p()
        L.check(L.lib().ptl_lut_create(device_id, self.H, self.W, L.dptr(alt), L.dptr(az),
                                       float(lidar_origin_to_beam_origin_mm), L.dptr(T),
                                       None if E is None else L.dptr(E), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ptl_lut_destroy(self._h)
            self._h = None

    __del__ = close

    def __call__(self, range_mm):
        r = np.ascontiguousarray(range_mm, dtype=np.uint32).reshape(-1)
        if r.size != self.H * self.W:
            raise ValueError("range image size mismatch")
        out = np.empty((self.H * self.W, 3))
        L.check(L.lib().ptl_lut_apply(self._h, r.ctypes.data_as(C.POINTER(C.c_uint32)), L.dptr(out)))
        return out


    def dewarp(self, range_mm, col_poses):
        """client.dewarp(XYZLut(scan), column_poses=scan.pose): (H*W, 3) world xyz and the number of returns"""
        r = np.ascontiguousarray(range_mm, dtype=np.uint32).reshape(-1)
        P = L.as_f64(col_poses).reshape(-1, 16)
        if r.size != self.H * self.W or len(P) != self.W:
            raise ValueError("need an H x W range image and one 4x4 pose per column")
        out = np.empty((self.H * self.W, 3))
        nv = C.c_int64()
        L.check(L.lib().ptl_lut_dewarp(self._h, r.ctypes.data_as(C.POINTER(C.c_uint32)), L.dptr(P), L.dptr(out), C.byref(nv)))
        return out, nv.value


def traj_poses_at(knot_ts, knot_poses, ts, bound_before=0.0, bound_after=0.0, device_id=0):
    """TrajectoryEvaluator.poses_at on device: (n, 4, 4) poses and the number of timestamps outside the bounds"""
    kt, kp, t = L.as_f64(knot_ts), L.as_f64(knot_poses).reshape(-1, 16), L.as_f64(ts).reshape(-1)
    if len(kt) != len(kp):
        raise ValueError("one pose per knot timestamp")
    out = np.empty((len(t), 4, 4))
    nout = C.c_int64()
    L.check(L.lib().ptl_traj_poses_at(device_id, L.dptr(kt), L.dptr(kp), len(kt), float(bound_before), float(bound_after),
                                      L.dptr(t), len(t), L.dptr(out), C.byref(nout)))
    return out, nout.value


class Ekf:
    def __init__(self, init_grav=None, init_bacc=None, init_bgyr=None, device_id=0):
        self.cfg = ekf_cfg(init_grav, init_bacc, init_bgyr, device_id)
        self._h = C.c_void_p()
        L.check(L.lib().ptl_ekf_create(C.byref(self.cfg), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ptl_ekf_destroy(self._h)
            self._h = None

    __del__ = close

    def process_imu(self, lacc, avel, ts):
        a, w = L.as_f64(lacc), L.as_f64(avel)
        L.check(L.lib().ptl_ekf_process_imu(self._h, L.dptr(a), L.dptr(w), float(ts)))

    def process_imu_batch(self, rows):
        r = L.as_f64(rows)
        L.check(L.lib().ptl_ekf_process_imu_batch(self._h, L.dptr(r), len(r)))

    def process_pose(self, pose, meas_cov=None):
        p = L.as_f64(pose).reshape(16)
        c = None if meas_cov is None else L.as_f64(meas_cov).reshape(36)
        L.check(L.lib().ptl_ekf_process_pose(self._h, L.dptr(p), None if c is None else L.dptr(c)))

    def state(self):
        nav, cov = np.empty(19), np.empty((18, 18))
        L.check(L.lib().ptl_ekf_get_state(self._h, L.dptr(nav), L.dptr(cov)))
        return nav, cov

    @property
    def nav(self):
        return self.state()[0]

    @property
    def cov(self):
        return self.state()[1]

    def pose_mat(self):
        out = np.empty((4, 4))
        L.check(L.lib().ptl_ekf_pose_mat(self._h, L.dptr(out)))
        return out

    @property
    def ts(self):
        t = C.c_double()
        L.check(L.lib().ptl_ekf_ts(self._h, C.byref(t)))
        return t.value


def icp_ekf_step(icp: "Icp", ekf: "Ekf", imu_rows, xyz, t01=None, guess=None, use_imu_prediction=False):
    """One scan of the reference's loop body (cli/ekf_bench.py:493-563) in one host round trip (include/ptudes_mi.h ptl_icp_ekf_step):
    the IMU rows [ts, lacc, avel] that precede the scan, the registration (guess: the filter's pose when use_imu_prediction, else
    `guess` / constant velocity), the filter's update with the new pose.  Returns (kiss_pose, ekf_pose, ekf_ts); the stats row is
    appended to icp.stats like register_frame does."""
    rows = L.as_f64(imu_rows).reshape(-1, 7) if len(imu_rows) else np.zeros((0, 7))
    xyz = np.asarray(xyz)
    if xyz.dtype == np.float32:
        x, dt = np.ascontiguousarray(xyz), L.PTL_F32
    else:
        x, dt = L.as_f64(xyz), L.PTL_F64
    if x.ndim != 2 or x.shape[1] != 3:
        raise ValueError("xyz must be (N, 3)")
    t = None if t01 is None else L.as_f64(t01)
    if t is not None and len(t) != len(x):
        raise ValueError("t01 must have one entry per point")
    g = None if guess is None else L.as_f64(guess).reshape(16)
    kiss, pose, ts, st = np.empty((4, 4)), np.empty((4, 4)), C.c_double(), L.IcpStats()
    L.check(L.lib().ptl_icp_ekf_step(icp._h, ekf._h, L.dptr(rows) if len(rows) else None, len(rows), x.ctypes.data_as(C.c_void_p), dt, len(x),
                                     None if t is None else L.dptr(t), None if g is None else L.dptr(g), int(bool(use_imu_prediction)),
                                     L.dptr(kiss), L.dptr(pose), C.byref(ts), C.byref(st)))
    icp.stats.append(st.as_dict())
    return kiss, pose, ts.value


def device_sync(device_id=0):
    L.check(L.lib().ptl_device_sync(device_id))


def host_pin(array, device_id=0):
    """page-lock a C-contiguous numpy array the sweeps will be uploaded from (include/ptudes_mi.h ptl_host_pin); keep the array alive until host_unpin"""
    if not array.flags["C_CONTIGUOUS"]:
        raise ValueError("host_pin needs a C-contiguous array")
    L.check(L.lib().ptl_host_pin(device_id, C.c_void_p(array.ctypes.data), array.nbytes))


def host_unpin(array):
    L.check(L.lib().ptl_host_unpin(C.c_void_p(array.ctypes.data)))


class SeqRunner:
    """Whole sequence in HBM, no host round trip per scan."""

    def __init__(self, n_scans, points_per_scan, n_imu, *, max_range=70.0, min_range=1.0, use_imu_prediction=False,
                 with_ekf=True, device_id=0, ekf=None, **icp_over):
        cfg = L.SeqCfg()
        cfg.icp = icp_cfg(max_range, min_range, device_id=device_id, **icp_over)
        cfg.ekf = ekf if ekf is not None else ekf_cfg(device_id=device_id)
        cfg.n_scans, cfg.points_per_scan, cfg.n_imu = n_scans, points_per_scan, n_imu
        cfg.use_imu_prediction = int(bool(use_imu_prediction))
        cfg.with_ekf = int(bool(with_ekf))
        self.cfg = cfg
        self._h = C.c_void_p()
        L.check(L.lib().ptl_seq_create(C.byref(cfg), C.byref(self._h)))
        self.n_scans = n_scans

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ptl_seq_destroy(self._h)
            self._h = None

    __del__ = close

    def upload_scan(self, k, xyz_f32):
        x = np.ascontiguousarray(xyz_f32, dtype=np.float32)
        if x.size != self.cfg.points_per_scan * 3:
            raise ValueError("scan size mismatch")
        L.check(L.lib().ptl_seq_upload_scan(self._h, k, x.ctypes.data_as(C.POINTER(C.c_float))))

    def upload_range(self, k, range_mm):
        r = np.ascontiguousarray(range_mm, dtype=np.uint32).reshape(-1)
        if r.size != self.cfg.points_per_scan:
            raise ValueError("range image size mismatch")
        L.check(L.lib().ptl_seq_upload_range(self._h, k, r.ctypes.data_as(C.POINTER(C.c_uint32))))

    def set_lut(self, lut, active_beams=0):
        self._lut = lut  # keep alive
        L.check(L.lib().ptl_seq_set_lut(self._h, lut._h, int(active_beams)))

    def upload_imu(self, imu_rows, imu_end):
        r = L.as_f64(imu_rows).reshape(-1, 7) if len(imu_rows) else np.zeros((0, 7))
        e = np.ascontiguousarray(imu_end, dtype=np.int64)
        if len(e) != self.n_scans:
            raise ValueError("imu_end needs one entry per scan")
        L.check(L.lib().ptl_seq_upload_imu(self._h, L.dptr(r) if len(r) else None, e.ctypes.data_as(L.c_i64_p)))

    def run(self, n=None):
        L.check(L.lib().ptl_seq_run(self._h, self.n_scans if n is None else n))

    def advance(self, n):
        L.check(L.lib().ptl_seq_advance(self._h, n))

    def enqueue(self, n):
        L.check(L.lib().ptl_seq_enqueue(self._h, n))

    def wait(self):
        L.check(L.lib().ptl_seq_wait(self._h))

    def copy_traj(self, dst_device_ptr, max_rows):
        rows = C.c_int64()
        L.check(L.lib().ptl_seq_copy_traj(self._h, C.c_void_p(dst_device_ptr), max_rows, C.byref(rows)))
        return rows.value

    def results(self):
        n = self.n_scans
        res_poses, res_t, kiss = np.empty((n, 4, 4)), np.empty(n), np.empty((n, 4, 4))
        stats = (L.IcpStats * n)()
        w = C.c_int64()
        L.check(L.lib().ptl_seq_results(self._h, L.dptr(res_poses), L.dptr(res_t), L.dptr(kiss), stats, n,
                                        C.byref(w)))
        m = w.value
        out = dict(kiss_poses=kiss[:m], stats=[stats[i].as_dict() for i in range(m)])
        if self.cfg.with_ekf:
            out.update(res_poses=res_poses[:m], res_t=res_t[:m])
        return out

    def traj_device(self):
        p, rows = C.c_void_p(), C.c_int64()
        L.check(L.lib().ptl_seq_traj_device(self._h, C.byref(p), C.byref(rows)))
        return p.value, rows.value

    def profile(self, enable=True, reset=False):
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().ptl_seq_profile(self._h, int(enable), C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value


class BatchRunner:
    """Up to 256 independent sequences on one GPU (lockstep: 32); sequence s lives on XCD s & 7 (the workgroups with
    blockIdx & 7 == s & 7).  Two drivers (`free_running`): True (the default with the 8-lane Gauss-Newton kernel) - one
    persistent launch carries up to `scans_per_launch` scans of every sequence, teams of `team_workgroups` workgroups take
    the scans of their XCD's sequences as they come free; False - lockstep, one launch per stage for all sequences, a step
    lasts as long as its slowest sequence.  Either way the per-sequence results are bit-identical to
    `SeqRunner(..., gn_workgroups=<workgroups per team>, gn_lanes_per_point=<the batch's>)`."""

    def __init__(self, n_sequences, n_scans, points_per_scan, n_imu, *, max_range=70.0, min_range=1.0,
                 use_imu_prediction=False, with_ekf=True, device_id=0, ekf=None, free_running=None, scans_per_launch=0,
                 team_workgroups=0, range_input=False, resident_scans=0, **icp_over):
        """range_input: every sweep will arrive as a raw range image (set_lut + upload_range) and stays one in HBM - 4 bytes per pixel
        resident instead of 12 (include/ptudes_mi.h ptl_seq_cfg.range_input).  resident_scans: R >= 2 = a ring of R sweep slots per sequence
        instead of all n_scans - sweeps are uploaded in order, later ones while a launch works on earlier ones (ptl_seq_cfg.resident_scans)"""
        cfg = L.SeqCfg()
        cfg.range_input = int(bool(range_input))
        cfg.resident_scans = int(resident_scans)
        icp_over.setdefault("gn_lanes_per_point", 8)  # a workgroup walks ~200 points per iteration here: the throughput form
        if icp_over["gn_lanes_per_point"] == 8:
            icp_over.setdefault("gn_threads", 512)
        cfg.icp = icp_cfg(max_range, min_range, device_id=device_id, **icp_over)
        cfg.ekf = ekf if ekf is not None else ekf_cfg(device_id=device_id)
        cfg.n_scans, cfg.points_per_scan, cfg.n_imu = n_scans, points_per_scan, n_imu
        cfg.use_imu_prediction = int(bool(use_imu_prediction))
        cfg.with_ekf = int(bool(with_ekf))
        self.cfg, self.S, self.n_scans = cfg, int(n_sequences), n_scans
        self._h = C.c_void_p()
        L.check(L.lib().ptl_batch_create(C.byref(cfg), self.S, C.byref(self._h)))
        self.free_running = (cfg.icp.gn_lanes_per_point == 8) if free_running is None else bool(free_running)
        if free_running is not None or scans_per_launch:
            L.check(L.lib().ptl_batch_set_driver(self._h, int(self.free_running), int(scans_per_launch)))
        if team_workgroups:
            L.check(L.lib().ptl_batch_set_team_workgroups(self._h, int(team_workgroups)))

    def set_driver(self, free_running, scans_per_launch=0, team_workgroups=None):
        """choose the driver (and the team size) again for the same handle and sweeps: back to the cold start first"""
        L.check(L.lib().ptl_batch_reset(self._h))
        L.check(L.lib().ptl_batch_set_driver(self._h, int(bool(free_running)), int(scans_per_launch)))
        self.free_running = bool(free_running)
        if team_workgroups is not None:
            L.check(L.lib().ptl_batch_set_team_workgroups(self._h, int(team_workgroups)))

    def team_geometry(self):
        """(workgroups per team, teams that can get work) of the free-running kernel"""
        g, t = C.c_int32(), C.c_int32()
        L.check(L.lib().ptl_batch_team_workgroups(self._h, C.byref(g), C.byref(t)))
        return g.value, t.value

    EXEC_COUNTERS = ("searches", "rows_rebuilt", "map_points_read", "gn_iterations", "vds1_claims", "vds2_claims",
                     "point_iterations", "scans", "settled_nothing_in_reach")

    def exec_counters(self, s):
        """executed-work counters of sequence s, cumulative since the cold start (include/ptudes_mi.h ptl_batch_exec_counters)"""
        out = (C.c_uint64 * 16)()
        L.check(L.lib().ptl_batch_exec_counters(self._h, s, out))
        return dict(zip(self.EXEC_COUNTERS, (int(v) for v in out)))

    def sched_counters(self, s):
        """where the scans of sequence s ran: scans taken by a team of another XCD, cross-XCD hand-overs, last XCC id + 1"""
        out = (C.c_uint64 * 4)()
        L.check(L.lib().ptl_batch_sched_counters(self._h, s, out))
        return dict(stolen=int(out[0]), cross_xcd_handovers=int(out[1]), last_xcc=int(out[2]) - 1)

    def status(self):
        """sticky status word of the free-running driver (include/ptudes_mi.h ptl_batch_status)"""
        v = C.c_uint32()
        L.check(L.lib().ptl_batch_status(self._h, C.byref(v)))
        return v.value

    def debug_stall_block(self, block, round=0):
        L.check(L.lib().ptl_batch_debug_stall_block(self._h, int(block), int(round)))

    def debug_map_points_per_thread(self, points=0):
        """test hook: the free-running map update's points per thread (0 = ask); returns the value in effect"""
        v = C.c_int32()
        L.check(L.lib().ptl_batch_debug_set_map_points_per_thread(self._h, int(points), C.byref(v)))
        return v.value

    def close(self):
        if getattr(self, "_h", None):
            L.lib().ptl_batch_destroy(self._h)
            self._h = None

    __del__ = close

    def upload_scan(self, s, k, xyz_f32):
        x = np.ascontiguousarray(xyz_f32, dtype=np.float32)
        if x.size != self.cfg.points_per_scan * 3:
            raise ValueError("scan size mismatch")
        L.check(L.lib().ptl_batch_upload_scan(self._h, s, k, x.ctypes.data_as(C.POINTER(C.c_float))))

    def set_lut(self, lut, active_beams=0):
        self._lut = lut
        L.check(L.lib().ptl_batch_set_lut(self._h, lut._h, int(active_beams)))

    def upload_range(self, s, k, range_mm):
        r = np.ascontiguousarray(range_mm, dtype=np.uint32).reshape(-1)
        L.check(L.lib().ptl_batch_upload_range(self._h, s, k, r.ctypes.data_as(C.POINTER(C.c_uint32))))

    def upload_imu(self, s, imu_rows, imu_end):
        r = L.as_f64(imu_rows).reshape(-1, 7) if len(imu_rows) else np.zeros((0, 7))
        e = np.ascontiguousarray(imu_end, dtype=np.int64)
        if len(e) != self.n_scans:
            raise ValueError("imu_end needs one entry per scan")
        L.check(L.lib().ptl_batch_upload_imu(self._h, s, L.dptr(r) if len(r) else None, e.ctypes.data_as(L.c_i64_p)))

    def run(self, n=None):
        L.check(L.lib().ptl_batch_run(self._h, self.n_scans if n is None else n))

    def enqueue(self, n):
        L.check(L.lib().ptl_batch_enqueue(self._h, n))

    def wait(self):
        L.check(L.lib().ptl_batch_wait(self._h))

    def results(self, s):
        n = self.n_scans
        res_poses, res_t, kiss = np.empty((n, 4, 4)), np.empty(n), np.empty((n, 4, 4))
        stats = (L.IcpStats * n)()
        w = C.c_int64()
        L.check(L.lib().ptl_batch_results(self._h, s, L.dptr(res_poses), L.dptr(res_t), L.dptr(kiss), stats, n,
                                          C.byref(w)))
        m = w.value
        out = dict(kiss_poses=kiss[:m], stats=[stats[i].as_dict() for i in range(m)])
        if self.cfg.with_ekf:
            out.update(res_poses=res_poses[:m], res_t=res_t[:m])
        return out

    def copy_traj(self, s, dst_device_ptr, max_rows):
        rows = C.c_int64()
        L.check(L.lib().ptl_batch_copy_traj(self._h, s, C.c_void_p(dst_device_ptr), max_rows, C.byref(rows)))
        return rows.value

    def seq_clocks(self, s):
        """per-scan mean microseconds of sequence s in the free-running kernel: (K0-K4, wait, GN, wait, map update, filter)"""
        out = (C.c_int64 * 8)()
        L.check(L.lib().ptl_batch_seq_clocks(self._h, s, out))
        n = max(out[6], 1)
        return tuple(out[i] / n / 100.0 for i in range(6))

    def seq_clocks_raw(self, s):
        """the same clocks as they are kept: (six sums of 100 MHz ticks since the cold start, scans) - a caller that wants the means over
        SOME of the scans (the timed ones, without the warm-up) takes differences"""
        out = (C.c_int64 * 8)()
        L.check(L.lib().ptl_batch_seq_clocks(self._h, s, out))
        return tuple(int(out[i]) for i in range(6)), int(out[6])

    def profile(self, enable=True, reset=False):
        ms, n = C.c_double(), C.c_int64()
        L.check(L.lib().ptl_batch_profile(self._h, int(enable), C.byref(ms), C.byref(n), int(reset)))
        return ms.value, n.value


BUILD_INFO = ("kcand", "ans_row_doubles", "lds_points", "seq_u", "seq_u2", "gn8_threads", "lanes_per_point", "spec", "surv",
              "prefetch", "keep_x1000", "tab_entry_bytes", "vds_entry_bytes", "diagnostics", "fast")


def build_info():
    """compile-time constants of the loaded library (include/ptudes_mi.h ptl_build_info)"""
    out = (C.c_int32 * 16)()
    L.check(L.lib().ptl_build_info(out))
    return dict(zip(BUILD_INFO, (int(v) for v in out)))

