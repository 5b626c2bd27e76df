"""The reference's driver loop (cli/ekf_bench.py:493-563) over an event stream, two ways:

`run_events`   per-call API (KissICPWrapper + ESEKF objects, one host round trip per event) - what the
               reference's loop does, for arbitrary feeds;
`run_resident` the same loop on a sequence uploaded to HBM once (`core.SeqRunner`, no host round trip per scan).

Events are ("imu", IMU) and ("scan", xyz (N,3), t01 (N,) or None, ts) in packet order.
"""
import time

import numpy as np

from . import core
from .ins.data import IMU
from .ins.es_ekf import ESEKF
from .kiss import KissICPWrapper


def run_events(events, metadata, *, kiss_min_range=1.0, kiss_max_range=70.0, use_imu_prediction=False,
               guess_fn=None, logging=False, device_id=0, stats=None, lazy_map_stats=True, fused=None):
    """Returns dict(res_t, res_poses, kiss_poses, kiss_icp, ekf, timings).  `guess_fn(ts)` (optional) supplies an
    external guess (the reference's --use-gt-guess, ekf_bench.py:536-542); `stats` (optional) is the StreamStatsTracker
    the loop feeds (ekf_bench.py:497-499, :522-524), its time goes into timings["track"].
    fused (default: on unless the filter logs every sample): the IMU samples between two scans are held back and go to the device
    together with the scan - ptl_icp_ekf_step: predicts, registration with the filter's pose read on the device, update, ONE wait
    per scan instead of a read-back for the guess, one for the pose and one for the filter's state.  Same kernels in the same order:
    the trajectory is the device-resident runner's bit for bit (the guess never leaves the device in either) and the call-by-call
    loop's to 1e-9 m (there the guess makes a round trip through host arithmetic) - tests/test_gpu_dropin.py.  Any number of IMU
    samples between two scans is served (the entry point feeds what exceeds its staging buffer in chunks).
    timings: in the fused form `kiss` is the whole loop body of a scan (predicts + registration + update, one call) and `imu` /
    `corr` only count the host's hold-back (about 0): the reference's three per-stage means (ekf_bench.py:590-595) exist
    separately only with fused=False; timings["fused"] says which form ran."""
    if fused is None:
        fused = not logging
    # (the loop reads poses only - ekf_bench.py:549-563: the registration need not wait for its map update, lazy_map_stats)
    kiss_icp = KissICPWrapper(metadata, _use_extrinsics=True, _min_range=kiss_min_range, _max_range=kiss_max_range,
                              device_id=device_id, lazy_map_stats=lazy_map_stats)
    ekf = ESEKF(_logging=logging, device_id=device_id)
    res_t, res_poses, kiss_poses = [], [], []
    t_imu = t_corr = t_kiss = t_track = 0.0
    n_imu = n_corr = 0
    imus_per_scan = 1  # ekf_bench.py:491
    held = []  # fused: the IMU samples since the last scan
    for ev in events:
        if ev[0] == "imu":
            if stats is not None:
                t1 = time.monotonic()
                stats.trackImu(ev[1])
                t_track += time.monotonic() - t1
            t1 = time.monotonic()
            if fused:
                # what ESEKF.processImu does to the sample on the host before anything else (es_ekf.py:194-196): imu.dt = ts - the previous sample's ts (0.0 before the first)
                ev[1].dt = ev[1].ts - (held[-1].ts if held else ekf.ts)
                held.append(ev[1])
            else:
                ekf.processImu(ev[1])
            t_imu += time.monotonic() - t1
            n_imu += 1
            imus_per_scan += 1
            continue
        if not imus_per_scan:  # ekf_bench.py:512-518
            continue
        imus_per_scan = 0
        if stats is not None:  # ekf_bench.py:522-524 (before the beams are reduced)
            t1 = time.monotonic()
            if ev[0] == "lidar_scan":
                stats.trackScan(ev[1])
            else:  # points: the ranges are their norms (mm, like the sensor's RANGE field)
                W = int(getattr(getattr(metadata, "format", None), "columns_per_frame", 0)) or len(ev[1])
                rng = np.round(np.linalg.norm(np.asarray(ev[1], dtype=np.float64), axis=1) * 1000.0).astype(np.uint32).reshape(-1, W)
                stats.trackScan(rng, last_valid_column_ts_ns=int(round(ev[3] * 1e9)))
            t_track += time.monotonic() - t1
        if ev[0] == "lidar_scan":  # an ouster LidarScan from the packet feed (data.OusterLidarData)
            xyz, t01, ts = None, None, float(getattr(ev[1], "timestamp", [0])[-1]) * 1e-9 if hasattr(ev[1], "timestamp") else 0.0
        else:
            _, xyz, t01, ts = ev
        if fused and ev[0] != "lidar_scan":
            # the whole loop body in one call (the guess of --use-imu-prediction is read on the device)
            if use_imu_prediction:
                guess = None
            elif guess_fn is not None:
                guess = guess_fn(ts)
            else:
                guess = None  # constant velocity, formed on the device (kiss.py:102-105)
            t1 = time.monotonic()
            rows = np.array([[i.ts, *i.lacc, *i.avel] for i in held], dtype=np.float64).reshape(-1, 7)
            kiss_pose, ekf_pose, ekf_ts = core.icp_ekf_step(kiss_icp._icp, ekf._ekf, rows, xyz, t01, guess, use_imu_prediction)
            kiss_icp._log_pose(kiss_pose, ts)
            ekf._after_fused_step(len(held), ekf_ts)
            held = []
            t_kiss += time.monotonic() - t1
            n_corr += 1
            kiss_poses.append(kiss_pose)
            res_poses.append(ekf_pose)
            res_t.append(ekf.ts)
            continue
        if held:  # (a scan type the fused entry does not take: flush the samples the call-by-call way)
            for i in held:
                ekf.processImu(i)
            held = []
        if use_imu_prediction:
            guess = ekf.nav.pose_mat()
        elif guess_fn is not None:
            guess = guess_fn(ts)
        else:
            last = kiss_icp._kiss.poses[-1] if kiss_icp._kiss.poses else np.eye(4)
            guess = last @ kiss_icp._kiss.get_prediction_model()
        t1 = time.monotonic()
        if ev[0] == "lidar_scan":
            kiss_icp.register_frame(ev[1], initial_guess=guess)  # reference ekf_bench.py:549
        else:
            kiss_icp.register_points(xyz, t01, ts, initial_guess=guess)
        t_kiss += time.monotonic() - t1
        t1 = time.monotonic()
        ekf.processPose(kiss_icp.pose)
        t_corr += time.monotonic() - t1
        n_corr += 1
        kiss_poses.append(kiss_icp.pose)
        res_poses.append(ekf.nav.pose_mat())
        res_t.append(ekf.ts)
    if held:  # samples behind the last scan
        for i in held:
            ekf.processImu(i)
    timings = dict(imu=t_imu / max(n_imu, 1), corr=t_corr / max(n_corr, 1), kiss=t_kiss / max(n_corr, 1),
                   track=t_track / max(n_corr, 1), n_imu=n_imu, n_corr=n_corr, fused=bool(fused))
    return dict(res_t=res_t, res_poses=res_poses, kiss_poses=kiss_poses, kiss_icp=kiss_icp, ekf=ekf, timings=timings)


def synthetic_events(seq, n_scans=None):
    """event stream of a synth.Sequence in the reference feed's order"""
    n = seq.n_scans if n_scans is None else n_scans
    for k in range(n):
        a, b = seq.imu_range_for_scan(k)
        for i in range(a, b):
            yield ("imu", IMU(seq.imu[i, 1:4].copy(), seq.imu[i, 4:7].copy(), float(seq.imu[i, 0])))
        yield ("scan", seq.scan(k), None, float(seq.t_base + (k + 1) * seq.scan_dt))


def run_resident(seq, n_scans=None, *, use_imu_prediction=False, with_ekf=True, device_id=0, **icp_over):
    """Upload a synth.Sequence and run the loop on device.  Returns the SeqRunner results dict + 'seconds'."""
    n = seq.n_scans if n_scans is None else n_scans
    n_imu = seq.imu_range_for_scan(n - 1)[1] if with_ekf else 0
    r = core.SeqRunner(n, seq.H * seq.W, n_imu, max_range=seq.max_range, min_range=seq.min_range,
                       use_imu_prediction=use_imu_prediction, with_ekf=with_ekf, device_id=device_id,
                       scan_cols=seq.W, **icp_over)
    for k in range(n):
        r.upload_scan(k, seq.scan(k))
    ends = [seq.imu_range_for_scan(k)[1] if with_ekf else 0 for k in range(n)]
    r.upload_imu(seq.imu[:n_imu] if with_ekf else np.zeros((0, 7)), ends)
    t0 = time.perf_counter()
    r.run()
    out = r.results()
    out["seconds"] = time.perf_counter() - t0
    out["runner"] = r
    return out
