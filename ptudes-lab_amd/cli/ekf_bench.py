"""`ekf-bench` commands: drop-in for the pose path of reference src/ptudes/cli/ekf_bench.py.

  sim     EKF with simulated IMU (reference :107-179) - the filter's known-answer harness
  nc      Newer College bag IMU + ground-truth pose corrections (reference :183-323)
  ouster  IMU + scans -> KissICP poses -> EKF smoothing, KITTI / NC-GT pose files out (reference :381-666)
  cmp     compare trajectories in Newer College format (reference :686-760)

Same option names, printed lines and output files.  Differences, all because ouster-sdk / rosbags are not
installable offline: `ouster` reads .pcap/.bag only when ouster-sdk is importable and otherwise (or with
--synthetic SEED) runs on a synthetic 128x1024 sequence; plotting options (-p) are not provided (the
OpenGL / matplotlib viewers are outside the path).  `nc` reads the bag with the package's own ROS1 reader (bag.py)
instead of rosbags.
"""
import os
from datetime import datetime
from itertools import count
from types import SimpleNamespace
from typing import List, Optional

import click
import numpy as np
import numpy.random as npr

from ..ins.data import GRAV, IMU, calc_ate, ekf_traj_ate
from ..ins.es_ekf import ESEKF
from ..utils import (filter_nc_gt_by_close_ts, filter_nc_gt_by_cmp, read_newer_college_gt,
                     save_poses_kitti_format, save_poses_nc_gt_format)

DOWN = np.array([0, 0, -1])
UP = np.array([0, 0, 1])


@click.group(name="ekf-bench")
def ptudes_ekf_bench() -> None:
    """ES EKF benchmarks and experiments (MI355X-native pose path)."""


def sim_imu(acc_mean=np.zeros(3), acc_std=1.5, acc_noise_std=0.4, acc_bias=np.array([0.9, -0.2, -0.4]),
            gyr_mean=np.zeros(3), gyr_std=1.0, gyr_noise_std=0.2, gyr_bias=np.array([0.01, 0.03, -0.012]),
            gravity=GRAV * DOWN, freq=100):
    """(ideal, noisy) IMU pairs; the commanded acc/gyr are redrawn every 10 samples.  Draws from the legacy
    global numpy RNG in the reference's order, so `np.random.seed(s)` reproduces its streams (reference :44-79)."""
    dt = 1 / freq

    def draw():
        a = npr.normal(0.0, acc_std, 3) + acc_mean - gravity
        g = npr.normal(0.0, gyr_std, 3) + gyr_mean
        return a, g

    acc, gyr = draw()
    for idx in count():
        if idx % 10 == 0:
            acc, gyr = draw()
        acc_noise = npr.normal(0, acc_noise_std, 3)
        gyr_noise = npr.normal(0.0, gyr_noise_std, 3)
        yield (IMU(acc, gyr, idx * dt), IMU(acc + acc_noise + acc_bias, gyr + gyr_noise + gyr_bias, idx * dt))


@click.command(name="sim")
@click.option("-t", "--duration", type=float, default=2.0, help="seconds of simulated IMU samples to run (default 2.0)")
@click.option("-f", "--freq", type=float, default=100.0, help="IMU rate, Hz")
@click.option("--corr-t", type=float, default=0.1, help="seconds between two pose corrections of the filter (default 0.1)")
@click.option("--acc-noise-std", type=float, default=0.4, help="standard deviation of the accelerometer noise")
@click.option("--gyr-noise-std", type=float, default=0.4, help="standard deviation of the gyroscope noise")
def ptudes_ekf_sim(duration: float, corr_t: float, freq: float, acc_noise_std: float, gyr_noise_std: float) -> None:
    """EKF with simulated IMU measurements; the noise-free filter is the ground truth for pose corrections."""
    print("Using sim IMUs with params:")
    print(f"  freq: {freq} Hz")
    print(f"  acc_noise_std: {acc_noise_std}")
    print(f"  gyr_noise_std: {gyr_noise_std}")
    print(f"  correction dt: {corr_t:.02} s")
    print("Running EKF ... \n")
    ekf_gt, ekf = ESEKF(_logging=True), ESEKF(_logging=True)
    start_ts = last_corr_t = ts = None
    for imu_ideal, imu_noisy in sim_imu(freq=freq, acc_noise_std=acc_noise_std, gyr_noise_std=gyr_noise_std):
        ts = imu_ideal.ts
        if start_ts is None:
            start_ts = last_corr_t = ts
        ekf_gt.processImu(imu_ideal)
        ekf.processImu(imu_noisy)
        if ts - last_corr_t > corr_t:
            ekf.processPose(ekf_gt.nav.pose_mat())
            last_corr_t = ts
        if ts - start_ts > duration:
            break
    print("Results:")
    print(f"processed duration: {ts - start_ts:0.04} s")
    print(f"updates num: {len(ekf._nav_update_idxs)}\n")
    print("NAV GT:\n", ekf_gt.nav)
    print("NAV:\n", ekf.nav)
    ate_rot, ate_trans = ekf_traj_ate(ekf_gt, ekf)
    print(f"ATE_rot:   {ate_rot:.04f} deg")
    print(f"ATE trans: {ate_trans:.04f} m")


@click.command(name="nc")
@click.argument("file", required=True, type=click.Path(exists=True))
@click.option("-m", "--meta", required=False, type=click.Path(exists=True, dir_okay=False, readable=True),
              help="sensor metadata .json of the BAG (unused by this command, kept for the option set)")
@click.option("-g", "--gt-file", required=True, type=click.Path(exists=True, dir_okay=False, readable=True),
              help="Newer College ground-truth CSV: its poses correct the filter and are what the result is compared with")
@click.option("-t", "--duration", type=float, default=0.0,
              help="seconds of data to process; 0 = to the end of the bag")
@click.option("--start-ts", type=float, default=0.0,
              help="seconds to skip from the first IMU sample (default 0.0)")
@click.option("-p", "--plot", required=False, type=str, help="Plotting option [graphs, point_viz] (not provided here)")
@click.option("--xy-plot", is_flag=True, help="(plots are not provided; accepted for the option set)")
@click.option("-i", "--imu-topic", required=False, default="/os_node/imu_packets", type=str,
              help="IMU topic: a sensor_msgs/Imu topic or an Ouster imu_packets topic")
def ptudes_ekf_nc(file: str, meta: Optional[str] = None, gt_file: Optional[str] = None, duration: float = 2.0,
                  start_ts: float = 0.0, plot: Optional[str] = None, xy_plot: bool = False,
                  imu_topic: Optional[str] = None) -> None:
    """EKF with Newer College Dataset IMUs topics.

    Ground truth (--gt-file) is used for pose correction.
    """
    from .. import bag  # IMUBagSource looked up at call time, as the reference imports it inside the command (:235)

    # the Ouster IMU's nav frame has gravity along -Z, the alphasense IMU's along +Z (reference :236-239)
    ouster_frame = imu_topic in ("/os_cloud_node/imu", "/os_node/imu_packets")
    init_grav = GRAV * (DOWN if ouster_frame else UP)
    print("init_grav = ", init_grav)
    print("Reading NC dataset:")
    print(f"  file: {file}")
    print(f"  topic: {imu_topic}")
    print(f"  gt file: {gt_file}")
    source = bag.IMUBagSource(file, imu_topic=imu_topic)
    if not gt_file:
        print("need gt now")
        return
    gt_rows = read_newer_college_gt(gt_file)
    gt_stamp = [row[0] for row in gt_rows]
    last_row = len(gt_rows) - 1
    print("Running EKF ... \n")
    ekf = ESEKF(init_grav=init_grav, _logging=bool(plot))
    corrections, estimates = [], []
    row = 0          # GT row that corrects the filter next
    origin = None    # inverse of the GT pose everything is expressed relative to
    ts, t_first = 0, -1
    for imu in source:
        ts = imu.ts
        if t_first < 0:
            t_first = ts
        elapsed = ts - t_first - start_ts
        if ts - t_first < start_ts:  # --start-ts: not there yet
            continue
        if origin is None:
            # the first GT row newer than the first sample that is processed (reference :283-287; a stream that starts
            # past the last row fails here, as it does there)
            while row < len(gt_rows) and ts >= gt_stamp[row]:
                row += 1
            origin = np.linalg.inv(gt_rows[row][1])
        ekf.processImu(imu)
        if ts >= gt_stamp[row]:
            target = origin @ gt_rows[row][1]
            ekf.processPose(target)
            corrections.append(target)
            estimates.append(ekf.nav.pose_mat())
            # past the last GT row the index stays: every further IMU sample is followed by a correction with that row
            row = min(row + 1, last_row)
        if duration > 0 and elapsed > duration:
            break
    print(f"scanned duration: {ts - t_first - start_ts:0.04} s")
    print(f"updates num: {len(estimates)}\n")
    if estimates:
        ate_rot, ate_trans = calc_ate(estimates, corrections)
        print(f"ATE_rot:   {ate_rot:.04f} deg")
        print(f"ATE trans: {ate_trans:.04f} m")
    if plot:
        print(f"WARNING: plot param '{plot}' doesn't supported")


def _synthetic_source(seed: int, n_scans: int):
    from .. import synth
    from ..sequence import synthetic_events
    seq = synth.make_sequence(seed=seed, n_scans=n_scans)
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=seq.W, pixels_per_column=seq.H),
                           prod_line="SYNTH-OS-0-128", mode=f"{seq.W}x10")
    return seq, meta, synthetic_events(seq)


@click.command(name="ouster")
@click.argument("file", required=False, type=click.Path())
@click.option("-m", "--meta", required=False, type=click.Path(exists=True, dir_okay=False, readable=True),
              help="sensor metadata .json of the PCAP / BAG (needed when it is not found next to FILE)")
@click.option("--start-scan", type=int, default=0, help="first scan to use (0-based)")
@click.option("--end-scan", type=int, help="last scan to use (inclusive)")
@click.option("-p", "--plot", required=False, type=str,
              help="Plotting option [graphs, point_viz]: `graphs` prints the ground-truth comparison; the figures and the "
                   "viewer themselves are not part of this build")
@click.option("--use-imu-prediction", is_flag=True,
              help="hand the filter's predicted pose to KissICP as initial guess (loosely coupled lidar-inertial odometry)")
@click.option("--use-gt-guess", is_flag=True,
              help="hand the ground-truth pose (--gt-file, interpolated at the scan's timestamp) to KissICP as initial "
                   "guess (sanity testing only)")
@click.option("-g", "--gt-file", required=False, type=click.Path(exists=True, dir_okay=False, readable=True),
              help="Newer College ground-truth CSV to compare the result with")
@click.option("--kiss-min-range", type=float, default=1, help="KissICP: shortest range kept, metres (default 1)")
@click.option("--kiss-max-range", type=float, default=70, help="KissICP: longest range kept, metres (default 70)")
@click.option("--beams", type=int, default=0, help="keep only NUM evenly spaced beams (rows) of every scan; 0 = all")
@click.option("--save-kitti-poses", required=False, type=click.Path(exists=False, dir_okay=False),
              help="write the resulting poses to this file, KITTI format")
@click.option("--save-nc-gt-poses", required=False, type=click.Path(exists=False, dir_okay=False),
              help="write the resulting poses to this file, Newer College ground-truth format")
@click.option("--synthetic", type=int, default=None,
              help="Run on the synthetic 128x1024 sequence with this seed instead of FILE (no ouster-sdk needed)")
def ptudes_ekf_ouster(file: Optional[str], meta: Optional[str], start_scan: int, end_scan: Optional[int],
                      plot: Optional[str], use_imu_prediction: bool, use_gt_guess: bool, gt_file: Optional[str], beams: int,
                      save_kitti_poses: Optional[str], save_nc_gt_poses: Optional[str], kiss_min_range: float,
                      kiss_max_range: float, synthetic: Optional[int]) -> None:
    """EKF with Ouster IMUs and scan KissICP poses updates (smoothing of the KissICP trajectory)."""
    from ..ins.data import StreamStatsTracker
    from ..sequence import run_events
    from ..utils import TrajectoryEvaluator, active_beam_rows
    if not gt_file and use_gt_guess:  # reference :416-418
        print("ERROR: --use-gt-guess requires the GT poses (--gt-file)")
        raise SystemExit(1)
    if synthetic is None:
        try:
            import ouster.client  # noqa: F401
            from ouster.sdk.util import resolve_metadata
        except Exception:
            raise click.ClickException("reading .pcap/.bag needs ouster-sdk, which is not installed; "
                                       "use --synthetic SEED to run the same path on a synthetic sequence")
        # the reference's feed (ekf_bench.py:426-448): packets -> OusterLidarData -> (scan idx, LidarScan | IMU)
        from ..data import OusterLidarData
        from ..utils import read_metadata_json, read_packet_source
        meta = resolve_metadata(file, meta)
        if not meta:
            raise click.ClickException("File not found, please specify a metadata file with `-m`")
        info = read_metadata_json(meta)
        data_source = OusterLidarData(read_packet_source(file, meta=info))
        seq = None
        events = ((("imu", d) if hasattr(d, "lacc") else ("lidar_scan", d))
                  for _, d in data_source.withScanIdx(start_scan=start_scan, end_scan=end_scan))
        start_scan_feed = 0  # withScanIdx already applied the range
    else:
        n_scans = (end_scan + 1) if end_scan is not None else 100
        seq, info, events = _synthetic_source(synthetic, n_scans)
        start_scan_feed = start_scan
    if synthetic is not None:
        file = f"synthetic:{synthetic}"
    display_header = f"data path: {file}\n"
    display_header += f"metadata path: {meta}\n\n"
    display_header += f"scans range: {start_scan} - {end_scan}\n"
    display_header += f"kiss min/max: {kiss_min_range} - {kiss_max_range}\n"
    display_header += f"use-imu-prediction: {use_imu_prediction}, use-gt-guess: {use_gt_guess}\n"
    display_header += f"beams: {beams or info.format.pixels_per_column}\n"
    display_header += f"sensor: {info.prod_line}, {info.mode}\n"
    print(display_header)
    log_metrics = bool(plot)
    print(f"metrics logging: {log_metrics}")

    # --use-gt-guess (reference :486-489, :536-542): the GT pose at the scan's last column, relative to the first one used
    gts = read_newer_college_gt(gt_file) if gt_file else []
    guess_fn = None
    if use_gt_guess:
        gt_traj = TrajectoryEvaluator(gts, time_bounds=1.0)
        origin = []

        def guess_fn(ts):
            gt_guess = gt_traj.pose_at(ts)
            if not origin:
                origin.append(np.linalg.inv(gt_guess))
            return origin[0] @ gt_guess

    def feed():
        scan_idx = 0
        for ev in events:
            if ev[0] == "scan":
                if scan_idx >= start_scan_feed:
                    xyz = ev[1]
                    if beams:
                        img = xyz.reshape(seq.H, seq.W, 3)
                        drop = np.ones(seq.H, dtype=bool)
                        drop[active_beam_rows(seq.H, beams)] = False
                        img[drop] = 0
                    yield ev
                scan_idx += 1
            elif ev[0] == "lidar_scan":  # ouster LidarScan (the packet feed applied start_scan / end_scan)
                if beams:
                    from ..utils import reduce_active_beams
                    reduce_active_beams(ev[1], beams)  # reference ekf_bench.py:520-521
                yield ev
            elif scan_idx >= start_scan_feed:  # IMUs before start_scan are dropped (reference data.py:76)
                yield ev

    stats = StreamStatsTracker(use_beams_num=32, metadata=info)  # reference :457
    out = run_events(feed(), info, kiss_min_range=kiss_min_range, kiss_max_range=kiss_max_range,
                     use_imu_prediction=use_imu_prediction, guess_fn=guess_fn, logging=log_metrics, stats=stats)
    res_t, res_poses, kiss_poses = out["res_t"], out["res_poses"], out["kiss_poses"]
    header = display_header + f"(scans/updates num: {len(res_poses)})\n"
    header += "time: " + datetime.now().strftime("%Y%m%d_%H%M%S")
    if save_kitti_poses:
        save_poses_kitti_format(save_kitti_poses, res_poses, header=header)
        print(f"Kitti poses saved to: {save_kitti_poses}")
    if save_nc_gt_poses:
        save_poses_nc_gt_format(save_nc_gt_poses, t=res_t, poses=res_poses, header=header)
        print(f"NC GT poses saved to: {save_nc_gt_poses}")
    tm = out["timings"]
    if tm["n_imu"] and tm["n_corr"]:  # reference :590-595
        print("\nTimings:")
        print(f"  ESEKF imu process:      {tm['imu']:.05f} s per step")
        print(f"  ESEKF update:           {tm['corr']:.05f} s per update")
        print(f"  KissICP register frame: {tm['kiss']:.05f} s per frame")
        print(f"  Stats tracking:         {tm['track']:.05f} s per frame")
    if plot == "graphs":  # reference :599-660: the comparison is printed, the figures are not drawn here
        if gts and res_poses:
            gts, res_t_matched = filter_nc_gt_by_close_ts(gts, res_t)
            idx, kiss_m, res_m = 0, [], []
            for t_m in res_t_matched:
                while res_t[idx] != t_m:
                    idx += 1
                kiss_m.append(kiss_poses[idx])
                res_m.append(res_poses[idx])
                idx += 1
            if gts:
                pose0 = res_m[0] @ np.linalg.inv(gts[0][1])
                gt2 = [pose0 @ g[1] for g in gts]
                for label, poses in ((f"with ES EKF smoothing {len(gt2)} poses", res_m),
                                     (f"no-EKF, only KissICP {len(gt2)} poses", kiss_m)):
                    ate_rot, ate_trans = calc_ate(poses, gt2)
                    print(f"\nGround truth comparison ({label}):")
                    print(f"ATE_rot:   {ate_rot:.04f} deg")
                    print(f"ATE trans: {ate_trans:.04f} m")
        print("NOTE: the graphs themselves (matplotlib) are not part of this build")
    elif plot == "point_viz":
        print("NOTE: the point viewer (ouster-sdk PointViz) is not part of this build")
    elif plot:
        print(f"WARNING: plot param '{plot}' doesn't supported")


@click.command(name="cmp")
@click.argument("gt_file", required=True, type=click.Path(exists=True))
@click.argument("gt_file_cmp", required=False, type=click.Path(exists=True), nargs=-1)
def ptudes_ekf_cmp(gt_file: str, gt_file_cmp: List[str]) -> None:
    """Compare trajectories in Newer College Dataset Formats"""
    gts_all = read_newer_college_gt(gt_file)
    for cmp_file in gt_file_cmp:
        gts, gts_cmp = filter_nc_gt_by_cmp(gts_all, read_newer_college_gt(cmp_file))
        ate_rot, ate_trans = calc_ate([p for _, p in gts], [p for _, p in gts_cmp])
        name = os.path.splitext(os.path.basename(cmp_file))[0]
        print(f"\nTraj poses comparisons GT v. {name} ({len(gts)} poses):")
        print(f"ATE_rot:   {ate_rot:.04f} deg")
        print(f"ATE trans: {ate_trans:.04f} m")


ptudes_ekf_bench.add_command(ptudes_ekf_sim)
ptudes_ekf_bench.add_command(ptudes_ekf_nc)
ptudes_ekf_bench.add_command(ptudes_ekf_ouster)
ptudes_ekf_bench.add_command(ptudes_ekf_cmp)
