"""`ptudes stat` (reference src/ptudes/cli/stat.py): range and IMU statistics of a data source.

Same options and report as the reference command.  Packet files need ouster-sdk (absent here, SURVEY.md 8(f) rank
3); `--synthetic SEED` runs the command on the synthetic 128x1024 sequence, whose sweeps are turned into range
images in millimetres (|xyz| rounded), the quantity `StreamStatsTracker.trackScan` reduces on the device."""
from typing import Optional

import click
import numpy as np

from ..ins.data import StreamStatsTracker


@click.command(name="stat")
@click.argument("file", required=False, type=click.Path())
@click.option("-m", "--meta", required=False, type=click.Path(exists=True, dir_okay=False, readable=True),
              help="sensor metadata .json of the PCAP / BAG (needed when it is not found next to FILE)")
@click.option("--start-scan", type=int, default=0, help="first scan to use (0-based)")
@click.option("--end-scan", type=int, help="last scan to use (inclusive)")
@click.option("--beams", type=int, default=0,
              help="keep only NUM evenly spaced beams (rows) of every scan; 0 = all")
@click.option("-t", "--duration", type=float, default=3.0,
              help="seconds of data (IMU / scans) to take into the statistics; 0 = the whole source (default 3.0)")
@click.option("--kiss-run", is_flag=True, help="also register every scan with KissICP (timing runs)")
@click.option("--kiss-min-range", type=float, default=1, help="KissICP: shortest range kept, metres (default 1)")
@click.option("--kiss-max-range", type=float, default=70, help="KissICP: longest range kept, metres (default 70)")
@click.option("--synthetic", type=int, default=None,
              help="Run on the synthetic 128x1024 sequence with this seed instead of FILE (no ouster-sdk needed)")
def ptudes_stat(file: Optional[str], meta: Optional[str], start_scan: int = 0, end_scan: Optional[int] = None,
                beams: int = 0, duration: float = 3, kiss_run: bool = False, kiss_min_range: float = 1.0,
                kiss_max_range: float = 70, synthetic: Optional[int] = None) -> None:
    """Range and IMU statistics of a data source (mean / std of the scans' ranges, of accelerometer and gyroscope
    samples) over --duration seconds, plus the gravity direction the accelerometer mean suggests."""
    if synthetic is None:
        raise click.ClickException("reading .pcap/.bag needs ouster-sdk, which is not installed; "
                                   "use --synthetic SEED to run the same command on a synthetic sequence")
    from .ekf_bench import _synthetic_source
    n_scans = (end_scan + 1) if end_scan is not None else max(2, int(np.ceil((duration or 10.0) * 10)) + 2)
    seq, info, events = _synthetic_source(synthetic, n_scans)
    stats = StreamStatsTracker(use_beams_num=beams, metadata=None)
    kiss_icp = None
    if kiss_run:
        from ..kiss import KissICPWrapper
        kiss_icp = KissICPWrapper(info, _min_range=kiss_min_range, _max_range=kiss_max_range)
    scan_idx = 0
    for ev in events:
        if ev[0] == "imu":
            if scan_idx >= start_scan:  # IMUs before start_scan are dropped (reference data.py:76)
                stats.trackImu(ev[1])
        else:
            if scan_idx >= start_scan:
                xyz, ts = ev[1], ev[3]
                rng = np.round(np.linalg.norm(np.asarray(xyz, dtype=np.float64).reshape(seq.H, seq.W, 3), axis=2) * 1000.0)
                stats.trackScan(rng.astype(np.uint32), int(round(ts * 1e9)))
                if kiss_icp is not None:
                    from types import SimpleNamespace
                    kiss_icp.register_frame(SimpleNamespace(xyz=xyz, ts=ts))
            scan_idx += 1
        if duration and stats.dt > duration:
            break
    print()
    print(stats)
    grav_est = stats.acc_mean / np.linalg.norm(stats.acc_mean)
    print("Gravity vector estimation: ", grav_est)
