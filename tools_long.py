"""Where the host time of feeding a SeqRunner goes (pose interpolation / rendering / upload), per scan."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import ptudes_lab_amd
from ptudes_lab_amd import synth, core
n = int(sys.argv[1]); H = int(sys.argv[2]); W = int(sys.argv[3])
sq = synth.make_sequence(seed=77, n_scans=n, H=H, W=W)
r = core.SeqRunner(n, sq.H * sq.W, sq.imu_range_for_scan(n - 1)[1], max_range=70.0, min_range=1.0, use_imu_prediction=True, with_ekf=True, scan_cols=sq.W)
t = time.time()
for k in range(n):
    tt = (k + np.arange(sq.W) / sq.W) * sq.scan_dt
    T = sq.pose_at(tt)
print('pose_at ms', (time.time() - t) / n * 1e3, flush=True)
t = time.time(); xs = [sq.scan(k) for k in range(n)]; print('scan ms', (time.time() - t) / n * 1e3, flush=True)
t = time.time()
for k in range(n):
    r.upload_scan(k, xs[k])
print('upload ms', (time.time() - t) / n * 1e3, flush=True)
