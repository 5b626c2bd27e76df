"""How much does the lockstep of the batched runner cost?  GN iterations per (sequence, scan) from one run:
sum over scans of the max over sequences (what a lockstep step waits for) against the mean and the per-sequence totals."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa
from ptudes_lab_amd import core, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 220
seqs = [synth.make_sequence(seed=1000 + s, n_scans=n) for s in range(S)]
n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True)
for s, sq in enumerate(seqs):
    for k in range(n):
        b.upload_scan(s, k, sq.scan(k))
    b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
b.run()
it = np.array([[st["iterations"] for st in b.results(s)["stats"]] for s in range(S)], dtype=float)[:, 20:]
ns = np.array([[st["n_src"] for st in b.results(s)["stats"]] for s in range(S)], dtype=float)[:, 20:]
w = it * ns  # point-iterations
print("iterations: mean %.2f  mean of per-step max %.2f  (ratio %.3f)" % (it.mean(), it.max(0).mean(), it.max(0).mean() / it.mean()))
print("per-sequence totals: min %.0f mean %.0f max %.0f (max/mean %.3f)" % (it.sum(1).min(), it.sum(1).mean(), it.sum(1).max(), it.sum(1).max() / it.sum(1).mean()))
print("point-iterations: per-step max / mean %.3f; per-sequence total max / mean %.3f" % (w.max(0).mean() / w.mean(), w.sum(1).max() / w.sum(1).mean()))
if S >= 16:
    pair = it[:8] + it[8:16]
    print("XCD pair sums: per-step max / mean %.3f; total max/mean %.3f" % (pair.max(0).mean() / pair.mean(), pair.sum(1).max() / pair.sum(1).mean()))
    pm = np.maximum(it[:8], it[8:16])
    print("XCD pair max : per-step max / mean of it %.3f" % (pm.max(0).mean() / it.mean()))
np.save("gpurun_out/lockstep_it.npy", it)
