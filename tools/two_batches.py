"""Two (or more) batched runners side by side on one GPU, each on its own streams with a share of the workgroups: while one
half's Gauss-Newton kernel runs, the other half's stage kernels find free CUs.   python tools/two_batches.py NB S_each WGS [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import core, synth

NB, S, WGS = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 120
W = 20
seqs = [synth.make_sequence(seed=1000 + s, n_scans=n) for s in range(NB * S)]
n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
bs = []
for b in range(NB):
    r = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=WGS)
    for j in range(S):
        sq = seqs[b * S + j]
        for k in range(n):
            r.upload_scan(j, k, sq.scan(k))
        r.upload_imu(j, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
    bs.append(r)
for r in bs:
    r.run(W)
core.device_sync(0)
t0 = time.perf_counter()
for r in bs:
    r.enqueue(n - W)
for r in bs:
    r.wait()
core.device_sync(0)
dt = time.perf_counter() - t0
print(f"NB={NB} S={S} WGS={WGS}: {(n - W) * NB * S / dt:.1f} scans/s ({1e3 * dt / (n - W):.3f} ms per step of {NB * S})")
