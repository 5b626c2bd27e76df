"""several independent sequences on one GPU, one persistent-GN stream each (ptl_seq_enqueue / ptl_seq_wait):
throughput against workgroups per sequence.   python tools/multi_stream.py S G [T]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa
from ptudes_lab_amd import core, synth

S, G = int(sys.argv[1]), int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
N, W = 120, 20
seqs = [synth.make_sequence(seed=1000 + s, n_scans=N) for s in range(S)]
rs = []
for sq in seqs:
    n_imu = sq.imu_range_for_scan(N - 1)[1]
    r = core.SeqRunner(N, sq.H * sq.W, n_imu, max_range=70.0, min_range=1.0, use_imu_prediction=True, with_ekf=True,
                       scan_cols=sq.W, gn_workgroups=G, gn_threads=T)
    for k in range(N):
        r.upload_scan(k, sq.scan(k))
    r.upload_imu(sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(N)])
    rs.append(r)
for r in rs:
    r.enqueue(W)
for r in rs:
    r.wait()
core.device_sync()
t0 = time.perf_counter()
for r in rs:
    r.enqueue(N - W)
for r in rs:
    r.wait()
dt = time.perf_counter() - t0
print(f"S={S} G={G} T={T}: {S * (N - W) / dt:.1f} scans/s")
