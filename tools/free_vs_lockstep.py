"""Both batch drivers on the same sweeps: results must agree bit for bit; scans/s of each."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa
from ptudes_lab_amd import core, synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 220
W = 20
extra = {}
for a in sys.argv[3:]:
    k, v = a.split("=")
    extra[k] = int(v)
with_ekf = bool(extra.pop("ekf", 1))
seqs = [synth.make_sequence(seed=1000 + s, n_scans=n) for s in range(S)]
n_imu = seqs[0].imu_range_for_scan(n - 1)[1] if with_ekf else 0
out = {}
for free in (True, False):
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=with_ekf, with_ekf=with_ekf, free_running=free, **extra)
    for s, sq in enumerate(seqs):
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu] if with_ekf else np.zeros((0, 7)), [sq.imu_range_for_scan(k)[1] if with_ekf else 0 for k in range(n)])
    b.run(W)
    core.device_sync(0)
    t0 = time.perf_counter()
    b.enqueue(n - W)
    b.wait()
    dt = time.perf_counter() - t0
    res = [b.results(s) for s in range(S)]
    out[free] = res
    print("free-running" if free else "lockstep    ", "%.0f scans/s" % ((n - W) * S / dt), "(%.3f ms per step of %d scans)" % (1e3 * dt / (n - W), S), flush=True)
    if free:
        clk = np.array([b.seq_clocks(s) for s in range(S)])
        print("  per scan, us (mean over sequences | min | max):  K0-K4 %s  wait %s  GN %s  wait %s  map %s  filter %s" % tuple(
            "%.0f|%.0f|%.0f" % (clk[:, i].mean(), clk[:, i].min(), clk[:, i].max()) for i in range(6)))
        print("  per-sequence total us/scan:", np.round(clk[:, :5].sum(1)).astype(int).tolist())
        import ctypes as C
        from ptudes_lab_amd import _lib as L
        icp = C.c_void_p(); L.check(L.lib().ptl_batch_icp(b._h, 0, C.byref(icp)))
        ds = (C.c_double * 32)(); L.check(L.lib().ptl_icp_debug_sums(icp, ds))
        d = np.array(list(ds)) / n / 100.0
        if d[:20].sum() > 0:  # built with make STAGES=1
            names = ["prologue", "w", "deskew+vds1", "w", "vds2", "w", "compact_fd", "w", "compact_src", "-", "insert_a", "w", "insert_b", "w", "insert_c", "w", "prune"]
            print("  stages of sequence 0, us/scan:", "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, d[:17])))
            print("  inside K1 (thread 0 of workgroup 0, its own waits): release %.0f | load + deskew + store %.0f | claim %.0f | bid %.0f | slot1 %.0f | count %.0f" % tuple(d[20:26]))
    b.close()
worst = 0.0
for s in range(S):
    a, c = out[True][s], out[False][s]
    worst = max(worst, float(np.abs(a["kiss_poses"] - c["kiss_poses"]).max()))
    if with_ekf:
        worst = max(worst, float(np.abs(a["res_poses"] - c["res_poses"]).max()))
    assert [st["iterations"] for st in a["stats"]] == [st["iterations"] for st in c["stats"]], s
    assert [st["map_points"] for st in a["stats"]] == [st["map_points"] for st in c["stats"]], s
print("max |free - lockstep| over all poses:", worst)
