"""Per-stage device clocks of the free-running kernel (a `make STAGES=1` build, loaded with PTL_LIB_PATH):

    PTL_LIB_PATH=.../libptudes_mi_stages.so python tools/stage_clocks.py S TEAM_WGS [N_SCANS]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa: E402,F401
from ptudes_lab_amd import _lib as L, core, synth  # noqa: E402

S, G = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
# PTL_TOOL_WORKLOAD=config5: BASELINE config 5 (64 x 2048 sweeps, 0.1 m voxels, 100 m range, two block classes) instead of the default workload
if os.environ.get("PTL_TOOL_WORKLOAD") == "config5":
    SEQ_KW = dict(H=64, W=2048, max_range=100.0)
    RUN_KW = dict(max_range=100.0, voxel_size=0.1, scan_cols=2048, map_block_capacity=600000, map_small_blocks=2200000, map_table_capacity=1 << 25)
    RUN_KW.update({k[4:].lower(): int(v) for k, v in os.environ.items() if k.startswith("PTL_KW_")})  # e.g. PTL_KW_MAP_TABLE_CAPACITY=...
else:
    SEQ_KW, RUN_KW = {}, {}
W = 10
seqs = [synth.make_sequence(seed=1000 + s, n_scans=n, **SEQ_KW) for s in range(S)]
n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, team_workgroups=G, **RUN_KW)
for s, sq in enumerate(seqs):
    for k in range(n):
        b.upload_scan(s, k, sq.scan(k))
    b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
b.run(W)
core.device_sync(0)
t0 = time.perf_counter()
b.enqueue(n - W)
b.wait()
dt = time.perf_counter() - t0
print("S %d, teams of %d x %d: %.0f scans/s" % ((S,) + b.team_geometry() + ((n - W) * S / dt,)))
clk = np.array([b.seq_clocks(s) for s in range(S)])
print("per scan, us (mean | min | max over sequences):  K0-K4 %s  wait %s  GN %s  wait %s  map %s  filter %s" % tuple(
    "%.0f|%.0f|%.0f" % (clk[:, i].mean(), clk[:, i].min(), clk[:, i].max()) for i in range(6)))
icp = C.c_void_p()
L.check(L.lib().ptl_batch_icp(b._h, 0, C.byref(icp)))
ds = (C.c_double * 32)()
L.check(L.lib().ptl_icp_debug_sums(icp, ds))
d = np.array(list(ds)) / n / 100.0
if d[:20].sum() > 0:
    names = ["prologue", "w", "deskew+vds1", "w", "compact_fd (look-back)", "w", "vds2 on frame_down", "w", "compact_src (look-back)", "-", "insert_a", "w", "insert_b", "w", "insert_c", "w", "prune"]
    print("stages of sequence 0 (workgroup 0), us/scan:", "  ".join("%s %.0f" % (nm, v) for nm, v in zip(names, d[:17])))
    print("inside K1 (thread 0 of workgroup 0, its own waits): - %.0f | load + deskew + store %.0f | claim %.0f | bid %.0f | slot1 %.0f | count %.0f" % tuple(d[20:26]))
else:
    print("(no stage clocks: not a STAGES=1 build)")
