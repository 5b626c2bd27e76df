"""per-call drop-in API rate: KissICPWrapper.register_frame + ESEKF.processImu / processPose driven from host arrays,
one call per event as the reference's loop does (PCIe upload and a host round trip per call included)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa
from ptudes_lab_amd.cli.ekf_bench import _synthetic_source
from ptudes_lab_amd.sequence import run_events
n = 120
seq, info, events = _synthetic_source(1000, n)
events = list(events)
run_events(iter(events[:60]), info, kiss_min_range=1.0, kiss_max_range=70.0, use_imu_prediction=True)  # library warm-up
for lazy in (False, True, False, True):
    t0 = time.perf_counter()
    out = run_events(iter(events), info, kiss_min_range=1.0, kiss_max_range=70.0, use_imu_prediction=True, lazy_map_stats=lazy)
    dt = time.perf_counter() - t0
    print("per-call API (%s): %d scans in %.2f s = %.1f scans/s (timings %s)" % ("registration returns with the pose, map update still running" if lazy else
          "registration returns after its map update", len(out["res_poses"]), dt, len(out["res_poses"]) / dt, out.get("timings")))
