"""A recording of any length at the full batch width: the sweep ring (ptl_seq_cfg.resident_scans) fed while the kernel runs.

    python tools/stream_feed.py [S] [T] [RING] [LAUNCH]        default 240 sequences x 120 sweeps, ring of 32, launches of 16

The reference walks a file scan by scan (src/ptudes/data.py:31-77, cli/ekf_bench.py:493-563).  Here every sequence's sweeps are rendered
ONCE into host memory as raw range images (what a recording holds: 512 KB per 128 x 1024 sweep), then registered twice:
  resident   every sweep uploaded before the run (what bench.py times: inputs in HBM when the timed region starts),
  streamed   a ring of RING sweep slots per sequence; launch n + 1's sweeps are uploaded between ptl_batch_enqueue(n) and ptl_batch_wait(n),
             i.e. while launch n runs - the PCIe-inclusive rate of a feed that delivers as fast as the host can copy.
The two must agree bit for bit.  Prints both rates and the host-to-device rate the streamed run sustained.  (bench.py's `value` never includes
host copies; this is the figure DESIGN.md quotes beside it.)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa: E402,F401
from ptudes_lab_amd import core, synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 240
T = int(sys.argv[2]) if len(sys.argv) > 2 else 120
RING = int(sys.argv[3]) if len(sys.argv) > 3 else 32
LAUNCH = int(sys.argv[4]) if len(sys.argv) > 4 else 16
H, W = 128, 1024
assert RING >= 2 * LAUNCH, "the ring holds the launch that runs and the one that is uploaded beside it"

l2s = np.diag([1.0, -1.0, 1.0, 1.0])  # (the synthetic sensor counts its columns the other way round: bench.py RangeFeed)
lut = core.Lut(H, W, np.linspace(45.0, -45.0, H), np.zeros(H), 0.0, l2s, None)
t0 = time.perf_counter()
seqs = [synth.make_sequence(seed=1000 + s, n_scans=T) for s in range(S)]
img = np.empty((S, T, H * W), dtype=np.uint32)
for s, sq in enumerate(seqs):
    for k in range(T):
        x = np.asarray(sq.scan(k), dtype=np.float64)
        img[s, k] = np.round(np.sqrt((x * x).sum(axis=1)) * 1000.0).astype(np.uint32)
n_imu = seqs[0].imu_range_for_scan(T - 1)[1]
print("rendered %d x %d range images (%.1f GB of host memory) in %.0f s" % (S, T, img.nbytes / 1e9, time.perf_counter() - t0), flush=True)
PIN = os.environ.get("PTL_STREAM_PIN", "1") != "0"
if PIN:  # the recording's buffer page-locked once: the uploads go by DMA straight from it (PTL_STREAM_PIN=0: pageable, through the runtime's staging buffer)
    t0 = time.perf_counter()
    core.host_pin(img)
    print("host buffer page-locked in %.1f s" % (time.perf_counter() - t0), flush=True)
kw = dict(use_imu_prediction=True, with_ekf=True, scans_per_launch=LAUNCH, range_input=True, scan_cols=W, map_block_capacity=65536)


def imu(b):
    for s, sq in enumerate(seqs):
        b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(T)])


# ---- resident: everything uploaded first, then one timed run
full = core.BatchRunner(S, T, H * W, n_imu, **kw)
full.set_lut(lut)
imu(full)
t0 = time.perf_counter()
for s in range(S):
    for k in range(T):
        full.upload_range(s, k, img[s, k])
t_up = time.perf_counter() - t0
full.run(LAUNCH)  # (warm-up: the first launch builds the maps)
core.device_sync(0)
t0 = time.perf_counter()
full.enqueue(T - LAUNCH)
full.wait()
dt_full = time.perf_counter() - t0
want = [full.results(s) for s in range(S)]
full.close()
print("resident: upload of all sweeps %.2f s (%.1f GB/s), then %d scans in %.3f s = %.0f scans/s" %
      (t_up, img.nbytes / t_up / 1e9, (T - LAUNCH) * S, dt_full, (T - LAUNCH) * S / dt_full), flush=True)

# ---- streamed: a ring of RING slots, the next launch's sweeps uploaded while this one runs
ring = core.BatchRunner(S, T, H * W, n_imu, resident_scans=RING, **kw)
ring.set_lut(lut)
imu(ring)


def feed(k0, k1):
    n = 0
    for k in range(k0, min(k1, T)):  # (sweep-major: every sequence's sweep k, then k + 1 - the order a multi-recording feed delivers)
        for s in range(S):
            ring.upload_range(s, k, img[s, k])
            n += 1
    return n


feed(0, RING)
ring.run(LAUNCH)
core.device_sync(0)
done, fed, t_feed = LAUNCH, RING, 0.0
t0 = time.perf_counter()
while done < T:
    m = min(LAUNCH, T - done)
    ring.enqueue(m)
    ta = time.perf_counter()
    feed(fed, done + RING)  # the slots of the scans known to be done take the next sweeps while the launch runs
    t_feed += time.perf_counter() - ta
    fed = max(fed, min(done + RING, T))
    ring.wait()
    done += m
dt_ring = time.perf_counter() - t0
bad = [s for s in range(S) if not (np.array_equal(ring.results(s)["res_poses"], want[s]["res_poses"]) and ring.results(s)["stats"] == want[s]["stats"])]
ring.close()
up_bytes = (T - RING) * S * H * W * 4
print("streamed (ring of %d, launches of %d): %d scans in %.3f s = %.0f scans/s; %.1f GB uploaded beside the launches in %.2f s of host time "
      "(%.1f GB/s); sequences that differ from the resident run: %s" %
      (RING, LAUNCH, (T - LAUNCH) * S, dt_ring, (T - LAUNCH) * S / dt_ring, up_bytes / 1e9, t_feed, up_bytes / max(t_feed, 1e-9) / 1e9, bad or "none"), flush=True)
print("resident memory for sweeps: %.1f GB (all %d) against %.1f GB (ring)" % (S * T * H * W * 4 / 1e9, T, S * RING * H * W * 4 / 1e9))
if PIN:
    core.host_unpin(img)
