"""Does the restated KISS-ICP pipeline track under the reference's DEFAULT constant-velocity guess (kiss.py:102-105)?
CPU oracle only (no GPU): controlled drives (synth.make_path_sequence) and the SURVEY 8(d) random walk, error against
ground truth per sweep.   N=200 python tools/track_experiments.py [case ...]
Since the poses are re-orthonormalised the way upstream's Sophus::SE3d are it does: `vehicle_jitter` drifts
0.25 % of the 190 m it drives, the random walks stay at their bootstrap offset; the level noise-free drives (`cv_*`) still
show ring locking (a lag, not a loss)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa: E402,F401
from ptudes_lab_amd import synth  # noqa: E402
from oracle import cpu as orc  # noqa: E402


def rel_gt(seq):
    gt = seq.gt_poses(0.5)
    g0i = np.linalg.inv(gt[0])
    return np.array([g0i @ g for g in gt])


def run(seq, n, guess="cv", log_every=20, **icp_over):
    icp = orc.ICP(max_range=seq.max_range, min_range=seq.min_range, **icp_over)
    t01 = seq.column_times()
    gt = rel_gt(seq)
    err, rows = [], []
    for k in range(n):
        g = gt[k] if guess == "gt" else None
        pose = icp.register_frame(seq.scan(k).astype(np.float64), t01, g)
        d = np.linalg.inv(gt[k]) @ pose
        et = float(np.linalg.norm(d[:3, 3]))
        er = float(np.arccos(np.clip((np.trace(d[:3, :3]) - 1) / 2, -1, 1)))
        err.append(et)
        st = icp.stats[-1]
        rows.append((k, et, er, st["sigma"], st["iterations"], st["n_src"], st["n_corr_last"]))
        if log_every and (k % log_every == 0 or k == n - 1):
            print(f"   k={k:4d} |dt|={et:8.4f} m  rot={er:8.5f} rad  sigma={st['sigma']:7.3f} it={st['iterations']:3d} "
                  f"n_src={st['n_src']:5d} ncorr={st['n_corr_last']:5d}", flush=True)
    return np.array(err), rows


CASES = {
    # noise-free, rotation-free constant translation from t = 0
    "cv_0.05": dict(kind="path", step_m=0.05), "cv_0.5": dict(kind="path", step_m=0.5), "cv_1.0": dict(kind="path", step_m=1.0),
    # the same with two static sweeps first and a 5-sweep ramp (nothing un-deskewed moves into the first map)
    "static2_0.5": dict(kind="path", step_m=0.5, static_sweeps=2, ramp_sweeps=5),
    "static2_1.0": dict(kind="path", step_m=1.0, static_sweeps=2, ramp_sweeps=5),
    # vehicle speed with 1 cm range noise + 2 % dropout
    "vehicle_noise": dict(kind="path", step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02),
    # ... and the same vehicle rocking on its suspension (1 deg roll / pitch, 3 cm heave): the rings move over the ground
    "vehicle_wobble": dict(kind="path", step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02,
                           wobble_deg=1.0, heave_m=0.03),
    "vehicle_wobble_flat": dict(kind="path", step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02,
                                wobble_deg=1.0, heave_m=0.03, rough_amp=0.0),
    # the vehicle drive with the ray pattern jittered by 0.3 deg per sweep (no sampling lattice): the cleanest track
    "vehicle_jitter": dict(kind="path", step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02, wobble_deg=1.0,
                           heave_m=0.03, rough_amp=0.0, n_boxes=400, n_cyls=200, room_size=(280.0, 60.0, 200.0), ray_jitter_deg=0.3),
    "vehicle_flat": dict(kind="path", step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02, rough_amp=0.0),
    "walk1000": dict(kind="walk", seed=1000), "walk1003": dict(kind="walk", seed=1003),
}

if __name__ == "__main__":
    orc.set_threads(min(8, synth.usable_cores()))
    names = sys.argv[1:] or list(CASES)
    n = int(os.environ.get("N", "200"))
    for nm in names:
        c = dict(CASES[nm])
        kind = c.pop("kind")
        guess = c.pop("guess", "cv")
        t0 = time.time()
        seq = synth.make_path_sequence(n_scans=n, **c) if kind == "path" else synth.make_sequence(n_scans=n, **c)
        print(f"== {nm}: {kind} {c} guess={guess}", flush=True)
        err, _ = run(seq, n, guess)
        print(f"== {nm}: rmse {np.sqrt(np.mean(err ** 2)):.4f} m, max {err.max():.4f} m, final {err[-1]:.4f} m  ({time.time() - t0:.0f} s)", flush=True)
