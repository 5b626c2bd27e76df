"""Why the restated KISS-ICP v0.2.10 pipeline does not hold a track under the reference's DEFAULT constant-velocity guess
(reference kiss.py:102-105) on the synthetic worlds - generator artefact or restatement bug?  (VERDICT r1, weak #1.)

The experiments below take the sequence loop out of the picture and look at single registrations (Registration.cpp
RegisterFrame as restated in oracle/oracle_icp.c; the HIP twin runs the same cases through ptl_icp_map_add /
ptl_icp_align and must give the same numbers).  What they pin:

 1. Ring / twin locking.  A level sensor translating over a flat floor and under a flat ceiling sees rings that are
    invariant under that translation: every floor / ceiling return of the new sweep has a "twin" return of the same ray in
    the map, displaced by nothing at all, and nearest-neighbour point-to-point residuals pull the estimate back to zero
    motion.  Started AT THE TRUTH, the registration slides back to well under 70 % of a 0.5 m step; from a zero guess it
    recovers under a third; a 5 cm step (walking pace at 10 Hz) is recovered to under a half.
 2. The lock is the rings', not the solver's: the same 0.5 m step with 2 deg of pitch and 10 cm of heave (the rings sweep
    over the floor) is recovered from a ZERO guess to within 3 cm.
 3. Sampling-lattice forces remain even then.  Against a map built from GROUND-TRUTH poses and perfectly deskewed sweeps,
    from the TRUE guess, the fixed point of the nearest-neighbour iteration sits centimetres off the truth (median > 1.5 cm,
    worst > 5 cm over 14 sweeps at the initial threshold): one source point per 1.05 m voxel against 0.35 m map samples
    on large regular planes is that accurate and no better.  A constant-velocity guess extrapolates this noise (x 2), the
    deskew twist and the map inherit it: the free-running loop amplifies it until the track is lost after some tens of
    sweeps (test_tracking_free_running below), while a guess that does not depend on the previous registrations (IMU
    filter, ground truth) keeps it bounded.
So: a property of (nearest-neighbour point-to-point ICP x idealised planar world), shared bit for bit by oracle and HIP path -
not a divergence between them, and nothing an upstream KISS-ICP would be spared on the same input.
"""
import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import synth
from oracle import cpu as orc
from tests.helpers import tracking as tk

SIGMA0 = 2.0  # initial_threshold: max distance 3 sigma, kernel sigma / 3 (reference kiss.py:112-113)


def _hall():
    return synth.make_path_sequence(n_scans=2, step_m=0.5)  # noise-free flat hall, obstacles of make_sequence's density


def _oracle_case(seq, T1, guess, sigma=SIGMA0):
    fd0, _ = tk.voxelize(tk.static_scan(seq, np.eye(4)))
    _, src1 = tk.voxelize(tk.static_scan(seq, T1))
    m = orc.Map(0.7, 70.0, 20)
    m.update(fd0, np.eye(4))
    T, it, nc, _ = m.register(src1, guess, 3 * sigma, sigma / 3)
    return T, it, (fd0, src1)


CASES = {
    "level_0.5_true_guess": (tk.pose(0.5), tk.pose(0.5)),
    "level_0.5_zero_guess": (tk.pose(0.5), np.eye(4)),
    "level_0.05_zero_guess": (tk.pose(0.05), np.eye(4)),
    "pitch2_heave_zero_guess": (tk.pose(0.5, dz=0.1, pitch=2.0), np.eye(4)),
}


def test_level_translation_locks_onto_the_rings_and_tilt_releases_it():
    seq = _hall()
    T, _, _ = _oracle_case(seq, *CASES["level_0.5_true_guess"])
    assert T[0, 3] < 0.7 * 0.5, T[:3, 3]     # started at the truth, slid back
    T, _, _ = _oracle_case(seq, *CASES["level_0.5_zero_guess"])
    assert T[0, 3] < 0.5 / 3, T[:3, 3]
    T, _, _ = _oracle_case(seq, *CASES["level_0.05_zero_guess"])
    assert T[0, 3] < 0.5 * 0.05, T[:3, 3]
    T1, g = CASES["pitch2_heave_zero_guess"]
    T, _, _ = _oracle_case(seq, T1, g)
    et, er = tk.err_of(T, T1)
    assert et < 0.03 and er < 0.15, (et, er)


def test_fixed_point_against_a_ground_truth_map_is_centimetres_off():
    n = 15
    seq = synth.make_path_sequence(n_scans=n, step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02,
                                   wobble_deg=1.0, heave_m=0.03, rough_amp=0.0, n_boxes=400, n_cyls=200, room_size=(180.0, 60.0, 200.0))
    gt = seq.gt_poses(0.5)
    gt = np.array([np.linalg.inv(gt[0]) @ g for g in gt])
    m = orc.Map(0.7, 70.0, 20)
    errs = []
    for k in range(n):
        fd, src = tk.voxelize(tk.perfectly_deskewed(seq, k))
        if k >= 1:
            T, _, _, _ = m.register(src, gt[k], 3 * SIGMA0, SIGMA0 / 3)
            errs.append(tk.err_of(T, gt[k])[0])
        m.update(fd, gt[k])
    errs = np.array(errs)
    assert errs[:2].max() < 0.01            # standing still: exact
    assert np.median(errs[2:]) > 0.015 and errs.max() > 0.05, errs
    assert errs.max() < 0.5                 # ... but bounded: no registration runs away by itself


@pytest.mark.parametrize("guess,ok", [("cv", False), ("gt", True)])
def test_tracking_free_running(guess, ok):
    """the whole pipeline (deskew, adaptive threshold, map from own poses) on a vehicle-speed drive: with the default
    constant-velocity guess the error is metres after 60 sweeps, with an external (ground-truth) guess it stays bounded"""
    n = 60
    seq = synth.make_path_sequence(n_scans=n, step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02,
                                   wobble_deg=1.0, heave_m=0.03, rough_amp=0.0, n_boxes=400, n_cyls=200, room_size=(180.0, 60.0, 200.0))
    gt = seq.gt_poses(0.5)
    gt = np.array([np.linalg.inv(gt[0]) @ g for g in gt])
    icp = orc.ICP(max_range=70.0, min_range=1.0)
    t01 = seq.column_times()
    orc.set_threads(min(8, synth.usable_cores()))
    try:
        errs = [tk.err_of(icp.register_frame(seq.scan(k).astype(np.float64), t01, gt[k] if guess == "gt" else None), gt[k])[0]
                for k in range(n)]
    finally:
        orc.set_threads(1)
    errs = np.array(errs)
    if ok:
        assert np.nanmax(errs) < 0.5 and np.sqrt(np.mean(errs ** 2)) < 0.2, errs
    else:
        assert not np.all(np.isfinite(errs)) or np.nanmax(errs) > 2.0, errs


@pytest.mark.gpu
def test_hip_path_agrees_with_the_oracle_on_the_locking_cases():
    """the same registrations through the C-ABI stage entry points: same fixed points, same iteration counts"""
    from ptudes_lab_amd import core
    seq = _hall()
    for name, (T1, g) in CASES.items():
        T_ref, it_ref, (fd0, src1) = _oracle_case(seq, T1, g)
        icp = core.Icp(70.0, 1.0)
        icp.map_add(fd0)
        T, it = icp.align(src1, g, 3 * SIGMA0, SIGMA0 / 3)
        assert abs(it - it_ref) <= 1, (name, it, it_ref)
        assert np.abs(T - T_ref).max() < 2e-4, (name, np.abs(T - T_ref).max())
