"""GPU tests of the drop-in surface: ESEKF / KissICPWrapper classes and the `ekf-bench` commands."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
from click.testing import CliRunner

import ptudes_lab_amd  # noqa: F401
from oracle import cpu as orc
from ptudes_lab_amd import sequence, synth
from ptudes_lab_amd import utils as pu
from ptudes_lab_amd.cli.run import ptudes_cli
from ptudes_lab_amd.ins.data import IMU, ekf_traj_ate
from ptudes_lab_amd.ins.es_ekf import ESEKF
from ptudes_lab_amd.kiss import KissICPWrapper

pytestmark = pytest.mark.gpu


def test_ekf_bench_sim_stdout_matches_reference(golden_dir):
    """`ptudes ekf-bench sim -t 2.0` with the legacy RNG seeded to 0 prints what the reference printed"""
    np.random.seed(0)
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "sim", "-t", "2.0"])
    assert res.exit_code == 0, res.output
    ref = open(os.path.join(golden_dir, "ekf_sim_stdout.txt")).read()
    assert res.output == ref


def test_esekf_class_logging_and_ate(golden_dir):
    g = np.load(os.path.join(golden_dir, "ekf_sim.npz"))
    ekf_gt, ekf = ESEKF(_logging=True), ESEKF(_logging=True)
    last = g["ts"][0]
    for i, ts in enumerate(g["ts"]):
        ekf_gt.processImu(IMU(g["ideal_lacc"][i], g["ideal_avel"][i], ts))
        ekf.processImu(IMU(g["noisy_lacc"][i], g["noisy_avel"][i], ts))
        if ts - last > 0.1:
            ekf.processPose(ekf_gt.nav.pose_mat())
            last = ts
    assert len(ekf._nav_update_idxs) == int(g["n_updates"])
    assert len(ekf._navs) == len(ekf._navs_pred) == len(ekf._navs_t)
    r, t = ekf_traj_ate(ekf_gt, ekf)
    assert abs(r - float(g["ate_rot"])) < 1e-9 and abs(t - float(g["ate_trans"])) < 1e-9
    assert ekf.ts == g["ts"][-1]
    assert np.abs(ekf._cov - g["cov"]).max() <= 1e-9 * np.abs(g["cov"]).max()


@pytest.fixture(scope="module")
def seq():
    return synth.make_sequence(seed=1001, n_scans=12)


def test_kiss_wrapper_matches_oracle(seq):
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=seq.W, pixels_per_column=seq.H))
    w = KissICPWrapper(meta, _min_range=1.0, _max_range=70.0)
    ref = orc.ICP(70.0, 1.0)
    t01 = seq.column_times()
    gt = seq.gt_poses(0.5)
    for k in range(6):
        x = seq.scan(k)
        guess = np.linalg.inv(gt[0]) @ gt[k] if k % 2 else None  # alternate external / constant-velocity guess
        scan = SimpleNamespace(xyz=x.reshape(seq.H, seq.W, 3), ts=100.0 + 0.1 * k)
        T = w.register_frame(scan, initial_guess=guess)
        Tr = ref.register_frame(x.astype(np.float64), t01, guess)
        assert np.abs(T - Tr).max() < 2e-4
        assert abs(w._sigmas[-1] - ref.stats[-1]["sigma"]) < 1e-9
        assert abs(w._err_dt[-1] - ref.stats[-1]["err_dt"]) < 2e-4
    # KissICPWrapper.deskew == Deskew.cpp on the last two poses
    pts = seq.scan(5).astype(np.float64)[::50]
    tt = t01[::50]
    assert np.abs(w.deskew(pts, tt) - orc.deskew(pts, tt, ref.pose(-2), ref.pose(-1))).max() < 1e-9
    assert len(w.poses) == 6 and w.poses_ts[-1] == 100.5
    assert np.abs(w._kiss.get_prediction_model() - ref.prediction()).max() < 4e-4
    assert np.allclose(w.velocity, w._kiss.get_prediction_model()[:3, 3] / 0.1)
    assert w.local_map_points.shape == (ref.stats[-1]["map_points"], 3)
    assert w._config.mapping.voxel_size == 0.7 and w._config.data.max_range == 70.0
    # masked input form (reference kiss.py:59-61): only RANGE != 0 pixels, with their per-pixel times
    x = seq.scan(6)
    rng = (np.abs(x).sum(1) > 0).astype(np.uint32)
    T = w.register_frame(SimpleNamespace(xyz=x, range=rng, ts=100.6))
    Tr = ref.register_frame(x[rng != 0].astype(np.float64), t01[rng != 0], None)
    assert np.abs(T - Tr).max() < 2e-4


def test_event_loop_equals_resident_runner(seq):
    """per-call driver loop (host round trip per event) == device-resident runner, both --use-imu-prediction"""
    n = 8
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=seq.W, pixels_per_column=seq.H))
    a = sequence.run_events(sequence.synthetic_events(seq, n), meta, use_imu_prediction=True)
    b = sequence.run_resident(seq, n, use_imu_prediction=True)
    assert np.array_equal(np.array(a["res_t"]), b["res_t"])
    assert np.abs(np.array(a["kiss_poses"]) - b["kiss_poses"]).max() < 1e-9
    assert np.abs(np.array(a["res_poses"]) - b["res_poses"]).max() < 1e-9


def test_lazy_map_stats_gives_the_same_poses_without_waiting_for_the_map_update(seq):
    """ptl_icp_set_lazy_map_stats: a registration returns when its pose is there, the map update still under way; poses and every
    other statistic are those of the default mode, the rows carry no map size, ptl_icp_map_size gives it on demand"""
    n = 6
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=seq.W, pixels_per_column=seq.H))
    a = KissICPWrapper(meta, _min_range=1.0, _max_range=70.0, lazy_map_stats=True)
    b = KissICPWrapper(meta, _min_range=1.0, _max_range=70.0)
    for k in range(n):
        x = seq.scan(k)
        Ta = a.register_frame(SimpleNamespace(xyz=x, ts=100.0 + 0.1 * k))
        Tb = b.register_frame(SimpleNamespace(xyz=x, ts=100.0 + 0.1 * k))
        assert np.array_equal(Ta, Tb), k
        sa, sb = a.stats[-1], b.stats[-1]
        assert (sa["map_voxels"], sa["map_points"]) == (-1, -1) and sb["map_voxels"] > 0
        assert all(sa[q] == sb[q] for q in ("sigma", "err_dt", "err_drot", "iterations", "n_corr_last", "n_valid", "n_down", "n_src", "sum_cand"))
    assert a._icp.map_size() == b._icp.map_size() == (b.stats[-1]["map_voxels"], b.stats[-1]["map_points"])


def test_calls_right_after_a_lazy_registration_are_ordered_behind_its_map_update(seq):
    """lazy_map_stats lets register_frame return while the scan's map update (K7-K10) still runs on the map stream: whatever touches the
    map, the device state or the staging buffers next - map_add, align, linear_system, deskew, the prediction - has to wait for it
    (they used to start beside it).  A lazy handle and a default one go through the same calls: identical results."""
    from ptudes_lab_amd import core
    rng = np.random.default_rng(11)
    gt = seq.gt_poses(0.5)
    g0i = np.linalg.inv(gt[0])
    a, b = core.Icp(70.0, 1.0, lazy_map_stats=True), core.Icp(70.0, 1.0)
    extra = rng.uniform(-20, 20, (3000, 3))
    for k in range(5):
        x = seq.scan(k)
        guess = g0i @ gt[k]
        Ta, Tb = a.register_frame(x, None, guess), b.register_frame(x, None, guess)
        assert np.array_equal(Ta, Tb), k
        if k == 1:  # straight into the map while the lazy handle's update of scan 1 is still in flight
            a.map_add(extra + 0.01 * k, origin=Ta[:3, 3])
            b.map_add(extra + 0.01 * k, origin=Tb[:3, 3])
            assert a.map_size() == b.map_size()
        if k == 2:  # a registration against the map as that scan left it: needs the complete update
            src = x[np.linalg.norm(x, axis=1) > 1.0][::37].astype(np.float64)
            Pa, ia = a.align(src, Ta, 6.0, 2.0 / 3.0)
            Pb, ib = b.align(src, Tb, 6.0, 2.0 / 3.0)
            assert np.array_equal(Pa, Pb) and ia == ib
        if k == 3:
            w = (Ta @ np.c_[x[::53].astype(np.float64), np.ones(len(x[::53]))].T).T[:, :3].copy()
            sa, sb = a.linear_system(w, 6.0, 2.0 / 3.0), b.linear_system(w, 6.0, 2.0 / 3.0)
            assert sa[1:] == sb[1:] and np.array_equal(sa[0], sb[0])
            assert np.array_equal(a.prediction(), b.prediction())
            t01 = seq.column_times()[::53]
            assert np.array_equal(a.deskew(x[::53].astype(np.float64), t01), b.deskew(x[::53].astype(np.float64), t01))
    assert a.map_size() == b.map_size()
    pa, pb = a.map_points(), b.map_points()
    assert np.array_equal(pa[np.lexsort(pa.T[::-1])], pb[np.lexsort(pb.T[::-1])])


def test_fused_loop_body_equals_the_resident_run_and_the_call_by_call_loop(seq):
    """sequence.run_events(fused=True): IMU samples held back and sent with the scan (ptl_icp_ekf_step: predicts, registration with the
    filter's pose read on the device, update - one host wait per scan).  Against the device-resident runner (same kernels, the guess
    never leaves the device): every pose and timestamp bit for bit; against the call-by-call loop (reference ekf_bench.py:493-563
    object by object): 1e-9 m; the wrappers' logs (poses, sigmas, innovation) are filled the same way; constant-velocity mode too."""
    n = 10
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=seq.W, pixels_per_column=seq.H))
    for use_imu in (True, False):
        ev = list(sequence.synthetic_events(seq, n))
        a = sequence.run_events(iter(ev), meta, use_imu_prediction=use_imu, fused=True)
        b = sequence.run_events(iter(ev), meta, use_imu_prediction=use_imu, fused=False)
        r = sequence.run_resident(seq, n, use_imu_prediction=use_imu, map_table_capacity=1 << 22)
        assert len(a["res_poses"]) == n and a["timings"]["n_corr"] == n and a["timings"]["n_imu"] == b["timings"]["n_imu"]
        assert np.array_equal(np.array(a["res_t"]), r["res_t"]) and np.array_equal(np.array(a["res_t"]), np.array(b["res_t"]))
        assert np.array_equal(np.array(a["kiss_poses"]), r["kiss_poses"]), use_imu
        assert np.array_equal(np.array(a["res_poses"]), r["res_poses"]), use_imu
        assert np.abs(np.array(a["kiss_poses"]) - np.array(b["kiss_poses"])).max() < 1e-9
        assert np.abs(np.array(a["res_poses"]) - np.array(b["res_poses"])).max() < 1e-9
        ka, kb = a["kiss_icp"], b["kiss_icp"]
        assert len(ka.poses) == len(kb.poses) == n and ka.poses_ts == kb.poses_ts
        assert np.abs(np.array(ka._sigmas) - np.array(kb._sigmas)).max() < 1e-9 and np.abs(np.array(ka._err_dt) - np.array(kb._err_dt)).max() < 1e-9
        assert a["ekf"].ts == b["ekf"].ts and a["ekf"]._imu_idx == b["ekf"]._imu_idx
        assert np.abs(a["ekf"].nav.pos - b["ekf"].nav.pos).max() < 1e-9  # (the filter object is usable afterwards: state read on demand)


def test_fused_loop_takes_any_number_of_imu_samples_between_two_scans(seq):
    """A feed with a long IMU-only prefix / a 1 kHz IMU / dropped lidar frames: more samples between two scans than the filter's staging
    buffer holds (1024).  The fused loop body must serve them like the call-by-call loop does (reference cli/ekf_bench.py:493-563 takes
    any count) - the same filter state bit for bit (same kernel, same rows, chunked) - and stamp every held sample's `dt` the way
    ESEKF.processImu does (es_ekf.py:194-196)."""
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=seq.W, pixels_per_column=seq.H))
    rng = np.random.default_rng(3)
    n_pre = 2500
    t0 = float(seq.imu[0, 0]) - n_pre * 0.001

    def feed():
        for i in range(n_pre):  # standing still for 2.5 s at 1 kHz before the first sweep
            yield ("imu", IMU(np.array([0.0, 0.0, 9.78]) + 1e-3 * rng.standard_normal(3), 1e-4 * rng.standard_normal(3), t0 + 0.001 * i))
        yield from sequence.synthetic_events(seq, 3)

    rng = np.random.default_rng(3)
    ev = list(feed())
    a = sequence.run_events(iter(ev), meta, use_imu_prediction=True, fused=True)
    dts_fused = [e[1].dt for e in ev if e[0] == "imu"]
    b = sequence.run_events(iter(ev), meta, use_imu_prediction=True, fused=False)
    dts_calls = [e[1].dt for e in ev if e[0] == "imu"]
    assert dts_fused == dts_calls and dts_fused[0] == ev[0][1].ts and abs(dts_fused[5] - 0.001) < 1e-6
    assert a["timings"]["n_imu"] == b["timings"]["n_imu"] == n_pre + 30 and a["timings"]["fused"] and not b["timings"]["fused"]
    assert a["ekf"]._imu_idx == b["ekf"]._imu_idx and a["ekf"].ts == b["ekf"].ts
    assert np.array_equal(np.array(a["res_t"]), np.array(b["res_t"]))
    assert np.abs(np.array(a["res_poses"]) - np.array(b["res_poses"])).max() < 1e-9
    assert np.abs(np.array(a["kiss_poses"]) - np.array(b["kiss_poses"])).max() < 1e-9


def test_ouster_command_on_synthetic_writes_pose_files(tmp_path):
    fk, fn = tmp_path / "k.txt", tmp_path / "n.csv"
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "ouster", "--synthetic", "1002", "--end-scan", "5",
                                          "--use-imu-prediction", "--save-kitti-poses", str(fk),
                                          "--save-nc-gt-poses", str(fn)])
    assert res.exit_code == 0, res.output
    assert "Timings:" in res.output and "KissICP register frame:" in res.output
    back = pu.read_newer_college_gt(str(fn))
    assert len(back) == 6 and np.loadtxt(str(fk)).shape == (6, 12)
    assert fn.read_text().startswith("# data path: synthetic:1002")
    # cmp of the file against itself: zero error
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "cmp", str(fn), str(fn)])
    assert res.exit_code == 0 and "ATE trans: 0.0000 m" in res.output


def test_ouster_command_gt_guess_plot_and_timings(tmp_path):
    """`--use-gt-guess` (GT pose interpolated at the scan's timestamp as KissICP guess, reference ekf_bench.py:536-542),
    `-p graphs` (the ground-truth comparison block, :599-633) and the four-line Timings block (:590-595)"""
    from ptudes_lab_amd import synth
    n = 8
    seq = synth.make_sequence(seed=1002, n_scans=n)
    t = np.arange(0.0, (n + 1) * seq.scan_dt, 0.02)
    gt_csv = tmp_path / "gt.csv"
    pu.save_poses_nc_gt_format(str(gt_csv), t=list(seq.t_base + t), poses=list(seq.pose_at(t)))
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "ouster", "--synthetic", "1002", "--end-scan", str(n - 1),
                                          "--use-gt-guess", "-g", str(gt_csv), "-p", "graphs"])
    assert res.exit_code == 0, res.output
    out = res.output
    assert "use-imu-prediction: False, use-gt-guess: True" in out and "metrics logging: True" in out
    for line in ("ESEKF imu process:", "ESEKF update:", "KissICP register frame:", "Stats tracking:"):
        assert line in out, line
    assert "Ground truth comparison (with ES EKF smoothing" in out and "Ground truth comparison (no-EKF, only KissICP" in out
    ate = [float(ln.split()[2]) for ln in out.splitlines() if ln.startswith("ATE trans:")]
    assert len(ate) == 2 and max(ate) < 0.05  # m^2 (reference-style mean squares): the guess is the truth
    # without -p the comparison block is not printed (it lives under `plot == "graphs"` in the reference)
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "ouster", "--synthetic", "1002", "--end-scan", "3", "-g", str(gt_csv)])
    assert res.exit_code == 0 and "Ground truth comparison" not in res.output and "metrics logging: False" in res.output
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "ouster", "--synthetic", "1002", "--end-scan", "2", "-p", "nonsense"])
    assert res.exit_code == 0 and "WARNING: plot param 'nonsense' doesn't supported" in res.output
