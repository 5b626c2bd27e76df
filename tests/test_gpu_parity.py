"""GPU parity: the HIP path (through the C-ABI of include/ptudes_mi.h) against the CPU oracle on the
same seeded inputs, and against the golden vectors generated from the reference itself.

Tolerances: integer / index work bit-exact (point selections, counts); EKF state and
covariance 1e-9; ICP teacher-forced |dt| <= 2e-4 m, angle <= 2e-5 rad; free-running trajectories
RMSE <= 1 cm over the compared span.
"""
import os

import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401  (import shim)
from oracle import cpu as orc
from ptudes_lab_amd import core, synth

pytestmark = pytest.mark.gpu

EKF_TOL = 1e-9
ICP_T_TOL, ICP_R_TOL = 2e-4, 2e-5


def _quat_close(a, b, tol):
    return min(np.abs(a - b).max(), np.abs(a + b).max()) <= tol


def _nav_close(a, b, tol=EKF_TOL):
    assert np.abs(a[:3] - b[:3]).max() <= tol
    assert _quat_close(a[3:7], b[3:7], tol)
    assert np.abs(a[7:] - b[7:]).max() <= tol


def _pose_diff(A, B):
    D = np.linalg.inv(A) @ B
    return np.linalg.norm(D[:3, 3]), orc.rot_angle(D)


# ------------------------------------------------------------------------------------------------ EKF
@pytest.mark.parametrize("variant", ["default", "init", "cov"])
def test_ekf_golden_steps(golden_dir, variant):
    g = np.load(os.path.join(golden_dir, f"ekf_steps_{variant}.npz"))
    default = variant != "init"
    ekf = core.Ekf(None if default else g["init_grav"], None if default else g["init_bacc"],
                   None if default else g["init_bgyr"])
    assert np.abs(ekf.cov - g["cov0"]).max() <= 1e-12
    upd = {int(i): k for k, i in enumerate(g["upd_idx"])}
    for i in range(len(g["imu_ts"])):
        ekf.process_imu(g["imu_lacc"][i], g["imu_avel"][i], g["imu_ts"][i])
        nav, cov = ekf.state()
        _nav_close(nav, g["nav_after_imu"][i])
        if i in upd:
            k = upd[i]
            ref = g["cov_pre"][k]
            assert np.abs(cov - ref).max() <= EKF_TOL * max(1.0, np.abs(ref).max())
            ekf.process_pose(g["upd_pose"][k], g["upd_cov"][k] if bool(g["has_cov"]) else None)
            nav, cov = ekf.state()
            ref = g["cov_post"][k]
            assert np.abs(cov - ref).max() <= EKF_TOL * max(1.0, np.abs(ref).max())
            _nav_close(nav, g["nav_after_upd"][k])
    assert ekf.ts == float(g["final_ts"])


def test_ekf_golden_sim_batched(golden_dir):
    """ekf-bench sim (reference cli/ekf_bench.py:107-179): IMU batches between updates in one launch each."""
    g = np.load(os.path.join(golden_dir, "ekf_sim.npz"))
    ekf_gt, ekf = core.Ekf(), core.Ekf()
    ts = g["ts"]
    rows_i = np.c_[ts, g["ideal_lacc"], g["ideal_avel"]]
    rows_n = np.c_[ts, g["noisy_lacc"], g["noisy_avel"]]
    last_corr, start, n_upd = ts[0], 0, 0
    for i in range(len(ts)):
        if ts[i] - last_corr > 0.1:
            ekf_gt.process_imu_batch(rows_i[start:i + 1])
            ekf.process_imu_batch(rows_n[start:i + 1])
            start = i + 1
            ekf.process_pose(ekf_gt.pose_mat())
            last_corr = ts[i]
            n_upd += 1
    ekf_gt.process_imu_batch(rows_i[start:])
    ekf.process_imu_batch(rows_n[start:])
    assert n_upd == int(g["n_updates"])
    _nav_close(ekf_gt.nav, g["nav_gt"])
    _nav_close(ekf.nav, g["nav"])
    assert np.abs(ekf.cov - g["cov"]).max() <= EKF_TOL * np.abs(g["cov"]).max()


# ------------------------------------------------------------------------------------------------ ICP stages
@pytest.fixture(scope="module")
def seq():
    return synth.make_sequence(seed=1000, n_scans=40)


def _sorted_rows(a):
    return a[np.lexsort(a.T[::-1])]


def test_map_and_linear_system_teacher_forced(seq):
    """VoxelHashMap::AddPoints/prune + GetCorrespondences/BuildLinearSystem on identical inputs."""
    rng = np.random.default_rng(3)
    gt = seq.gt_poses(0.5)
    icp = core.Icp(70.0, 1.0)
    m = orc.Map(0.7, 70.0, 20)
    for k in range(3):
        x = seq.scan(k).astype(np.float64)
        fd = orc.voxel_downsample(orc.preprocess(x, 70.0, 1.0), 0.35)
        T = np.linalg.inv(gt[0]) @ gt[k]
        w = (T @ np.c_[fd, np.ones(len(fd))].T).T[:, :3].copy()
        m.add_points(w)
        m.prune(T[:3, 3])
        icp.map_add(w, origin=T[:3, 3])
    assert icp.map_size() == (m.num_voxels, m.num_points)
    assert np.array_equal(_sorted_rows(icp.map_points()), _sorted_rows(m.points()))  # bit-exact content
    x = seq.scan(3).astype(np.float64)
    src = orc.voxel_downsample(orc.voxel_downsample(orc.preprocess(x, 70.0, 1.0), 0.35), 1.05)
    T = np.linalg.inv(gt[0]) @ gt[3] @ orc.se3_exp(rng.normal(0, 0.02, 6))
    sw = (T @ np.c_[src, np.ones(len(src))].T).T[:, :3].copy()
    for sigma in (2.0, 0.4):
        s_ref, nc_ref, cand_ref = m.linear_system(sw, 3 * sigma, sigma / 3)
        s_gpu, nc, cand = icp.linear_system(sw, 3 * sigma, sigma / 3)
        assert (nc, cand) == (nc_ref, cand_ref)
        assert np.abs(s_gpu - s_ref).max() <= 1e-9 * np.abs(s_ref).max()
    # full Gauss-Newton loop on the same map
    out_ref, it_ref, _, _ = m.register(src, T, 6.0, 2 / 3)
    out_gpu, it_gpu = icp.align(src, T, 6.0, 2 / 3)
    dt, dr = _pose_diff(out_ref, out_gpu)
    assert dt <= ICP_T_TOL and dr <= ICP_R_TOL
    assert abs(it_gpu - it_ref) <= 1


def test_map_prune_and_capacity(seq):
    """cap of 20 points per voxel in insertion order, first-point pruning, empty / ragged batches"""
    rng = np.random.default_rng(5)
    icp = core.Icp(30.0, 1.0, max_points_per_scan=20000)
    m = orc.Map(0.3, 30.0, 20)
    pos = np.zeros(3)
    for step in range(12):
        pos = pos + np.array([4.0, 1.0, 0.0])
        n = [0, 1, 257, 5000, 19999][step % 5]
        P = pos + rng.normal(0, 6, (n, 3))
        P[: n // 3] = np.round(P[: n // 3] * 2) / 2 + rng.normal(0, 0.02, (n // 3, 3))  # crowded voxels
        m.add_points(P)
        m.prune(pos)
        icp.map_add(P, origin=pos)
        assert icp.map_size() == (m.num_voxels, m.num_points)
    assert np.array_equal(_sorted_rows(icp.map_points()), _sorted_rows(m.points()))


def test_register_frame_teacher_forced_guess(seq):
    """per-scan pipeline (reference kiss.py:83-131) with the same external guess on both sides:
    selections bit-exact, poses within the ICP tolerance, stats identical"""
    gt = seq.gt_poses(0.5)
    g0i = np.linalg.inv(gt[0])
    t01 = seq.column_times()
    icp = core.Icp(70.0, 1.0)
    ref = orc.ICP(70.0, 1.0)
    for k in range(8):
        x32 = seq.scan(k)
        guess = g0i @ gt[k]
        Tr = ref.register_frame(x32.astype(np.float64), t01, guess)
        Tg = icp.register_frame(x32, None, guess)  # f32 input, column-implicit times
        sr, sg = ref.stats[-1], icp.stats[-1]
        for key in ("n_in", "n_valid", "n_down", "n_src"):
            assert sr[key] == sg[key], (k, key, sr[key], sg[key])
        if k < 2:  # no deskew yet: identical fp64 arithmetic => identical clouds
            assert np.array_equal(icp.last_frame_down(), ref.last_frame_down())
            assert np.array_equal(icp.last_source(), ref.last_source())
        else:
            assert np.abs(icp.last_frame_down() - ref.last_frame_down()).max() <= 1e-9
        dt, dr = _pose_diff(Tr, Tg)
        assert dt <= ICP_T_TOL and dr <= ICP_R_TOL, (k, dt, dr)
        assert abs(sr["sigma"] - sg["sigma"]) <= 1e-9
        assert abs(sr["iterations"] - sg["iterations"]) <= 1
        assert (sr["map_voxels"], sr["map_points"]) == (sg["map_voxels"], sg["map_points"])


def test_register_frame_edge_cases():
    icp = core.Icp(70.0, 1.0)
    # empty scan, all-invalid scan: first pose is the guess (identity), nothing enters the map
    T = icp.register_frame(np.zeros((0, 3)), np.zeros(0))
    assert np.array_equal(T, np.eye(4))
    T = icp.register_frame(np.zeros((500, 3), dtype=np.float32), None)
    assert np.array_equal(T, np.eye(4)) and icp.map_size() == (0, 0)
    with pytest.raises(RuntimeError):
        icp.register_frame(np.zeros((200000, 3)), np.zeros(200000))  # above max_points_per_scan
    with pytest.raises(ValueError):
        icp.register_frame(np.zeros((10, 2)), None)


# ------------------------------------------------------------------------------------------------ sequence
def _upload(seq, runner, n):
    for k in range(n):
        runner.upload_scan(k, seq.scan(k))
    ends = [seq.imu_range_for_scan(k)[1] for k in range(n)]
    runner.upload_imu(seq.imu[: ends[-1]], ends)


def test_sequence_imu_prediction_vs_oracle(seq):
    """driver loop (reference cli/ekf_bench.py:493-563, --use-imu-prediction) free-running on device
    against the oracle's loop on the same events"""
    n = 30
    ref = orc.run_sequence(seq.events(n), max_range=70.0, min_range=1.0, use_imu_prediction=True)
    r = core.SeqRunner(n, seq.H * seq.W, seq.imu_range_for_scan(n - 1)[1], max_range=70.0, min_range=1.0,
                       use_imu_prediction=True, with_ekf=True)
    _upload(seq, r, n)
    r.run()
    out = r.results()
    assert len(out["kiss_poses"]) == n == len(ref["kiss_poses"])
    assert np.array_equal(out["res_t"], ref["res_t"])
    d_kiss = [np.linalg.norm(out["kiss_poses"][k][:3, 3] - ref["kiss_poses"][k][:3, 3]) for k in range(n)]
    d_ekf = [np.linalg.norm(out["res_poses"][k][:3, 3] - ref["res_poses"][k][:3, 3]) for k in range(n)]
    assert np.sqrt(np.mean(np.square(d_kiss))) <= 0.01, d_kiss
    assert np.sqrt(np.mean(np.square(d_ekf))) <= 0.01, d_ekf
    ate_r, ate_t = orc.calc_ate(out["res_poses"], ref["res_poses"])
    assert ate_t <= 1e-4


def test_sequence_icp_only_matches_per_call(seq):
    """SeqRunner (no host sync per scan) == the per-call API on the same scans (constant-velocity guess)"""
    n = 12
    r = core.SeqRunner(n, seq.H * seq.W, 0, max_range=70.0, min_range=1.0, with_ekf=False)
    for k in range(n):
        r.upload_scan(k, seq.scan(k))
    r.upload_imu(np.zeros((0, 7)), [0] * n)
    r.run()
    out = r.results()
    icp = core.Icp(70.0, 1.0)
    per_call = [icp.register_frame(seq.scan(k), None) for k in range(n)]
    assert np.array_equal(out["kiss_poses"], np.array(per_call))  # same kernels, same order => bit-exact
    # deterministic across runs
    r.run()
    assert np.array_equal(r.results()["kiss_poses"], out["kiss_poses"])


def test_dense_map_config_vs_oracle():
    """BASELINE config 5 shape (64 x 2048 sweeps, 0.1 m voxels, max range 100 m) on a few sweeps: the regime
    where N_s (~45 k) exceeds the points in flight and every workgroup walks several points per iteration"""
    sq = synth.make_sequence(seed=1005, n_scans=4, H=64, W=2048, max_range=100.0)
    t01 = sq.column_times()
    icp = core.Icp(100.0, 1.0, voxel_size=0.1, scan_cols=2048, map_block_capacity=1 << 20, map_table_capacity=1 << 22)
    ref = orc.ICP(100.0, 1.0, voxel_size=0.1)
    gt = sq.gt_poses(0.5)
    for k in range(4):
        x = sq.scan(k)
        guess = np.linalg.inv(gt[0]) @ gt[k]
        Tg = icp.register_frame(x, None, guess)
        Tr = ref.register_frame(x.astype(np.float64), t01, guess)
        sr, sg = ref.stats[-1], icp.stats[-1]
        for key in ("n_valid", "n_down", "n_src", "map_voxels", "map_points", "sum_cand"):
            assert sr[key] == sg[key], (k, key, sr[key], sg[key])
        dt, dr = _pose_diff(Tr, Tg)
        assert dt <= ICP_T_TOL and dr <= ICP_R_TOL, (k, dt, dr)
    assert sg["n_src"] > 20000 and sg["map_voxels"] > 100000


@pytest.mark.parametrize("wgs,threads", [(1, 256), (3, 512), (8, 1024), (20, 256), (64, 1024), (256, 512)])
def test_gn_launch_shapes_agree(seq, wgs, threads):
    """The Gauss-Newton kernel's exchange (group leaders, idle leaders, multi-pass point walk, fewer than 8 groups)
    must not depend on the launch shape beyond the association of the sums: poses within 1e-10 of the default shape,
    integer statistics identical."""
    gt = seq.gt_poses(0.5)
    g0i = np.linalg.inv(gt[0])
    ref = core.Icp(70.0, 1.0)
    alt = core.Icp(70.0, 1.0, gn_workgroups=wgs, gn_threads=threads)
    for k in range(5):
        x = seq.scan(k)
        guess = g0i @ gt[k]
        Tr = ref.register_frame(x, None, guess)
        Ta = alt.register_frame(x, None, guess)
        dt, dr = _pose_diff(Tr, Ta)
        assert dt <= 1e-10 and dr <= 1e-10, (k, dt, dr)
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points"):
            assert ref.stats[-1][key] == alt.stats[-1][key], (k, key)


def _room_points(rng, n):
    """points on the walls / floor / ceiling of a 40 x 30 x 8 m room (the oracle KATs' scene)"""
    face = rng.integers(0, 6, n)
    u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
    L = np.array([40.0, 30.0, 8.0])
    P = np.empty((n, 3))
    for f in range(6):
        a, s = f // 2, f % 2
        b, c_ = (a + 1) % 3, (a + 2) % 3
        m = face == f
        P[m, a] = s * L[a]
        P[m, b] = u[m] * L[b]
        P[m, c_] = v[m] * L[c_]
    return P - L / 2 + rng.normal(0, 0.003, (n, 3))


def test_known_answer_registration_on_device():
    """the algebraic known-answer cases of the ICP restatement, on the HIP path itself: a known rigid
    motion is recovered (exactly when the source is the map's own points, to the sampling texture otherwise), an
    empty map and a scan without correspondences return the guess."""
    rng = np.random.default_rng(5)
    M = _room_points(rng, 300000)
    icp = core.Icp(1e6, 0.0, voxel_size=0.7, map_block_capacity=1 << 17, max_points_per_scan=300000)
    icp.map_add(M)
    ref = orc.Map(0.7, 1e9, 20)
    ref.add_points(M)
    sub = ref.points()[::9]
    Tt = orc.se3_exp(np.array([0.2, -0.1, 0.05, 0.01, -0.02, 0.03]))
    body = (np.linalg.inv(Tt) @ np.c_[sub, np.ones(len(sub))].T).T[:, :3]
    out, it = icp.align(body, np.eye(4), 6.0, 2 / 3)
    assert np.abs(out - Tt).max() < 1e-4 and it < 60
    S = orc.voxel_downsample(orc.voxel_downsample(_room_points(rng, 60000), 0.35), 1.05)
    body = (np.linalg.inv(Tt) @ np.c_[S, np.ones(len(S))].T).T[:, :3]
    out, it = icp.align(body, np.eye(4), 6.0, 2 / 3)
    assert np.linalg.norm(out[:3, 3] - Tt[:3, 3]) < 0.03 and orc.rot_angle(np.linalg.inv(Tt) @ out) < 2e-3
    out_ref, _, _, _ = ref.register(body, np.eye(4), 6.0, 2 / 3)
    dt, dr = _pose_diff(out_ref, out)
    assert dt <= ICP_T_TOL and dr <= ICP_R_TOL
    g = orc.se3_exp(np.array([1.0, 2.0, 3.0, 0.1, 0.2, 0.3]))
    out, it = core.Icp(100.0, 1.0).align(body, g, 6.0, 2 / 3)       # empty map => the guess (as an SE3d holds it)
    assert np.abs(out - g).max() < 1e-15 and it == 0
    g_bad = g.copy()
    g_bad[:3, :3] += 1e-6 * rng.normal(size=(3, 3))                   # a guess that has drifted off SO(3) comes back a rotation
    out, _ = core.Icp(100.0, 1.0).align(body, g_bad, 6.0, 2 / 3)
    assert np.abs(out[:3, :3] @ out[:3, :3].T - np.eye(3)).max() < 1e-15 and np.abs(out - g).max() < 1e-5
    ref_bad, _, _, _ = orc.Map(0.7, 100.0, 20).register(body, g_bad, 6.0, 2 / 3)
    assert np.abs(out - ref_bad).max() < 1e-15
    out, it = icp.align(body + 1000.0, np.eye(4), 6.0, 2 / 3)        # no correspondences => the guess
    assert np.array_equal(out, np.eye(4)) and it == 1


def test_threshold_landmarks_on_device():
    """sigma stays at the initial 2.0 until the sensor has moved 0.5 m from its first pose, then
    becomes the RMS of the model deviations above min_motion_th (Threshold.cpp), as in the oracle scan by scan."""
    rng = np.random.default_rng(7)
    scan = _room_points(rng, 30000)
    t01 = rng.uniform(0, 1, len(scan))
    icp, ref = core.Icp(100.0, 1.0), orc.ICP(100.0, 1.0)
    T0 = icp.register_frame(scan, t01)
    ref.register_frame(scan, t01)
    assert np.array_equal(T0, np.eye(4)) and icp.stats[0]["iterations"] == 0
    T1 = icp.register_frame(scan, t01)
    ref.register_frame(scan, t01)
    assert np.abs(T1 - np.eye(4)).max() < 1e-12 and icp.stats[1]["iterations"] == 1 and icp.stats[1]["sigma"] == 2.0
    for k in range(1, 4):
        G = np.eye(4)
        G[0, 3] = 1.0 * k
        guess = G @ orc.se3_exp(np.array([0.15, 0, 0, 0, 0, 0]))
        icp.register_frame(scan - G[:3, 3], t01, guess)
        ref.register_frame(scan - G[:3, 3], t01, guess=guess)
    sig = [s["sigma"] for s in icp.stats]
    assert sig[2] == 2.0 and 0.1 < sig[-1] < 0.6
    assert np.allclose(sig, [s["sigma"] for s in ref.stats], rtol=0, atol=1e-9)


@pytest.mark.parametrize("seed,min_range,max_range", [(1003, 1.0, 70.0), (1006, 2.0, 50.0)])
def test_sequence_other_seeds_and_ranges_vs_oracle(seed, min_range, max_range):
    """the free-running loop on other sequences and range settings (voxel size follows max_range / 100): GPU and
    oracle stay together far below the 1 cm bar, and every integer statistic of every scan is identical"""
    n = 24
    sq = synth.make_sequence(seed=seed, n_scans=n, min_range=min_range, max_range=max_range)
    orc.set_threads(8)
    try:
        ref = orc.run_sequence(sq.events(n), max_range=max_range, min_range=min_range, use_imu_prediction=True)
    finally:
        orc.set_threads(1)
    r = core.SeqRunner(n, sq.H * sq.W, sq.imu_range_for_scan(n - 1)[1], max_range=max_range, min_range=min_range,
                       use_imu_prediction=True, with_ekf=True)
    _upload(sq, r, n)
    r.run()
    out = r.results()
    d = [np.linalg.norm(out["res_poses"][k][:3, 3] - ref["res_poses"][k][:3, 3]) for k in range(n)]
    assert max(d) <= 1e-9, d
    for k in range(n):
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "map_voxels", "map_points"):
            assert out["stats"][k][key] == ref["stats"][k][key], (k, key)


def test_exchange_epoch_wraparound(seq):
    """the flag of the Gauss-Newton exchange carries a 22-bit launch epoch; crossing its wrap-around (buffers are cleared
    and flags restart) must not change a single bit of the result"""
    from ptudes_lab_amd import _lib as L
    gt = seq.gt_poses(0.5)
    g0i = np.linalg.inv(gt[0])
    ref, alt = core.Icp(70.0, 1.0), core.Icp(70.0, 1.0)
    L.check(L.lib().ptl_icp_debug_set_epoch(alt._h, 0x3FFFFD))
    for k in range(6):
        x = seq.scan(k)
        guess = g0i @ gt[k]
        Tr = ref.register_frame(x, None, guess)
        Ta = alt.register_frame(x, None, guess)
        assert np.array_equal(Tr, Ta), k
        assert ref.stats[-1]["iterations"] == alt.stats[-1]["iterations"]


def test_long_run_grows_the_trajectory_buffers():
    """more sweeps than the initial trajectory capacity (4096): the buffers are re-allocated in the middle of a resident
    run, with the EKF stream still reading rows of the old ones - nothing may change"""
    n = 4200
    sq = synth.make_sequence(seed=77, n_scans=n, H=16, W=64)
    n_imu = sq.imu_range_for_scan(n - 1)[1]

    def run(m):
        r = core.SeqRunner(m, sq.H * sq.W, sq.imu_range_for_scan(m - 1)[1], max_range=70.0, min_range=1.0,
                           use_imu_prediction=True, with_ekf=True, scan_cols=sq.W)
        _upload(sq, r, m)
        r.run()
        return r.results()

    full, head = run(n), run(64)
    assert len(full["kiss_poses"]) == n and np.isfinite(full["res_poses"]).all()
    assert np.array_equal(full["kiss_poses"][:64], head["kiss_poses"])
    assert np.array_equal(full["res_poses"][:64], head["res_poses"])
    assert n_imu > 0 and full["res_t"][-1] > full["res_t"][0]


# ------------------------------------------------------------------------------------------------ long free-running parity
def _threaded_oracle(events, **kw):
    orc.set_threads(min(16, synth.usable_cores()))  # fixed 128-point chunks reduced in chunk order: deterministic
    try:
        return orc.run_sequence(events, **kw)
    finally:
        orc.set_threads(1)


def test_free_running_220_sweeps_imu_mode_vs_oracle():
    """BASELINE config 3's mode (ICP + IMU-EKF, --use-imu-prediction) free-running for 220 sweeps: the HIP loop and the
    oracle's loop stay together to 1e-9 m on every pose, and every integer statistic of every scan is identical"""
    n = 220
    sq = synth.make_sequence(seed=1000, n_scans=n)
    ref = _threaded_oracle(sq.events(n), max_range=70.0, min_range=1.0, use_imu_prediction=True)
    r = core.SeqRunner(n, sq.H * sq.W, sq.imu_range_for_scan(n - 1)[1], max_range=70.0, min_range=1.0,
                       use_imu_prediction=True, with_ekf=True)
    _upload(sq, r, n)
    r.run()
    out = r.results()
    assert len(out["res_poses"]) == n == len(ref["res_poses"])
    d = np.linalg.norm(out["res_poses"][:, :3, 3] - ref["res_poses"][:, :3, 3], axis=1)
    dk = np.linalg.norm(out["kiss_poses"][:, :3, 3] - ref["kiss_poses"][:, :3, 3], axis=1)
    assert d.max() <= 1e-9 and dk.max() <= 1e-9, (d.max(), dk.max())
    assert np.abs(out["res_poses"][:, :3, :3] - ref["res_poses"][:, :3, :3]).max() <= 1e-9
    for k in range(n):
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points"):
            assert out["stats"][k][key] == ref["stats"][k][key], (k, key)
    ate_r, ate_t = orc.calc_ate(out["res_poses"], ref["res_poses"])
    assert ate_t <= 1e-16 and ate_r <= 1e-16


def test_free_running_icp_only_vs_oracle():
    """BASELINE config 2's mode: ICP only, the reference's DEFAULT constant-velocity guess (kiss.py:102-105), no filter -
    against the ORACLE over 200 sweeps of the benchmark's sequence.  (Until round 2 re-orthonormalised the poses the way
    upstream's Sophus::SE3d does, this loop amplified rounding differences 2.4 x per sweep and lost the track around
    sweep 40; it is now as tight as the IMU mode.)  The trajectory stays on the ground truth up to the bootstrap offset."""
    n = 200
    sq = synth.make_sequence(seed=1000, n_scans=n)
    ref = _threaded_oracle(sq.events(n), max_range=70.0, min_range=1.0, with_ekf=False)
    r = core.SeqRunner(n, sq.H * sq.W, 0, max_range=70.0, min_range=1.0, with_ekf=False)
    for k in range(n):
        r.upload_scan(k, sq.scan(k))
    r.upload_imu(np.zeros((0, 7)), [0] * n)
    r.run()
    out = r.results()
    assert len(out["kiss_poses"]) == n == len(ref["kiss_poses"])
    d = np.linalg.norm(out["kiss_poses"][:, :3, 3] - ref["kiss_poses"][:, :3, 3], axis=1)
    assert d.max() <= 1e-9, d.max()
    for k in range(n):
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points"):
            assert out["stats"][k][key] == ref["stats"][k][key], (k, key)
        assert abs(out["stats"][k]["sigma"] - ref["stats"][k]["sigma"]) <= 1e-9
    R = out["kiss_poses"][:, :3, :3]
    assert np.abs(np.einsum("nij,nkj->nik", R, R) - np.eye(3)).max() < 1e-14  # every pose a rotation to rounding
    gt = sq.gt_poses(0.5)
    gt = np.array([np.linalg.inv(gt[0]) @ g for g in gt])
    err = np.linalg.norm(out["kiss_poses"][:, :3, 3] - gt[:, :3, 3], axis=1)
    assert err.max() < 0.6 and abs(err[-1] - err[60]) < 0.15, (err.max(), err[60], err[-1])  # flat: the bootstrap offset


def test_constant_velocity_vehicle_drive_on_device():
    """200 sweeps at 10 m/s (190 m) in the reference's default mode through the device-resident loop: holds the track,
    drift under 0.8 % of the distance travelled, and equals the oracle's trajectory to 1e-9 m"""
    n = 200
    sq = synth.make_path_sequence(n_scans=n, step_m=1.0, static_sweeps=2, ramp_sweeps=10, noise_std=0.01, dropout=0.02,
                                  wobble_deg=1.0, heave_m=0.03, rough_amp=0.0, n_boxes=400, n_cyls=200, room_size=(280.0, 60.0, 200.0),
                                  ray_jitter_deg=0.3)
    ref = _threaded_oracle(sq.events(n), max_range=70.0, min_range=1.0, with_ekf=False)
    r = core.SeqRunner(n, sq.H * sq.W, 0, max_range=70.0, min_range=1.0, with_ekf=False)
    for k in range(n):
        r.upload_scan(k, sq.scan(k))
    r.upload_imu(np.zeros((0, 7)), [0] * n)
    r.run()
    out = r.results()
    d = np.linalg.norm(out["kiss_poses"][:, :3, 3] - ref["kiss_poses"][:, :3, 3], axis=1)
    assert d.max() <= 1e-9, d.max()
    gt = sq.gt_poses(0.5)
    gt = np.array([np.linalg.inv(gt[0]) @ g for g in gt])
    err = np.linalg.norm(out["kiss_poses"][:, :3, 3] - gt[:, :3, 3], axis=1)
    travelled = np.linalg.norm(gt[-1, :3, 3])
    assert travelled > 180.0 and err.max() < 0.008 * travelled, (travelled, err.max())


def test_dense_map_config5_full_size_vs_oracle():
    """BASELINE config 5 at its stated size: 64 x 2048 sweeps, 0.1 m voxels, max range 100 m, run until the local map
    holds >= 5 M points.  Counts of every scan identical to the oracle's, poses to 1e-9 m."""
    n = 72
    sq = synth.make_sequence(seed=1000, n_scans=n, H=64, W=2048, max_range=100.0)
    over = dict(voxel_size=0.1)
    ref = _threaded_oracle(sq.events(n), max_range=100.0, min_range=1.0, use_imu_prediction=True, **over)
    r = core.SeqRunner(n, sq.H * sq.W, sq.imu_range_for_scan(n - 1)[1], max_range=100.0, min_range=1.0,
                       use_imu_prediction=True, with_ekf=True, scan_cols=2048, map_block_capacity=1 << 21,
                       map_table_capacity=1 << 23, **over)
    _upload(sq, r, n)
    r.run()
    out = r.results()
    assert out["stats"][-1]["map_points"] >= 5_000_000, out["stats"][-1]
    d = np.linalg.norm(out["res_poses"][:, :3, 3] - ref["res_poses"][:, :3, 3], axis=1)
    assert d.max() <= 1e-9, d.max()
    for k in range(n):
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points"):
            assert out["stats"][k][key] == ref["stats"][k][key], (k, key)


def test_throughput_kernel_8_lanes_per_point_vs_oracle():
    """the 8-lanes-per-point Gauss-Newton kernel (moment accumulation, one-hop exchange; what the batched runner runs per
    sequence) free-running against the oracle: same correspondences (pair and candidate counts of every scan identical),
    poses to 1e-9 m; and against the 32-lane kernel on the same scans"""
    n = 40
    sq = synth.make_sequence(seed=1002, n_scans=n)
    ref = _threaded_oracle(sq.events(n), max_range=70.0, min_range=1.0, use_imu_prediction=True)
    outs = {}
    for lanes in (8, 32):
        r = core.SeqRunner(n, sq.H * sq.W, sq.imu_range_for_scan(n - 1)[1], max_range=70.0, min_range=1.0,
                           use_imu_prediction=True, with_ekf=True, gn_workgroups=32, gn_lanes_per_point=lanes,
                           gn_threads=512 if lanes == 8 else 1024)
        _upload(sq, r, n)
        r.run()
        outs[lanes] = r.results()
    out = outs[8]
    d = np.linalg.norm(out["res_poses"][:, :3, 3] - ref["res_poses"][:, :3, 3], axis=1)
    assert d.max() <= 1e-9, d
    assert np.abs(out["kiss_poses"] - outs[32]["kiss_poses"]).max() <= 1e-9
    for k in range(n):
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points"):
            assert out["stats"][k][key] == ref["stats"][k][key], (k, key)


def test_throughput_kernel_with_many_points_per_workgroup_vs_oracle():
    """the 8-lanes-per-point kernel on 4 workgroups: ~1750 source points each, i.e. four phase-A chunks whose misses share
    the search queue - which overflows (1024 entries) and is flushed between the chunks in the first iterations"""
    n = 14
    sq = synth.make_sequence(seed=1005, n_scans=n)
    ref = _threaded_oracle(sq.events(n), max_range=70.0, min_range=1.0, use_imu_prediction=True)
    r = core.SeqRunner(n, sq.H * sq.W, sq.imu_range_for_scan(n - 1)[1], max_range=70.0, min_range=1.0,
                       use_imu_prediction=True, with_ekf=True, gn_workgroups=4, gn_lanes_per_point=8, gn_threads=512)
    _upload(sq, r, n)
    r.run()
    out = r.results()
    assert out["stats"][-1]["n_src"] > 4 * 1024
    d = np.linalg.norm(out["res_poses"][:, :3, 3] - ref["res_poses"][:, :3, 3], axis=1)
    assert d.max() <= 1e-9, d
    for k in range(n):
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points"):
            assert out["stats"][k][key] == ref["stats"][k][key], (k, key)


def test_throughput_kernel_over_the_whole_chip_vs_oracle():
    """the 8-lanes-per-point kernel for ONE sequence on all 256 workgroups (two-hop exchange; the dense-scan form)"""
    n = 24
    sq = synth.make_sequence(seed=1004, n_scans=n)
    ref = _threaded_oracle(sq.events(n), max_range=70.0, min_range=1.0, use_imu_prediction=True)
    r = core.SeqRunner(n, sq.H * sq.W, sq.imu_range_for_scan(n - 1)[1], max_range=70.0, min_range=1.0,
                       use_imu_prediction=True, with_ekf=True, gn_lanes_per_point=8, gn_threads=512)
    _upload(sq, r, n)
    r.run()
    out = r.results()
    d = np.linalg.norm(out["res_poses"][:, :3, 3] - ref["res_poses"][:, :3, 3], axis=1)
    assert d.max() <= 1e-9, d
    for k in range(n):
        for key in ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points"):
            assert out["stats"][k][key] == ref["stats"][k][key], (k, key)


def test_persistent_launch_is_checked_and_a_missing_workgroup_does_not_hang(seq):
    """(1) a Gauss-Newton grid that cannot be co-resident is refused at create (occupancy x CUs), not discovered
    through a poll time-out; (2) a workgroup that never arrives (test hook) makes the others time out ONCE, raise the abort
    word and leave together - seconds, not max_iter x time-out - and the error surfaces at the next synchronising call;
    (3) the handle works again afterwards"""
    import time
    from ptudes_lab_amd import _lib as L
    with pytest.raises(RuntimeError, match="co-resident"):
        core.Icp(70.0, 1.0, gn_workgroups=512)  # 512 x 1024 threads: two workgroups per CU do not fit
    icp = core.Icp(70.0, 1.0)
    x0, x1 = seq.scan(0), seq.scan(1)
    icp.register_frame(x0, None)
    L.check(L.lib().ptl_icp_debug_stall_workgroup(icp._h, 37))
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="0x10|timeout"):
        icp.register_frame(x1, None)
    assert time.perf_counter() - t0 < 120.0
    L.check(L.lib().ptl_icp_debug_stall_workgroup(icp._h, -1))
    ref = core.Icp(70.0, 1.0)
    ref.register_frame(x0, None)
    T = ref.register_frame(x1, None)
    assert np.isfinite(T).all()
    # the 8-lanes-per-point kernel (one-hop exchange, what a batch member runs) leaves the same way
    import ctypes as C
    r = core.SeqRunner(3, seq.H * seq.W, 0, max_range=70.0, min_range=1.0, with_ekf=False, gn_workgroups=32,
                       gn_lanes_per_point=8, gn_threads=512)
    for k in range(3):
        r.upload_scan(k, seq.scan(k))
    r.upload_imu(np.zeros((0, 7)), [0] * 3)
    h = C.c_void_p()
    L.check(L.lib().ptl_seq_icp(r._h, C.byref(h)))
    r.run(1)
    L.check(L.lib().ptl_icp_debug_stall_workgroup(h, 5))
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="0x10|timeout"):
        r.advance(2)
    assert time.perf_counter() - t0 < 120.0
