"""Batched runner: S sequences on one GPU, each on its share of one XCD (the workgroups with blockIdx & 7 == s & 7), in
either driver - the free-running kernel (kx_seq_run: one persistent launch, every sequence at its own pace) or lockstep
(one launch per stage) - == S independent device-resident runs with as many Gauss-Newton workgroups each, bit for bit."""
import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import core, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("use_imu,S,lanes,driver", [(True, 3, 8, "free"), (False, 3, 8, "free"), (True, 8, 8, "free"), (True, 3, 32, "lockstep"),
                                                    (True, 12, 8, "free"), (True, 20, 8, "free"), (True, 3, 8, "lockstep"),
                                                    (True, 12, 8, "lockstep"), (True, 5, 8, "free3"), (True, 43, 8, "free")])
def test_batch_equals_independent_runs(use_imu, S, lanes, driver):
    # (43 sequences: more than the 32 teams of the chip - five or six per XCD, whose four teams take the scans as they come free)
    n = 10 if S <= 8 else 6 if S <= 32 else 5
    seqs = [synth.make_sequence(seed=1010 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    # "free3": launches of 3 scans (several per run) and a hash-table rebuild every 4 scans inside the kernel
    more = dict(scans_per_launch=3) if driver == "free3" else {}
    rebuild = dict(rebuild_every=4) if driver == "free3" else {}
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=use_imu, with_ekf=True, gn_lanes_per_point=lanes,
                         gn_threads=512 if lanes == 8 else 1024, free_running=driver != "lockstep", **more, **rebuild)
    assert b.free_running == (driver != "lockstep")
    singles = []
    for s, sq in enumerate(seqs):
        # one sequence of the batch runs on 256 / 8 = 32 workgroups (16 / 8 when two / four sequences share an XCD): the
        # independent run it must equal uses as many
        r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=use_imu, with_ekf=True, gn_workgroups=32 if S <= 8 else 16 if S <= 16 else 8,
                           gn_lanes_per_point=lanes, gn_threads=512 if lanes == 8 else 1024, **rebuild)
        ends = [sq.imu_range_for_scan(k)[1] for k in range(n)]
        for k in range(n):
            x = sq.scan(k)
            b.upload_scan(s, k, x)
            r.upload_scan(k, x)
        b.upload_imu(s, sq.imu[:n_imu], ends)
        r.upload_imu(sq.imu[:n_imu], ends)
        r.run()
        singles.append(r.results())
    b.run(4)       # cold start + 4 scans ...
    b.enqueue(n - 4)  # ... then the rest without a reset
    b.wait()
    for s in range(S):
        out = b.results(s)
        assert np.array_equal(out["kiss_poses"], singles[s]["kiss_poses"]), s
        assert np.array_equal(out["res_poses"], singles[s]["res_poses"]), s
        assert np.array_equal(out["res_t"], singles[s]["res_t"])
        assert out["stats"] == singles[s]["stats"]
    # sequences differ from each other (the batch is not computing one thing S times)
    assert not np.array_equal(b.results(0)["kiss_poses"], b.results(1)["kiss_poses"])


@pytest.mark.parametrize("free", [True, False])
def test_batch_icp_only_equals_independent_runs(free):
    """no filter (BASELINE config 2): every workgroup of a team works on the scan, constant-velocity guess"""
    S, n = 3, 8
    seqs = [synth.make_sequence(seed=1040 + s, n_scans=n) for s in range(S)]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, 0, with_ekf=False, free_running=free)
    for s, sq in enumerate(seqs):
        r = core.SeqRunner(n, sq.H * sq.W, 0, with_ekf=False, gn_workgroups=32, gn_lanes_per_point=8, gn_threads=512)
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
            r.upload_scan(k, sq.scan(k))
        b.upload_imu(s, np.zeros((0, 7)), [0] * n)
        r.upload_imu(np.zeros((0, 7)), [0] * n)
        r.run()
        seqs[s] = r.results()
    b.run()
    for s in range(S):
        out = b.results(s)
        assert np.array_equal(out["kiss_poses"], seqs[s]["kiss_poses"]), s
        assert out["stats"] == seqs[s]["stats"]


def test_free_running_other_sensor_and_voxel_capacity():
    """a 64 x 2048 sensor at 100 m range and 12 points per voxel (the kernels' run-time-P instances, another block count per
    scan, another deskew table width): free-running batch == single runs"""
    S, n = 3, 6
    seqs = [synth.make_sequence(seed=1070 + s, n_scans=n, H=64, W=2048, max_range=100.0) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    kw = dict(max_range=100.0, min_range=1.0, use_imu_prediction=True, with_ekf=True, max_points_per_voxel=12, scan_cols=2048)
    b = core.BatchRunner(S, n, 64 * 2048, n_imu, **kw)
    assert b.free_running
    for s, sq in enumerate(seqs):
        r = core.SeqRunner(n, 64 * 2048, n_imu, gn_workgroups=32, gn_lanes_per_point=8, gn_threads=512, **kw)
        ends = [sq.imu_range_for_scan(k)[1] for k in range(n)]
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
            r.upload_scan(k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu], ends)
        r.upload_imu(sq.imu[:n_imu], ends)
        r.run()
        seqs[s] = r.results()
    b.run()
    for s in range(S):
        out = b.results(s)
        assert np.array_equal(out["kiss_poses"], seqs[s]["kiss_poses"]) and np.array_equal(out["res_poses"], seqs[s]["res_poses"]), s
        assert out["stats"] == seqs[s]["stats"]
        assert out["stats"][-1]["iterations"] > 0 and out["stats"][-1]["map_points"] > 0


def test_free_running_kernel_leaves_when_a_workgroup_never_arrives():
    """a workgroup of one team that skips the Gauss-Newton loop (test hook): the team's polls run out, the abort word goes
    up, every workgroup of that team leaves the persistent kernel and the host gets the time-out flag - no hang; the
    other sequence of the batch finishes its scans"""
    import ctypes as C
    import time
    from ptudes_lab_amd import _lib as L
    S, n = 2, 4
    seqs = [synth.make_sequence(seed=1050 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True)
    for s, sq in enumerate(seqs):
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
    b.run(2)
    h = C.c_void_p()
    L.check(L.lib().ptl_batch_icp(b._h, 1, C.byref(h)))
    L.check(L.lib().ptl_icp_debug_stall_workgroup(h, 3))
    t0 = time.perf_counter()
    b.enqueue(2)
    with pytest.raises(RuntimeError, match="sequence 1.*0x10"):
        b.wait()
    assert time.perf_counter() - t0 < 120.0
    assert all(st["iterations"] > 0 for st in b.results(0)["stats"][1:n])  # sequence 0 went through all four scans (the first one meets an empty map)


def test_free_running_equals_lockstep_over_a_longer_run():
    """24 sequences (three per XCD on its four teams: the scheduler hands sequences from team to team all the time), 60
    sweeps in launches of 25: every pose, every filter output and every per-scan counter equal to the lockstep run's"""
    S, n = 24, 60
    seqs = [synth.make_sequence(seed=1100 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    outs = {}
    for free in (True, False):
        b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, free_running=free,
                             scans_per_launch=25 if free else 0, gn_workgroups=256)
        for s, sq in enumerate(seqs):
            for k in range(n):
                b.upload_scan(s, k, sq.scan(k))
            b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
        b.run()
        outs[free] = [b.results(s) for s in range(S)]
        b.close()
    for s in range(S):
        a, c = outs[True][s], outs[False][s]
        assert np.array_equal(a["kiss_poses"], c["kiss_poses"]) and np.array_equal(a["res_poses"], c["res_poses"]), s
        assert np.array_equal(a["res_t"], c["res_t"]) and a["stats"] == c["stats"], s
    # (lockstep with 24 sequences runs four per XCD on 8 workgroups each - the team size of the free-running run)


def test_lockstep_serves_at_most_32_sequences():
    b = core.BatchRunner(33, 2, 1024, 2, with_ekf=True, max_points_per_scan=1024, scan_cols=64, free_running=False)
    for s in range(33):
        b.upload_imu(s, np.zeros((2, 7)), [1, 2])
    with pytest.raises(ValueError, match="lockstep"):
        b.run(1)


def test_free_running_needs_the_8_lane_kernel():
    with pytest.raises(ValueError):
        core.BatchRunner(2, 3, 1024, 4, with_ekf=True, max_points_per_scan=1024, scan_cols=64, gn_lanes_per_point=32, gn_threads=1024,
                         free_running=True)


def test_batch_rejects_missing_imu():
    b = core.BatchRunner(2, 3, 1024, 4, with_ekf=True, max_points_per_scan=1024, scan_cols=64)
    with pytest.raises(ValueError):
        b.upload_imu(0, np.zeros((4, 7)), [2, 2, 4])  # no IMU between scans 0 and 1


def test_batch_that_does_not_fit_the_device_is_refused_with_the_numbers():
    """ptl_batch_create adds up the batch's device memory before it allocates anything: 256 sequences with 8 M voxel
    blocks each (4 GB of blocks per sequence) cannot fit 288 GB - a capacity error that says how much is needed and how much is free"""
    with pytest.raises(RuntimeError, match=r"needs .* GB of device memory .* per sequence"):
        core.BatchRunner(256, 2, 131072, 2, with_ekf=True, map_block_capacity=8 * 1024 * 1024)


def test_batch_rejects_bad_workgroup_count():
    with pytest.raises(ValueError):
        core.BatchRunner(2, 3, 1024, 4, with_ekf=True, max_points_per_scan=1024, scan_cols=64, gn_workgroups=100)


def test_points_per_voxel_above_32_is_rejected_and_32_matches_the_oracle():
    """the search maps one lane of a 32-lane group to one stored point of a voxel: 33 is refused; at 32 a
    voxel's 32nd point is found"""
    from oracle import cpu as orc
    with pytest.raises(ValueError):
        core.Icp(70.0, 1.0, max_points_per_voxel=33)
    rng = np.random.default_rng(5)
    pts = rng.uniform(0.0, 0.69, (400, 3)) + np.array([7.0, 7.0, 0.7])  # one voxel, far more than 32 candidates
    src = pts[31] + np.array([1e-3, 0.0, 0.0])                          # nearest stored point is the 32nd
    icp = core.Icp(70.0, 1.0, max_points_per_voxel=32)
    icp.map_add(pts)
    m = orc.Map(0.7, 70.0, 32)
    m.add_points(pts)
    assert icp.map_size() == (m.num_voxels, m.num_points) == (1, 32)
    sums, nc, cand = icp.linear_system(src[None], 2.0, 0.5)
    ref = m.linear_system(src[None], 2.0, 0.5)
    assert nc == 1 and cand == 32
    assert np.allclose(sums, ref[0], rtol=1e-12, atol=1e-15)


def _load(b, seqs, n, n_imu, with_ekf=True):
    for s, sq in enumerate(seqs):
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, sq.imu[:n_imu] if with_ekf else np.zeros((0, 7)),
                     [sq.imu_range_for_scan(k)[1] if with_ekf else 0 for k in range(n)])


@pytest.mark.parametrize("team,S", [(4, 9), (2, 21), (1, 11)])
def test_small_teams_equal_independent_runs(team, S):
    """teams of 4 / 2 / 1 workgroups (the geometry of 96 - 256 sequences per GPU: 8 / 16 / 32 teams per XCD), more sequences than
    some XCDs have teams and fewer than others: every sequence equals its run alone with as many Gauss-Newton workgroups, bit for
    bit.  (A team of one is a workgroup on its own - no exchange partner, the filter step after the map update.)"""
    n = 5
    seqs = [synth.make_sequence(seed=1300 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, team_workgroups=team, scans_per_launch=3)
    assert b.team_geometry()[0] == team
    _load(b, seqs, n, n_imu)
    b.run()
    for s in (0, 1, S // 2, S - 1):
        sq = seqs[s]
        r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=team, gn_lanes_per_point=8, gn_threads=512)
        for k in range(n):
            r.upload_scan(k, sq.scan(k))
        r.upload_imu(sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
        r.run()
        one, out = r.results(), b.results(s)
        assert np.array_equal(out["kiss_poses"], one["kiss_poses"]) and np.array_equal(out["res_poses"], one["res_poses"]), s
        assert out["stats"] == one["stats"], s
    c = b.exec_counters(0)
    st = b.results(0)["stats"]
    assert c["scans"] == n - 1 and c["gn_iterations"] == sum(q["iterations"] for q in st)  # (the first scan meets an empty map: no loop)
    assert c["point_iterations"] == sum(q["iterations"] * q["n_src"] for q in st)
    assert c["searches"] >= sum(q["n_src"] for q in st[1:]) and c["rows_rebuilt"] >= sum(q["n_src"] for q in st[1:])  # every point searches in a scan's first iteration
    assert 0 < c["vds2_claims"] <= c["vds1_claims"] <= sum(q["n_valid"] for q in st)


@pytest.mark.parametrize("what,free", [("pool", True), ("pool", False), ("table", True), ("table", False)])
def test_capacity_exhaustion_is_reported_for_that_sequence_and_the_others_finish(what, free):
    """one sequence of a batch runs out of voxel blocks / of map-table room in the middle of a run (test hook): the wait names it
    with PTL_ERR_CAPACITY and the flag, nothing hangs, and the other sequences' results are those of their runs alone"""
    import ctypes as C
    from ptudes_lab_amd import _lib as L
    S, n = 10, 6
    seqs = [synth.make_sequence(seed=1400 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, free_running=free)
    _load(b, seqs, n, n_imu)
    b.run(3)
    h = C.c_void_p()
    L.check(L.lib().ptl_batch_icp(b._h, 4, C.byref(h)))
    L.check(L.lib().ptl_icp_debug_limit_capacity(h, 50 if what == "pool" else -1, -1 if what == "pool" else 1 << 30))
    b.enqueue(n - 3)
    with pytest.raises(RuntimeError, match=r"sequence 4.*0x%d" % (2 if what == "pool" else 4)):
        b.wait()
    for s in (0, 5, 9):
        sq = seqs[s]
        r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=16, gn_lanes_per_point=8, gn_threads=512)
        for k in range(n):
            r.upload_scan(k, sq.scan(k))
        r.upload_imu(sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
        r.run()
        one, out = r.results(), b.results(s)
        assert np.array_equal(out["res_poses"], one["res_poses"]) and out["stats"] == one["stats"], s
    bad = b.results(4)
    assert np.all(np.isfinite(bad["kiss_poses"])) and len(bad["stats"]) == n  # it went on (with a map that stopped growing)


def _oracle_run(sq, n, use_imu, with_ekf):
    from oracle import cpu as orc
    icp, ekf = orc.ICP(max_range=sq.max_range, min_range=sq.min_range), orc.EKF()
    t01 = sq.column_times()
    kiss, res = [], []
    for k in range(n):
        a, e = sq.imu_range_for_scan(k)
        if with_ekf:
            for i in range(a, e):
                ekf.process_imu(sq.imu[i, 1:4], sq.imu[i, 4:7], sq.imu[i, 0])
        pose = icp.register_frame(sq.scan(k).astype(np.float64), t01, ekf.pose_mat() if use_imu else None)
        if with_ekf:
            ekf.process_pose(pose)
        kiss.append(pose)
        res.append(ekf.pose_mat() if with_ekf else pose)
    return np.array(kiss), np.array(res), icp.stats


_INT_STATS = ("n_valid", "n_down", "n_src", "iterations", "n_corr_last", "sum_cand", "map_voxels", "map_points")


@pytest.mark.parametrize("S,team,n", [(48, 0, 100), (40, 2, 60)])
def test_free_running_batch_against_the_oracle_over_a_long_run(S, team, n):
    """the bench's kind of run inside the test suite: 48 sequences x 100 sweeps on teams of 8 (and 40 x 60 on teams of 2), ICP +
    IMU-EKF with the filter's pose as guess - three sequences other than sequence 0 against the CPU oracle: every pose within
    1e-9 m, every integer statistic of every sweep identical"""
    from oracle import cpu as orc
    seqs = [synth.make_sequence(seed=1500 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, team_workgroups=team)
    _load(b, seqs, n, n_imu)
    b.run()
    orc.set_threads(1)  # the sequential pass (the parity form of the oracle)
    for s in (7, S // 2 + 1, S - 1):
        kiss, res, stats = _oracle_run(seqs[s], n, True, True)
        out = b.results(s)
        assert np.abs(out["kiss_poses"][:, :3, 3] - kiss[:, :3, 3]).max() < 1e-9 and np.abs(out["res_poses"][:, :3, 3] - res[:, :3, 3]).max() < 1e-9, s
        assert np.abs(out["kiss_poses"] - kiss).max() < 1e-9, s
        for k in range(n):
            assert all(out["stats"][k][q] == stats[k][q] for q in _INT_STATS), (s, k, out["stats"][k], stats[k])


def test_icp_only_batch_against_the_oracle():
    """BASELINE config 2 through the batched runner: 48 sequences, no filter, constant-velocity guess, 40 sweeps; two of them
    against the oracle (poses 1e-9, integer statistics identical)"""
    from oracle import cpu as orc
    S, n = 48, 40
    seqs = [synth.make_sequence(seed=1600 + s, n_scans=n) for s in range(S)]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, 0, with_ekf=False)
    _load(b, seqs, n, 0, with_ekf=False)
    b.run()
    orc.set_threads(1)
    for s in (3, 29):
        kiss, _, stats = _oracle_run(seqs[s], n, False, False)
        out = b.results(s)
        assert np.abs(out["kiss_poses"] - kiss).max() < 1e-9, s
        for k in range(n):
            assert all(out["stats"][k][q] == stats[k][q] for q in _INT_STATS), (s, k)


def test_one_handle_both_drivers_and_several_team_sizes():
    """the driver and the team size chosen again for the same handle and the same uploaded sweeps (ptl_batch_reset): lockstep and
    the free-running kernel agree bit for bit; teams of 4 / 2 give what single runs with 4 / 2 workgroups give"""
    S, n = 5, 5
    seqs = [synth.make_sequence(seed=1700 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True)
    _load(b, seqs, n, n_imu)
    b.run()
    free = [b.results(s) for s in range(S)]
    b.set_driver(False)
    b.run()
    for s in range(S):
        lock = b.results(s)
        assert np.array_equal(lock["res_poses"], free[s]["res_poses"]) and lock["stats"] == free[s]["stats"], s
    for team in (4, 2):
        b.set_driver(True, team_workgroups=team)
        assert b.team_geometry()[0] == team
        b.run()
        sq = seqs[3]
        r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=team, gn_lanes_per_point=8, gn_threads=512)
        for k in range(n):
            r.upload_scan(k, sq.scan(k))
        r.upload_imu(sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
        r.run()
        assert np.array_equal(b.results(3)["res_poses"], r.results()["res_poses"]) and b.results(3)["stats"] == r.results()["stats"], team


def test_empty_and_nearly_empty_sweeps_inside_a_batch():
    """ragged inputs through the look-back compactions of the free-running kernel: one sequence of the batch gets a sweep without a
    single return, then a sweep with a handful of returns, then normal sweeps again (frame_downsample of 0 / a few entries: no block,
    one block for K3b and K4).  Every sequence equals its run alone (whose stages are separate launches with counting passes) bit for
    bit, and the disturbed one follows the oracle."""
    from oracle import cpu as orc
    S, n = 3, 7
    seqs = [synth.make_sequence(seed=1750 + s, n_scans=n) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    sweeps = [[sq.scan(k).copy() for k in range(n)] for sq in seqs]
    sweeps[1][2][:] = 0.0                      # no return at all
    keep = np.zeros(len(sweeps[1][3]), dtype=bool)
    keep[5000:5040] = True                     # forty returns of one beam
    sweeps[1][3][~keep] = 0.0
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, n_imu, use_imu_prediction=True, with_ekf=True, team_workgroups=2)
    singles = []
    for s, sq in enumerate(seqs):
        r = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=2, gn_lanes_per_point=8, gn_threads=512)
        ends = [sq.imu_range_for_scan(k)[1] for k in range(n)]
        for k in range(n):
            b.upload_scan(s, k, sweeps[s][k])
            r.upload_scan(k, sweeps[s][k])
        b.upload_imu(s, sq.imu[:n_imu], ends)
        r.upload_imu(sq.imu[:n_imu], ends)
        r.run()
        singles.append(r.results())
    b.run()
    assert b.status() == 0
    for s in range(S):
        out = b.results(s)
        assert np.array_equal(out["kiss_poses"], singles[s]["kiss_poses"]) and np.array_equal(out["res_poses"], singles[s]["res_poses"]), s
        assert out["stats"] == singles[s]["stats"], s
    st = b.results(1)["stats"]
    # (an empty source: one Gauss-Newton iteration without a pair, dx = 0, converged - as upstream's loop does)
    assert st[2]["n_valid"] == 0 and st[2]["n_down"] == 0 and st[2]["n_src"] == 0 and st[2]["iterations"] == 1 and st[2]["n_corr_last"] == 0
    assert 0 < st[3]["n_valid"] <= 40 and 0 < st[3]["n_src"] <= st[3]["n_down"] <= 40
    assert st[4]["n_src"] > 1000 and st[4]["iterations"] > 0
    # the disturbed sequence against the oracle (same sweeps, same IMU)
    orc.set_threads(1)
    icp, ekf = orc.ICP(max_range=seqs[1].max_range, min_range=seqs[1].min_range), orc.EKF()
    t01 = seqs[1].column_times()
    for k in range(n):
        a, e = seqs[1].imu_range_for_scan(k)
        for i in range(a, e):
            ekf.process_imu(seqs[1].imu[i, 1:4], seqs[1].imu[i, 4:7], seqs[1].imu[i, 0])
        pose = icp.register_frame(sweeps[1][k].astype(np.float64), t01, ekf.pose_mat())
        ekf.process_pose(pose)
        assert np.abs(b.results(1)["kiss_poses"][k] - pose).max() < 1e-9, k
        assert all(st[k][q] == icp.stats[k][q] for q in _INT_STATS), (k, st[k], icp.stats[k])



def _map_rows(b, s):
    """sorted points of sequence s's local map (KissICPWrapper.local_map_points of that member)"""
    import ctypes as C
    from ptudes_lab_amd import _lib as L
    h = C.c_void_p()
    L.check(L.lib().ptl_batch_icp(b._h, s, C.byref(h)))
    nv, npnt = C.c_int64(), C.c_int64()
    L.check(L.lib().ptl_icp_map_size(h, C.byref(nv), C.byref(npnt)))
    pts = np.empty((npnt.value + 8, 3))
    w = C.c_int64()
    L.check(L.lib().ptl_icp_map_points(h, L.dptr(pts), len(pts), C.byref(w)))
    p = pts[:w.value]
    return (nv.value, npnt.value), p[np.lexsort(p.T[::-1])]


def test_free_running_map_update_does_not_depend_on_its_points_per_thread():
    """VoxelHashMap::AddPoints + RemovePointsFarFromLocation of the free-running kernel (reference kiss.py:129) at BOTH instances the build
    carries - 8 points per thread (default) and 4 - and in the lockstep driver (one point per thread): the local map of every sequence
    after every launch, every pose and every per-scan counter bit-equal between the three; the map after the first sweep (identity pose:
    no Gauss-Newton sum in it) bit-equal to the CPU oracle's, after the last sweep the oracle's voxel and point counts and its points
    within 1e-9 m.  Rounds 3-4 failed this at 4 points per thread (a miscompiled branch chain in insert a): with the table
    rebuild inside the kernel every 3 scans and launches of 1 / 3 / 4 scans."""
    from oracle import cpu as orc
    S, n = 5, 8
    seqs = [synth.make_sequence(seed=1300 + s, n_scans=n) for s in range(S)]
    b = core.BatchRunner(S, n, seqs[0].H * seqs[0].W, 0, with_ekf=False, rebuild_every=3)
    for s, sq in enumerate(seqs):
        for k in range(n):
            b.upload_scan(s, k, sq.scan(k))
        b.upload_imu(s, np.zeros((0, 7)), [0] * n)
    assert b.debug_map_points_per_thread() == 8
    runs = {}
    for name in ("u8", "u4", "lockstep"):
        b.set_driver(name != "lockstep")
        if name != "lockstep":
            assert b.debug_map_points_per_thread(8 if name == "u8" else 4) == (8 if name == "u8" else 4)
        maps = []
        b.run(1)
        maps.append([_map_rows(b, s) for s in range(S)])
        for m in (3, 4):
            b.enqueue(m)
            b.wait()
            maps.append([_map_rows(b, s) for s in range(S)])
        runs[name] = (maps, [b.results(s) for s in range(S)])
    b.debug_map_points_per_thread(8)
    for name in ("u4", "lockstep"):
        for launch in range(3):
            for s in range(S):
                assert runs[name][0][launch][s][0] == runs["u8"][0][launch][s][0], (name, launch, s)
                assert np.array_equal(runs[name][0][launch][s][1], runs["u8"][0][launch][s][1]), (name, launch, s)
        for s in range(S):
            assert np.array_equal(runs[name][1][s]["kiss_poses"], runs["u8"][1][s]["kiss_poses"]), (name, s)
            assert runs[name][1][s]["stats"] == runs["u8"][1][s]["stats"], (name, s)
    # ... and against the oracle (sequences 0 and 3)
    orc.set_threads(1)
    for s in (0, 3):
        ref = orc.ICP(70.0, 1.0)
        t01 = seqs[s].column_times()
        for k in range(n):
            ref.register_frame(seqs[s].scan(k).astype(np.float64), t01)
            if k == 0:
                p = ref.map.points()
                assert np.array_equal(p[np.lexsort(p.T[::-1])], runs["u4"][0][0][s][1])  # bit for bit
                assert (ref.map.num_voxels, ref.map.num_points) == runs["u4"][0][0][s][0]
        p = ref.map.points()
        p = p[np.lexsort(p.T[::-1])]
        (nv, npnt), q = runs["u4"][0][2][s]
        assert (ref.map.num_voxels, ref.map.num_points) == (nv, npnt)
        # (sorted by coordinates that differ in the 14th digit: compare as sets through a rounded key)
        kp, kq = np.round(p, 6), np.round(q, 6)
        p, q = p[np.lexsort(kp.T[::-1])], q[np.lexsort(kq.T[::-1])]
        assert np.abs(p - q).max() <= 1e-9
        for k in range(n):
            for key in ("n_down", "n_src", "iterations", "map_voxels", "map_points"):
                assert ref.stats[k][key] == runs["u4"][1][s]["stats"][k][key], (s, k, key)


@pytest.mark.parametrize("case", ["sparse_map", "small_pool_runs_out"])
def test_two_block_classes_give_the_one_class_map(case):
    """map_small_blocks: voxels start in 128-byte blocks of 5 points and move to full blocks when a batch takes them past that (prune
    pass notes them, d_map_migrate moves them behind a barrier).  Same sweeps through a batch with ONE block class: every pose, every
    per-scan counter and the local map of every sequence after every launch bit-equal; counters against the CPU oracle.
    sparse_map: 64 x 2048 sweeps into 0.1 m voxels (BASELINE config 5's kind of map: most voxels never leave their small block);
    small_pool_runs_out: the default 0.7 m voxels (nearly every voxel outgrows five points) with so few small blocks that new
    voxels fall back to full blocks from the first scan on."""
    from oracle import cpu as orc
    if case == "sparse_map":
        S, n, H, W, mr, over = 3, 6, 64, 2048, 100.0, dict(voxel_size=0.1, scan_cols=2048, map_table_capacity=1 << 22)
        small, full = 600_000, 150_000
    else:
        S, n, H, W, mr, over = 5, 9, 128, 1024, 70.0, dict(rebuild_every=4, map_table_capacity=1 << 20)
        small, full = 3_000, 60_000
    seqs = [synth.make_sequence(seed=1400 + s, n_scans=n, H=H, W=W, max_range=mr) for s in range(S)]
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    runs = {}
    for name, kw in (("one", dict(map_block_capacity=small + full)), ("two", dict(map_block_capacity=full, map_small_blocks=small))):
        b = core.BatchRunner(S, n, H * W, n_imu, max_range=mr, min_range=1.0, use_imu_prediction=True, with_ekf=True, **over, **kw)
        _load(b, seqs, n, n_imu)
        maps = []
        b.run(2)
        maps.append([_map_rows(b, s) for s in range(S)])
        b.enqueue(n - 2)
        b.wait()
        maps.append([_map_rows(b, s) for s in range(S)])
        runs[name] = (maps, [b.results(s) for s in range(S)])
        b.close()
    for s in range(S):
        assert np.array_equal(runs["two"][1][s]["kiss_poses"], runs["one"][1][s]["kiss_poses"]), s
        assert np.array_equal(runs["two"][1][s]["res_poses"], runs["one"][1][s]["res_poses"]), s
        assert runs["two"][1][s]["stats"] == runs["one"][1][s]["stats"], s
        for launch in range(2):
            assert runs["two"][0][launch][s][0] == runs["one"][0][launch][s][0], (launch, s)
            assert np.array_equal(runs["two"][0][launch][s][1], runs["one"][0][launch][s][1]), (launch, s)
    assert runs["two"][1][0]["stats"][-1]["map_points"] > runs["two"][1][0]["stats"][-1]["map_voxels"] > 0  # (voxels hold more than one point each)
    orc.set_threads(4)
    kiss, res, stats = _oracle_run_cfg(seqs[1], n, mr, over.get("voxel_size"))
    out = runs["two"][1][1]
    assert np.abs(out["kiss_poses"] - kiss).max() < 1e-9
    for k in range(n):
        assert all(out["stats"][k][q] == stats[k][q] for q in _INT_STATS), (k, out["stats"][k], stats[k])
    orc.set_threads(1)
    # the one-class drivers refuse the option instead of mis-addressing blocks
    with pytest.raises((ValueError, RuntimeError)):
        core.SeqRunner(n, H * W, n_imu, max_range=mr, min_range=1.0, map_small_blocks=1000)
    with pytest.raises((ValueError, RuntimeError)):
        core.BatchRunner(2, n, H * W, n_imu, max_range=mr, min_range=1.0, free_running=False, map_small_blocks=1000)


def _oracle_run_cfg(sq, n, max_range, voxel_size):
    from oracle import cpu as orc
    over = {"voxel_size": voxel_size} if voxel_size else {}
    icp, ekf = orc.ICP(max_range=max_range, min_range=1.0, **over), orc.EKF()
    t01 = sq.column_times()
    kiss, res = [], []
    for k in range(n):
        a, e = sq.imu_range_for_scan(k)
        for i in range(a, e):
            ekf.process_imu(sq.imu[i, 1:4], sq.imu[i, 4:7], sq.imu[i, 0])
        pose = icp.register_frame(sq.scan(k).astype(np.float64), t01, ekf.pose_mat())
        ekf.process_pose(pose)
        kiss.append(pose)
        res.append(ekf.pose_mat())
    return np.array(kiss), np.array(res), icp.stats



def test_sweep_ring_streams_a_long_run_and_equals_the_resident_run():
    """ptl_seq_cfg.resident_scans (round 6): a ring of R sweep slots per sequence instead of every sweep of the run - a recording of any length at
    the full batch width (the reference walks a file scan by scan, data.py:31-77, cli/ekf_bench.py:493-563).  22 sweeps of 9 sequences through a ring
    of 8 in launches of 4, the sweeps of launch n + 1 uploaded WHILE launch n runs (between enqueue and wait): every pose, filter state and statistic
    bit-equal to the fully resident batch; the guards (order, a slot still in use, scans without their sweeps, the lockstep driver) refuse."""
    import pytest
    S, n, R, L = 9, 22, 8, 4
    seqs = [synth.make_sequence(seed=2300 + s, n_scans=n) for s in range(S)]
    pps = seqs[0].H * seqs[0].W
    n_imu = seqs[0].imu_range_for_scan(n - 1)[1]
    sweeps = [[sq.scan(k) for k in range(n)] for sq in seqs]
    kw = dict(use_imu_prediction=True, with_ekf=True, scans_per_launch=L)

    def imu(b):
        for s, sq in enumerate(seqs):
            b.upload_imu(s, sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])

    full = core.BatchRunner(S, n, pps, n_imu, **kw)
    for s in range(S):
        for k in range(n):
            full.upload_scan(s, k, sweeps[s][k])
    imu(full)
    full.run()
    want = [full.results(s) for s in range(S)]
    full.close()

    ring = core.BatchRunner(S, n, pps, n_imu, resident_scans=R, **kw)
    imu(ring)

    def feed(k0, k1):
        for s in range(S):
            for k in range(k0, min(k1, n)):
                ring.upload_scan(s, k, sweeps[s][k])

    with pytest.raises(RuntimeError, match="uploaded"):
        ring.run(L)  # nothing there yet
    with pytest.raises(RuntimeError, match="in order"):
        ring.upload_scan(0, 1, sweeps[0][1])
    feed(0, R)  # the ring is full ...
    with pytest.raises(RuntimeError, match="still holds"):
        ring.upload_scan(0, R, sweeps[0][R])  # ... and its first slot is not free before scan 0 is known to be done
    ring.run(L)  # cold start + scans [0, L), waits
    done = L
    while done < n:
        m = min(L, n - done)
        ring.enqueue(m)           # scans [done, done + m) on their way ...
        feed(done + R - L, done + R)  # ... while the slots of the scans BEFORE them (known to be done) take the next sweeps
        ring.wait()
        done += m
    assert ring.status() == 0
    for s in range(S):
        got = ring.results(s)
        assert np.array_equal(got["kiss_poses"], want[s]["kiss_poses"]) and np.array_equal(got["res_poses"], want[s]["res_poses"]), s
        assert np.array_equal(got["res_t"], want[s]["res_t"]) and got["stats"] == want[s]["stats"], s
    ring.close()
    with pytest.raises(ValueError, match="free-running"):
        core.BatchRunner(4, n, pps, n_imu, resident_scans=R, gn_lanes_per_point=32, **kw)


def test_uploads_from_a_page_locked_host_buffer_give_the_same_run():
    """ptl_host_pin / ptl_host_unpin: the caller's recording buffer page-locked once, the (synchronous) uploads then go by DMA straight from it.
    Same bytes on the device: the run equals the one fed from pageable memory."""
    S, n = 3, 4
    seqs = [synth.make_sequence(seed=2400 + s, n_scans=n) for s in range(S)]
    pps = seqs[0].H * seqs[0].W
    buf = np.ascontiguousarray(np.stack([np.stack([sq.scan(k) for k in range(n)]) for sq in seqs]))  # (S, n, pps, 3) f32
    outs = []
    for pinned in (False, True):
        if pinned:
            core.host_pin(buf)
        b = core.BatchRunner(S, n, pps, 0, with_ekf=False)
        for s in range(S):
            for k in range(n):
                b.upload_scan(s, k, buf[s, k])
            b.upload_imu(s, np.zeros((0, 7)), [0] * n)
        b.run()
        outs.append([b.results(s) for s in range(S)])
        b.close()
        if pinned:
            core.host_unpin(buf)
    for s in range(S):
        assert np.array_equal(outs[0][s]["kiss_poses"], outs[1][s]["kiss_poses"]) and outs[0][s]["stats"] == outs[1][s]["stats"]
    import pytest
    with pytest.raises((RuntimeError, ValueError)):
        core.host_unpin(buf)  # (not registered any more)
    # ... and that failure must not stay behind in the runtime's sticky "last error": the next, unrelated call that checks it after a kernel
    # launch - a handle's reset here - would report it as its own (it did: round 6, found by the suite's next test)
    core.SeqRunner(2, pps, 0, with_ekf=False).close()
