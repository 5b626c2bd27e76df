"""Host-side pieces of bench.py and the data tooling that need no GPU: the workload key that ties a PMC summary to the
run it was collected on, the trajectory-row helpers of the gather, the controlled drives of the tracking experiments."""
import importlib.util
import json
import os
import types

import numpy as np

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import parallel, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _args(**over):
    a = dict(rows=128, cols=1024, min_range=1.0, max_range=70.0, voxel_size=0.0, seed_base=1000, warmup=20, steps=200,
             const_velocity=False, icp_only=False, gn_lanes=0, lockstep=False, gn_wgs=0, gn_threads=0, team_wgs=0)
    a.update(over)
    return types.SimpleNamespace(**a)


def test_roofline_traffic_only_from_a_pmc_pass_of_the_same_workload(tmp_path, monkeypatch):
    b = _bench()
    k16, k8 = b.workload_key(_args(), 16), b.workload_key(_args(), 8)
    assert k16 != k8 and b.workload_key(_args(voxel_size=0.1), 16) != k16
    assert b.workload_key(_args(gn_lanes=8), 1) != b.workload_key(_args(), 1)
    # the driver is part of the workload: counters of the lockstep Gauss-Newton launches say nothing about the free-running kernel
    lock16 = b.workload_key(_args(lockstep=True), 16)
    assert "_free" in k16 and "_free" not in lock16
    assert "_free" not in b.workload_key(_args(), 1) and "_free" not in b.workload_key(_args(gn_lanes=32), 16)
    # per-launch kernels: warm-up and step counts decide what the timed launches see.  The free-running kernel's summaries are per
    # scan: the same workload at another step count (the driver's --steps 20 --warmup 5) finds them - the team geometry must match
    assert b.workload_key(_args(lockstep=True, steps=60), 16) != lock16 and b.workload_key(_args(lockstep=True, warmup=5), 16) != lock16
    assert b.workload_key(_args(steps=20, warmup=5), 16) == k16
    assert b.workload_key(_args(team_wgs=4), 16) != k16 and b.workload_key(_args(gn_threads=256, gn_wgs=512), 16) != k16
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r09_x_pmc_hbm_traffic_a.json").write_text(json.dumps({"workload_key": lock16, "traffic_bytes_per_launch": 1.0e9}))
    (prof / "r09_x_pmc_hbm_traffic_b.json").write_text(json.dumps({"workload_key": k8, "traffic_bytes_per_launch": 9.6e11, "traffic_bytes_per_scan": 1.0e8}))
    (prof / "r01_n_pmc_hbm_traffic.json").write_text(json.dumps({"k_gn_loop_traffic_bytes_per_launch": 2.5e7}))  # round-1 form: no key
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    def got(p):
        return None if p is None else (p["bytes"], p["file"], p["stale"])

    assert got(b.pmc_traffic_for(lock16)) == (1.0e9, "r09_x_pmc_hbm_traffic_a.json", False)
    assert got(b.pmc_traffic_for(k8, scans_per_launch=160)) == (1.6e10, "r09_x_pmc_hbm_traffic_b.json", False)  # 8 sequences x 20 steps in one launch
    assert b.pmc_traffic_for(k16, scans_per_launch=320) is None  # another workload's counters are not this run's
    # ... and to the kernel sources it ran on: a pass of another build is reported as stale, a pass of this build wins
    (prof / "r09_y_pmc_hbm_traffic_c.json").write_text(json.dumps({"workload_key": k8, "code_id": "abc", "traffic_bytes_per_scan": 2.0e8}))
    assert got(b.pmc_traffic_for(k8, 160, code="abc")) == (3.2e10, "r09_y_pmc_hbm_traffic_c.json", False)
    assert b.pmc_traffic_for(k8, 160, code="zzz")["stale"] is True
    (prof / "r09_z_pmc_hbm_traffic_d.json").write_text(json.dumps({"workload_key": k8, "code_id": "old", "traffic_bytes_per_scan": 3.0e8}))
    assert got(b.pmc_traffic_for(k8, 160, code="abc")) == (3.2e10, "r09_y_pmc_hbm_traffic_c.json", False)
    assert len(b.code_id()) == 12 and b.code_id() == b.source_code_id()  # the in-tree library is built from the tree (csrc/Makefile CODE_ID)
    # a per-scan summary speaks for a run of OTHER warm-up / step counts only through the executed-byte ratio
    (prof / "r09_zz_pmc_hbm_traffic_e.json").write_text(json.dumps({"workload_key": k8, "code_id": "abc", "traffic_bytes_per_scan": 1.0e8,
                                                                       "warmup": 5, "steps": 20, "executed_bytes_per_scan": 8.0e7}))
    p = b.pmc_traffic_for(k8, 160, code="abc")
    assert p["file"] == "r09_zz_pmc_hbm_traffic_e.json" and (p["warmup"], p["steps"]) == (5, 20)
    assert b.pmc_for_this_run(p, 5, 20, 8.2e7) == ("same scans", 1.6e10)
    how, nbytes = b.pmc_for_this_run(p, 10, 200, 7.2e7)
    assert how == "scaled" and abs(nbytes - 1.6e10 * 0.9) < 1.0
    assert b.pmc_for_this_run(p, 10, 200, None) == (None, None)  # nothing to scale by: the line keeps the executed model
    assert b.pmc_for_this_run(dict(p, stale=True), 5, 20, 8.0e7) == (None, None)
    old = b.pmc_traffic_for(lock16)  # per-launch kernels: the key itself carries the counts
    assert b.pmc_for_this_run(old, 1, 2, None) == ("same scans", 1.0e9)


def test_executed_byte_model_follows_the_loaded_build():
    """bench.py's cost table must describe the library that is loaded - its build-dependent terms come from
    ptl_build_info (a host-only call: no GPU needed)"""
    from ptudes_lab_amd import core
    b = _bench()
    info = core.build_info()
    assert info["kcand"] >= 2 and info["ans_row_doubles"] >= 3 + 3 * info["kcand"] + 3 and info["tab_entry_bytes"] == 16  # (s0, K candidates, bound, ids, voxel key)
    cost, notes = b.exec_cost_for_build(info)
    assert cost["point_iteration_later"] == 8 * info["ans_row_doubles"] and cost["lds_points_per_workgroup"] == info["lds_points"]  # (the key travels in the row)
    other = dict(info, ans_row_doubles=14, kcand=3, lds_points=1536, diagnostics=1)
    cost2, notes2 = b.exec_cost_for_build(other)
    assert cost2["point_iteration_later"] == 112 and cost2["search"] == 128 + 8 + 112 + 4 and len(notes2) == 3


def test_committed_pmc_summaries_name_their_workload():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_*pmc_hbm_traffic*.json")))
    assert files
    for f in files:
        d = json.load(open(f))
        assert d.get("workload_key") and d.get("traffic_bytes_per_launch", 0) > 0 and d["hbm_bytes_per_launch"]["kernel"]


def test_scan_and_icp_byte_models_follow_survey_8d():
    b = _bench()
    st = dict(iterations=10, n_src=1000, sum_cand=50000, n_valid=100000, n_down=30000, map_voxels=20000)
    assert b.icp_bytes(st) == 10 * (12 + 27 * 16) * 1000 + 12 * 50000
    n_raw = 131072
    assert b.scan_bytes(st, n_raw) == (12 * n_raw + 12 * 100000) + ((12 + 16) * 100000 + 12 * 30000 + (12 + 16) * 30000 + 12 * 1000) \
        + b.icp_bytes(st) + ((12 + 16 + 12) * 30000 + (16 + 12) * 20000)


def test_executed_byte_model():
    """the as-executed model of the free-running kernel: unit costs x the kernel's counters + the per-scan statistics"""
    b = _bench()
    c = b.EXEC_COST
    cnt = dict(searches=5000, rows_rebuilt=10, map_points_read=5000, gn_iterations=7, vds1_claims=300, vds2_claims=40, point_iterations=7000, scans=1)
    st = [dict(iterations=7, n_src=1000, n_valid=100000, n_down=30000, map_voxels=20000)]
    tot, gn, stages = b.executed_bytes(cnt, st, 131072, 8, 1024)
    assert gn == 24 * 1000 + c["point_iteration_later"] * 6000 + c["search"] * 5000 - 136 * 1000 + c["row_rebuilt"] * 10 + 24 * 5000 + 7 * 8 * 9 * 36 * 8  # (the 1000 searches of the first iteration read no row)
    # a scan with more source points than the team's LDS holds (1 workgroup: 3072): the rest goes through src_cur every iteration
    big = [dict(st[0], n_src=4072)]
    _, gn_big, _ = b.executed_bytes(dict(cnt, point_iterations=7 * 4072), big, 131072, 1, 1024)
    assert c["point_iteration_later"] == 144 and gn_big == 24 * 4072 + 48 * 1000 * 7 + 144 * 6 * 4072 + c["search"] * 5000 - 136 * 4072 + c["row_rebuilt"] * 10 + 24 * 5000 + 7 * 1 * 2 * 36 * 8
    assert stages == c["raw_point"] * 131072 + c["valid_point"] * 100000 + c["down_point"] * 30000 + c["source_point"] * 1000 \
        + c["map_voxel"] * 20000 + 96 * 1024 + c["vds_claim"] * 340
    # (round 4: insert c is part of the prune pass, the stored count comes with the table entry; round 6: only a run's head keeps its voxel
    # slot, so the winner index is read per claim - not per valid point / frame_down entry)
    assert c["map_voxel"] == 36 and c["down_point"] == 236 and c["valid_point"] == 24 and c["vds_claim"] == 28
    assert tot == gn + stages
    # a first scan (empty map: no iteration) costs its stages only
    tot0, gn0, _ = b.executed_bytes(dict(cnt, searches=0, rows_rebuilt=0, map_points_read=0, gn_iterations=0, point_iterations=0),
                                    [dict(st[0], iterations=0)], 131072, 8, 1024)
    assert gn0 == 0 and tot0 == stages


def test_pose_rows_round_trip():
    rng = np.random.default_rng(2)
    from scipy.spatial.transform import Rotation
    T = np.tile(np.eye(4), (7, 1, 1))
    T[:, :3, :3] = Rotation.from_rotvec(rng.normal(0, 0.5, (7, 3))).as_matrix()
    T[:, :3, 3] = rng.normal(0, 10, (7, 3))
    t = 1000.0 + np.arange(7) * 0.1
    rows = parallel.poses_to_rows(t, T)
    assert rows.shape == (7, 8) and np.allclose(np.linalg.norm(rows[:, 4:8], axis=1), 1.0)
    t2, T2 = parallel.rows_to_poses(rows)
    assert np.array_equal(t2, t) and np.abs(T2 - T).max() < 1e-12
    assert parallel.poses_to_rows([], np.zeros((0, 4, 4))).shape == (0, 8)


def test_controlled_drive_is_what_it_says():
    seq = synth.make_path_sequence(n_scans=30, step_m=0.5, static_sweeps=3, ramp_sweeps=4, wobble_deg=1.0, heave_m=0.03)
    gt = seq.gt_poses(0.5)
    assert np.abs(gt[:3, :3, 3] - gt[0, :3, 3]).max() < 1e-12              # standing still for the first sweeps
    steps = np.linalg.norm(np.diff(gt[:, :3, 3], axis=0), axis=1)
    assert abs(steps[-1] - 0.5) < 5e-3 and np.all(np.diff(steps[2:8]) > -1e-9)  # ramps up to 0.5 m per sweep
    tilt = np.degrees(np.arccos(np.clip(gt[:, 2, 2], -1, 1)))
    assert tilt[:3].max() < 1e-9 and 0.2 < tilt[10:].max() < 2.0           # rocks once it moves
    x = seq.scan(12)
    assert x.shape == (128 * 1024, 3) and (np.linalg.norm(x, axis=1) > 0).mean() > 0.5
    # the lane around the path is clear of obstacles
    assert np.all(np.abs(seq.boxes[:, 1]) - seq.boxes[:, 4] > 3.9) and np.all(np.abs(seq.cyls[:, 1]) - seq.cyls[:, 2] > 3.9)


def test_pmc_summary_takes_the_timed_launches_of_the_dominant_kernel(tmp_path):
    """tools/pmc_summary.py: the free-running kernel has a warm-up launch and the timed one - the per-launch traffic is the
    timed launch's (the last `roofline.launches` dispatches), fetch side doubled (gfx950), write side as reported"""
    import subprocess
    import sys
    line = {"value": 1.0, "steps": 200, "config": {"workload_key": "K"}, "roofline": {"kernel": "kx_seq_run", "launches": 1, "scans_per_launch": 9600.0}}
    (tmp_path / "line.json").write_text("some library noise\n" + json.dumps(line) + "\n")
    for name, vals in (("FETCH_SIZE", (10.0, 100.0)), ("WRITE_SIZE", (4.0, 40.0))):
        d = tmp_path / name / "x"
        d.mkdir(parents=True)
        rows = ["Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value"]
        rows.append(f'1,"void k_other(int)",{name},7.0')
        # two rows per dispatch (a counter reported per dimension): summed; dispatch 2 = warm-up, 5 = timed
        rows += [f'2,"void kx_seq_run<20, 8>(SeqCtx const*, SeqRun)",{name},{vals[0] / 2}'] * 2
        rows += [f'5,"void kx_seq_run<20, 8>(SeqCtx const*, SeqRun)",{name},{vals[1] / 2}'] * 2
        (d / "f_counter_collection.csv").write_text("\n".join(rows) + "\n")
    out = tmp_path / "out.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), str(out), str(tmp_path / "line.json"),
                    f"FETCH_SIZE={tmp_path / 'FETCH_SIZE'}", f"WRITE_SIZE={tmp_path / 'WRITE_SIZE'}"], check=True, capture_output=True)
    r = json.load(open(out))
    assert r["workload_key"] == "K" and r["dominant_kernel"] == "kx_seq_run"
    assert r["FETCH_SIZE_timed"] == {"launches": 1, "mean_per_launch_KB": 100.0}
    assert r["traffic_bytes_per_launch"] == 2 * 100.0 * 1024 + 40.0 * 1024
    assert r["traffic_bytes_per_scan"] == (2 * 100.0 * 1024 + 40.0 * 1024) / 9600.0
