"""bench.py keeps the driver's contract: one JSON line with the agreed keys, the roofline and cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "24", "--warmup", "8",
                          "--cpu-budget", "3", "--seqs-per-gpu", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 8 and d["unit"] == "scans/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] <= 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "scans/s" and cb["sample"]
    assert d["parity_vs_oracle"]["max_dpos_m"] < 1e-9


@pytest.mark.gpu
def test_bench_launches_its_own_ranks_on_distinct_sequences(tmp_path):
    """the driver's plain `python bench.py --gpus 2`: two fresh ranks (sequences 1000 and 1001), no launcher.
    On a 1-GPU box both ranks share GPU 0 (64 Gauss-Newton workgroups each; the gather then goes over gloo because RCCL
    refuses two ranks on one device).  Each gathered trajectory equals the single-process run of that sequence."""
    import numpy as np
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import core, synth
    dump = str(tmp_path / "traj.npz")
    K, W = 10, 4
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(K), "--warmup", str(W),
                          "--gn-wgs", "64", "--seqs-per-gpu", "1", "--dump-traj", dump], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == K and d["cpu_baseline"] is None
    assert d["gathered_trajectories"]["sequences"] == 2 and d["gathered_trajectories"]["rows_each"] == [K + W]
    assert len(d["per_rank_scans_per_s"]["values"]) == 2
    assert abs(d["value"] - 2 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]  # one step = one sweep on each of the 2 ranks
    got = np.load(dump)
    n = K + W
    for r in range(2):
        sq = synth.make_sequence(seed=1000 + r, n_scans=n)
        n_imu = sq.imu_range_for_scan(n - 1)[1]
        run = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=64)
        for k in range(n):
            run.upload_scan(k, sq.scan(k))
        run.upload_imu(sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
        run.run()
        o = run.results()
        rows = got[f"rank{r}_seq0"]
        assert rows.shape == (n, 8)
        assert np.array_equal(rows[:, 0], o["res_t"]) and np.array_equal(rows[:, 1:4], o["res_poses"][:, :3, 3])
    assert not np.array_equal(got["rank0_seq0"][:, 1:4], got["rank1_seq0"][:, 1:4])


@pytest.mark.gpu
def test_two_ranks_with_free_running_batches_share_the_gpu(tmp_path):
    """two self-launched ranks, three sequences each (rank r owns the seeds 1000 + r + 2 j) in the free-running kernel: on the
    1-GPU box the two persistent launches are time-sliced on GPU 0 (their waits are bounded by executed polls, not by wall
    clock) - both finish, all six trajectories are gathered, and a sequence equals its single-process run"""
    import numpy as np
    import ptudes_lab_amd  # noqa: F401
    from ptudes_lab_amd import core, synth
    dump = str(tmp_path / "traj.npz")
    K, W = 8, 4
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(K), "--warmup", str(W),
                          "--seqs-per-gpu", "3", "--dump-traj", dump], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert d["n_gpus"] == 2 and d["config"]["driver"].startswith("free-running") and d["config"]["sequences_per_gpu"] == 3
    assert d["gathered_trajectories"]["sequences"] == 6 and d["gathered_trajectories"]["rows_each"] == [K + W]
    assert abs(d["value"] - 6 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    got = np.load(dump)
    seeds = {(int(r), int(j)): int(sd) for r, j, sd in got["seeds"]}
    assert seeds == {(0, 0): 1000, (0, 1): 1002, (0, 2): 1004, (1, 0): 1001, (1, 1): 1003, (1, 2): 1005}
    n = K + W
    sq = synth.make_sequence(seed=1003, n_scans=n)
    n_imu = sq.imu_range_for_scan(n - 1)[1]
    run = core.SeqRunner(n, sq.H * sq.W, n_imu, use_imu_prediction=True, with_ekf=True, gn_workgroups=32, gn_lanes_per_point=8, gn_threads=512)
    for k in range(n):
        run.upload_scan(k, sq.scan(k))
    run.upload_imu(sq.imu[:n_imu], [sq.imu_range_for_scan(k)[1] for k in range(n)])
    run.run()
    o = run.results()
    rows = got["rank1_seq1"]
    assert np.array_equal(rows[:, 0], o["res_t"]) and np.array_equal(rows[:, 1:4], o["res_poses"][:, :3, 3])


@pytest.mark.gpu
def test_bench_default_is_the_batched_runner():
    """the driver's command (`--gpus 1 --steps 20 --warmup 5`, here with a short CPU budget): 240 independent sequences on the GPU
    (seeds 1000..1239, 30 per XCD served by its 16 teams of 2 workgroups) in the free-running kernel (one persistent launch for
    the timed steps), `value` = 240 scans per step; the roofline figure is the kernel's executed bytes and a fraction of the
    peak, with the HBM traffic of a committed PMC pass of this workload beside it; sequence 0 and two more checked against
    the oracle inside the run"""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--cpu-budget", "3"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["sequences_per_gpu"] == 240 and d["config"]["sequence_seeds"].startswith("1000..1239")
    assert d["config"]["team_workgroups"] == 2 and d["config"]["teams"] == 128
    assert abs(d["value"] - 240 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    # three timed repeats, each one persistent launch for its 20 steps; the median is reported, all are listed,
    # and the repeats reproduce each other bit for bit
    rp = d["repeats"]
    assert rp["n"] == 3 and len(rp["values"]) == 3 and rp["reported"] == "median" and rp["bit_identical_trajectories"] is True
    assert sorted(rp["values"])[1] == pytest.approx(d["value"], rel=1e-9) and rp["spread_rel"] < 0.2
    assert r["kernel"] == "kx_seq_run" and r["launches"] == 3 and r["scans_per_launch"] == 240 * 20
    assert r["executed_model_build"]["differs_from_default_build"] == [] and len(d["config"]["code_id"]) == 12
    sc = d["config"]["scheduling"]
    assert sc["cross_xcd_handovers"] >= 0 and sc["scans_run_by_a_team_of_another_xcd"] >= 0
    # a roofline fraction is a fraction: HBM traffic of a counter pass of THIS build / launch time / peak when such a pass is committed
    # (else the executed-byte model, which always sits beside it as executed_frac); SURVEY 8(d)'s brute-force figure is there too
    assert 0 < r["frac"] <= 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["algorithmic_frac"] > r["executed_frac"]
    assert abs(r["executed_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 8e12 - r["executed_frac"]) < 1e-9
    assert r["frac"] == pytest.approx(r["executed_frac"] if r["traffic_stale"] else r["measured_frac"], rel=1e-12)
    assert r["frac_of_copy_rate"] == pytest.approx(r["frac"] * 8.0 / 6.29, rel=1e-12)
    sp = r["executed_split_per_scan"]
    assert abs(sp["gauss_newton"] + sp["stages"] - r["executed_bytes_per_scan"]) < 1e-3 and sp["stages"] > 5e6 and sp["gauss_newton"] > 5e6
    # the PMC pass committed for this workload (profiles/, bytes per scan) speaks for the driver's step count too
    assert r["traffic"] is not None and r["traffic_source"].startswith("r0") and 0 < r["measured_frac"] <= 1
    assert r["traffic_stale"] in (True, False)  # (true while the committed pass predates the current kernel sources)
    assert r["traffic"] > r["executed_bytes_per_launch"]  # HBM moves whole lines: more than the lanes asked for
    assert d["config"]["driver"].startswith("free-running") and "_free" in d["config"]["workload_key"]
    ph = d["sequence_phases_us_per_scan"]
    assert ph["slowest_sequence_total"] >= ph["mean_sequence_total"] > 0
    assert d["cpu_baseline"]["value"] > 0 and d["parity_vs_oracle"]["max_dpos_m"] < 1e-9
    ex = d["parity_vs_oracle"]["extra_sequences"]
    assert len(ex["sequences"]) == 8 and 0 not in ex["sequences"] and ex["max_dpos_m"] < 1e-9
    one = d["single_sequence"]  # k sequences per GPU and one - sequence 0 alone through the latency pipeline
    assert one["value"] > 0 and one["kernel"] == "k_gn_loop" and one["max_dpos_vs_batched_m"] < 1e-9


@pytest.mark.gpu
def test_bench_lockstep_driver_line():
    """--lockstep: the per-stage driver, the per-XCD Gauss-Newton launch dominant (one launch per step)"""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "6", "--seqs-per-gpu", "8",
                          "--lockstep", "--no-cpu-baseline", "--no-single-sequence", "--repeats", "1"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert d["roofline"]["kernel"] == "kx_gn_loop8" and d["roofline"]["launches"] == 12 and d["config"]["driver"].startswith("lockstep")
    assert "_free" not in d["config"]["workload_key"] and d["sequence_phases_us_per_scan"] is None
    assert 0 < d["roofline"]["frac"] <= 1 and d["roofline"]["executed_bytes_per_launch"] is None  # (algorithmic bytes: frac_kind says so)


@pytest.mark.gpu
def test_bench_under_the_launcher_gathers_over_rccl():
    """the driver's N > 1 command form with one rank: `python -m torch.distributed.run ... bench.py --gpus 1` - the rank
    takes the launcher's environment, the product library is loaded first (system HIP runtime), torch only carries the gloo
    control plane, and the final trajectory gather is the library's own ncclAllGather (ptl_batch_gather_trajectories: device rows
    in, host rows out; the communicator's id made on rank 0 and carried over gloo)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8",
                          "--warmup", "4", "--no-cpu-baseline"],  # the default 240 sequences: the gather's real tensor sizes
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["sequences_per_gpu"] == 240
    g = d["gathered_trajectories"]
    assert g["sequences"] == 240 and g["rows_each"] == [12] and g["backend"].startswith("RCCL ncclAllGather through libptudes_mi.so")
    assert d["repeats"]["n"] == 3 and d["roofline"]["launches"] == 3
    assert "rendering" in res.stderr and "timed region" in res.stderr  # the per-rank start-up times a slow many-rank start is read from


@pytest.mark.gpu
def test_bench_falls_back_to_host_rows_when_the_rccl_gather_raises():
    """first contact with a node whose RCCL does not come up must not void the measurement: the library's gather raises (here: PTL_RCCL_PATH
    names a library that does not exist), every rank falls back to host rows over the gloo control plane, the line says so, exit code 0"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6",
                          "--warmup", "3", "--seqs-per-gpu", "16", "--no-cpu-baseline", "--no-single-sequence", "--repeats", "1"],
                         capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, PTL_RCCL_PATH="/nonexistent/librccl.so.1"))
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")][0])
    g = d["gathered_trajectories"]
    assert g["sequences"] == 16 and g["rows_each"] == [9] and g["backend"].startswith("gloo, host rows - FALLBACK") and "librccl" in g["backend"]
    assert "trajectory gather (rccl) failed" in res.stderr


@pytest.mark.gpu
def test_bench_verify_all_compares_every_sequence_with_its_single_run():
    """`--verify-all`: after the batched run every sequence is registered once more ALONE (single-sequence runner with a team's
    workgroup count) and compared bit for bit - the whole batch, not a sample"""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--seqs-per-gpu", "20", "--team-wgs", "2", "--steps", "6", "--warmup", "3",
                          "--repeats", "2", "--verify-all", "--no-cpu-baseline", "--no-single-sequence"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-2000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][0])
    assert d["config"]["sequences_per_gpu"] == 20 and d["config"]["team_workgroups"] == 2
    assert d["verify_all"] == {"sequences": 20, "differ_from_their_single_runs": []}
    assert d["repeats"]["n"] == 2 and d["repeats"]["bit_identical_trajectories"] is True



@pytest.mark.gpu
def test_bench_range_input_keeps_range_images_resident_and_matches_the_oracle():
    """`--range-input` (SURVEY 8(f) rank 1 inside the bench): sweeps enter and STAY as u32 range images (ptl_seq_cfg.range_input: 4
    resident bytes per pixel), the LUT runs on the device, the oracle is fed the points the device LUT makes of the same images -
    parity as on the xyz path; `--ray-jitter-deg` is part of the workload (key and text)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "8", "--warmup", "4", "--seqs-per-gpu", "16", "--range-input",
                          "--repeats", "1", "--cpu-budget", "4", "--no-single-sequence"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert "_range" in d["config"]["workload_key"] and "range images" in d["config"]["workload"]
    assert d["parity_vs_oracle"]["scans"] >= 12 and d["parity_vs_oracle"]["max_dpos_m"] < 1e-9
    assert d["parity_vs_oracle"]["extra_sequences"]["max_dpos_m"] < 1e-9
    assert d["config"]["library_built_from_this_source_tree"] is True and len(d["config"]["code_id"]) == 12
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "3", "--seqs-per-gpu", "8", "--ray-jitter-deg", "0.3",
                          "--repeats", "1", "--cpu-budget", "3", "--no-single-sequence"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert "_jit0.3" in d["config"]["workload_key"] and "jittered" in d["config"]["workload"]
    assert d["parity_vs_oracle"]["max_dpos_m"] < 1e-9


@pytest.mark.gpu
def test_icp_only_ranks_gather_the_registrations_own_poses(tmp_path):
    """`--icp-only` multi-rank runs gather too: no filter, so the rows are the registration's own poses stamped with the
    scan index.  Two self-launched ranks share GPU 0 here (host rows over gloo); a one-rank batch under the launcher takes the library's
    path (ptl_batch_gather_trajectories -> k_kiss_rows -> ncclAllGather)."""
    import numpy as np
    dump = str(tmp_path / "traj.npz")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "3", "--icp-only",
                          "--gn-wgs", "64", "--seqs-per-gpu", "1", "--dump-traj", dump], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert d["gathered_trajectories"]["sequences"] == 2 and d["gathered_trajectories"]["rows_each"] == [9]
    got = np.load(dump)
    assert np.array_equal(got["rank0_seq0"][:, 0], np.arange(9.0)) and np.abs(got["rank0_seq0"][-1, 1:4]).max() > 0
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29617")
    dump1 = str(tmp_path / "traj1.npz")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "3", "--icp-only", "--seqs-per-gpu", "16",
                          "--repeats", "1", "--no-cpu-baseline", "--no-single-sequence", "--dump-traj", dump1], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    d = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    g = d["gathered_trajectories"]
    assert g["sequences"] == 16 and g["rows_each"] == [9] and "RCCL" in g["backend"] and "FALLBACK" not in g["backend"], g
    rows = np.load(dump1)["rank0_seq3"]
    assert np.array_equal(rows[:, 0], np.arange(9.0)) and abs(np.linalg.norm(rows[-1, 4:8]) - 1.0) < 1e-12
