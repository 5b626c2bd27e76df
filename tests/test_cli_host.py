"""CLI / host pieces that need no GPU: sim_imu stream, `ekf-bench cmp`, library symbol surface."""
import ctypes
import os
import re

import numpy as np
from click.testing import CliRunner

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import _lib
from ptudes_lab_amd import utils as pu
from ptudes_lab_amd.cli import ekf_bench as eb
from ptudes_lab_amd.cli.run import ptudes_cli

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sim_imu_matches_reference_stream(golden_dir):
    g = np.load(os.path.join(golden_dir, "sim_imu.npz"))["rows"]
    np.random.seed(7)
    it = eb.sim_imu(freq=50.0, acc_noise_std=0.1, gyr_noise_std=0.05)
    rows = []
    for _ in range(len(g)):
        a, b = next(it)
        rows.append(np.concatenate([[a.ts], a.lacc, a.avel, [b.ts], b.lacc, b.avel]))
    assert np.array_equal(np.array(rows), g)


def test_cmp_command(golden_dir, tmp_path):
    gt = os.path.join(golden_dir, "poses_ncgt_hdr.csv")
    poses = pu.read_newer_college_gt(gt)
    # a second trajectory: same timestamps, perturbed poses
    rng = np.random.default_rng(0)
    t = [p[0] for p in poses]
    moved = []
    for _, T in poses:
        D = np.eye(4)
        D[:3, 3] = rng.normal(0, 0.05, 3)
        moved.append(T @ D)
    other = tmp_path / "run_a.csv"
    pu.save_poses_nc_gt_format(str(other), t, moved, header="x")
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "cmp", gt, str(other)])
    assert res.exit_code == 0, res.output
    assert "Traj poses comparisons GT v. run_a" in res.output
    m = re.search(r"ATE trans: ([0-9.]+) m", res.output)
    assert m and 0.0 < float(m.group(1)) < 0.05


def test_ouster_without_sdk_fails_loudly():
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "ouster", "nofile.pcap"])
    assert res.exit_code != 0 and "ouster-sdk" in res.output


def test_library_exports_every_declared_symbol():
    """libptudes_mi.so loads and exports every function include/ptudes_mi.h declares (no compute calls)."""
    hdr = open(os.path.join(ROOT, "include", "ptudes_mi.h")).read()
    declared = set(re.findall(r"\b(ptl_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = _lib.lib()
    assert lib.ptl_backend() in (0, 1)
    if lib.ptl_backend() == 0:  # no GPU here: constructors must fail loudly, never fall back to a CPU path
        import pytest
        from ptudes_lab_amd import core
        with pytest.raises(RuntimeError):
            core.Icp(70.0, 1.0)
        with pytest.raises(RuntimeError):
            core.Ekf()


def test_squared_gate_is_equivalent_to_the_square_root_test():
    """The GN kernel replaces `sqrt(d2) < M` (correspondence gate, convergence test) by `d2 < T` with
    T = min{x : sqrt(x) >= M} (icp_kernels.h sqrt_gate): the construction, restated with numpy, agrees with the square
    root for every double around the threshold - sqrt is correctly rounded and monotone."""
    rng = np.random.default_rng(5)

    def gate(M):
        T = np.float64(M) * np.float64(M)
        while np.sqrt(np.nextafter(T, 0.0)) >= M:
            T = np.nextafter(T, 0.0)
        while np.sqrt(T) < M:
            T = np.nextafter(T, np.inf)
        return T

    for M in list(rng.uniform(1e-6, 300.0, 200)) + [1e-4, 6.0, 3 * 2.0, 0.1, 2.1 * 3]:
        M = np.float64(M)
        T = gate(M)
        x = T
        for _ in range(40):  # 40 doubles below T ...
            x = np.nextafter(x, 0.0)
            assert np.sqrt(x) < M and x < T
        x = T
        for _ in range(40):  # ... and T and 39 above
            assert not (np.sqrt(x) < M) and not (x < T)
            x = np.nextafter(x, np.inf)


def test_ouster_gt_guess_needs_a_gt_file_and_plot_is_an_option():
    """reference cli/ekf_bench.py:337-349, :416-418: the option set a user's existing command line relies on"""
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "ouster", "nofile.pcap", "--use-gt-guess"])
    assert res.exit_code == 1 and "ERROR: --use-gt-guess requires the GT poses (--gt-file)" in res.output
    res = CliRunner().invoke(ptudes_cli, ["ekf-bench", "ouster", "--help"])
    assert res.exit_code == 0
    for opt in ("-p, --plot", "--use-gt-guess", "--use-imu-prediction", "-g, --gt-file", "--beams", "--save-nc-gt-poses"):
        assert opt in res.output, opt
