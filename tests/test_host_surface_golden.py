"""Host-side mirror of the reference's pose-file / metric / types surface against golden vectors produced
by the reference itself (tests/golden/gen_golden.py)."""
import os
from types import SimpleNamespace

import numpy as np

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import utils as pu
from ptudes_lab_amd.ins import data as idata


def test_pose_files_byte_identical(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "pose_files.npz"))
    poses, t, hdr = list(g["poses"]), list(g["t"]), str(g["header"])
    for name, h in (("hdr", hdr), ("nohdr", "")):
        fk, fn = tmp_path / f"k_{name}.txt", tmp_path / f"n_{name}.csv"
        pu.save_poses_kitti_format(str(fk), poses, header=h)
        pu.save_poses_nc_gt_format(str(fn), t, poses, header=h)
        assert fk.read_bytes() == open(os.path.join(golden_dir, f"poses_kitti_{name}.txt"), "rb").read()
        assert fn.read_bytes() == open(os.path.join(golden_dir, f"poses_ncgt_{name}.csv"), "rb").read()
    back = pu.read_newer_college_gt(os.path.join(golden_dir, "poses_ncgt_hdr.csv"))
    assert np.array_equal(np.array([b[0] for b in back]), g["read_t"])
    assert np.array_equal(np.array([b[1] for b in back]), g["read_poses"])
    raw = pu.read_newer_college_gt(os.path.join(golden_dir, "poses_ncgt_hdr.csv"), to_os_imu=False)
    assert np.array_equal(np.array([b[1] for b in raw]), g["read_poses_raw"])
    # save o read = identity up to the text precision of the file
    assert np.abs(np.array([b[1] for b in back]) - g["poses"]).max() < 1e-12
    assert np.array_equal(pu.NC_OS_IMU_TO_BASE, g["nc_os_imu_to_base"])
    assert np.array_equal(pu.vee(g["vee_in"]), g["vee_out"])


def test_calc_ate(golden_dir):
    g = np.load(os.path.join(golden_dir, "calc_ate.npz"))
    for k in range(3):
        r, t = idata.calc_ate(list(g[f"a{k}"]), list(g[f"b{k}"]))
        assert abs(r - g[f"ate{k}"][0]) <= 1e-12 * max(1.0, g[f"ate{k}"][0])
        assert abs(t - g[f"ate{k}"][1]) <= 1e-12 * max(1.0, g[f"ate{k}"][1])


def test_timestamp_filters(golden_dir):
    g = np.load(os.path.join(golden_dir, "ts_filters.npz"))
    for k in range(4):
        gt = [(float(t), p) for t, p in zip(g[f"gt_t{k}"], g[f"gt_p{k}"])]
        cmp_ = [(float(t), p) for t, p in zip(g[f"cmp_t{k}"], g[f"cmp_p{k}"])]
        m_gt, m_t = pu.filter_nc_gt_by_close_ts(gt, [c[0] for c in cmp_])
        assert np.array_equal(np.array([x[0] for x in m_gt]), g[f"m_gt_t{k}"])
        assert np.array_equal(np.array(m_t), g[f"m_t{k}"])
        g2, c2 = pu.filter_nc_gt_by_cmp(gt, cmp_)
        assert np.array_equal(np.array([x[0] for x in g2]), g[f"g2_t{k}"])
        assert np.array_equal(np.array([x[0] for x in c2]), g[f"c2_t{k}"])
        assert np.array_equal(np.array([x[1] for x in c2]), g[f"c2_p{k}"])
        assert len(m_t) > 10
    assert pu.filter_nc_gt_by_close_ts([], [1.0]) == []
    assert pu.filter_nc_gt_by_close_ts([(0.0, None), (1.0, None)], []) == []


def test_imu_and_navstate(golden_dir):
    g = np.load(os.path.join(golden_dir, "imu_nav.npz"))
    assert idata.GRAV == float(g["grav"])
    pk = SimpleNamespace(sys_ts=int(g["pk_sys_ts"]), accel=g["pk_accel"], angular_vel=g["pk_gyro"])
    a = idata.IMU.from_packet(pk)
    b = idata.IMU.from_packet(pk, _intr_rot=g["rot"])
    assert np.array_equal(np.concatenate([[a.ts], a.lacc, a.avel, [a.dt]]), g["imu_a"])
    assert np.array_equal(np.concatenate([[b.ts], b.lacc, b.avel, [b.dt]]), g["imu_b"])
    for M, q, Mb, rv in zip(g["mats"], g["quats"], g["mats_back"], g["rotvecs"]):
        ns = idata.NavState()
        ns.att_h = M
        assert np.array_equal(ns.att_q, q) and np.array_equal(ns.att_h, Mb) and np.array_equal(ns.att_v, rv)
    ns = idata.NavState()
    ns.pos = np.array([1.0, 2.0, 3.0])
    ns.att_v = np.array([0.3, -0.2, 0.1])
    assert np.array_equal(ns.pose_mat(), g["pose_mat"])
    # instances do not share state (the reference's dataclass defaults do)
    n1, n2 = idata.NavState(), idata.NavState()
    n1.pos += 1.0
    assert np.all(n2.pos == 0.0)
    for h, n in ((128, 32), (128, 48), (64, 16), (128, 100)):
        assert np.array_equal(pu.active_beam_rows(h, n), g[f"beams_{h}_{n}"])
    img = np.ones((128, 8), dtype=np.uint32)
    pu.reduce_active_beams(img, 32)
    assert img.sum() == 32 * 8 and np.all(img[g["beams_128_32"]] == 1)
