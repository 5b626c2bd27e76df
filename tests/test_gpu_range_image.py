"""range image -> xyz on device (XYZLut equivalent) + reduce_active_beams, against the
numpy restatement (oracle/lut.py) and the oracle pipeline."""
import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from oracle import cpu as orc
from oracle import lut as olut
from ptudes_lab_amd import core, synth
from ptudes_lab_amd import utils as pu

pytestmark = pytest.mark.gpu


def _sensor(H):
    rng = np.random.default_rng(9)
    alt = np.linspace(44.5, -44.5, H) + rng.normal(0, 0.02, H)
    az = np.tile([4.2, 1.4, -1.4, -4.2], H // 4) + rng.normal(0, 0.01, H)
    l2s = np.eye(4)
    l2s[:3, :3] = np.diag([-1.0, -1.0, 1.0])  # ouster lidar->sensor is a 180 deg yaw
    l2s[:3, 3] = [0.0, 0.0, 36.18]
    ext = np.eye(4)
    ext[:3, 3] = [0.014, -0.012, -0.015]
    return alt, az, 15.806, l2s, ext


def test_lut_matches_numpy_restatement():
    H, W = 128, 1024
    alt, az, n, l2s, ext = _sensor(H)
    rng = np.random.default_rng(1)
    r = rng.integers(0, 60000, H * W).astype(np.uint32)
    r[rng.random(H * W) < 0.1] = 0
    for e in (None, ext):
        lut = core.Lut(H, W, alt, az, n, l2s, e)
        d, o = olut.xyz_lut(H, W, alt, az, n, l2s, e)
        ref = olut.apply(d, o, r)
        got = lut(r)
        assert np.abs(got - ref).max() < 1e-11
        assert np.all(got[r == 0] == 0.0)


def test_register_range_matches_oracle_pipeline():
    seq = synth.make_sequence(seed=1003, n_scans=6)
    H, W = seq.H, seq.W
    alt = np.linspace(45.0, -45.0, H)
    az = np.zeros(H)
    lut = core.Lut(H, W, alt, az, 0.0, np.eye(4), None)
    d, o = olut.xyz_lut(H, W, alt, az, 0.0, np.eye(4), None)
    t01 = seq.column_times()
    icp = core.Icp(70.0, 1.0)
    ref = orc.ICP(70.0, 1.0)
    for k in range(5):
        x = seq.scan(k).reshape(H, W, 3)
        # ouster column v looks along 2 pi (1 - v / W): the synthetic column (W - v) % W
        rng_img = np.round(np.linalg.norm(x[:, (W - np.arange(W)) % W, :], axis=2) * 1000.0).astype(np.uint32)
        T = icp.register_range(lut, rng_img)
        xyz = olut.apply(d, o, rng_img.reshape(-1))
        sel = rng_img.reshape(-1) != 0  # reference kiss.py:59-61
        Tr = ref.register_frame(xyz[sel], t01[sel], None)
        assert np.abs(T - Tr).max() < 2e-4, k
        for key in ("n_valid", "n_down", "n_src"):
            assert icp.stats[-1][key] == ref.stats[-1][key]
    # reduce_active_beams: 32 of 128 rows
    icp.set_active_beams(H, 32)
    rng_img = np.round(np.linalg.norm(seq.scan(5).reshape(H, W, 3)[:, (W - np.arange(W)) % W, :], axis=2) * 1000.0).astype(np.uint32)
    T = icp.register_range(lut, rng_img)
    masked = rng_img.copy()
    pu.reduce_active_beams(masked, 32)
    xyz = olut.apply(d, o, masked.reshape(-1))
    sel = masked.reshape(-1) != 0
    Tr = ref.register_frame(xyz[sel], t01[sel], None)
    assert icp.stats[-1]["n_valid"] == ref.stats[-1]["n_valid"] <= 32 * W
    assert np.abs(T - Tr).max() < 2e-4


def test_wrapper_and_runner_take_range_images():
    """KissICPWrapper with beam geometry in the metadata and SeqRunner.upload_range == the xyz path"""
    from types import SimpleNamespace
    from ptudes_lab_amd.kiss import KissICPWrapper
    seq = synth.make_sequence(seed=1004, n_scans=5)
    H, W = seq.H, seq.W
    alt, az = np.linspace(45.0, -45.0, H), np.zeros(H)
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=W, pixels_per_column=H), beam_altitude_angles=alt,
                           beam_azimuth_angles=az, lidar_origin_to_beam_origin_mm=0.0,
                           lidar_to_sensor_transform=np.eye(4))
    w = KissICPWrapper(meta, _min_range=1.0, _max_range=70.0)
    r = core.SeqRunner(5, H * W, 0, max_range=70.0, min_range=1.0, with_ekf=False)
    r.set_lut(core.Lut(H, W, alt, az))
    imgs = []
    for k in range(5):
        x = seq.scan(k).reshape(H, W, 3)
        img = np.round(np.linalg.norm(x[:, (W - np.arange(W)) % W, :], axis=2) * 1000.0).astype(np.uint32)
        imgs.append(img)
        r.upload_range(k, img)
    r.upload_imu(np.zeros((0, 7)), [0] * 5)
    r.run()
    res = r.results()["kiss_poses"]
    for k in range(5):
        T = w.register_frame(SimpleNamespace(range=imgs[k], ts=float(k)))
        assert np.array_equal(T, res[k])
    assert len(w._sigmas) == 5


@pytest.mark.parametrize("free,compact", [(True, False), (False, False), (True, True), (False, True)])
def test_batch_takes_range_images(free, compact):
    """range images + reduce_active_beams through the batched runner (both drivers: the free-running kernel converts four
    pixels per thread) == the single-sequence runner on the same images.  compact: ptl_seq_cfg.range_input (round 6) - the sweeps
    stay u32 range images in HBM, 4 resident bytes per pixel instead of a 12-byte slot; xyz uploads are then refused."""
    S, n = 3, 5
    seqs = [synth.make_sequence(seed=1060 + s, n_scans=n) for s in range(S)]
    H, W = seqs[0].H, seqs[0].W
    alt, az = np.linspace(45.0, -45.0, H), np.zeros(H)
    lut = core.Lut(H, W, alt, az)
    b = core.BatchRunner(S, n, H * W, 0, with_ekf=False, free_running=free, range_input=compact)
    b.set_lut(lut, active_beams=64)
    if compact:
        with pytest.raises(RuntimeError, match="range images"):
            b.upload_scan(0, 0, seqs[0].scan(0))
    singles = []
    for s, sq in enumerate(seqs):
        r = core.SeqRunner(n, H * W, 0, max_range=70.0, min_range=1.0, with_ekf=False, gn_workgroups=32, gn_lanes_per_point=8, gn_threads=512)
        r.set_lut(lut, active_beams=64)
        for k in range(n):
            x = sq.scan(k).reshape(H, W, 3)
            img = np.round(np.linalg.norm(x[:, (W - np.arange(W)) % W, :], axis=2) * 1000.0).astype(np.uint32)
            b.upload_range(s, k, img)
            r.upload_range(k, img)
        b.upload_imu(s, np.zeros((0, 7)), [0] * n)
        r.upload_imu(np.zeros((0, 7)), [0] * n)
        r.run()
        singles.append(r.results())
    b.run()
    for s in range(S):
        out = b.results(s)
        assert np.array_equal(out["kiss_poses"], singles[s]["kiss_poses"]), s
        assert out["stats"] == singles[s]["stats"]
        assert 0 < out["stats"][-1]["n_valid"] <= 64 * W
