#!/usr/bin/env python3
"""Golden-vector generator: runs the *reference* (bexcite/ptudes-lab @ /root/reference)
in THIS container and records inputs + expected outputs as small fixtures.

Run only where /root/reference exists (the build container):

    python tests/golden/gen_golden.py            # regenerates every fixture

The reference cannot travel to the GPU box, so the fixtures written next to this
script are what the tests consume.  Nothing here is imported by the product.

How the reference is imported: `ouster.*`, `rosbags.*` and
`kiss_icp.*` are absent from the image, so they are replaced by MagicMock modules;
the two SO(3) helpers the EKF takes from ouster-sdk (`exp_rot_vec`, `log_rot_mat`)
are replaced by their mathematical definition via scipy.  Python 3.10 is required
(the reference's ndarray dataclass defaults are rejected by >= 3.11).

Every case runs in a FRESH interpreter: the reference's `NavErrState` dataclass
shares one class-level ndarray per field between all instances
(ins/es_ekf.py:23-32), so a filter created after another filter's first update
would start from a polluted error state.
"""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"


def _mock_reference_deps():
    from unittest.mock import MagicMock
    import numpy as np
    from scipy.spatial.transform import Rotation as R
    sys.dont_write_bytecode = True
    for m in ["ouster", "ouster.client", "ouster.client._client", "ouster.viz",
              "ouster.viz.scans_accum", "ouster.pcap", "ouster.sdk", "ouster.sdk.util",
              "ouster.sdk.pose_util", "rosbags", "rosbags.highlevel", "kiss_icp",
              "kiss_icp.config", "kiss_icp.config.parser", "kiss_icp.registration",
              "kiss_icp.kiss_icp"]:
        mm = MagicMock(name=m)
        mm.__path__ = []
        sys.modules[m] = mm

    class PacketSource:  # bag.py subclasses it, must be a real class
        pass

    sys.modules["ouster.client"].PacketSource = PacketSource
    pu = sys.modules["ouster.sdk.pose_util"]
    pu.exp_rot_vec = lambda v: R.from_rotvec(np.asarray(v, float)).as_matrix()
    pu.log_rot_mat = lambda M: R.from_matrix(M).as_rotvec()
    sys.modules["ouster"].client = sys.modules["ouster.client"]
    sys.modules["ouster.sdk"].pose_util = pu
    sys.path.insert(0, REF_SRC)
    import matplotlib
    matplotlib.use("Agg")


def _nav_vec(nav):
    """19 doubles: pos(3) quat_xyzw(4) vel(3) bg(3) ba(3) grav(3)"""
    import numpy as np
    return np.concatenate([nav.pos, nav.att_q, nav.vel, nav.bias_gyr, nav.bias_acc,
                           nav.grav]).astype(np.float64)


def _rand_pose(rng, trans_scale=1.0, rot_scale=0.3):
    import numpy as np
    from scipy.spatial.transform import Rotation as R
    T = np.eye(4)
    T[:3, :3] = R.from_rotvec(rng.normal(0, rot_scale, 3)).as_matrix()
    T[:3, 3] = rng.normal(0, trans_scale, 3)
    return T


# ----------------------------------------------------------------------------- cases


def case_ekf_steps(variant):
    """Step-by-step ESEKF trace: seeded IMU stream with interleaved pose updates."""
    import numpy as np
    from scipy.spatial.transform import Rotation as R
    from ptudes.ins.es_ekf import ESEKF
    from ptudes.ins.data import IMU, GRAV

    rng = np.random.default_rng({"default": 11, "init": 12, "cov": 13}[variant])
    n_imu, upd_every = 300, 10
    kwargs = {}
    if variant == "init":
        kwargs = dict(init_grav=GRAV * np.array([0.02, -0.01, -0.9997]),
                      init_bacc=np.array([0.05, -0.02, 0.03]),
                      init_bgyr=np.array([0.002, -0.001, 0.0015]))
    ekf = ESEKF(**kwargs)

    # smooth-ish body motion: specific force + gyro, irregular dt
    ts = 100.0 + np.cumsum(rng.uniform(0.008, 0.012, n_imu))
    lacc = np.array([0, 0, GRAV]) + rng.normal(0, 0.8, (n_imu, 3))
    avel = rng.normal(0, 0.3, (n_imu, 3))

    navs = np.zeros((n_imu, 19))
    upd_idx, upd_pose, upd_cov = [], [], []
    navs_post = []
    covs_pre, covs_post, covs_imu = [], [], []
    for i in range(n_imu):
        ekf.processImu(IMU(lacc[i].copy(), avel[i].copy(), float(ts[i])))
        navs[i] = _nav_vec(ekf.nav)
        if i % 25 == 0:
            covs_imu.append(ekf._cov.copy())
        if i > 0 and i % upd_every == 0:
            # measured pose = current estimate perturbed (so residuals are realistic)
            T = ekf.nav.pose_mat()
            T[:3, 3] += rng.normal(0, 0.05, 3)
            T[:3, :3] = T[:3, :3] @ R.from_rotvec(rng.normal(0, 0.02, 3)).as_matrix()
            cov = None
            if variant == "cov":
                A = rng.normal(0, 0.03, (6, 6))
                cov = A @ A.T + np.diag([4e-4] * 3 + [1e-4] * 3)
            covs_pre.append(ekf._cov.copy())
            ekf.processPose(T.copy(), None if cov is None else cov.copy())
            covs_post.append(ekf._cov.copy())
            navs_post.append(_nav_vec(ekf.nav))
            upd_idx.append(i)
            upd_pose.append(T)
            upd_cov.append(np.zeros((6, 6)) if cov is None else cov)
    np.savez_compressed(
        os.path.join(HERE, f"ekf_steps_{variant}.npz"),
        imu_ts=ts, imu_ts_ns=ts_ns, imu_lacc=lacc, imu_avel=avel, nav_after_imu=navs,
        upd_idx=np.array(upd_idx), upd_pose=np.array(upd_pose),
        upd_cov=np.array(upd_cov), has_cov=np.array(variant == "cov"),
        nav_after_upd=np.array(navs_post), cov_pre=np.array(covs_pre),
        cov_post=np.array(covs_post), cov_imu_every25=np.array(covs_imu),
        init_grav=np.asarray(kwargs.get("init_grav", GRAV * np.array([0, 0, -1.0]))),
        init_bacc=np.asarray(kwargs.get("init_bacc", np.zeros(3))),
        init_bgyr=np.asarray(kwargs.get("init_bgyr", np.zeros(3))),
        cov0=ESEKF()._cov_init, final_ts=np.array(ekf.ts))


def case_ekf_sim():
    """`ptudes ekf-bench sim -t 2.0` with the legacy numpy RNG seeded to 0
    (cli/ekf_bench.py:107-179).  Records the IMU stream the CLI drew, the final
    states of both filters, update count and the printed ATE."""
    import numpy as np
    import ptudes.cli.ekf_bench as eb
    from ptudes.ins.data import ekf_traj_ate

    # capture the filters + imu stream through thin wrappers around the reference's own symbols
    made = []
    orig_esekf = eb.ESEKF

    def mk(*a, **k):
        f = orig_esekf(*a, **k)
        made.append(f)
        return f

    stream = []
    orig_sim = eb.sim_imu

    def sim(**k):
        for ideal, noisy in orig_sim(**k):
            stream.append((ideal.ts, ideal.lacc.copy(), ideal.avel.copy(),
                           noisy.lacc.copy(), noisy.avel.copy()))
            yield ideal, noisy

    eb.ESEKF = mk
    eb.sim_imu = sim
    from click.testing import CliRunner
    np.random.seed(0)
    res = CliRunner().invoke(eb.ptudes_ekf_sim, ["-t", "2.0"])
    assert res.exit_code == 0, res.output
    ekf_gt, ekf = made
    ate_r, ate_t = ekf_traj_ate(ekf_gt, ekf)
    np.savez_compressed(
        os.path.join(HERE, "ekf_sim.npz"),
        ts=np.array([s[0] for s in stream]),
        ideal_lacc=np.array([s[1] for s in stream]), ideal_avel=np.array([s[2] for s in stream]),
        noisy_lacc=np.array([s[3] for s in stream]), noisy_avel=np.array([s[4] for s in stream]),
        nav_gt=_nav_vec(ekf_gt.nav), nav=_nav_vec(ekf.nav),
        cov=ekf._cov, n_updates=np.array(len(ekf._nav_update_idxs)),
        upd_ts=np.array([ekf._navs_t[i] for i in ekf._nav_update_idxs]),
        ate_rot=np.array(ate_r), ate_trans=np.array(ate_t))
    with open(os.path.join(HERE, "ekf_sim_stdout.txt"), "w") as f:
        f.write(res.output)


def case_sim_imu():
    """First 40 draws of sim_imu with non-default args (cli/ekf_bench.py:44-79)."""
    import numpy as np
    import ptudes.cli.ekf_bench as eb
    np.random.seed(7)
    out = []
    for k, (a, b) in enumerate(eb.sim_imu(freq=50.0, acc_noise_std=0.1, gyr_noise_std=0.05)):
        out.append(np.concatenate([[a.ts], a.lacc, a.avel, [b.ts], b.lacc, b.avel]))
        if k == 39:
            break
    np.savez_compressed(os.path.join(HERE, "sim_imu.npz"), rows=np.array(out))


def case_pose_files():
    """KITTI / NC-GT writers + reader (utils.py:191-252) on fixed poses."""
    import numpy as np
    import ptudes.utils as pu
    rng = np.random.default_rng(21)
    poses = [_rand_pose(rng, 20.0, 1.0) for _ in range(6)]
    poses[0] = np.eye(4)
    t = [1626432215.25 + 0.1 * i + 1e-4 * rng.random() for i in range(6)]
    hdr = "data path: synthetic\nscans range: 0 - None\ntime: 20260101_000000"
    files = {}
    for name, h in (("hdr", hdr), ("nohdr", "")):
        fk = os.path.join(HERE, f"poses_kitti_{name}.txt")
        fn = os.path.join(HERE, f"poses_ncgt_{name}.csv")
        pu.save_poses_kitti_format(fk, poses, header=h)
        pu.save_poses_nc_gt_format(fn, t, poses, header=h)
        files[name] = (fk, fn)
    back = pu.read_newer_college_gt(files["hdr"][1])
    back_raw = pu.read_newer_college_gt(files["hdr"][1], to_os_imu=False)
    np.savez_compressed(
        os.path.join(HERE, "pose_files.npz"), poses=np.array(poses), t=np.array(t),
        header=np.array(hdr), read_t=np.array([b[0] for b in back]),
        read_poses=np.array([b[1] for b in back]),
        read_poses_raw=np.array([b[1] for b in back_raw]),
        nc_os_imu_to_base=pu.NC_OS_IMU_TO_BASE,
        vee_in=np.array([0.3, -1.2, 2.5]), vee_out=pu.vee(np.array([0.3, -1.2, 2.5])))


def case_ate():
    """calc_ate (ins/data.py:124-153) on fixed trajectory pairs."""
    import numpy as np
    from ptudes.ins.data import calc_ate
    rng = np.random.default_rng(31)
    out = {}
    for k, n in enumerate((1, 5, 40)):
        a = [_rand_pose(rng, 10.0, 0.8) for _ in range(n)]
        off = _rand_pose(rng, 3.0, 0.5)
        b = [off @ p @ _rand_pose(rng, 0.05, 0.01) for p in a]
        r, t = calc_ate(a, b)
        out[f"a{k}"] = np.array(a)
        out[f"b{k}"] = np.array(b)
        out[f"ate{k}"] = np.array([r, t])
    np.savez_compressed(os.path.join(HERE, "calc_ate.npz"), **out)


def case_ts_filters():
    """filter_nc_gt_by_close_ts / filter_nc_gt_by_cmp (utils.py:255-325)."""
    import numpy as np
    import ptudes.utils as pu
    rng = np.random.default_rng(41)
    out = {}
    for k in range(4):
        n_gt = 60 + 10 * k
        gt_t = 1000.0 + np.cumsum(rng.uniform(0.09, 0.11, n_gt))
        # compare stream: different rate, jitter, starts later / ends earlier
        start = gt_t[3 + k] + rng.uniform(-0.02, 0.02)
        cmp_t = start + np.cumsum(rng.uniform(0.045 * (k + 1), 0.055 * (k + 1), 80))
        cmp_t = cmp_t[cmp_t < gt_t[-2]]
        gt = [(float(t), _rand_pose(rng)) for t in gt_t]
        cmp_ = [(float(t), _rand_pose(rng)) for t in cmp_t]
        m_gt, m_t = pu.filter_nc_gt_by_close_ts(gt, [c[0] for c in cmp_])
        g2, c2 = pu.filter_nc_gt_by_cmp(gt, cmp_)
        out[f"gt_t{k}"] = gt_t
        out[f"gt_p{k}"] = np.array([g[1] for g in gt])
        out[f"cmp_t{k}"] = np.array([c[0] for c in cmp_])
        out[f"cmp_p{k}"] = np.array([c[1] for c in cmp_])
        out[f"m_gt_t{k}"] = np.array([g[0] for g in m_gt])
        out[f"m_t{k}"] = np.array(m_t)
        out[f"g2_t{k}"] = np.array([g[0] for g in g2])
        out[f"c2_t{k}"] = np.array([c[0] for c in c2])
        out[f"c2_p{k}"] = np.array([c[1] for c in c2])
    np.savez_compressed(os.path.join(HERE, "ts_filters.npz"), **out)


def case_imu_nav():
    """IMU.from_packet unit conversion (ins/data.py:18-31), NavState attitude
    accessors (ins/data.py:70-90), reduce_active_beams row choice (utils.py:328-341)."""
    import numpy as np
    from types import SimpleNamespace
    from ptudes.ins.data import IMU, NavState, GRAV
    from scipy.spatial.transform import Rotation as R
    rng = np.random.default_rng(51)
    pk = SimpleNamespace(sys_ts=1626432215123456789, accel=np.array([0.01, -0.02, 1.003]),
                         angular_vel=np.array([1.5, -2.25, 0.75]))
    rot = R.from_rotvec([0.1, 0.2, -0.3]).as_matrix()
    a = IMU.from_packet(pk)
    b = IMU.from_packet(pk, _intr_rot=rot)
    mats, quats, back, rotvecs = [], [], [], []
    for _ in range(8):
        M = R.from_rotvec(rng.normal(0, 1.2, 3)).as_matrix()
        ns = NavState()
        ns.att_h = M
        mats.append(M)
        quats.append(np.array(ns.att_q))
        back.append(ns.att_h)
        rotvecs.append(ns.att_v)
    ns = NavState()
    ns.pos = np.array([1.0, 2.0, 3.0])
    ns.att_v = np.array([0.3, -0.2, 0.1])
    beam = {f"beams_{h}_{n}": np.linspace(0, h, num=n, endpoint=False, dtype=int)
            for h, n in ((128, 32), (128, 48), (64, 16), (128, 100))}
    np.savez_compressed(
        os.path.join(HERE, "imu_nav.npz"), grav=np.array(GRAV),
        pk_sys_ts=np.array(pk.sys_ts), pk_accel=pk.accel, pk_gyro=pk.angular_vel, rot=rot,
        imu_a=np.concatenate([[a.ts], a.lacc, a.avel, [a.dt]]),
        imu_b=np.concatenate([[b.ts], b.lacc, b.avel, [b.dt]]),
        mats=np.array(mats), quats=np.array(quats), mats_back=np.array(back),
        rotvecs=np.array(rotvecs), pose_mat=ns.pose_mat(), **beam)


def case_stream_stats():
    """StreamStatsTracker (ins/data.py:207-369): running range mean / std over scans (with and without the
    `use_beams_num` row selection), IMU mean / std, time span, and the text `ptudes stat` prints."""
    import numpy as np
    import ouster.client as client
    from ptudes.ins.data import IMU, StreamStatsTracker
    rng = np.random.default_rng(77)
    H, W, T = 128, 96, 6
    scans = rng.integers(300, 60000, size=(T, H, W)).astype(np.uint32)
    scans[rng.random((T, H, W)) < 0.08] = 0            # missing returns
    scans[2, 5:9] = 0                                  # whole rows missing
    scan_ts_ns = (1_700_000_000_000_000_000 + np.arange(T) * 100_000_000 + 99_000_000).astype(np.int64)
    imu_ts = 1_700_000_000.0 + np.arange(10 * T) * 0.01
    imu_a = rng.normal([0.1, -0.2, 9.8], 0.05, (10 * T, 3))
    imu_w = rng.normal([0.01, 0.0, -0.02], 0.01, (10 * T, 3))

    class FakeScan:  # what trackScan touches: .h, .field(RANGE); last_valid_column_ts(ls) is mocked below
        def __init__(self, r, ts):
            self.h, self._r, self.ts = r.shape[0], r, ts

        def field(self, _):
            return self._r

    client.last_valid_column_ts = lambda ls: ls.ts
    out = {}
    for name, beams in (("all", None), ("b32", 32)):
        st = StreamStatsTracker(use_beams_num=beams, metadata=None)
        rows = []
        for k in range(T):
            for i in range(10 * k, 10 * (k + 1)):
                st.trackImu(IMU(imu_a[i].copy(), imu_w[i].copy(), float(imu_ts[i])))
            st.trackScan(FakeScan(scans[k], int(scan_ts_ns[k])))
            rows.append(np.concatenate([[st.range_mean, st.range_std, st._min_range, st._max_range, st._points_num,
                                         st._scans_num, st.dt], st.acc_mean, st.acc_std, st.gyr_mean, st.gyr_std]))
        out[f"rows_{name}"] = np.array(rows)
        with open(os.path.join(HERE, f"stream_stats_{name}.txt"), "w") as f:
            f.write(repr(st))
    np.savez_compressed(os.path.join(HERE, "stream_stats.npz"), scans=scans, scan_ts_ns=scan_ts_ns, imu_ts=imu_ts,
                        imu_a=imu_a, imu_w=imu_w, **out)


def case_packet_feed():
    """OusterLidarData.withScanIdx (data.py:31-77): event order of a packet stream - scans batched from lidar packets,
    IMU packets passed through, start_scan / end_scan, the trailing partial scan.  ouster-sdk is absent, so its types are
    replaced by minimal stand-ins here: packets carry (kind, frame_id, payload); the stand-in ScanBatcher completes a scan
    when the frame id changes, which is all the reference's loop relies on."""
    import numpy as np
    import ouster.client as client
    import ouster.client._client as _client
    from types import SimpleNamespace

    class LidarPacket:
        def __init__(self, frame, col0):
            self.frame, self.col0 = frame, col0

    class ImuPacket:
        def __init__(self, ts):
            self.sys_ts, self.accel, self.angular_vel = ts, np.array([0.0, 0.0, 1.0]), np.array([1.0, 2.0, 3.0])

    class LidarScan:
        def __init__(self, h, w, fields, cpp):
            self.h, self.w, self.frame_id, self.cols = h, w, -1, []

    class ScanBatcher:  # completes when a packet of the next frame arrives (that packet starts the next scan)
        def __init__(self, w, pf):
            self.cur = None

        def __call__(self, packet, ls):
            if self.cur is not None and packet.frame != self.cur:
                self.cur = packet.frame
                return True
            self.cur = packet.frame
            ls.frame_id = packet.frame
            ls.cols.append(packet.col0)
            return False

    client.LidarPacket, client.ImuPacket, client.LidarScan = LidarPacket, ImuPacket, LidarScan
    _client.ScanBatcher = ScanBatcher
    _client.PacketFormat = SimpleNamespace(from_info=lambda info: None)
    import importlib
    import ptudes.data as pdata
    importlib.reload(pdata)
    meta = SimpleNamespace(format=SimpleNamespace(columns_per_frame=64, pixels_per_column=8, columns_per_packet=16,
                                                   udp_profile_lidar="LEGACY"))
    packets = []
    ts = 1000
    for frame in range(6):
        for k in range(4):
            packets.append(LidarPacket(frame, 16 * k))
            if k % 2 == 1:
                ts += 5_000_000
                packets.append(ImuPacket(ts))
    packets = packets[:-3]  # the stream ends inside the last frame
    enc = [("L", p.frame, p.col0) if isinstance(p, LidarPacket) else ("I", p.sys_ts, 0) for p in packets]
    out = {}
    for name, kw in (("all", {}), ("from2", dict(start_scan=2)), ("from1to3", dict(start_scan=1, end_scan=3))):
        class Src:
            metadata = meta

            def __iter__(self):
                return iter(packets)

        src = Src()
        ev = []
        for idx, d in pdata.OusterLidarData(src).withScanIdx(**kw):
            ev.append([idx, 0, d.frame_id] if isinstance(d, LidarScan) else [idx, 1, int(round(d.ts * 1e9))])
        out[name] = ev
    with open(os.path.join(HERE, "packet_feed.json"), "w") as f:
        json.dump({"packets": enc, "events": out}, f)


def case_ekf_nc(variant):
    """`ptudes ekf-bench nc` (cli/ekf_bench.py:220-323): IMU stream from a bag + GT poses used as pose corrections.
    rosbags is absent, so `ptudes.bag.IMUBagSource` is replaced by a source yielding the reference's own `IMU` objects
    from a seeded stream; the GT file is written with the reference's NC-GT writer.  Records the stream, the GT file
    text, what the command printed and the filter pose after every correction.  One option set per process (App. C1)."""
    import numpy as np
    import ptudes.cli.ekf_bench as eb
    import ptudes.bag as pbag
    from ptudes.ins.data import IMU
    from ptudes.utils import save_poses_nc_gt_format
    from scipy.spatial.transform import Rotation as R
    from click.testing import CliRunner

    rng = np.random.default_rng(20240)
    n = 400
    t0 = 1583836591.0
    # stamps on the nanosecond grid, turned into seconds the way the reference does for sensor_msgs/Imu (bag.py:139)
    ts_ns = np.int64(t0) * 10**9 + 10_000_000 * np.arange(n, dtype=np.int64) + rng.integers(-200_000, 200_000, n)
    ts = np.array([int(v) // 10**9 + (int(v) % 10**9) * 1e-9 for v in ts_ns])
    grav = 9.782940329221166
    lacc = np.array([0.0, 0.0, grav]) + rng.normal(0, 0.3, (n, 3)) + 0.4 * np.sin(np.arange(n)[:, None] * 0.05 + np.array([0, 1, 2]))
    avel = rng.normal(0, 0.05, (n, 3)) + 0.1 * np.cos(np.arange(n)[:, None] * 0.03 + np.array([0.5, 1.5, 2.5]))
    # GT at ~10 Hz, starting before the IMU stream and ending before its end
    m = 36
    gt_t = t0 - 0.25 + 0.1 * np.arange(m) + rng.uniform(-3e-3, 3e-3, m)
    gt_poses = []
    for k in range(m):
        T = np.eye(4)
        T[:3, :3] = R.from_rotvec([0.01 * k, -0.004 * k, 0.02 * k]).as_matrix()
        T[:3, 3] = [0.05 * k, 0.02 * k * np.sin(0.3 * k), 0.01 * k]
        gt_poses.append(T)
    gt_path = os.path.join(HERE, "ekf_nc_gt.csv")
    save_poses_nc_gt_format(gt_path, t=list(gt_t), poses=gt_poses, header="ekf-bench nc golden")

    class Src:
        def __init__(self, data_path, imu_topic=None):
            self.imu_topic = imu_topic

        def __iter__(self):
            for i in range(n):
                yield IMU(lacc[i].copy(), avel[i].copy(), float(ts[i]))

    pbag.IMUBagSource = Src
    poses_log = []
    orig = eb.ESEKF

    class Rec(orig):
        def processPose(self, *a, **k):
            super().processPose(*a, **k)
            poses_log[-1].append((self.ts, self.nav.pose_mat().copy()))

    eb.ESEKF = Rec
    bag_path = os.path.join(HERE, "ekf_nc_gt.csv")  # any existing path satisfies click.Path(exists=True); never opened
    runs = {"default": ["-g", gt_path], "start": ["-g", gt_path, "--start-ts", "0.7", "-t", "1.5"],
            "alphasense": ["-g", gt_path, "-i", "/alphasense/imu", "-t", "2.0"]}
    args = runs[variant]
    poses_log.append([])
    res = CliRunner().invoke(eb.ptudes_ekf_nc, [bag_path] + args)
    assert res.exit_code == 0, (res.output, res.exception)
    np.savez_compressed(os.path.join(HERE, f"ekf_nc_{variant}.npz"), imu_ts=ts, imu_ts_ns=ts_ns, imu_lacc=lacc, imu_avel=avel, gt_t=gt_t,
                        gt_poses=np.array(gt_poses), upd_t=np.array([p[0] for p in poses_log[-1]]),
                        upd_pose=np.array([p[1] for p in poses_log[-1]]))
    with open(os.path.join(HERE, f"ekf_nc_{variant}.json"), "w") as f:
        json.dump({"args": args[2:], "stdout": res.output.replace(gt_path, "<GT>").replace(bag_path, "<BAG>")}, f, indent=1)


CASES = {
    "ekf_steps_default": lambda: case_ekf_steps("default"),
    "ekf_steps_init": lambda: case_ekf_steps("init"),
    "ekf_steps_cov": lambda: case_ekf_steps("cov"),
    "ekf_sim": case_ekf_sim,
    "sim_imu": case_sim_imu,
    "pose_files": case_pose_files,
    "ate": case_ate,
    "ts_filters": case_ts_filters,
    "imu_nav": case_imu_nav,
    "stream_stats": case_stream_stats,
    "packet_feed": case_packet_feed,
    "ekf_nc_default": lambda: case_ekf_nc("default"),
    "ekf_nc_start": lambda: case_ekf_nc("start"),
    "ekf_nc_alphasense": lambda: case_ekf_nc("alphasense"),
}

if __name__ == "__main__":
    if len(sys.argv) == 2:
        _mock_reference_deps()
        CASES[sys.argv[1]]()
        print("golden:", sys.argv[1], "ok")
    else:
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        for name in CASES:
            subprocess.run([sys.executable, os.path.abspath(__file__), name], check=True, env=env)
        import numpy, scipy
        with open(os.path.join(HERE, "MANIFEST.json"), "w") as f:
            json.dump({"reference": "bexcite/ptudes-lab v0.0.3 (/root/reference)",
                       "python": sys.version.split()[0], "numpy": numpy.__version__,
                       "scipy": scipy.__version__, "cases": sorted(CASES)}, f, indent=1)
