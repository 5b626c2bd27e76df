"""ROS1 bag reader + IMUBagSource (reference bag.py:96-156) on bags written by tests/bagwriter.py."""
import numpy as np
import pytest

import ptudes_lab_amd  # noqa: F401
from ptudes_lab_amd import bag
from ptudes_lab_amd.ins.data import GRAV

import bagwriter as bw

IMU_MD5 = "6a62c6daae103f4ff57a132d6f95cec2"


def _stream(n=50, seed=3):
    rng = np.random.default_rng(seed)
    ts = 1_583_836_591_000_000_000 + (np.arange(n) * 10_000_000) + rng.integers(-200_000, 200_000, n)
    return ts, rng.normal(0, 1, (n, 3)), rng.normal(0, 0.1, (n, 3))


@pytest.mark.parametrize("compression,with_index,chunk_msgs", [("none", True, 8), ("bz2", True, 5), ("none", False, 7), ("bz2", False, 64)])
def test_imu_msgs_round_trip_in_time_order(tmp_path, compression, with_index, chunk_msgs):
    ts, lacc, avel = _stream()
    conns = [("/alphasense/imu", "sensor_msgs/Imu", IMU_MD5), ("/os_node/lidar_packets", "ouster_ros/PacketMsg", bag.OUSTER_PACKETMSG_MD5)]
    msgs = []
    for i in range(len(ts)):
        # bag time lags the header stamp, and neighbours are swapped in the file: the reader orders by bag time
        msgs.append((0, int(ts[i]) + 1_000_000, bw.imu_msg(i, int(ts[i]), lacc[i], avel[i])))
        if i % 3 == 0:
            msgs.append((1, int(ts[i]) + 500, bw.packet_msg(bytes([i % 256]) * 100)))
    for a in range(0, len(msgs) - 1, 4):
        msgs[a], msgs[a + 1] = msgs[a + 1], msgs[a]
    p = tmp_path / "x.bag"
    bw.write_bag(p, conns, msgs, chunk_msgs=chunk_msgs, compression=compression, with_index=with_index)
    rd = bag.Ros1BagReader([p])
    rd.open()
    assert sorted((c.topic, c.msgtype) for c in rd.connections) == [("/alphasense/imu", "sensor_msgs/msg/Imu"),
                                                                    ("/os_node/lidar_packets", "ouster_ros/msg/PacketMsg")]
    stamps = [t for _, t, _ in rd.messages()]
    assert stamps == sorted(stamps) and len(stamps) == len(msgs)
    rd.close()
    out = list(bag.IMUBagSource(str(p)))
    assert len(out) == len(ts)
    for i, imu in enumerate(out):
        assert imu.ts == int(ts[i]) // 10**9 + (int(ts[i]) % 10**9) * 1e-9  # the reference's stamp arithmetic (bag.py:139)
        assert np.array_equal(imu.lacc, lacc[i]) and np.array_equal(imu.avel, avel[i])


def test_ouster_imu_packets_and_topic_selection(tmp_path):
    ts, lacc, avel = _stream(20)
    conns = [("/os_node/imu_packets", "ouster_ros/PacketMsg", bag.OUSTER_PACKETMSG_MD5), ("/alphasense/imu", "sensor_msgs/Imu", IMU_MD5)]
    msgs = []
    for i in range(len(ts)):
        pk = bw.ouster_imu_packet(int(ts[i]), int(ts[i]) + 10, int(ts[i]) + 20, np.float32(lacc[i]), np.float32(avel[i] * 100))
        msgs.append((0, int(ts[i]) + 2_000_000, bw.packet_msg(pk)))
        msgs.append((1, int(ts[i]) + 1_000_000, bw.imu_msg(i, int(ts[i]), lacc[i], avel[i])))
    p = tmp_path / "y.bag"
    bw.write_bag(p, conns, msgs, chunk_msgs=6, compression="bz2")
    # no topic: the first IMU-capable connection in the bag (bag.py:124-125)
    first = bag.IMUBagSource(str(p))
    assert [c.topic for c in first._conns] == ["/os_node/imu_packets"]
    got = list(first)
    assert len(got) == len(ts)
    for i, imu in enumerate(got):
        assert imu.ts == int(ts[i]) / 10**9
        assert np.array_equal(imu.lacc, GRAV * np.float64(np.float32(lacc[i])))
        assert np.array_equal(imu.avel, np.pi * np.float64(np.float32(avel[i] * 100)) / 180.0)
    named = list(bag.IMUBagSource([str(p)], imu_topic="/alphasense/imu"))
    assert np.array_equal(named[3].lacc, lacc[3])
    with pytest.raises(AssertionError, match="'/nope' name but found None"):
        bag.IMUBagSource(str(p), imu_topic="/nope")


def test_bag_without_imu_and_broken_files(tmp_path):
    p = tmp_path / "z.bag"
    bw.write_bag(p, [("/os_node/lidar_packets", "ouster_ros/PacketMsg", bag.OUSTER_PACKETMSG_MD5)],
                 [(0, 10**9 + k, bw.packet_msg(b"ab")) for k in range(5)])
    with pytest.raises(AssertionError, match="found None"):
        bag.IMUBagSource(str(p))
    q = tmp_path / "notabag.bag"
    q.write_bytes(b"hello")
    with pytest.raises(bag.BagError, match="not a ROS1 bag"):
        bag.IMUBagSource(str(q))
    data = p.read_bytes()
    (tmp_path / "cut.bag").write_bytes(data[: 13 + 4096 + 90])  # inside the first chunk
    with pytest.raises(bag.BagError):
        bag.Ros1BagReader([tmp_path / "cut.bag"]).open()


def test_raw_packet_source_with_stand_in_sdk(tmp_path):
    """OusterRawBagSource (bag.py:21-93): topic filter, md5 check, packet construction order"""
    class Sdk:
        class LidarPacket:
            def __init__(self, buf, info, ts): self.kind, self.buf, self.ts = "L", bytes(buf), ts
        class ImuPacket:
            def __init__(self, buf, info, ts): self.kind, self.buf, self.ts = "I", bytes(buf), ts
    conns = [("/os_node/lidar_packets", "ouster_ros/PacketMsg", bag.OUSTER_PACKETMSG_MD5),
             ("/os_node/imu_packets", "ouster_ros/PacketMsg", bag.OUSTER_PACKETMSG_MD5),
             ("/other/lidar_packets", "ouster_ros/PacketMsg", "0" * 32), ("/camera", "sensor_msgs/Image", "1" * 32)]
    msgs = [(k % 4, 10**9 + 1000 * k, bw.packet_msg(bytes([k]) * 4)) for k in range(16)]
    p = tmp_path / "raw.bag"
    bw.write_bag(p, conns, msgs, chunk_msgs=5)
    src = bag.OusterRawBagSource(str(p), info="META", _sdk=Sdk)
    assert sorted(src.topics) == ["/os_node/imu_packets", "/os_node/lidar_packets", "/other/lidar_packets"]
    got = [(x.kind, x.buf[0], x.ts) for x in src]
    assert got == [("L" if k % 4 == 0 else "I", k, (10**9 + 1000 * k) / 10**9) for k in range(16) if k % 4 in (0, 1)]
    only = bag.OusterRawBagSource(str(p), info="META", lidar_topic="/os_node/lidar_packets", _sdk=Sdk)
    assert [x.buf[0] for x in only] == [0, 4, 8, 12] and only.metadata == "META"


def test_several_bags_merge_by_time_and_unknown_compression_is_reported(tmp_path):
    conns = [("/alphasense/imu", "sensor_msgs/Imu", IMU_MD5)]
    ts, lacc, avel = _stream(30, seed=9)
    # even samples in one file, odd ones in the other: the source yields them interleaved by bag time
    for part in (0, 1):
        msgs = [(0, int(ts[i]), bw.imu_msg(i, int(ts[i]), lacc[i], avel[i])) for i in range(part, len(ts), 2)]
        bw.write_bag(tmp_path / f"p{part}.bag", conns, msgs, chunk_msgs=4, compression="bz2" if part else "none")
    got = list(bag.IMUBagSource([str(tmp_path / "p1.bag"), str(tmp_path / "p0.bag")], imu_topic="/alphasense/imu"))
    assert [g.lacc[0] for g in got] == [lacc[i][0] for i in range(len(ts))]
    # a chunk compressed with something this reader does not know
    data = (tmp_path / "p0.bag").read_bytes().replace(b"compression=none", b"compression=zstd")
    (tmp_path / "odd.bag").write_bytes(data)
    with pytest.raises(bag.BagError, match="unknown chunk compression"):
        list(bag.IMUBagSource(str(tmp_path / "odd.bag")))


def test_packet_sources_need_ouster_sdk_and_say_so(tmp_path):
    from ptudes_lab_amd import utils as pu
    p = tmp_path / "raw.bag"
    bw.write_bag(p, [("/os_node/lidar_packets", "ouster_ros/PacketMsg", bag.OUSTER_PACKETMSG_MD5)],
                 [(0, 10**9 + k, bw.packet_msg(b"ab")) for k in range(3)])
    try:
        import ouster.client  # noqa: F401
        pytest.skip("ouster-sdk is installed here")
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match="ouster-sdk is not installed"):
        pu.read_packet_source(str(p), meta=None)
    with pytest.raises(RuntimeError, match="ouster-sdk is not installed"):
        pu.read_packet_source(str(tmp_path), meta=None)  # a directory of bags
    assert pu.read_packet_source(str(tmp_path / "missing.xyz")) is None  # like the reference: anything else yields None
