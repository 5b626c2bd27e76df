"""ROS bag sources of the pose path: reference src/ptudes/bag.py (`OusterRawBagSource`, `IMUBagSource`).

The reference reads bags through the third-party `rosbags` package (`AnyReader`), which is not installable offline.
What the path needs from a bag is small - connections, and the serialized messages of a few of them in time order -
so this module reads the ROS1 bag format 2.0 directly (`Ros1BagReader`, written from the published format
description: records of `<header_len><header><data_len><data>`, op codes 2 message / 3 bag header / 4 index /
5 chunk / 7 connection; chunk compression none / bz2, lz4 when the `lz4` module exists) and decodes the two message types the
reference consumes itself:

  sensor_msgs/Imu          header stamp + angular_velocity + linear_acceleration   (bag.py:138-148)
  ouster_ros/PacketMsg     `uint8[] buf`; an Ouster IMU packet is 48 bytes: sys_ts, accel_ts, gyro_ts (u64 ns),
                           accel xyz (f32, g), angular velocity xyz (f32, deg/s)    (bag.py:149-156, ins/data.py:18-31)

Lidar packets need ouster-sdk's `LidarPacket` (packet format tables) and are handed to it when it is importable.
Message type names are normalised the way rosbags does (`pkg/Type` -> `pkg/msg/Type`) so that the reference's
selection rules read the same here.  Parity: rosbags itself cannot run here, so the reader is pinned by round trips
through `tests/bagwriter.py` (format-level, unpinned against rosbags); the loop that consumes `IMUBagSource`
(`ekf-bench nc`) is pinned by vectors the reference produced (`tests/golden/ekf_nc_*.npz`).
"""
import bz2
import struct
import time
from pathlib import Path
from types import SimpleNamespace
from typing import Iterator, List, Optional, Union

import numpy as np

from .ins.data import IMU

# Ouster ROS PacketMsg MD5 sum (reference bag.py:19)
OUSTER_PACKETMSG_MD5 = "4f7b5949e76f86d01e96b0e33ba9b5e3"

_MAGIC = b"#ROSBAG V2.0\n"
_OP_MSG, _OP_BAG_HEADER, _OP_INDEX, _OP_CHUNK, _OP_CHUNK_INFO, _OP_CONNECTION = 2, 3, 4, 5, 6, 7


class BagError(RuntimeError):
    pass


def _norm_msgtype(t: str) -> str:
    """`sensor_msgs/Imu` -> `sensor_msgs/msg/Imu` (rosbags' naming, which the reference's rules are written in)"""
    parts = t.split("/")
    return t if len(parts) != 2 else f"{parts[0]}/msg/{parts[1]}"


def _parse_fields(buf: bytes) -> dict:
    """`<len:4><name>=<value>` repeated"""
    out, off, n = {}, 0, len(buf)
    while off < n:
        if off + 4 > n:
            raise BagError("truncated header field")
        (ln,) = struct.unpack_from("<I", buf, off)
        off += 4
        fld = buf[off:off + ln]
        if len(fld) != ln or b"=" not in fld:
            raise BagError("malformed header field")
        k, v = fld.split(b"=", 1)
        out[k.decode("ascii", "replace")] = v
        off += ln
    return out


def _records(buf: bytes, off: int = 0):
    """(header fields, data bytes, next offset) for every record in buf[off:]"""
    n = len(buf)
    while off < n:
        if off + 4 > n:
            raise BagError("truncated record")
        (hl,) = struct.unpack_from("<I", buf, off)
        hdr = _parse_fields(buf[off + 4:off + 4 + hl])
        off += 4 + hl
        if off + 4 > n:
            raise BagError("truncated record")
        (dl,) = struct.unpack_from("<I", buf, off)
        data = buf[off + 4:off + 4 + dl]
        if len(data) != dl:
            raise BagError("truncated record data")
        off += 4 + dl
        yield hdr, data, off


def _decompress(kind: bytes, data: bytes, size: int) -> bytes:
    if kind == b"none":
        return data
    if kind == b"bz2":
        return bz2.decompress(data)
    if kind == b"lz4":
        try:
            import lz4.frame
        except ImportError as e:
            raise BagError("lz4-compressed chunk and no `lz4` module in this environment") from e
        return lz4.frame.decompress(data)
    raise BagError(f"unknown chunk compression {kind!r}")


class Ros1BagReader:
    """Connections and time-ordered messages of one or several ROS1 bags (format 2.0).

    The interface is the part of rosbags' `AnyReader` the reference uses (bag.py:41-44,66-67,135-136):
    `.connections` (objects with id, topic, msgtype, digest), `.messages(connections=...)` yielding
    (connection, timestamp_ns, rawdata) ordered by timestamp, `.open()`, `.close()`.

    `open()` walks the top-level records once without touching chunk payloads (a chunk's index records list, per
    connection, the time and offset of every message in it), so a multi-gigabyte bag costs its index in memory, and
    `messages()` decompresses only the chunks that hold wanted messages, each once while it is in use."""

    _CACHE = 4  # decompressed chunks kept (messages are nearly chunk-sequential in time)

    def __init__(self, paths: List[Path]):
        self._paths = [Path(p) for p in paths]
        self.connections: List[SimpleNamespace] = []
        self._files = None
        self._chunks = []   # (file index, data position, data length, compression, uncompressed size)
        self._index = []    # (ts_ns, chunk index, offset in chunk, connection index), sorted
        self._cache = {}

    # -- open: connections + message index ---------------------------------------------------------------------
    def open(self) -> None:
        self._files = [open(p, "rb") for p in self._paths]
        for fi, f in enumerate(self._files):
            if f.read(len(_MAGIC)) != _MAGIC:
                raise BagError(f"{self._paths[fi]}: not a ROS1 bag (format 2.0)")
            fsize = self._paths[fi].stat().st_size
            local = {}      # connection id in this file -> index into self.connections
            pending = []    # (ts, chunk, offset, connection id in this file); connection records may come later
            indexed = set()  # chunks that were followed by index records
            last_chunk = None
            while True:
                head = f.read(4)
                if not head:
                    break
                if len(head) < 4:
                    raise BagError("truncated record")
                (hl,) = struct.unpack("<I", head)
                raw_hdr, raw_dl = f.read(hl), f.read(4)
                if len(raw_hdr) != hl or len(raw_dl) != 4:
                    raise BagError("truncated record")
                hdr = _parse_fields(raw_hdr)
                (dl,) = struct.unpack("<I", raw_dl)
                op = hdr.get("op", b"\xff")[0]
                if op == _OP_CHUNK:
                    (size,) = struct.unpack("<I", hdr["size"])
                    last_chunk = len(self._chunks)
                    self._chunks.append((fi, f.tell(), dl, hdr["compression"], size))
                    if f.tell() + dl > fsize:
                        raise BagError(f"{self._paths[fi]}: truncated chunk")
                    f.seek(dl, 1)
                    continue
                data = f.read(dl)
                if len(data) != dl:
                    raise BagError("truncated record data")
                if op == _OP_CONNECTION:
                    self._add_conn(hdr, data, local, fi)
                elif op == _OP_INDEX and last_chunk is not None:
                    (ver,) = struct.unpack("<I", hdr["ver"])
                    (cid,) = struct.unpack("<I", hdr["conn"])
                    (cnt,) = struct.unpack("<I", hdr["count"])
                    if ver != 1 or len(data) < 12 * cnt:
                        raise BagError("unsupported index record")
                    indexed.add(last_chunk)
                    for k in range(cnt):
                        sec, nsec, off = struct.unpack_from("<III", data, 12 * k)
                        pending.append((sec * 10**9 + nsec, last_chunk, off, cid))
                elif op == _OP_MSG:
                    raise BagError("message record outside a chunk (unchunked bags are not supported)")
            # chunks without index records (an unindexed / truncated bag): walk their payload
            for ci, ch in enumerate(self._chunks):
                if ch[0] == fi and ci not in indexed:
                    for h2, d2, off in self._chunk_records(ci):
                        op2 = h2.get("op", b"\xff")[0]
                        if op2 == _OP_CONNECTION:
                            self._add_conn(h2, d2, local, fi)
                        elif op2 == _OP_MSG:
                            (cid,) = struct.unpack("<I", h2["conn"])
                            sec, nsec = struct.unpack("<II", h2["time"])
                            pending.append((sec * 10**9 + nsec, ci, off, cid))
            for ts, ci, off, cid in pending:
                if cid not in local:
                    raise BagError("message on an undeclared connection")
                self._index.append((ts, ci, off, local[cid]))
        self._index.sort()  # by time; equal stamps in file order

    def _add_conn(self, hdr, data, local, fi):
        (cid,) = struct.unpack("<I", hdr["conn"])
        if cid in local:
            return
        info = _parse_fields(data)
        local[cid] = len(self.connections)
        self.connections.append(SimpleNamespace(
            id=len(self.connections), topic=hdr["topic"].decode(), digest=info.get("md5sum", b"").decode(),
            msgtype=_norm_msgtype(info.get("type", b"").decode()), path=str(self._paths[fi])))

    def _chunk_bytes(self, ci) -> bytes:
        buf = self._cache.get(ci)
        if buf is None:
            fi, pos, dl, comp, size = self._chunks[ci]
            f = self._files[fi]
            f.seek(pos)
            buf = _decompress(comp, f.read(dl), size)
            if len(self._cache) >= self._CACHE:
                self._cache.pop(next(iter(self._cache)))
            self._cache[ci] = buf
        return buf

    def _chunk_records(self, ci):
        """(header, data, offset of the record) for every record of chunk ci"""
        buf = self._chunk_bytes(ci)
        start = 0
        for hdr, data, nxt in _records(buf):
            yield hdr, data, start
            start = nxt

    # -- messages --------------------------------------------------------------------------------------------
    def messages(self, connections=None):
        if self._files is None:
            raise BagError("reader is not open")
        want = None if connections is None else {c.id for c in connections}
        for ts, chunk, off, ci_conn in self._index:
            if want is not None and ci_conn not in want:
                continue
            buf = self._chunk_bytes(chunk)
            hdr, data, _ = next(_records(buf, off))
            if hdr.get("op", b"\xff")[0] != _OP_MSG:
                raise BagError("index entry does not point at a message record")
            yield self.connections[ci_conn], ts, data

    def close(self) -> None:
        if self._files is not None:
            for f in self._files:
                f.close()
        self._files = None
        self._cache = {}


# ---------------------------------------------------------------------------------------------- message decoding
def decode_packet_msg(raw: bytes) -> bytes:
    """ouster_ros/PacketMsg: `uint8[] buf`"""
    (n,) = struct.unpack_from("<I", raw, 0)
    if 4 + n > len(raw):
        raise BagError("PacketMsg shorter than its length prefix")
    return raw[4:4 + n]


def decode_imu_msg(raw: bytes) -> IMU:
    """sensor_msgs/Imu (ROS1 serialization) -> IMU(lacc, avel, header stamp) as the reference builds it (bag.py:138-148)"""
    _seq, sec, nsec, flen = struct.unpack_from("<IIII", raw, 0)
    off = 16 + flen
    # orientation (4) + covariance (9), angular velocity (3) + covariance (9), linear acceleration (3) + covariance (9)
    v = struct.unpack_from("<37d", raw, off)
    avel = np.array(v[13:16])
    lacc = np.array(v[25:28])
    return IMU(lacc, avel, sec + nsec * 1e-9)


def decode_ouster_imu_packet(buf: bytes) -> SimpleNamespace:
    """48-byte Ouster IMU packet (the layout has not changed between packet formats, as the reference notes at
    bag.py:128-131): the fields `IMU.from_packet` reads"""
    if len(buf) < 48:
        raise BagError(f"IMU packet of {len(buf)} bytes (expected 48)")
    sys_ts, accel_ts, gyro_ts = struct.unpack_from("<QQQ", buf, 0)
    f = struct.unpack_from("<6f", buf, 24)
    return SimpleNamespace(sys_ts=sys_ts, accel_ts=accel_ts, gyro_ts=gyro_ts,
                           accel=np.array(f[:3], dtype=np.float64), angular_vel=np.array(f[3:], dtype=np.float64))


def _paths(data_path: Union[str, list]) -> List[Path]:
    return [Path(p) for p in data_path] if isinstance(data_path, list) else [Path(data_path)]


# ---------------------------------------------------------------------------------------------- sources
class IMUBagSource:
    """Read imu msgs from ROS bags (reference bag.py:96-156): `sensor_msgs/msg/Imu` topics or Ouster `imu_packets`."""

    def __init__(self, data_path: Union[str, list], imu_topic: Optional[str] = None, _reader=None):
        self._bag_reader = _reader if _reader is not None else Ros1BagReader(_paths(data_path))
        self._bag_reader.open()
        self._conns = []
        imu_conns = [
            c for c in self._bag_reader.connections
            if (c.msgtype == "sensor_msgs/msg/Imu" or (c.msgtype == "ouster_ros/msg/PacketMsg"
                                                       and c.topic.endswith("imu_packets")))
        ]
        assert len(imu_conns), "Expect any topic with msgtype: " \
            "sensor_msgs/msg/Imu or Ouster imu_packets types but found None"
        if imu_topic is not None:
            self._conns += [c for c in imu_conns if c.topic == imu_topic]
            assert len(self._conns), "Expect a topic with msgtype: " \
                f"sensor_msgs/msg/Imu and '{imu_topic}' name but found None"
        else:
            self._conns += [imu_conns[0]]

    def __iter__(self) -> Iterator[IMU]:
        for conn, _ts, rawdata in self._bag_reader.messages(connections=self._conns):
            if conn.msgtype == "sensor_msgs/msg/Imu":
                yield decode_imu_msg(rawdata)
            elif conn.msgtype == "ouster_ros/msg/PacketMsg":
                # the IMU's time is the packet's own sys_ts (ins/data.py:31); the bag time the reference passes
                # along as the host timestamp of the packet is not used downstream
                yield IMU.from_packet(decode_ouster_imu_packet(decode_packet_msg(rawdata)))

    def close(self) -> None:
        self._bag_reader.close()


class OusterRawBagSource:
    """Read an Ouster raw sensor packet stream from ROS bag(s) (reference bag.py:21-93).

    Yields ouster-sdk `LidarPacket` / `ImuPacket` objects, so it needs ouster-sdk (or a stand-in passed as `_sdk`
    with the same two constructors); topic selection, pacing (`rate`) and the md5 check follow the reference."""

    def __init__(self, data_path: Union[str, list], info, *, rate: float = 0.0, lidar_topic: str = "",
                 imu_topic: str = "", _sdk=None, _reader=None) -> None:
        if _sdk is None:
            try:
                import ouster.client as _sdk  # noqa: F811
            except ImportError as e:
                raise RuntimeError("OusterRawBagSource builds ouster-sdk packets: ouster-sdk is not installed "
                                   "(IMUBagSource and `ekf-bench nc` do not need it)") from e
        self._sdk = _sdk
        self._bag_reader = _reader if _reader is not None else Ros1BagReader(_paths(data_path))
        self._bag_reader.open()
        if not lidar_topic and not imu_topic:
            # Use any lidar/imu_packets topics if not set anything in ctor
            self._conns = [c for c in self._bag_reader.connections
                           if c.topic.endswith("lidar_packets") or c.topic.endswith("imu_packets")]
        else:
            topics = [t for t in [lidar_topic, imu_topic] if t]
            self._conns = [c for c in self._bag_reader.connections if c.topic in topics]
        self._metadata = info
        self._rate = rate

    def __iter__(self):
        real_start_ts = time.monotonic()
        bag_start_ts = None
        for conn, ts, rawdata in self._bag_reader.messages(connections=self._conns):
            msg_ts_sec = ts / 10**9
            if self._rate:
                if not bag_start_ts:
                    bag_start_ts = msg_ts_sec
                real_delta = time.monotonic() - real_start_ts
                bag_delta = (msg_ts_sec - bag_start_ts) / self._rate
                time.sleep(max(0, bag_delta - real_delta))
            if conn.digest != OUSTER_PACKETMSG_MD5:
                continue
            if conn.topic.endswith("lidar_packets"):
                yield self._sdk.LidarPacket(decode_packet_msg(rawdata), self._metadata, msg_ts_sec)
            elif conn.topic.endswith("imu_packets"):
                yield self._sdk.ImuPacket(decode_packet_msg(rawdata), self._metadata, msg_ts_sec)

    @property
    def topics(self) -> List[str]:
        return [c.topic for c in self._conns]

    @property
    def metadata(self):
        return self._metadata

    def close(self) -> None:
        self._bag_reader.close()
