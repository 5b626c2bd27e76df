"""Test infrastructure: writes small ROS1 bags (format 2.0) for the reader in ptudes_lab_amd/bag.py.

Written from the published format description like the reader, but the other way round; it shares no code with it.
Layout: magic, bag header record padded to 4096 bytes, chunks (connection + message records, optionally bz2) each
followed by its index records, then the connection records again and one chunk-info record per chunk."""
import bz2
import struct


def _field(name: str, value: bytes) -> bytes:
    body = name.encode() + b"=" + value
    return struct.pack("<I", len(body)) + body


def _record(fields, data: bytes) -> bytes:
    hdr = b"".join(_field(k, v) for k, v in fields)
    return struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", len(data)) + data


def _time(ts_ns: int) -> bytes:
    return struct.pack("<II", ts_ns // 10**9, ts_ns % 10**9)


def imu_msg(seq, ts_ns, lacc, avel, frame_id="imu") -> bytes:
    """sensor_msgs/Imu, ROS1 serialization"""
    fid = frame_id.encode()
    out = struct.pack("<III", seq, ts_ns // 10**9, ts_ns % 10**9) + struct.pack("<I", len(fid)) + fid
    vals = [0.0, 0.0, 0.0, 1.0] + [0.0] * 9 + list(avel) + [0.0] * 9 + list(lacc) + [0.0] * 9
    return out + struct.pack("<37d", *vals)


def packet_msg(buf: bytes) -> bytes:
    """ouster_ros/PacketMsg: uint8[] buf"""
    return struct.pack("<I", len(buf)) + buf


def ouster_imu_packet(sys_ts, accel_ts, gyro_ts, accel_g, gyro_dps) -> bytes:
    return struct.pack("<QQQ", sys_ts, accel_ts, gyro_ts) + struct.pack("<6f", *accel_g, *gyro_dps)


def write_bag(path, connections, messages, *, chunk_msgs=8, compression="none", with_index=True):
    """connections: list of (topic, msgtype, md5sum); messages: list of (connection index, bag time ns, bytes) in the
    order they are written (bag order need not be time order)."""
    chunks = []  # (bytes of the chunk record + index records, per-connection counts, start, end)
    seen = set()
    body = bytearray()
    for a in range(0, len(messages), chunk_msgs):
        part = messages[a:a + chunk_msgs]
        payload = bytearray()
        index = {}
        for ci, ts, data in part:
            if ci not in seen:
                seen.add(ci)
                payload += _conn_record(ci, connections[ci])
            index.setdefault(ci, []).append((ts, len(payload)))
            payload += _record([("op", b"\x02"), ("conn", struct.pack("<I", ci)), ("time", _time(ts))], data)
        raw = bytes(payload)
        stored = bz2.compress(raw) if compression == "bz2" else raw
        rec = _record([("op", b"\x05"), ("compression", compression.encode()), ("size", struct.pack("<I", len(raw)))], stored)
        if with_index:
            for ci, ent in index.items():
                d = b"".join(_time(ts) + struct.pack("<I", off) for ts, off in ent)
                rec += _record([("op", b"\x04"), ("ver", struct.pack("<I", 1)), ("conn", struct.pack("<I", ci)),
                                ("count", struct.pack("<I", len(ent)))], d)
        chunks.append((len(body), {ci: len(e) for ci, e in index.items()}, min(t for _, t, _ in part), max(t for _, t, _ in part)))
        body += rec
    tail = bytearray()
    for ci in sorted(seen):
        tail += _conn_record(ci, connections[ci])
    for pos, counts, t0, t1 in chunks:
        d = b"".join(struct.pack("<II", ci, n) for ci, n in counts.items())
        tail += _record([("op", b"\x06"), ("ver", struct.pack("<I", 1)), ("chunk_pos", struct.pack("<Q", 13 + 4096 + pos)),
                         ("start_time", _time(t0)), ("end_time", _time(t1)), ("count", struct.pack("<I", len(counts)))], d)
    hdr_fields = [("op", b"\x03"), ("index_pos", struct.pack("<Q", 13 + 4096 + len(body))),
                  ("conn_count", struct.pack("<I", len(seen))), ("chunk_count", struct.pack("<I", len(chunks)))]
    hdr = b"".join(_field(k, v) for k, v in hdr_fields)
    pad = 4096 - 4 - len(hdr) - 4
    head = struct.pack("<I", len(hdr)) + hdr + struct.pack("<I", pad) + b" " * pad
    with open(path, "wb") as f:
        f.write(b"#ROSBAG V2.0\n" + head + bytes(body) + bytes(tail))


def _conn_record(ci, conn):
    topic, msgtype, md5 = conn
    data = (_field("topic", topic.encode()) + _field("type", msgtype.encode()) + _field("md5sum", md5.encode())
            + _field("message_definition", b"# test"))
    return _record([("op", b"\x07"), ("conn", struct.pack("<I", ci)), ("topic", topic.encode())], data)
