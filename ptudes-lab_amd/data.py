"""Lidar data source: scans + IMUs with the scan index, from a stream of Ouster packets.

Host-side mirror of reference src/ptudes/data.py:12-92 (`OusterLidarData`): same constructor, `withScanIdx`
(start_scan / end_scan semantics, IMUs before `start_scan` dropped, the trailing partial scan yielded when the stream
ends), iteration, `close`, `metadata`.  Packet parsing and scan batching are ouster-sdk's (third party, absent from this
image): its types are imported when the class is used, or injected through `_sdk` (what the tests do with stand-ins;
tests/golden/packet_feed.json holds the event order the reference itself produces on the same packets)."""
from types import SimpleNamespace
from typing import Iterator, Optional, Tuple

from .ins.data import IMU


def _ouster_sdk():
    try:
        import ouster.client as client
        import ouster.client._client as _client
    except Exception as e:  # pragma: no cover - ouster-sdk is not installed here
        raise RuntimeError("reading Ouster packets needs ouster-sdk (>= 0.10), which is not installed") from e
    return SimpleNamespace(LidarPacket=client.LidarPacket, ImuPacket=client.ImuPacket, LidarScan=client.LidarScan,
                           PacketFormat=_client.PacketFormat, ScanBatcher=_client.ScanBatcher)


class OusterLidarData:
    """Lidar data source: LidarScan + IMUs iterator with scan index"""

    def __init__(self, source, *, fields=None, _sdk=None) -> None:
        self._source = source
        self._fields = fields if fields is not None else self._source.metadata.format.udp_profile_lidar
        self._sdk = _sdk
        self._scan_idx = 0

    def withScanIdx(self, *, start_scan: int = 0, end_scan: Optional[int] = None) -> Iterator[Tuple[int, object]]:
        """Make an iterator with (scanIdx, scan/imu)  (reference data.py:31-77)"""
        sdk = self._sdk or _ouster_sdk()
        fmt = self._source.metadata.format
        w, h, columns_per_packet = fmt.columns_per_frame, fmt.pixels_per_column, fmt.columns_per_packet
        batch = sdk.ScanBatcher(w, sdk.PacketFormat.from_info(self._source.metadata))
        ls_write = None
        scan_idx = 0
        for packet in self._source:
            if isinstance(packet, sdk.LidarPacket):
                if ls_write is None:
                    ls_write = sdk.LidarScan(h, w, self._fields, columns_per_packet)
                if batch(packet, ls_write):  # finished frame
                    if scan_idx >= start_scan:
                        yield scan_idx, ls_write
                    scan_idx += 1
                    if end_scan is not None and scan_idx > end_scan:
                        return
                    ls_write = None
            elif isinstance(packet, sdk.ImuPacket):
                if scan_idx >= start_scan:
                    yield scan_idx, IMU.from_packet(packet)
        if ls_write is not None:  # the stream ended inside a frame
            yield scan_idx, ls_write

    def __iter__(self):
        """Make an iterator just data"""
        for scan_idx, d in self.withScanIdx():
            yield scan_idx, d

    def close(self) -> None:
        """Close the underlying PacketSource."""
        self._source.close()

    @property
    def metadata(self):
        """Return metadata from the underlying PacketSource."""
        return self._source.metadata
