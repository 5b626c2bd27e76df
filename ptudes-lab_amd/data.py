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
    """Scans and IMU samples of an Ouster packet stream, each tagged with the index of the scan it belongs to."""

    def __init__(self, source, *, fields=None, _sdk=None) -> None:
        self._source, self._sdk = source, _sdk
        self._fields = source.metadata.format.udp_profile_lidar if fields is None else fields

    def withScanIdx(self, *, start_scan: int = 0, end_scan: Optional[int] = None) -> Iterator[Tuple[int, object]]:
        """(scan index, LidarScan | IMU) in packet order (reference data.py:31-77).  A scan is emitted when the batcher
        closes it, i.e. on the first packet of the next frame; IMU samples carry the index of the scan being assembled;
        nothing is emitted before `start_scan`; the stream stops after scan `end_scan`; a frame left open when the
        packets run out is emitted as it is."""
        sdk = self._sdk or _ouster_sdk()
        info = self._source.metadata
        geometry = (info.format.pixels_per_column, info.format.columns_per_frame)
        batcher = sdk.ScanBatcher(geometry[1], sdk.PacketFormat.from_info(info))
        open_scan, index = None, 0
        for pkt in self._source:
            emit = index >= start_scan
            if isinstance(pkt, sdk.ImuPacket):
                if emit:
                    yield index, IMU.from_packet(pkt)
                continue
            if not isinstance(pkt, sdk.LidarPacket):
                continue
            if open_scan is None:
                open_scan = sdk.LidarScan(geometry[0], geometry[1], self._fields, info.format.columns_per_packet)
            if not batcher(pkt, open_scan):
                continue
            if emit:
                yield index, open_scan
            open_scan, index = None, index + 1
            if end_scan is not None and index > end_scan:
                return
        if open_scan is not None:
            yield index, open_scan

    def __iter__(self):
        return self.withScanIdx()

    def close(self) -> None:
        self._source.close()

    @property
    def metadata(self):
        return self._source.metadata
