"""Multi-GPU layer: independent sequences sharded one rank per GPU, no data-path collective.

The path is a strict recurrence inside a sequence (pose / map / threshold / filter state of scan k feed scan
k+1), so it shards across sequences only.  The single collective is the final gather of each
rank's (T, 8) NC-GT rows [t, x, y, z, qx, qy, qz, qw]: `Comm` = the library's own RCCL entry points
(include/ptudes_mi.h ptl_comm_* / ptl_gather_trajectories: ncclAllGather over xGMI, device rows in, host rows out, no
torch in the data path).  The library does no bootstrap - the 128-byte id travels over whatever control plane the
caller has (`Comm.over`: a torch.distributed gloo group).  `gather_trajectories` is the host-memory twin over
`torch.distributed` (gloo in the CPU tests; several ranks that share one GPU, which RCCL refuses).
"""
import ctypes as C

import numpy as np


class Comm:
    """RCCL communicator of `world` ranks for the final trajectory gather (ptl_comm).  One rank makes the id
    (`Comm.unique_id()`), every rank constructs with the same bytes."""

    ID_BYTES = 128

    def __init__(self, id_bytes, world, rank, device_id=0):
        from . import _lib as L
        if len(id_bytes) != self.ID_BYTES:
            raise ValueError(f"the communicator id has {self.ID_BYTES} bytes, got {len(id_bytes)}")
        self.world, self.rank, self.device_id = int(world), int(rank), int(device_id)
        self._h = C.c_void_p()
        buf = (C.c_uint8 * self.ID_BYTES).from_buffer_copy(bytes(id_bytes))
        L.check(L.lib().ptl_comm_create(buf, self.world, self.rank, self.device_id, C.byref(self._h)))

    @staticmethod
    def unique_id():
        from . import _lib as L
        buf = (C.c_uint8 * Comm.ID_BYTES)()
        L.check(L.lib().ptl_comm_unique_id(buf))
        return bytes(buf)

    @classmethod
    def over(cls, dist, group, device_id=0, make_id=None):
        """bring the communicator up over a torch.distributed control group (gloo): rank 0 makes the id, a broadcast
        carries the bytes, every rank joins.  make_id: stand-in for `Comm.unique_id` (CPU tests of the plumbing)"""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        return cls(broadcast_id(dist, group, make_id or cls.unique_id), world, rank, device_id)

    def gather_rows(self, d_rows_ptr, S, T, counts):
        """every rank: device pointer to (S, T, 8) f64 rows + S counts -> {(rank, j): (count, 8) ndarray} on every rank"""
        from . import _lib as L
        cnt = np.ascontiguousarray(counts, dtype=np.int64)
        rows = np.empty((self.world, S, T, 8))
        cnts = np.empty((self.world, S), dtype=np.int64)
        L.check(L.lib().ptl_gather_trajectories(self._h, C.c_void_p(d_rows_ptr), S, T, cnt.ctypes.data_as(L.c_i64_p),
                                                L.dptr(rows), cnts.ctypes.data_as(L.c_i64_p)))
        return {(r, j): rows[r, j, : int(cnts[r, j])].copy() for r in range(self.world) for j in range(S)}

    def gather_batch(self, batch_runner):
        """the rows a BatchRunner's filter kernel wrote on device, of every rank's batch (same n_sequences / n_scans everywhere)"""
        from . import _lib as L
        S, T = batch_runner.S, batch_runner.n_scans
        rows = np.empty((self.world, S, T, 8))
        cnts = np.empty((self.world, S), dtype=np.int64)
        L.check(L.lib().ptl_batch_gather_trajectories(batch_runner._h, self._h, L.dptr(rows), cnts.ctypes.data_as(L.c_i64_p)))
        return {(r, j): rows[r, j, : int(cnts[r, j])].copy() for r in range(self.world) for j in range(S)}

    def close(self):
        if getattr(self, "_h", None):
            try:
                from . import _lib as L
                L.lib().ptl_comm_destroy(self._h)
            except Exception:  # noqa: BLE001  (interpreter shutdown: the library may be gone already)
                pass
            self._h = None

    __del__ = close


def broadcast_id(dist, group, make_id):
    """rank 0 of `group` calls make_id() -> bytes; every rank returns those bytes (hex through a broadcast of objects: any
    control-plane backend carries it)"""
    # A make_id() that raises on rank 0 (librccl cannot be opened: the case bench.py's gloo fallback exists for) must not leave the other
    # ranks blocked in this broadcast while rank 0 moves on to other collectives of the same group: the error travels IN the broadcast
    # and every rank raises together.
    box = [None]
    if dist.get_rank(group) == 0:
        try:
            box = [("id", make_id().hex())]
        except Exception as e:  # noqa: BLE001
            box = [("error", repr(e))]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if hasattr(dist, "get_global_rank") else 0, group=group)
    kind, payload = box[0]
    if kind != "id":
        raise RuntimeError(f"rank 0 could not make the communicator id: {payload}")
    return bytes.fromhex(payload)


def shard_sequences(n_sequences: int, rank: int, world: int):
    """sequence ids owned by `rank`: s with s % world == rank"""
    return [s for s in range(n_sequences) if s % world == rank]


def sequence_seed(seq_id: int, base: int = 1000) -> int:
    return base + seq_id


def gather_trajectories(rows, counts, dist, device=None):
    """All-gather per-rank trajectory rows.

    rows: torch tensor (S, T_max, 8) float64 on this rank's device, padded; counts: list of S valid lengths.
    Returns on every rank {sequence_slot (rank, j): ndarray (count, 8)}."""
    import torch
    world = dist.get_world_size()
    cnt = torch.tensor(counts, dtype=torch.int64, device=rows.device)
    all_cnt = [torch.empty_like(cnt) for _ in range(world)]
    all_rows = [torch.empty_like(rows) for _ in range(world)]
    dist.all_gather(all_cnt, cnt)
    dist.all_gather(all_rows, rows)
    out = {}
    for r in range(world):
        c = all_cnt[r].cpu().numpy()
        a = all_rows[r].cpu().numpy()
        for j in range(len(c)):
            out[(r, j)] = a[j, : int(c[j])].copy()
    return out


def max_over_ranks(value: float, dist, device="cpu", group=None) -> float:
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


def poses_to_rows(t, poses):
    """(T,), (T, 4, 4) -> (T, 8) NC-GT rows [t, x, y, z, qx, qy, qz, qw] (host-side twin of the rows the filter kernel
    writes on device; used when the gather has to go through host memory)"""
    from scipy.spatial.transform import Rotation
    poses = np.asarray(poses, dtype=np.float64).reshape(-1, 4, 4)
    rows = np.zeros((len(poses), 8))
    if len(poses):
        rows[:, 0] = np.asarray(t, dtype=np.float64)
        rows[:, 1:4] = poses[:, :3, 3]
        rows[:, 4:8] = Rotation.from_matrix(poses[:, :3, :3]).as_quat()
    return rows


def rows_to_poses(rows: np.ndarray):
    """(T, 8) NC-GT rows -> (t (T,), poses (T, 4, 4))"""
    from scipy.spatial.transform import Rotation
    T = np.tile(np.eye(4), (len(rows), 1, 1))
    if len(rows):
        T[:, :3, :3] = Rotation.from_quat(rows[:, 4:8]).as_matrix()
        T[:, :3, 3] = rows[:, 1:4]
    return rows[:, 0].copy(), T
