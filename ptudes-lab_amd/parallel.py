"""Multi-GPU layer: independent sequences sharded one rank per GPU, no data-path collective.

The path is a strict recurrence inside a sequence (pose / map / threshold / filter state of scan k feed scan
k+1), so it shards across sequences only.  The single collective is the final gather of each
rank's (T, 8) NC-GT rows [t, x, y, z, qx, qy, qz, qw]; `torch.distributed` carries it (backend "nccl" = RCCL
over xGMI on the GPU node, "gloo" in the CPU tests).
"""
import numpy as np


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
