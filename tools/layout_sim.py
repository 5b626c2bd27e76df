"""Map-layout arithmetic on the host (no GPU, no oracle): what a table size or a block size does to the LINES a search touches.

Builds an approximation of the local map the pipeline holds after N sweeps of a synthetic sequence - the sweeps placed with their
ground-truth poses, down-sampled like frame_downsample (first point per 0.5 vs voxel), inserted first-come up to 20 points per voxel
(KISS-ICP's VoxelHashMap.AddPoints as the reference uses it, kiss.py:129) - and reports

  * the fill of the voxels (points per voxel) and what a block of 512 B / a block sized by the fill costs in 128-byte lines per visit;
  * for a brick-addressed table (one 128-byte line per 2 x 2 x 2 brick of voxels, linear probing by lines) of 2^k slots: the share of
    bricks that do not sit on their home line, and the probability that the 8 brick lookups of a 27-voxel neighbourhood need a
    second dependent round.

    python tools/layout_sim.py [default|config5] [N_SWEEPS]

DESIGN.md section 8 quotes these figures as estimates for the next layout step; the measured effect of the table size is in
profiles/r04_y_map_table_and_rebuild.txt and profiles/r04_z_config5_tables.txt."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd  # noqa: E402,F401
from ptudes_lab_amd import synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "default"
n = int(sys.argv[2]) if len(sys.argv) > 2 else (25 if which == "default" else 30)
if which == "config5":
    H, W, rmax, vs = 64, 2048, 100.0, 0.1
else:
    H, W, rmax, vs = 128, 1024, 70.0, 0.7
seq = synth.make_sequence(seed=1000, n_scans=n, H=H, W=W, min_range=1.0, max_range=rmax)
gt = seq.gt_poses()
P = 20


def keys(p, s):
    return np.trunc(p / s).astype(np.int64)  # (int)(p / s): truncation toward zero like the reference's voxel key


def pack(k):
    return ((k[:, 0] + (1 << 20)) << 42) | ((k[:, 1] + (1 << 20)) << 21) | (k[:, 2] + (1 << 20))


mk = np.zeros(0, np.int64)  # the map: packed voxel keys (sorted), stored counts, first points
mc = np.zeros(0, np.int64)
mf = np.zeros((0, 3))
for k in range(n):
    x = seq.scan(k).astype(np.float64)
    r = np.linalg.norm(x, axis=1)
    x = x[(r > 1.0) & (r < rmax)]
    _, idx = np.unique(pack(keys(x, 0.5 * vs)), return_index=True)  # first point per 0.5 vs voxel, scan order kept
    fd = x[np.sort(idx)]
    w = fd @ gt[k][:3, :3].T + gt[k][:3, 3]
    uk, ui, uc = np.unique(pack(keys(w, vs)), return_index=True, return_counts=True)  # (return_index: the first occurrence = the voxel's first point)
    pos = np.searchsorted(mk, uk)
    old = (pos < len(mk)) & (mk[np.minimum(pos, max(len(mk) - 1, 0))] == uk) if len(mk) else np.zeros(len(uk), bool)
    mc[pos[old]] = np.minimum(P, mc[pos[old]] + uc[old])
    mk = np.concatenate([mk, uk[~old]]); mc = np.concatenate([mc, np.minimum(P, uc[~old])]); mf = np.concatenate([mf, w[ui[~old]]])
    o = np.argsort(mk, kind="stable")
    mk, mc, mf = mk[o], mc[o], mf[o]
    keep = np.sum((mf - gt[k][:3, 3]) ** 2, axis=1) <= rmax * rmax  # RemovePointsFarFromLocation: by the voxel's first point
    mk, mc, mf = mk[keep], mc[keep], mf[keep]
cnt = mc
print(f"{which}: {n} sweeps of {H}x{W}, voxel {vs} m -> {len(cnt)} voxels, {cnt.sum()} points, {cnt.mean():.2f} points per voxel "
      f"(<= 4: {np.mean(cnt <= 4):.0%}, <= 9: {np.mean(cnt <= 9):.0%}, 20: {np.mean(cnt == 20):.0%})")
lines_r4 = np.ceil((16 + 24 * cnt) / 128)  # rounds 3-4: header + the stored points in the block
lines = np.ceil(24 * cnt / 128)            # round 5: the block holds points only (header and first point in the block directory)
print(f"  512-byte blocks: {lines.mean():.2f} lines per visit of a voxel (its points; {lines_r4.mean():.2f} with the round-4 header in front), pool {len(cnt) * 512 / 1e6:.0f} MB live; "
      f"the prune pass reads {len(cnt) * 40 / 1e6:.1f} MB of directory per scan instead of {len(cnt) * 128 / 1e6:.1f} MB of block lines")
cls = np.where(cnt <= 5, 128, 512)  # round 5's two classes: 128-byte blocks of 5 points, full blocks
print(f"  two block classes (<= 5 points: 128 B, else 512 B): {np.mean(cnt <= 5):.0%} of the voxels small, pool {cls.sum() / 1e6:.0f} MB live "
      f"({cls.sum() / (len(cnt) * 512):.0%} of the one-class layout); three classes (128 / 256 / 512 B by fill) would need "
      f"{np.where(cnt <= 5, 128, np.where(cnt <= 10, 256, 512)).sum() / 1e6:.0f} MB")

kk = mk
kx, ky, kz = (kk >> 42) - (1 << 20), ((kk >> 21) & ((1 << 21) - 1)) - (1 << 20), (kk & ((1 << 21) - 1)) - (1 << 20)
bricks = np.unique(np.stack([kx >> 1, ky >> 1, kz >> 1], 1), axis=0)
print(f"  {len(bricks)} bricks of 2x2x2 voxels ({len(cnt) / len(bricks):.2f} voxels per brick)")


def mix(b):  # splitmix64 finaliser over the packed brick key (the kernels hash the brick the same way in spirit: icp_kernels.h mix64 / brick_slot)
    h = ((b[:, 0].astype(np.int64) + (1 << 20)) << 42 | (b[:, 1].astype(np.int64) + (1 << 20)) << 21 | (b[:, 2].astype(np.int64) + (1 << 20))).astype(np.uint64)
    h ^= h >> np.uint64(30)
    h *= np.uint64(0xBF58476D1CE4E5B9)
    h ^= h >> np.uint64(27)
    h *= np.uint64(0x94D049BB133111EB)
    h ^= h >> np.uint64(31)
    return h


h = mix(bricks)
for lg in range(21, 29):
    nl = 1 << (lg - 3)  # lines of 8 slots
    home = (h % np.uint64(nl)).astype(np.int64)
    hs = np.sort(home)
    # linear probing over sorted home lines: brick i lands on max(home_i, previous landing + 1) (wrap-around ignored: the table is sparse)
    land = np.maximum.accumulate(hs - np.arange(len(hs))) + np.arange(len(hs))
    displaced = int(np.sum(land != hs))
    occ = len(bricks) / nl
    pd = displaced / len(bricks)
    # a 27-voxel neighbourhood spans 8 bricks; present ones find their line unless displaced, absent ones end on an empty line unless it is taken
    p8 = 1.0 - (1.0 - max(pd, occ)) ** 8
    print(f"  table 2^{lg} slots ({(1 << lg) * 16 / 1e6:.0f} MB): {occ:.1%} of the lines taken, {pd:.1%} of the bricks off their home line, "
          f"a row rebuild meets a second round with p ~ {p8:.0%}")
