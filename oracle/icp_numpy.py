"""Second, independent restatement of the KISS-ICP 0.2.10 per-scan pipeline as the reference drives it
(reference src/ptudes/kiss.py:83-131).  TEST INFRASTRUCTURE ONLY - a cross-check of oracle/oracle_icp.c.

Written from SURVEY.md Appendix A (A.1 - A.6) alone, not from the C oracle: numpy arrays, a dict of Python lists as
the voxel map, a brute-force search over the 27 neighbour voxels by padded gathers + argmin, scipy's Rotation for
SO(3).  Nothing here shares code, data structures or summation order with the C restatement or the HIP kernels;
tests/test_oracle_icp_kat.py runs both on the same sweeps and demands the same poses (1e-9) and the same integer
statistics.  It is not a pin (kiss-icp itself is absent: parity stays unpinned) - it is the test that would have
caught round 1's bare-matrix pose composition.

Deterministic choices where upstream leaves the order to its hash map (the same ones the oracle documents): a voxel
down-sampling keeps the points in scan order; a voxel keeps its first points in scan order.
"""
import numpy as np
from scipy.spatial.transform import Rotation

I3 = np.eye(3)


def hat(w):
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def se3_exp(xi):
    """Sophus order (upsilon, omega): [Exp(omega), V upsilon], V = I + (1 - cos t)/t^2 K + (t - sin t)/t^3 K^2"""
    v, w = np.asarray(xi[:3], float), np.asarray(xi[3:], float)
    th = float(np.linalg.norm(w))
    K = hat(w)
    if th < 1e-8:
        V = I3 + 0.5 * K + K @ K / 6.0
    else:
        V = I3 + (1.0 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * (K @ K)
    T = np.eye(4)
    T[:3, :3] = Rotation.from_rotvec(w).as_matrix()
    T[:3, 3] = V @ v
    return T


def se3_log(T):
    w = Rotation.from_matrix(T[:3, :3]).as_rotvec()
    th = float(np.linalg.norm(w))
    K = hat(w)
    if th < 1e-8:
        Vi = I3 - 0.5 * K + K @ K / 12.0
    else:
        Vi = I3 - 0.5 * K + (1.0 / th ** 2 - (1.0 + np.cos(th)) / (2.0 * th * np.sin(th))) * (K @ K)
    return np.concatenate([Vi @ T[:3, 3], w])


def as_se3(T):
    """what a Sophus::SE3d holds of a 4x4: a UNIT quaternion and the translation.  The quaternion comes from Eigen's
    matrix -> quaternion conversion (positive trace: w = sqrt(1 + tr) / 2 and the antisymmetric part; otherwise the branch of
    the largest diagonal entry), normalised by Sophus - NOT from an SVD: for a matrix that is not a rotation the two differ
    at second order (scipy's Rotation.from_matrix would orthogonalise first)."""
    m = np.asarray(T, float)[:3, :3]
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    q = np.zeros(4)  # x, y, z, w
    if tr > 0.0:
        t = np.sqrt(tr + 1.0)
        q[3] = 0.5 * t
        t = 0.5 / t
        q[0], q[1], q[2] = (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        q[i] = 0.5 * t
        t = 0.5 / t
        q[3] = (m[k, j] - m[j, k]) * t
        q[j] = (m[j, i] + m[i, j]) * t
        q[k] = (m[k, i] + m[i, k]) * t
    q /= np.linalg.norm(q)
    out = np.eye(4)
    out[:3, :3] = Rotation.from_quat(q).as_matrix()
    out[:3, 3] = np.asarray(T, float)[:3, 3]
    return out


def inv(T):
    out = np.eye(4)
    out[:3, :3] = T[:3, :3].T
    out[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return out


def voxel_index(p, size):
    return np.trunc(p / size).astype(np.int64)  # (int)(x / size): C truncation toward zero (A.3)


def voxel_downsample(pts, size):
    """first point of every voxel, in input order (A.3)"""
    if len(pts) == 0:
        return pts
    _, first = np.unique(voxel_index(pts, size), axis=0, return_index=True)
    return pts[np.sort(first)]


def _code(k):
    off = 1 << 20
    return ((k[..., 0] + off) << 42) | ((k[..., 1] + off) << 21) | (k[..., 2] + off)


class VoxelMap:
    def __init__(self, voxel_size, max_distance, cap):
        self.vs, self.max_distance, self.cap = voxel_size, max_distance, cap
        self.vox = {}

    def add(self, pts):
        for key, p in zip(map(tuple, voxel_index(pts, self.vs)), pts):
            lst = self.vox.get(key)
            if lst is None:
                self.vox[key] = [p]
            elif len(lst) < self.cap:
                lst.append(p)

    def prune(self, origin):
        far = [k for k, lst in self.vox.items() if np.sum((lst[0] - origin) ** 2) > self.max_distance ** 2]
        for k in far:
            del self.vox[k]

    def update(self, pts, pose):
        self.add(pts @ pose[:3, :3].T + pose[:3, 3])
        self.prune(pose[:3, 3])

    def num_points(self):
        return sum(len(v) for v in self.vox.values())

    def tables(self):
        """sorted voxel codes, padded point table (M, cap, 3) with +inf beyond a voxel's points, counts"""
        keys = np.array(list(self.vox.keys()), dtype=np.int64).reshape(-1, 3)
        code = _code(keys)
        order = np.argsort(code)
        pts = np.full((len(keys) + 1, self.cap, 3), np.inf)  # (last row: the absent voxel)
        cnt = np.zeros(len(keys) + 1, dtype=np.int64)
        for row, j in enumerate(order):
            lst = self.vox[tuple(keys[j])]
            pts[row, :len(lst)] = lst
            cnt[row] = len(lst)
        return code[order], pts, cnt


_OFFS = np.array([(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1)], dtype=np.int64)  # i, j, k ascending (A.5)


def nearest(src, tables, vs):
    """nearest stored point in the 27 voxels around each source point: candidates in visiting order (voxels i, j, k ascending,
    points in insertion order), the first of the strictly smallest squared distances wins.  Returns (points, d2, found, #candidates)."""
    code, pts, cnt = tables
    k0 = voxel_index(src, vs)
    rows = np.empty((len(src), 27), dtype=np.int64)
    for o, off in enumerate(_OFFS):
        c = _code(k0 + off)
        pos = np.minimum(np.searchsorted(code, c), max(len(code) - 1, 0))
        hit = (code[pos] == c) if len(code) else np.zeros(len(src), bool)
        rows[:, o] = np.where(hit, pos, len(code))
    cand = pts[rows].reshape(len(src), -1, 3)                      # (N, 27 * cap, 3)
    d2 = np.sum((cand - src[:, None, :]) ** 2, axis=2)
    best = np.argmin(d2, axis=1)                                    # the FIRST minimum: visiting order breaks ties
    ar = np.arange(len(src))
    return cand[ar, best], d2[ar, best], np.isfinite(d2[ar, best]), int(cnt[rows].sum())


def register(src, vmap, guess, max_dist, kernel, max_iter=500, conv=1e-4):
    """A.5; returns (pose, iterations, pairs of the last iteration, candidates examined)"""
    guess = as_se3(guess)
    if not vmap.vox:
        return guess, 0, 0, 0
    s = src @ guess[:3, :3].T + guess[:3, 3]
    T = np.eye(4)
    tables = vmap.tables()
    iters = ncorr = ncand = 0
    for _ in range(max_iter):
        nn, d2, found, c = nearest(s, tables, vmap.vs)
        ncand += c
        m = found & (np.sqrt(d2) < max_dist)
        sp, r = s[m], s[m] - nn[m]
        ncorr = int(m.sum())
        w = kernel ** 2 / (kernel + np.sum(r * r, axis=1)) ** 2
        J = np.zeros((len(sp), 3, 6))
        J[:, :, :3] = I3
        J[:, 0, 4], J[:, 0, 5] = sp[:, 2], -sp[:, 1]              # -hat(s)
        J[:, 1, 3], J[:, 1, 5] = -sp[:, 2], sp[:, 0]
        J[:, 2, 3], J[:, 2, 4] = sp[:, 1], -sp[:, 0]
        JTJ = np.einsum("n,nai,naj->ij", w, J, J)
        JTr = np.einsum("n,nai,na->i", w, J, r)
        dx = np.linalg.solve(JTJ, -JTr) if ncorr and np.linalg.matrix_rank(JTJ) == 6 else np.zeros(6)
        E = se3_exp(dx)
        s = s @ E[:3, :3].T + E[:3, 3]
        T = E @ T
        iters += 1
        if np.linalg.norm(dx) < conv:
            break
    return as_se3(T @ guess), iters, ncorr, ncand


class KissICP:
    """deskew, range filter, two voxel down-samplings, adaptive threshold, guess, registration, map update, pose list"""

    def __init__(self, max_range=100.0, min_range=5.0, deskew=True):
        self.max_range, self.min_range, self.deskew = max_range, min_range, deskew
        self.vs = max_range / 100.0                                  # A.1
        self.map = VoxelMap(self.vs, max_range, 20)
        self.poses = []
        self.initial_threshold, self.min_motion = 2.0, 0.1
        self.sse, self.n_samples = 0.0, 0
        self.model_deviation = np.eye(4)
        self.stats = []

    def has_moved(self):
        if not self.poses:
            return False
        return np.linalg.norm((inv(self.poses[0]) @ self.poses[-1])[:3, 3]) > 5.0 * self.min_motion

    def threshold(self):                                            # A.4
        if not self.has_moved():
            return self.initial_threshold
        theta = np.linalg.norm(Rotation.from_matrix(self.model_deviation[:3, :3]).as_rotvec())
        e = np.linalg.norm(self.model_deviation[:3, 3]) + 2.0 * self.max_range * np.sin(theta / 2.0)
        if e > self.min_motion:
            self.sse += e * e
            self.n_samples += 1
        return self.initial_threshold if self.n_samples < 1 else np.sqrt(self.sse / self.n_samples)

    def prediction(self):
        return inv(self.poses[-2]) @ self.poses[-1] if len(self.poses) >= 2 else np.eye(4)

    def register_frame(self, xyz, t01, guess=None):
        frame = np.asarray(xyz, float)
        if self.deskew and len(self.poses) >= 2:                    # A.2
            xi = se3_log(inv(self.poses[-2]) @ self.poses[-1])
            out = np.empty_like(frame)
            for t in np.unique(t01):
                M = se3_exp((t - 0.5) * xi)
                sel = t01 == t
                out[sel] = frame[sel] @ M[:3, :3].T + M[:3, 3]
            frame = out
        rng = np.linalg.norm(frame, axis=1)
        frame = frame[(rng > self.min_range) & (rng < self.max_range)]
        fd = voxel_downsample(frame, 0.5 * self.vs)
        src = voxel_downsample(fd, 1.5 * self.vs)
        sigma = self.threshold()
        if guess is None:
            guess = (self.poses[-1] if self.poses else np.eye(4)) @ self.prediction()
        pose, iters, ncorr, ncand = register(src, self.map, guess, 3.0 * sigma, sigma / 3.0)
        gain = np.linalg.inv(np.asarray(guess, float)) @ pose       # kiss.py:116, :128 - the RAW guess, a general 4x4 inverse
        err_dt = float(np.linalg.norm(gain[:3, 3]))                 # kiss.py:118
        err_drot = float(np.linalg.norm(Rotation.from_matrix(gain[:3, :3]).as_rotvec()))  # kiss.py:119-120
        self.model_deviation = as_se3(gain)                         # update_model_deviation takes it as a Sophus::SE3d
        self.map.update(fd, pose)
        self.poses.append(pose)
        self.stats.append(dict(sigma=float(sigma), err_dt=err_dt, err_drot=err_drot, iterations=iters, n_corr_last=ncorr, n_valid=len(frame), n_down=len(fd), n_src=len(src),
                               sum_cand=ncand, map_voxels=len(self.map.vox), map_points=self.map.num_points()))
        return pose
