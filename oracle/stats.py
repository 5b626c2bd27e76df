"""CPU restatement of the reference's StreamStatsTracker (TEST INFRASTRUCTURE ONLY - see oracle.h).

Follows /root/reference/src/ptudes/ins/data.py:207-369 statement by statement (numpy, like the reference):
trackImu :266-282 (Welford mean and sigma^2 * n), trackScan :284-321 (per-scan mean / population variance of the
non-zero ranges of the selected beams, pooled into a running sample variance), the properties :323-349 and the
report :351-366.  Pinned by tests/golden/stream_stats.npz (+ the two report texts), which gen_golden.py produced by
running the reference itself.
"""
import numpy as np


class StreamStats:
    def __init__(self, use_beams_num=None, range_to_m=0.001):
        self.mean, self.scans_num, self.points_num, self.sigma_sq = 0, 0, 0, 0
        self.use_beams_num, self.beams_sel = use_beams_num, None
        self.mean_acc, self.mean_gyr = np.zeros(3), np.zeros(3)
        self.sigman_acc, self.sigman_gyr = np.zeros(3), np.zeros(3)
        self.imu_num = 0
        self.max_ts = self.min_ts = 0
        self.min_range = self.max_range = 0
        self.range_to_m = range_to_m  # :242-252 (x8 for the RNG15 profile)

    def _ts(self, ts):  # :254-260
        if not self.imu_num and not self.scans_num:
            self.min_ts = self.max_ts = ts
        else:
            self.min_ts, self.max_ts = min(self.min_ts, ts), max(self.max_ts, ts)

    def track_imu(self, lacc, avel, ts):  # :266-282
        pa, pg = self.mean_acc.copy(), self.mean_gyr.copy()
        self.mean_acc = self.mean_acc + (lacc - self.mean_acc) / (self.imu_num + 1)
        self.sigman_acc = self.sigman_acc + (lacc - pa) * (lacc - self.mean_acc)
        self.mean_gyr = self.mean_gyr + (avel - self.mean_gyr) / (self.imu_num + 1)
        self.sigman_gyr = self.sigman_gyr + (avel - pg) * (avel - self.mean_gyr)
        self._ts(ts)
        self.imu_num += 1

    def scan_moments(self, range_img):
        """(n, mean, population variance, min, max) of the selected non-zero ranges in metres (:286-308)"""
        if self.use_beams_num:
            if self.beams_sel is None:
                self.beams_sel = np.linspace(0, range_img.shape[0], num=self.use_beams_num, endpoint=False, dtype=int)
            r = range_img[self.beams_sel, :]
        else:
            r = range_img
        r = r[r > 0] * self.range_to_m
        return r.size, np.mean(r), np.var(r), np.min(r), np.max(r)

    def track_scan(self, range_img, last_col_ts_ns):  # :284-321
        n, m, v, lo, hi = self.scan_moments(range_img)
        if not self.points_num:  # :262-264
            self.min_range, self.max_range = lo, hi
        else:
            self.min_range, self.max_range = min(self.min_range, lo), max(self.max_range, hi)
        s1 = 0 if not self.points_num else (self.points_num - 1) * self.sigma_sq
        corr = self.points_num * n * np.square(self.mean - m) / ((self.points_num + n) * (self.points_num + n - 1))
        self.sigma_sq = (s1 + n * v) / (self.points_num + n - 1) + corr
        self.mean = (self.mean * self.points_num + m * n) / (self.points_num + n)
        self._ts(last_col_ts_ns * 1e-9)
        self.scans_num += 1
        self.points_num += n

    range_mean = property(lambda self: self.mean)
    range_std = property(lambda self: np.sqrt(self.sigma_sq))
    acc_std = property(lambda self: np.sqrt(self.sigman_acc / self.imu_num))
    gyr_std = property(lambda self: np.sqrt(self.sigman_gyr / self.imu_num))
    dt = property(lambda self: self.max_ts - self.min_ts)

    def row(self):
        return np.concatenate([[self.range_mean, self.range_std, self.min_range, self.max_range, self.points_num,
                                self.scans_num, self.dt], self.mean_acc, self.acc_std, self.mean_gyr, self.gyr_std])
