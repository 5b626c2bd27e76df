"""ESEKF: drop-in for reference src/ptudes/ins/es_ekf.py:57-365, computed on the GPU.

Same constructor keywords, methods (`processImu`, `processPose`), properties (`nav`, `ts`) and logging
members (`_navs`, `_navs_pred`, `_navs_t`, `_nav_update_idxs`, `_lg_*`, used by ins/data.py:170-204 and the
reference's plots).  The filter state lives in device memory behind `ptl_ekf_*` (include/ptudes_mi.h); each
call is one kernel launch, `processImuBatch` runs a whole batch of samples in one launch.
"""
from copy import deepcopy
from typing import Optional

import numpy as np

from .. import core
from .data import GRAV, IMU, NavState

DOWN = np.array([0.0, 0.0, -1.0])
UP = np.array([0.0, 0.0, 1.0])


class ESEKF:
    STATE_RANK = 18
    POS_ID, VEL_ID, PHI_ID, BG_ID, BA_ID, G_ID = 0, 3, 6, 9, 12, 15

    def __init__(self, *, init_grav=None, init_bacc=None, init_bgyr=None, _logging: bool = False, device_id: int = 0):
        grav = GRAV * DOWN if init_grav is None else np.asarray(init_grav, dtype=np.float64)
        self._ekf = core.Ekf(grav, init_bacc, init_bgyr, device_id=device_id)
        self._logging = _logging
        self._cov_init = self._ekf.cov
        self._imu_idx = 0
        self._imu_initialized = False
        self._ts = 0.0
        self._nav_cache = None
        self._lg_t, self._lg_acc, self._lg_gyr = [], [], []
        self._navs, self._navs_pred, self._navs_t, self._nav_update_idxs = [], [], [], []

    # ------------------------------------------------------------------ state access
    def _refresh(self):
        nav, cov = self._ekf.state()
        self._nav_cache = (NavState.from_vector(nav), cov)

    @property
    def nav(self) -> NavState:
        """current NavState (es_ekf.py:181-184)"""
        if self._nav_cache is None:
            self._refresh()
        return self._nav_cache[0]

    @property
    def _cov(self) -> np.ndarray:
        if self._nav_cache is None:
            self._refresh()
        return self._nav_cache[1]

    @property
    def ts(self) -> float:
        """timestamp of the last processed IMU sample (es_ekf.py:186-189)"""
        return self._ts

    # ------------------------------------------------------------------ steps
    def processImu(self, imu: IMU) -> None:
        """predict step (es_ekf.py:191-237)"""
        imu.dt = imu.ts - self._ts
        self._ekf.process_imu(imu.lacc, imu.avel, imu.ts)
        self._ts = float(imu.ts)
        self._imu_idx += 1
        self._nav_cache = None
        if not self._imu_initialized:
            self._imu_initialized = True
            return
        if self._logging:
            self._lg_t.append(imu.ts)
            self._lg_acc.append(np.array(imu.lacc, dtype=np.float64))
            self._lg_gyr.append(np.array(imu.avel, dtype=np.float64))
            nav = deepcopy(self.nav)
            self._navs.append(nav)
            self._navs_t.append(imu.ts)
            pred = deepcopy(nav)
            pred.cov = self._cov.copy()
            self._navs_pred.append(pred)

    def processImuBatch(self, imus) -> None:
        """many predict steps in one launch; `imus` is a list of IMU or an (n, 7) array (ts, lacc, avel).
        No per-sample logging."""
        if len(imus) == 0:
            return
        rows = np.array([[i.ts, *i.lacc, *i.avel] for i in imus]) if isinstance(imus[0], IMU) else np.asarray(imus)
        self._ekf.process_imu_batch(rows)
        self._ts = float(rows[-1, 0])
        self._imu_idx += len(rows)
        self._imu_initialized = True
        self._nav_cache = None

    def _after_fused_step(self, n_imu: int, ts: float) -> None:
        """bookkeeping after core.icp_ekf_step ran `n_imu` predicts and one update on this filter's device state (sequence.run_events)"""
        if n_imu:
            self._imu_idx += n_imu
            self._imu_initialized = True
        self._ts = float(ts)
        self._nav_cache = None

    def processPose(self, pose_corr: np.ndarray, meas_cov: Optional[np.ndarray] = None) -> None:
        """update step with a 4x4 pose measurement and optional 6x6 covariance (es_ekf.py:259-329)"""
        if self._logging:
            pred = deepcopy(self.nav)
            pred.cov = self._cov.copy()
            self._navs_pred.append(pred)
        self._ekf.process_pose(pose_corr, meas_cov)
        self._nav_cache = None
        if self._logging:
            nav = deepcopy(self.nav)
            nav.cov = self._cov.copy()
            nav.update = True
            nav.kiss_pose = np.array(pose_corr)
            self._navs.append(nav)
            self._navs_t.append(self._ts)
            self._nav_update_idxs.append(len(self._navs) - 1)
            assert len(self._navs) == len(self._navs_pred)
