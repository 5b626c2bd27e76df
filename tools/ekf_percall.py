import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import ptudes_lab_amd
from ptudes_lab_amd.ins.es_ekf import ESEKF
from ptudes_lab_amd.ins.data import IMU
ekf = ESEKF()
_ = ekf.nav
rng = np.random.default_rng(0)
t0 = time.perf_counter()
for k in range(100):
    for i in range(10):
        ekf.processImu(IMU(np.array([0.1, 0.0, 9.8]), np.array([0.01, 0.0, 0.0]), 100.0 + k * 0.1 + i * 0.01))
    t1 = time.perf_counter(); p = ekf.nav.pose_mat(); t2 = time.perf_counter()
    ekf.processPose(p)
    if k % 25 == 0: print("nav read %.1f us" % ((t2 - t1) * 1e6))
print("per scan %.1f us" % ((time.perf_counter() - t0) / 100 * 1e6))
