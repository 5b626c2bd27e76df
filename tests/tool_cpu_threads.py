"""thread-scaling of the CPU oracle's multi-core timing mode (oracle.h orc_set_threads) on this host"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ptudes_lab_amd
from ptudes_lab_amd import synth
from oracle import cpu as orc
sq = synth.make_sequence(seed=2024, n_scans=40, H=128, W=1024, max_range=70.0)
t01 = sq.column_times()
print('cores', len(os.sched_getaffinity(0)), os.cpu_count())
for nt in [int(a) for a in sys.argv[1:]] or (1, 2, 4, 8):
    orc.set_threads(nt)
    icp = orc.ICP(max_range=70.0, min_range=1.0); ekf = orc.EKF()
    sp = 0; out=[]
    for k in range(40):
        x = sq.scan(k).astype(np.float64); a,b = sq.imu_range_for_scan(k)
        t0=time.perf_counter()
        for i in range(a,b): ekf.process_imu(sq.imu[i,1:4], sq.imu[i,4:7], sq.imu[i,0])
        pose = icp.register_frame(x, t01, ekf.pose_mat()); ekf.process_pose(pose)
        sp += time.perf_counter()-t0; out.append(pose)
    print(nt, 40/sp, np.array(out)[-1][:3,3])
